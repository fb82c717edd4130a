#!/usr/bin/env python3
"""Benchmark: device-resident decode of an INT64 DELTA_BINARY_PACKED column on MI355X.

Workload (per GPU, weak scaling): 5000 pages x 20 000 int64 values = 100 M values,
block size 128 / 4 miniblocks, delta - minDelta uniform in [0, 2^16), pages written by
the reference encoder's algorithm (parquet_go_amd.synth.delta_encode).  A "step" is
one pass of the decode kernels over all pages, inputs already resident in HBM.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment the process starts N ranks of itself (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU and waits for
them; launched under torchrun it is one of the ranks.  `--dry-run` forms the ranks over gloo,
partitions the pages and prints the partition without decoding anything (no GPU needed).

Rank 0 prints ONE JSON line: whole-job output GB/s, the roofline object of the
dominant kernel (algorithmic in+out bytes / HIP-event time, against 8 TB/s HBM) and
the CPU baseline (the scalar C oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, n):
    """Start n ranks of `script` (one process per GPU) and wait for them.  The caller must not have
    touched the GPU: the children are fresh interpreters, nothing is exec'ed over this process.
    Returns the largest exit status."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def rank_pages(total_pages, rank, world):
    """Global page indices [lo, hi) of one rank: the job's pages as one chunk cut by
    parquet_go_amd.shard.plan_shards (contiguous page ranges, nothing shared)."""
    from parquet_go_amd import shard

    pieces = shard.plan_shards([shard.ChunkInfo(total_pages, 0, 0)], world)[rank]
    return (pieces[0].page_lo, pieces[0].page_hi) if pieces else (0, 0)


def make_pages(n_pages, values_per_page, width, seed):
    """-> (src uint8, src_off[], src_len[], expected int64[n_pages, values_per_page])"""
    from parquet_go_amd import synth

    n = values_per_page
    total = n_pages * (n - 1)
    resid = (synth.splitmix64(seed, total) >> np.uint64(64 - width)).astype(np.int64).reshape(n_pages, n - 1)
    first = synth.splitmix64(seed ^ 0x5EED, n_pages).astype(np.int64) >> np.int64(8)
    min_delta = (np.arange(n_pages, dtype=np.int64) % 257) - 128          # per-page base step
    steps = resid + min_delta[:, None]
    expected = np.empty((n_pages, n), dtype=np.int64)
    expected[:, 0] = first
    np.cumsum(steps, axis=1, out=expected[:, 1:])
    expected[:, 1:] += first[:, None]
    streams = [synth.delta_encode(expected[p], 64, 128, 4) for p in range(n_pages)]
    lens = np.array([s.size for s in streams], dtype=np.int64)
    # page starts are 16-byte aligned, like separately allocated page buffers would be
    offs = np.zeros(n_pages, dtype=np.int64)
    pos = 0
    for p in range(n_pages):
        offs[p] = pos
        pos += (int(lens[p]) + 15) // 16 * 16
    src = np.zeros(pos + 16, dtype=np.uint8)
    for p in range(n_pages):
        src[offs[p]: offs[p] + lens[p]] = streams[p]
    return src, offs, lens, expected


def cpu_baseline(src, offs, lens, n, seconds=10.0):
    """Single-thread scalar oracle on the same pages; also an all-cores figure."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O

    lib = O.lib()
    fn = lib.pqo_delta_decode64
    n_pages = offs.size
    base = src.ctypes.data

    def decode_range(lo, hi):
        out = np.empty(n, dtype=np.int64)
        nd, cons, vc, ist = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_int(0)
        for p in range(lo, hi):
            st = fn(C.c_void_p(base + int(offs[p])), C.c_size_t(int(lens[p])), C.c_uint32(n),
                    C.c_void_p(out.ctypes.data), C.byref(nd), C.byref(cons), C.byref(vc), C.byref(ist))
            assert st == 0 and nd.value == n
        return hi - lo

    t0 = time.perf_counter()
    pages_done = 0
    while time.perf_counter() - t0 < seconds:
        pages_done += decode_range(0, n_pages)
    dt1 = time.perf_counter() - t0
    single = pages_done * n * 8 / dt1 / 1e9

    cores = min(os.cpu_count() or 1, 16, len(os.sched_getaffinity(0)))
    chunks = [(i * n_pages // cores, (i + 1) * n_pages // cores) for i in range(cores)]
    t0 = time.perf_counter()
    done = 0
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds / 2:
            done += sum(ex.map(lambda c: decode_range(*c), chunks))
    dtn = time.perf_counter() - t0
    multi = done * n * 8 / dtn / 1e9
    return dict(value=round(single, 4), unit="GB/s (output)", cores=1, kind="port",
                sample="%d pages x %d int64 (the whole GPU workload) decoded %d times by the scalar C "
                       "restatement of the reference decoder" % (n_pages, n, pages_done // n_pages),
                all_cores_value=round(multi, 4), all_cores=cores)


def library_sha256():
    import hashlib

    from parquet_go_amd import capi
    h = hashlib.sha256()
    with open(capi.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 counter passes
    (profiles/r*/traffic.json; the counters cannot be read from inside this process).  The file
    names the workload and the library build (sha256 of libpqdecode.so) it was collected on: any
    other workload or build gets null, so the figure cannot go stale silently.
    -> (bytes or None, dict describing where the figure comes from)"""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not paths:
        return None, {"traffic_source": None}
    path = paths[-1]
    src = {"traffic_source": os.path.relpath(path, ROOT)}
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, src
    src["traffic_library_sha256"] = t.get("library_sha256")
    if (t.get("pages"), t.get("values_per_page"), t.get("width")) != (args.pages, args.values_per_page, args.width):
        src["traffic_note"] = "collected on another workload"
        return None, src
    if t.get("library_sha256") != library_sha256():
        src["traffic_note"] = "collected on another build of libpqdecode.so"
        return None, src
    return t.get("hbm_bytes_per_launch"), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pages", type=int, default=5000)
    ap.add_argument("--values-per-page", type=int, default=20000)
    ap.add_argument("--width", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="form the ranks (gloo), print every rank's page range, decode nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: be the launcher.  Nothing here has imported torch or touched HIP.
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but the launcher formed WORLD_SIZE=%d ranks; using the latter"
              % (args.gpus, world), file=sys.stderr)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # pages are independent: the job's world x pages pages are cut into one range per rank
    page_lo, page_hi = rank_pages(world * args.pages, rank, world)

    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        mine = {"rank": rank, "world": world, "page_lo": page_lo, "page_hi": page_hi}
        ranges = [mine]
        if world > 1:
            ranges = [None] * world
            dist.all_gather_object(ranges, mine)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "total_pages": world * args.pages,
                              "ranks": ranges}), flush=True)
        return

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from parquet_go_amd import capi

    n = args.values_per_page
    assert page_hi - page_lo == args.pages
    src, offs, lens, expected = make_pages(args.pages, n, args.width, seed=0xC2 + 7919 * rank)
    descs = np.zeros(args.pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DELTA64
    descs["src_off"] = offs
    descs["src_len"] = lens
    descs["n_values"] = n
    descs["dst_off"] = np.arange(args.pages, dtype=np.uint64) * np.uint64(n * 8)

    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(local, stream.cuda_stream)
    t_src = torch.from_numpy(src).to(dev)
    out_bytes = args.pages * n * 8
    t_out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    plan = ctx.plan(descs)
    torch.cuda.synchronize()

    def step():
        plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), out_bytes)

    for _ in range(args.warmup):
        step()
    stream.synchronize()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        evs[0].record()
        for i in range(args.steps):
            step()
            evs[i + 1].record()
    stream.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # ---- verification (outside the timed region): statuses and bit-exact values ----
    res = plan.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == n).all(), "decode reported errors"
    got = t_out.view(torch.int64).view(args.pages, n)
    want = torch.from_numpy(expected).to(dev)
    assert torch.equal(got, want), "decoded values differ from the encoded input"
    del want

    kernel_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    in_bytes = int(lens.sum())
    alg_bytes = in_bytes + out_bytes
    k_avg_s = float(np.mean(kernel_ms)) / 1e3
    achieved = alg_bytes / k_avg_s / 1e9

    # ---- host-to-host rate: the same pages through pq_decode_chunks (the C ABI's pipelined entry
    # point for host memory: copy in, decode, copy back, three slots), as 10 jobs of pages / 10 pages,
    # from ordinary pageable memory (what a cgo caller passes) and from pinned memory ----
    def host_jobs(alloc):
        jobs, per = [], (args.pages + 9) // 10
        for lo in range(0, args.pages, per):
            hi = min(args.pages, lo + per)
            b0, b1 = int(offs[lo]), int(offs[hi - 1] + (lens[hi - 1] + 15) // 16 * 16)
            d = descs[lo:hi].copy()
            d["src_off"] -= np.uint64(b0)
            d["dst_off"] -= np.uint64(lo * n * 8)
            s_buf = alloc(b1 - b0)
            s_buf[:] = src[b0:b1]
            jobs.append(dict(descs=d, src=s_buf, values=alloc((hi - lo) * n * 8), lo=lo, hi=hi))
        return jobs

    hctx = capi.Context(local)
    h2h = {}
    for mode in ("pageable", "pinned"):
        held = []

        def alloc(nbytes, mode=mode, held=held):
            if mode == "pinned":
                held.append(hctx.host_alloc(nbytes))
                return held[-1]
            return np.empty(nbytes, dtype=np.uint8)
        jobs = host_jobs(alloc)
        hctx.decode_chunks(jobs, pinned=(mode == "pinned"))          # sizes the staging, faults the pages in
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            outcome = hctx.decode_chunks(jobs, pinned=(mode == "pinned"))
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        for job, (st, res) in zip(jobs, outcome):
            assert st == 0 and (res["status"] == 0).all()
            assert np.array_equal(job["values"].view(np.int64).reshape(-1, n), expected[job["lo"]:job["hi"]])
        h2h[mode] = out_bytes / best / 1e9
        del jobs
        for a in held:
            hctx.host_free(a)
    hctx.close()

    if rank == 0:
        traffic, traffic_src = measured_traffic(args)
        value = world * out_bytes * args.steps / elapsed / 1e9
        line = {
            "metric": "device-resident decode GB/s (output) for hybrid/bitpack/delta pages, 1-8 GPU",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": "INT64 DELTA_BINARY_PACKED column, %d pages x %d values per GPU "
                            "(%.0fM values), block 128 / 4 miniblocks, miniblock width %d"
                            % (args.pages, n, args.pages * n / 1e6, args.width),
                "pages_per_gpu": args.pages,
                "values_per_page": n,
                "encoded_bytes_per_gpu": in_bytes,
                "decoded_bytes_per_gpu": out_bytes,
                "parallelism": "pages partitioned over %d GPU(s), no collective" % world,
                "world_size": world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "pq::delta_wave_kernel<unsigned long>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, see profiles/)",
                **traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(k_avg_s * 1e3, 4),
            },
            "host_to_host": {"GBps_output_pageable": round(h2h["pageable"], 2),
                             "GBps_output_pinned": round(h2h["pinned"], 2),
                             "note": "pq_decode_chunks: H2D of the pages, decode, D2H of the values, 10 jobs "
                                     "through 3 pipeline slots; caller memory pageable / from pq_host_alloc; "
                                     "every value checked"},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU figure is taken once, on the single-GPU run
            line["cpu_baseline"] = cpu_baseline(src, offs, lens, n, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    plan.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
