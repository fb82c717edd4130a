#!/usr/bin/env python3
"""Multi-row-group workload: 64 column chunks (4 row groups x 16 columns) decoded from HOST memory.

Even columns are INT64 DELTA_BINARY_PACKED (miniblock width 16), odd columns INT32 RLE_DICTIONARY
(width 12, 4096-entry dictionary); every chunk has 800 pages x 20 000 values (16 M values).  Chunks are
assigned to the ranks by parquet_go_amd.shard.plan_shards; a rank hands its chunks to ONE call of
pq_decode_chunks (the C ABI's pipelined host-memory entry point: three slots, each with its own stream,
device staging and pinned staging, so that the copy in of one chunk, the decode of another and the copy
back of a third overlap).

Reports, per job (all ranks): the end-to-end rate (output GB/s including both copies) with the caller's
buffers in ordinary pageable memory (what a cgo caller passes) and in pinned memory from pq_host_alloc,
the device-resident rate of the same decode launches, and the box's plain pinned copy rates.

    python tools/bench_chunks.py [--gpus N]      (N > 1: starts its own N ranks, or runs under torchrun)

Page payloads repeat: a chunk is built from a pool of `--distinct` distinct pages (encoding 51 200
distinct pages on the host takes longer than everything measured here); offsets, sizes and the
bytes moved and decoded are those of the full workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ELEM = {"delta": 8, "dict": 4}


def build_pool(kind, distinct, n, seed):
    """-> list of (stream bytes, expected values)"""
    from parquet_go_amd import synth
    rng = np.random.default_rng(seed)
    pool = []
    for p in range(distinct):
        if kind == "delta":
            steps = rng.integers(0, 1 << 16, size=n - 1, dtype=np.int64) - 128
            v = np.concatenate([[p * 1000003], p * 1000003 + np.cumsum(steps)]).astype(np.int64)
            pool.append((synth.delta_encode(v, 64, 128, 4), v))
        else:
            keys = rng.integers(0, 4096, size=n)
            s = np.concatenate([[12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8)
            pool.append((s, keys))
    return pool


def build_chunk(pool, pages, shift=0):
    """One chunk image: pages 16-byte aligned, page p = pool entry (p + shift) % len(pool).
    -> (src uint8, offs, lens, page -> pool index)"""
    which = (np.arange(pages) + shift) % len(pool)
    lens = np.array([pool[i][0].size for i in which], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])
    src = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
    for p, i in enumerate(which):
        src[offs[p]: offs[p] + lens[p]] = pool[i][0]
    return src, offs[:-1], lens, which


def build_workload(row_groups=4, columns=16, pages=800, n=20000, distinct=50):
    """-> dict(kinds per chunk, pools, images per (kind, shift), infos (shard.ChunkInfo per chunk),
    dictionary, descs per kind)"""
    from parquet_go_amd import capi, shard, synth
    n_chunks = row_groups * columns
    kinds = ["delta" if (c % columns) % 2 == 0 else "dict" for c in range(n_chunks)]
    pools = {"delta": build_pool("delta", distinct, n, 11), "dict": build_pool("dict", distinct, n, 12)}
    # chunks of one kind differ by a rotation of the page pool (4 distinct images per kind)
    images = {(k, s): build_chunk(pools[k], pages, shift=7 * s) for k in pools for s in range(4)}
    infos = [shard.ChunkInfo(pages, int(images[(k, c % 4)][2].sum()), pages * n * ELEM[k]) for c, k in enumerate(kinds)]
    dictionary = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    descs = {}
    for key, (src, offs, lens, _) in images.items():
        k = key[0]
        d = np.zeros(pages, dtype=capi.DESC_DTYPE)
        d["kind"] = capi.KIND_DELTA64 if k == "delta" else capi.KIND_DICT
        d["src_off"], d["src_len"], d["n_values"] = offs, lens, n
        d["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(ELEM[k] * n)
        if k == "dict":
            d["elem_size"], d["dict_len"] = 4, 4096
        descs[key] = d
    return dict(kinds=kinds, pools=pools, images=images, infos=infos, dictionary=dictionary, descs=descs,
                pages=pages, n=n)


def chunk_jobs(wl, mine, alloc):
    """Jobs of pq_decode_chunks for the chunks `mine`; alloc(nbytes) -> uint8 array for inputs and outputs."""
    dict_bytes = wl["dictionary"].view(np.uint8)
    jobs = []
    for c in mine:
        k = wl["kinds"][c]
        key = (k, c % 4)
        src = alloc(wl["images"][key][0].size)
        src[:] = wl["images"][key][0]
        job = dict(descs=wl["descs"][key], src=src, values=alloc(wl["pages"] * wl["n"] * ELEM[k]), key=key, chunk=c)
        if k == "dict":
            job["dict"] = alloc(dict_bytes.size)
            job["dict"][:] = dict_bytes
        jobs.append(job)
    return jobs


def check_jobs(wl, jobs, outcome, every_value=True):
    """Statuses, counts and values of decoded jobs against what the pages were written from."""
    n, pages = wl["n"], wl["pages"]
    for job, (status, res) in zip(jobs, outcome):
        assert status == 0, (job["chunk"], status)
        assert (res["status"] == 0).all() and (res["n_decoded"] == n).all(), job["chunk"]
        k = job["key"][0]
        which = wl["images"][job["key"]][3]
        got = job["values"].view(np.int64 if k == "delta" else np.int32).reshape(pages, n)
        pool = wl["pools"][k]
        if every_value:
            want = np.stack([p[1] for p in pool])                      # (distinct, n)
            want = want if k == "delta" else wl["dictionary"][want]
            for lo in range(0, pages, 100):
                hi = min(pages, lo + 100)
                assert np.array_equal(got[lo:hi], want[which[lo:hi]]), (job["chunk"], lo)
        else:
            for p in (0, pages // 2, pages - 1):
                want = pool[which[p]][1]
                want = want if k == "delta" else wl["dictionary"][want]
                assert np.array_equal(got[p], want), (job["chunk"], p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--row-groups", type=int, default=4)
    ap.add_argument("--columns", type=int, default=16)
    ap.add_argument("--pages", type=int, default=800)
    ap.add_argument("--values-per-page", type=int, default=20000)
    ap.add_argument("--distinct", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--slots", type=int, default=0, help="pipeline slots of pq_decode_chunks (0 = the default)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # be the launcher: N fresh ranks of this script, started before anything here touches the GPU
        from bench import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    from parquet_go_amd import capi, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    wl = build_workload(args.row_groups, args.columns, args.pages, args.values_per_page, args.distinct)
    plan_of_rank = shard.plan_shards(wl["infos"], world)
    mine = [p.chunk for p in plan_of_rank[rank] if p.page_hi > p.page_lo]  # >= as many chunks as ranks: whole chunks
    ctx = capi.Context(local)
    if args.slots:
        ctx.set_option(capi.OPT_CHUNK_SLOTS, args.slots)

    def timed(fn):
        fn()                                           # warm-up (sizes the staging)
        best = None
        for _ in range(args.repeats):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            best = dt if best is None else min(best, dt)
        return best

    # ---- pageable caller memory (numpy arrays), then pinned memory from pq_host_alloc ----
    times = {}
    for mode in ("pageable", "pinned"):
        held = []

        def alloc(nbytes, mode=mode, held=held):
            if mode == "pinned":
                a = ctx.host_alloc(nbytes)
                held.append(a)
                return a
            return np.empty(nbytes, dtype=np.uint8)
        jobs = chunk_jobs(wl, mine, alloc)
        last = {}

        def run(jobs=jobs, mode=mode, last=last):
            last["outcome"] = ctx.decode_chunks(jobs, pinned=(mode == "pinned"))
        times[mode] = timed(run)
        check_jobs(wl, jobs, last["outcome"], every_value=False)
        del jobs
        for a in held:
            ctx.host_free(a)

    # ---- the same chunks resident in device memory, every chunk in buffers of its own ----
    stream = torch.cuda.Stream(device=dev)
    rctx = capi.Context(local, stream.cuda_stream)
    t_dict = torch.from_numpy(wl["dictionary"].copy()).to(dev)
    chunk_bufs, chunk_plans = [], []
    for c in mine:
        key = (wl["kinds"][c], c % 4)
        chunk_bufs.append((torch.from_numpy(wl["images"][key][0]).to(dev),
                           torch.empty(wl["pages"] * wl["n"] * ELEM[key[0]], dtype=torch.uint8, device=dev)))
        chunk_plans.append(rctx.plan(wl["descs"][key]))
    torch.cuda.synchronize()

    def set_of(c, bufs):
        d_src, d_out = bufs
        dict_args = (t_dict.data_ptr(), 4096 * 4) if wl["kinds"][c] == "dict" else (0, 0)
        return (d_src.data_ptr(), d_src.numel(), d_out.data_ptr(), d_out.numel(), 0, 0) + dict_args

    def resident_per_chunk():      # one plan per chunk, run one after the other (what pq_decode_chunks launches)
        for c, bufs, plan in zip(mine, chunk_bufs, chunk_plans):
            plan.run(*set_of(c, bufs))
        stream.synchronize()
    t_res_each = timed(resident_per_chunk)

    # one plan over all of the rank's chunks (pq_plan_create_sets): a single set of launches
    all_descs = np.concatenate([wl["descs"][(wl["kinds"][c], c % 4)] for c in mine])
    first = np.arange(len(mine) + 1, dtype=np.uint32) * wl["pages"]
    big = rctx.plan(all_descs, first_desc=first)
    sets = [set_of(c, bufs) for c, bufs in zip(mine, chunk_bufs)]

    def resident():
        big.run_sets(sets)
        stream.synchronize()
    t_res = timed(resident)
    res = big.results()
    assert (res["status"] == 0).all() and (res["n_decoded"] == wl["n"]).all()
    for j, c in enumerate(mine):       # first, middle and last page of every chunk
        k = wl["kinds"][c]
        which = wl["images"][(k, c % 4)][3]
        got = chunk_bufs[j][1].view(torch.int64 if k == "delta" else torch.int32).view(wl["pages"], wl["n"])
        for p in (0, wl["pages"] // 2, wl["pages"] - 1):
            want = wl["pools"][k][which[p]][1]
            want = want if k == "delta" else wl["dictionary"][want]
            assert np.array_equal(got[p].cpu().numpy(), want), (c, p)
    big.close()
    res_plans = dict(enumerate(chunk_plans))

    # ---- the box's plain copy rates (pinned, 1 GiB) ----
    h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    d = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def copy_rate(a, b):
        b.copy_(a, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.copy_(a, non_blocking=True)
        torch.cuda.synchronize()
        return (1 << 30) / (time.perf_counter() - t0) / 1e9
    h2d, d2h = copy_rate(h, d), copy_rate(d, h)

    if rank == 0:
        infos = wl["infos"]
        out_bytes = sum(i.decoded_bytes for i in infos)
        in_bytes = sum(i.encoded_bytes for i in infos)
        print(json.dumps({
            "workload": "%d row groups x %d columns = %d chunks (even: INT64 delta w=16, odd: INT32 dictionary w=12), "
                        "%d pages x %d values per chunk" % (args.row_groups, args.columns, len(infos), args.pages,
                                                             args.values_per_page),
            "entry_point": "pq_decode_chunks (%d pipeline slots)" % (args.slots or 3),
            "n_gpus": world, "chunks_per_rank": [len(p) for p in plan_of_rank],
            "imbalance": round(shard.imbalance(infos, plan_of_rank), 4),
            "encoded_GB": round(in_bytes / 1e9, 3), "decoded_GB": round(out_bytes / 1e9, 3),
            "host_to_host_pageable_s": round(times["pageable"], 4),
            "host_to_host_pageable_out_GBps": round(out_bytes / times["pageable"] / 1e9, 2),
            "host_to_host_pinned_s": round(times["pinned"], 4),
            "host_to_host_pinned_out_GBps": round(out_bytes / times["pinned"] / 1e9, 2),
            "resident_s": round(t_res, 5), "resident_out_GBps": round(out_bytes / t_res / 1e9, 1),
            "resident_alg_frac_of_8TBps": round((in_bytes + out_bytes) / t_res / 8e12, 4),
            "resident_one_plan_per_chunk_s": round(t_res_each, 5),
            "resident_one_plan_per_chunk_out_GBps": round(out_bytes / t_res_each / 1e9, 1),
            "pinned_copy_GBps": {"h2d": round(h2d, 1), "d2h": round(d2h, 1)},
            "pinned_over_d2h": round(out_bytes / times["pinned"] / 1e9 / d2h, 3),
            "pageable_over_d2h": round(out_bytes / times["pageable"] / 1e9 / d2h, 3),
            "distinct_pages_per_kind": args.distinct,
        }), flush=True)
    for p in res_plans.values():
        p.close()
    rctx.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
