#!/usr/bin/env python3
"""Multi-row-group workload: 64 column chunks (4 row groups x 16 columns) decoded from HOST memory.

Even columns are INT64 DELTA_BINARY_PACKED (miniblock width 16), odd columns INT32 RLE_DICTIONARY
(width 12, 4096-entry dictionary); every chunk has 800 pages x 20 000 values (16 M values).  Chunks are
assigned to the ranks by parquet_go_amd.shard.plan_shards; a rank streams its chunks through
`--slots` pipeline slots, each with its own HIP stream, context and device buffers:
H2D of the encoded chunk -> decode -> D2H of the decoded values, all in stream order, so the copies
of one slot overlap the decode of another.  Host buffers are pinned.

Reports, per job (all ranks): the end-to-end rate (output GB/s including both copies) and the
device-resident rate of the same decode launches with the data already in HBM.

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


def build_pool(kind, distinct, n, seed):
    """-> list of (stream bytes, expected values)"""
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


def build_chunk(pool, pages):
    """One chunk image: pages 16-byte aligned.  -> (src uint8, offs, lens, page->pool index)"""
    which = np.arange(pages) % len(pool)
    lens = np.array([pool[i][0].size for i in which], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])
    src = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
    for p, i in enumerate(which):
        src[offs[p]: offs[p] + lens[p]] = pool[i][0]
    return src, offs[:-1], lens, which


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--row-groups", type=int, default=4)
    ap.add_argument("--columns", type=int, default=16)
    ap.add_argument("--pages", type=int, default=800)
    ap.add_argument("--values-per-page", type=int, default=20000)
    ap.add_argument("--distinct", type=int, default=50)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--wave-min-pages", type=int, default=-1, help="override both kernel-selection thresholds")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # be the launcher: N fresh ranks of this script, started before anything here touches the GPU
        from bench import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    global torch, capi, shard, synth
    import torch
    from parquet_go_amd import capi, shard, synth

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

    n, pages = args.values_per_page, args.pages
    n_chunks = args.row_groups * args.columns
    kinds = ["delta" if (c % args.columns) % 2 == 0 else "dict" for c in range(n_chunks)]
    pools = {"delta": build_pool("delta", args.distinct, n, 11), "dict": build_pool("dict", args.distinct, n, 12)}
    images = {k: build_chunk(pools[k], pages) for k in pools}            # every chunk of a kind has this image
    elem = {"delta": 8, "dict": 4}
    infos = [shard.ChunkInfo(pages, int(images[k][2].sum()), pages * n * elem[k]) for k in kinds]
    plan_of_rank = shard.plan_shards(infos, world)
    mine = [p.chunk for p in plan_of_rank[rank] if p.page_hi > p.page_lo]  # >= as many chunks as ranks: whole chunks

    dictionary = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    t_dict = torch.from_numpy(dictionary.copy()).to(dev)

    # pinned host images (one per kind; every chunk of the kind reads it) and one pinned output per chunk
    h_src = {k: torch.from_numpy(images[k][0]).pin_memory() for k in images}
    h_out = [torch.empty(pages * n * elem[kinds[c]], dtype=torch.uint8).pin_memory() for c in mine]

    slots = []
    for s in range(args.slots):
        st = torch.cuda.Stream(device=dev)
        ctx = capi.Context(local, st.cuda_stream)
        if args.wave_min_pages >= 0:
            ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, args.wave_min_pages)
            ctx.set_option(capi.OPT_HYBRID_WAVE_MIN_PAGES, args.wave_min_pages)
        bufs, plans = {}, {}
        for k in images:
            src, offs, lens, _ = images[k]
            descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
            descs["kind"] = capi.KIND_DELTA64 if k == "delta" else capi.KIND_DICT
            descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
            descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(elem[k] * n)
            if k == "dict":
                descs["elem_size"], descs["dict_len"] = 4, 4096
            plans[k] = ctx.plan(descs)
            bufs[k] = (torch.empty(src.size, dtype=torch.uint8, device=dev),
                       torch.empty(pages * n * elem[k], dtype=torch.uint8, device=dev))
        slots.append((st, ctx, bufs, plans))
    torch.cuda.synchronize()

    def decode(slot, k):
        st, ctx, bufs, plans = slot
        d_src, d_out = bufs[k]
        if k == "delta":
            plans[k].run(d_src.data_ptr(), d_src.numel(), d_out.data_ptr(), d_out.numel())
        else:
            plans[k].run(d_src.data_ptr(), d_src.numel(), d_out.data_ptr(), d_out.numel(), 0, 0,
                         t_dict.data_ptr(), 4096 * 4)

    def end_to_end():
        for i, c in enumerate(mine):
            slot = slots[i % len(slots)]
            k = kinds[c]
            with torch.cuda.stream(slot[0]):
                slot[2][k][0].copy_(h_src[k], non_blocking=True)
                decode(slot, k)
                h_out[i].copy_(slot[2][k][1], non_blocking=True)
        for s in slots:
            s[0].synchronize()

    def resident():
        for i, c in enumerate(mine):
            decode(slots[i % len(slots)], kinds[c])
        for s in slots:
            s[0].synchronize()

    def timed(fn):
        fn()                                           # warm-up
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

    t_e2e = timed(end_to_end)
    # verification of what came back to the host (first and last chunk of this rank)
    for i in sorted({0, len(mine) - 1}):
        k = kinds[mine[i]]
        which = images[k][3]
        got = h_out[i].numpy().view(np.int64 if k == "delta" else np.int32).reshape(pages, n)
        for p in (0, pages // 2, pages - 1):
            want = pools[k][which[p]][1]
            want = want if k == "delta" else dictionary[want]
            assert np.array_equal(got[p], want), (k, p)
    ran = {(i % len(slots), kinds[c]) for i, c in enumerate(mine)}
    for si, k in sorted(ran):
        res = slots[si][3][k].results()
        assert (res["status"] == 0).all() and (res["n_decoded"] == n).all()
    t_res = timed(resident)

    if rank == 0:
        out_bytes = sum(i.decoded_bytes for i in infos)
        in_bytes = sum(i.encoded_bytes for i in infos)
        print(json.dumps({
            "workload": "%d row groups x %d columns = %d chunks (even: INT64 delta w=16, odd: INT32 dictionary w=12), "
                        "%d pages x %d values per chunk" % (args.row_groups, args.columns, n_chunks, pages, n),
            "n_gpus": world, "chunks_per_rank": [len(p) for p in plan_of_rank],
            "imbalance": round(shard.imbalance(infos, plan_of_rank), 4),
            "encoded_GB": round(in_bytes / 1e9, 3), "decoded_GB": round(out_bytes / 1e9, 3),
            "end_to_end_s": round(t_e2e, 4), "end_to_end_out_GBps": round(out_bytes / t_e2e / 1e9, 2),
            "resident_s": round(t_res, 5), "resident_out_GBps": round(out_bytes / t_res / 1e9, 1),
            "slots": args.slots, "distinct_pages_per_kind": args.distinct,
        }), flush=True)
    for s in slots:
        for k in images:
            s[3][k].close()
        s[1].close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
