#!/usr/bin/env python3
"""Controlled run structures through the hybrid wave kernel, one timed launch series per shape.

Every shape is an INT32 RLE_DICTIONARY page set (width 12, 4096-entry dictionary, `--pages` pages x
20 000 values) whose runs all have one length and kind, so that counters collected over this
program (tools/pmc_pass.sh) give the cost per run and per tile by differences:

    rle<L>      RLE runs of L values each
    bp<G>       bit-packed runs of G groups each
    mix<L>      alternating RLE run of L values and one bit-packed group
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parquet_go_amd import capi, synth  # noqa: E402

DEV = torch.device("cuda:0")


def build(shape, n, rng):
    """-> (stream bytes incl. the width byte, keys)"""
    out, keys = [12], []
    kind = "".join(c for c in shape if c.isalpha())
    arg = int("".join(c for c in shape if c.isdigit()))
    left = n
    while left > 0:
        if kind in ("rle", "mix"):
            k = int(rng.integers(0, 4096))
            out += synth.uleb128(arg << 1) + [k & 0xff, k >> 8]
            keys += [k] * arg
            left -= arg
        if kind in ("bp", "mix") and left > 0:
            g = arg if kind == "bp" else 1
            ks = rng.integers(0, 4096, size=8 * g)
            out += synth.uleb128((g << 1) | 1) + synth.pack_bits(ks, 12).tolist()
            keys += ks.tolist()
            left -= 8 * g
    return np.array(out, dtype=np.uint8), np.array(keys[:n], dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="rle16,rle32,rle64,rle128,rle512,bp1,bp2,bp4,bp8,bp16,mix8,mix24,mix56")
    ap.add_argument("--pages", type=int, default=8000)
    ap.add_argument("--values", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    stream = torch.cuda.Stream(device=DEV)
    ctx = capi.Context(0, stream.cuda_stream)
    d = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    t_dict = torch.from_numpy(d.copy()).to(DEV)
    n, pages = args.values, args.pages
    for shape in args.shapes.split(","):
        rng = np.random.default_rng(9)
        pool = [build(shape, n, rng) for _ in range(16)]
        lens = np.array([pool[p % 16][0].size for p in range(pages)], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])
        src = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
        for p in range(pages):
            src[offs[p]: offs[p] + lens[p]] = pool[p % 16][0]
        descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
        descs["kind"] = capi.KIND_DICT
        descs["src_off"], descs["src_len"], descs["n_values"] = offs[:-1], lens, n
        descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(4 * n)
        descs["elem_size"], descs["dict_len"] = 4, 4096
        t_src = torch.from_numpy(src).to(DEV)
        t_out = torch.empty(pages * n, dtype=torch.int32, device=DEV)
        plan = ctx.plan(descs)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        with torch.cuda.stream(stream):
            evs[0].record()
            for i in range(args.steps):
                plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * 4, 0, 0, t_dict.data_ptr(), 4096 * 4)
                evs[i + 1].record()
        stream.synchronize()
        ms = min(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        res = plan.results()
        assert (res["status"] == 0).all() and (res["n_decoded"] == n).all(), shape
        got = t_out.cpu().numpy().reshape(pages, n)
        for p in (0, pages - 1):
            assert np.array_equal(got[p], d[pool[p % 16][1]]), shape
        print(json.dumps({"shape": shape, "ms": round(ms, 4), "tiles": pages * n / 512,
                          "ns_per_tile_per_simd": round(ms * 1e6 * 1024 / (pages * n / 512), 1),
                          "out_GBps": round(pages * n * 4 / ms / 1e6, 1)}), flush=True)
        plan.close()
        del t_src, t_out
    ctx.close()


if __name__ == "__main__":
    main()
