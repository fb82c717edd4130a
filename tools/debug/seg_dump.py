#!/usr/bin/env python3
"""Decode large delta pages in segments and print, for the first segment that is not decoded by a wave
of its own, its record next to the true block headers around it (walked on the host)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parquet_go_amd import capi, synth  # noqa: E402


def true_headers(s):
    """(offsets of the block headers, offsets of their width bytes) of a 128/4 or 256/4 stream"""
    pos = 0

    def uv():
        nonlocal pos
        v, sh = 0, 0
        while True:
            b = int(s[pos]); pos += 1
            v |= (b & 0x7f) << sh; sh += 7
            if b < 0x80:
                return v
    bs, nm, total, first = uv(), uv(), uv(), uv()
    offs, keys = [], []
    nblocks = (total - 1 + bs - 1) // bs
    for b in range(nblocks):
        offs.append(pos)
        uv()
        keys.append(pos)
        w = s[pos:pos + 4].astype(np.int64); pos += 4
        pos += int(w.sum()) * (bs // 4) // 8
    return np.array(offs), np.array(keys)


def main():
    n = 1_000_000
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    pages = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dt = np.int64 if bits == 64 else np.int32
    rng = np.random.default_rng(3)
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    L = 8192
    broken = 0
    for p in range(pages):
        steps = rng.integers(0, 1 << width, size=n - 1, dtype=np.int64) - 5
        v = np.concatenate([[p * 7], p * 7 + np.cumsum(steps)]).astype(dt)
        s = synth.delta_encode(v, bits)
        descs = np.zeros(1, dtype=capi.DESC_DTYPE)
        descs["kind"] = capi.KIND_DELTA64 if bits == 64 else capi.KIND_DELTA32
        descs["src_len"], descs["n_values"] = s.size, n
        t_src = torch.from_numpy(np.concatenate([s, np.zeros(32, np.uint8)])).to(dev)
        t_out = torch.zeros(n, dtype=torch.int64 if bits == 64 else torch.int32, device=dev)
        plan = ctx.plan(descs)
        plan.run(t_src.data_ptr(), s.size, t_out.data_ptr(), n * (bits // 8))
        res = plan.results()
        segs = plan.segments()
        plan.close()
        ok = bool(np.array_equal(t_out.cpu().numpy(), v))
        skipped = np.flatnonzero(segs["decode"] == 0)
        print("page", p, "status", int(res[0]["status"]), "values ok", ok, "segments", segs.size, "serial", skipped.size)
        if skipped.size > 3:
            broken += 1
            k = int(skipped[0])
            hdr, keys = true_headers(s)
            j = int(np.searchsorted(keys, k * L))
            print("  first serial segment", k, "lo", k * L, {nm: int(segs[k][nm]) for nm in segs.dtype.names if nm != "reserved"})
            print("  segment in front    ", {nm: int(segs[k - 1][nm]) for nm in segs.dtype.names if nm != "reserved"})
            print("  true headers around ", hdr[j - 1:j + 2].tolist(), "width bytes at", keys[j - 1:j + 2].tolist())
            a = int(hdr[j]) - 12
            print("  bytes from", a, ":", s[a:a + 32].tolist())
    print("pages with a broken chain:", broken, "of", pages)


if __name__ == "__main__":
    main()
