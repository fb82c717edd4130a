#!/usr/bin/env python3
"""Decode front-coded value lists on the GPU and print where the first wrong byte is."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from parquet_go_amd import capi, synth
from byte_array_cases import random_values, total_len

ctx = capi.Context(0)
for shape in ("sorted", "runs", "short"):
    for seed in range(3):
        for n in (3, 63, 64, 65, 127, 128, 130, 200, 1000, 5000):
            rng = np.random.default_rng(seed * 100 + n)
            vals = random_values(rng, n, shape)
            fs = synth.delta_ba_encode(vals)
            room = total_len(vals)
            got = ctx.delta_ba_decode(fs, n, room)
            flat = np.frombuffer(b"".join(vals), dtype=np.uint8)
            offs = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int64)
            ok = got["status"] == 0 and np.array_equal(got["data"], flat)
            if ok:
                continue
            d = got["data"]
            bad = np.nonzero(d[: flat.size] != flat)[0]
            print(shape, seed, n, "status", got["status"], "nd", got["n_decoded"], "bad bytes", bad.size, "of", flat.size, "longest", max(len(v) for v in vals))
            if bad.size:
                for x in bad[:6]:
                    i = int(np.searchsorted(offs, x, side="right") - 1)
                    pl = 0
                    if i:
                        a, b = vals[i - 1], vals[i]
                        while pl < min(len(a), len(b)) and a[pl] == b[pl]:
                            pl += 1
                    print("   byte", int(x), "value", i, "pos", int(x - offs[i]), "len", len(vals[i]), "prefix", pl, "got", int(d[x]), "want", int(flat[x]))
print("done")
