#!/usr/bin/env python3
"""The CPU side of tools/bench_configs.py: the scalar C oracle (the restatement of the reference's
decoders, one thread, value at a time like the Go code) timed on pages of the same shape as each GPU
configuration.  Lives under tests/ because only test code may call the oracle.  One JSON line per
configuration: output GB/s of one host core.

    python tests/time_oracle_configs.py [--seconds 1.0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import oracle_lib as O  # noqa: E402
from parquet_go_amd import synth  # noqa: E402


def timed(fn, out_bytes, seconds):
    fn()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        reps += 1
    dt = time.perf_counter() - t0
    return reps * out_bytes / dt / 1e9, dt / reps


def report(name, gbps, per_call, extra=None):
    d = {"config": name, "cpu_out_GBps_one_core": round(gbps, 4), "ms_per_page": round(per_call * 1e3, 4)}
    d.update(extra or {})
    print(json.dumps(d), flush=True)


def runny(rng, n, hi, mean_run):
    out = np.empty(n, dtype=np.int64)
    i = 0
    while i < n:
        ln = int(rng.geometric(1.0 / mean_run))
        out[i:i + ln] = rng.integers(0, hi)
        i += ln
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    n = 20000
    d = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)

    keys = rng.integers(0, 4096, size=n)
    page = np.concatenate([[12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8)
    report("C1 dict w=12 uniform keys", *timed(lambda: O.dict_decode(page, d, n), 4 * n, a.seconds))
    keys = runny(rng, n, 4096, 32)
    page_b = np.concatenate([[12], synth.hybrid_encode(keys, 12)]).astype(np.uint8)
    report("C1b dict w=12 run-heavy keys (mean run 32)", *timed(lambda: O.dict_decode(page_b, d, n), 4 * n, a.seconds))

    rep = (rng.random(n) >= 0.25).astype(np.int64)
    deff = np.where(rng.random(n) < 0.8, 3, rng.integers(0, 3, size=n))
    k = int((deff == 3).sum())
    nested = np.concatenate([synth.length_prefixed(synth.hybrid_encode(rep, 1)),
                             synth.length_prefixed(synth.hybrid_encode(deff, 2)),
                             [12], synth.hybrid_encode_bitpacked(rng.integers(0, 4096, size=k), 12)]).astype(np.uint8)
    report("C4 nested V1 pages (rep w=1, def w=2, dict w=12)",
           *timed(lambda: O.page_read(nested, 1, n, 1, 3, O.VALUES_DICT, dictionary=d), 8 * n + 4 * k, a.seconds))

    for w in (4, 16, 30):
        v = np.cumsum(rng.integers(0, 1 << w, size=n, dtype=np.int64) - 5)
        s = synth.delta_encode(v, 64)
        report("C2x delta64 width %d" % w, *timed(lambda: O.delta_decode64(s, n), 8 * n, a.seconds))

    raw = rng.integers(0, 256, size=8 * n, dtype=np.uint8)
    report("P PLAIN int64", *timed(lambda: O.plain_decode(raw, n, np.int64), 8 * n, a.seconds))
    bits = rng.integers(0, 256, size=(n + 7) // 8, dtype=np.uint8)
    report("P BOOLEAN PLAIN", *timed(lambda: O.bool_plain_decode(bits, n), 4 * n, a.seconds))

    vals = [bytes(rng.integers(0, 256, size=int(rng.integers(4, 21)), dtype=np.uint8)) for _ in range(n)]
    ls = synth.delta_length_ba_encode(vals)
    ob = sum(map(len, vals)) + 4 * (n + 1)
    report("B DELTA_LENGTH_BYTE_ARRAY 4..20-byte values", *timed(lambda: O.delta_length_ba_decode(ls, n), ob, a.seconds))
    words = sorted(bytes(rng.integers(97, 123, size=int(rng.integers(3, 14)), dtype=np.uint8)) for _ in range(n))
    fs = synth.delta_ba_encode(words)
    room = sum(map(len, words))
    report("B DELTA_BYTE_ARRAY sorted words", *timed(lambda: O.delta_ba_decode(fs, n, room), room + 4 * (n + 1), a.seconds))

    body = 20000
    heads = [np.frombuffer(synth.page_header(0, body, body, data=(20000, 8, 3, 3, synth.statistics(rng))), dtype=np.uint8)
             for _ in range(16)]
    parts = []
    for i in range(800):
        parts += [heads[i % 16], np.zeros(body, np.uint8)]
    chunk = np.concatenate(parts)
    g, per = timed(lambda: O.walk_chunk(chunk, chunk.size, 4), 800, a.seconds)
    report("W page-header walk, one chunk of 800 pages", g, per,
           {"cpu_pages_per_s_one_core": round(800 / per), "us_per_page": round(per * 1e6 / 800, 4),
            "note": "cpu_out_GBps_one_core is not meaningful for this row"})

    m = 2_000_000
    rep = (rng.random(m) < 0.75).astype(np.int32)
    deff = np.where((rng.random(m) < 0.9) | (rep == 1), 3, rng.integers(0, 3, size=m)).astype(np.int32)
    g, per = timed(lambda: O.column_entries(deff, rep, 3, 1, True), 8 * (m // 4), a.seconds)
    report("L column entries, REPEATED (max_rep 1, max_def 3)", g, per,
           {"cpu_slots_per_s_one_core": round(m / per)})
    g, per = timed(lambda: O.levels_pack(deff, 2), m // 4, a.seconds)
    report("L PackedArray bytes of int32 levels, width 2", g, per, {"cpu_slots_per_s_one_core": round(m / per)})


if __name__ == "__main__":
    main()
