#!/usr/bin/env python3
"""Time the segmented decode of 100 x 1M-value int64 delta pages for several segment lengths."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from parquet_go_amd import capi, synth  # noqa: E402
from bench_configs import pack_pages, timed  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    pages, n, width = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 1_000_000, 16
    rng = np.random.default_rng(3)
    streams = []
    for p in range(pages):
        steps = rng.integers(0, 1 << width, size=n - 1, dtype=np.int64) - 5
        v = np.concatenate([[p * 7], p * 7 + np.cumsum(steps)]).astype(np.int64)
        streams.append(synth.delta_encode(v, 64))
    src, offs, lens = pack_pages(streams)
    descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
    descs["kind"] = capi.KIND_DELTA64
    descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
    descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(8 * n)
    t_src = torch.from_numpy(src).to(DEV)
    t_out = torch.empty(pages * n * 8, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.Stream(device=DEV)
    ctx = capi.Context(0, stream.cuda_stream)
    for L in (2048, 4096, 8192, 16384, 32768, 65536):
        ctx.set_option(capi.OPT_SEGMENT_BYTES, L)
        plan = ctx.plan(descs)
        t = timed(lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * 8), stream)
        res = plan.results()
        segs = plan.segments()
        assert (res["status"] == 0).all()
        print("L", L, "ms", round(t * 1e3, 4), "segments", segs.size, "serial", int((segs["decode"] == 0).sum()),
              "alg frac", round((int(lens.sum()) + pages * n * 8) / t / 8e12, 4), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
