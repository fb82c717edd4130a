#!/usr/bin/env python3
"""Kernel-selection sweep: decode time of P pages x 20 000 values under the one-wave-per-page and the
one-workgroup-per-page kernels, for INT64 delta (w=16) and INT32 dictionary pages (w=12)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parquet_go_amd import capi  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_chunks as BC  # noqa: E402
import bench_configs as BCFG  # noqa: E402

DEV = torch.device("cuda:0")
n = 20000
stream = torch.cuda.Stream(device=DEV)
ctx = capi.Context(0, stream.cuda_stream)
pools = {"delta": BC.build_pool("delta", 32, n, 1), "dict": BC.build_pool("dict", 32, n, 2)}
t_dict = torch.from_numpy(np.arange(4096, dtype=np.int32)).to(DEV)
for kind in ("delta", "dict"):
    es = 8 if kind == "delta" else 4
    for pages in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        src, offs, lens, _ = BC.build_chunk(pools[kind], pages)
        descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
        descs["kind"] = capi.KIND_DELTA64 if kind == "delta" else capi.KIND_DICT
        descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
        descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(es * n)
        if kind == "dict":
            descs["elem_size"], descs["dict_len"] = 4, 4096
        t_src = torch.from_numpy(src).to(DEV)
        t_out = torch.empty(pages * n * es, dtype=torch.uint8, device=DEV)
        row = {"kind": kind, "pages": pages}
        for name, thr in (("wave_ms", 0), ("workgroup_ms", 1 << 30)):
            ctx.set_option(capi.OPT_DELTA_WAVE_MIN_PAGES, thr)
            ctx.set_option(capi.OPT_HYBRID_WAVE_MIN_PAGES, thr)
            plan = ctx.plan(descs)
            torch.cuda.synchronize()
            if kind == "delta":
                fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), t_out.numel())
            else:
                fn = lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), t_out.numel(), 0, 0,
                                      t_dict.data_ptr(), 4096 * 4)
            row[name] = round(BCFG.timed(fn, stream) * 1e3, 4)
            assert (plan.results()["status"] == 0).all()
            plan.close()
        print(json.dumps(row), flush=True)
ctx.close()
