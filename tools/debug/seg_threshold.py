#!/usr/bin/env python3
"""20 000-value pages, few to many: decode time with the pages cut into segments and not."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(3)
    stream = torch.cuda.Stream(device=DEV)
    ctx = capi.Context(0, stream.cuda_stream)
    d = synth.splitmix64(0xD1C7, 4096).astype(np.uint32).view(np.int32)
    t_dict = torch.from_numpy(d.copy()).to(DEV)
    pool_delta, pool_dict = [], []
    for p in range(64):
        steps = rng.integers(0, 1 << 16, size=n - 1, dtype=np.int64) - 5
        v = np.concatenate([[p * 7], p * 7 + np.cumsum(steps)]).astype(np.int64)
        pool_delta.append(synth.delta_encode(v, 64))
        keys = rng.integers(0, 4096, size=n)
        pool_dict.append(np.concatenate([[12], synth.hybrid_encode_bitpacked(keys, 12)]).astype(np.uint8))
    for kind, pool, es in (("delta64", pool_delta, 8), ("dict", pool_dict, 4)):
        for pages in (250, 500, 1000, 2000, 4000, 8000):
            streams = [pool[p % 64] for p in range(pages)]
            src, offs, lens = pack_pages(streams)
            descs = np.zeros(pages, dtype=capi.DESC_DTYPE)
            descs["kind"] = capi.KIND_DELTA64 if kind == "delta64" else capi.KIND_DICT
            descs["src_off"], descs["src_len"], descs["n_values"] = offs, lens, n
            descs["dst_off"] = np.arange(pages, dtype=np.uint64) * np.uint64(es * n)
            if kind == "dict":
                descs["elem_size"], descs["dict_len"] = 4, 4096
            t_src = torch.from_numpy(src).to(DEV)
            t_out = torch.empty(pages * n * es, dtype=torch.uint8, device=DEV)
            row = []
            for seg in (1 << 30, 0):
                ctx.set_option(capi.OPT_SEGMENT_MAX_PAGES, seg)
                plan = ctx.plan(descs)
                t = timed(lambda: plan.run(t_src.data_ptr(), src.size, t_out.data_ptr(), pages * n * es, 0, 0,
                                           t_dict.data_ptr(), 4096 * 4), stream)
                assert (plan.results()["status"] == 0).all()
                plan.close()
                row.append(t)
            alg = int(lens.sum()) + pages * n * es
            print("%-8s pages %5d x %d: segmented %.4f ms (%.3f of 8 TB/s)   whole pages %.4f ms (%.3f)" %
                  (kind, pages, n, row[0] * 1e3, alg / row[0] / 8e12, row[1] * 1e3, alg / row[1] / 8e12), flush=True)


if __name__ == "__main__":
    main()
