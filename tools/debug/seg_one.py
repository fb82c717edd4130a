#!/usr/bin/env python3
"""One configuration for counter passes: 100 x 1M-value int64 delta pages, segmented, 5 runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from parquet_go_amd import capi  # noqa: E402
import bench_configs as B  # noqa: E402

stream = torch.cuda.Stream(device=B.DEV)
ctx = capi.Context(0, stream.cuda_stream)
B.bench_delta(ctx, stream, 100, 1_000_000, 64, 16, ", segmented")
ctx.close()
