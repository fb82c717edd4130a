#!/usr/bin/env python3
"""Practical HBM ceilings on this box: pure write, pure read, copy, and a 1:4 read:write mix
(the delta workload's ratio), measured with torch kernels."""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n / 1e3
N = 800_000_000
out = torch.empty(N, dtype=torch.uint8, device=dev)
src = torch.empty(N, dtype=torch.uint8, device=dev).random_()
o64, s64 = out.view(torch.int64), src.view(torch.int64)
t = timeit(lambda: o64.fill_(7)); print("write-only  %.0f GB/s" % (N / t / 1e9))
t = timeit(lambda: s64.sum()); print("read-only   %.0f GB/s" % (N / t / 1e9))
t = timeit(lambda: o64.copy_(s64)); print("copy        %.0f GB/s (read+write)" % (2 * N / t / 1e9))
q = s64[: N // 8 // 4]
t = timeit(lambda: torch.add(q.repeat_interleave(1), 1, out=o64[: N // 8 // 4])); 
# 1:4 mix: read N/4 bytes, write N bytes (expand each int64 into 4 outputs)
qq = s64[: N // 32]
t = timeit(lambda: torch.add(qq.unsqueeze(1).expand(-1, 4), 1, out=o64[: N // 8].view(-1, 4)))
print("1:4 mix     %.0f GB/s (read+write), write part %.0f GB/s" % ((N + N / 4) / t / 1e9, N / t / 1e9))
