#!/usr/bin/env python3
"""Pinned copy rates with both directions at once (two streams)."""
import time
import torch
dev = torch.device("cuda:0")
n = 1 << 30
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(both):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(s1):
        h1.copy_(d1, non_blocking=True)
    if both:
        with torch.cuda.stream(s2):
            d2.copy_(h2, non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t
run(True)
a = min(run(False) for _ in range(3)); b = min(run(True) for _ in range(3))
print("D2H alone %.1f GB/s; D2H + H2D at once: %.1f GB/s each direction (%.1f total)" % (n / a / 1e9, n / b / 1e9, 2 * n / b / 1e9))
# D2H of 6 x 128 MB pieces back to back on one stream against one 768 MB copy
pieces = [h1[i * (n // 8):(i + 1) * (n // 8)] for i in range(8)]
def chunks():
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(s1):
        for i, p in enumerate(pieces):
            p.copy_(d1[i * (n // 8):(i + 1) * (n // 8)], non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t
chunks(); c = min(chunks() for _ in range(3))
print("D2H as 8 copies of 128 MiB on one stream: %.1f GB/s" % (n / c / 1e9))
