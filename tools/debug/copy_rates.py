#!/usr/bin/env python3
"""Host <-> device copy rates on this box: pinned, pageable (the runtime's own staging), pageable
through hipHostRegister, and host memcpy with 1..16 threads (what a staging copy would cost)."""
import ctypes as C
import os
import threading
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
H2D, D2H = 1, 2


def chk(e, what):
    assert e == 0, (what, e)


def rate(fn, nbytes, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t)
    return nbytes / best / 1e9


def main():
    n = 1 << 30
    d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), n), "malloc")
    p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), n, 0), "hostmalloc")
    pag = np.ones(n, dtype=np.uint8)
    pag2 = np.ones(n, dtype=np.uint8)
    pa = pag.ctypes.data
    print("pinned   H2D %.1f GB/s  D2H %.1f GB/s" % (rate(lambda: hip.hipMemcpy(d, p, n, H2D), n), rate(lambda: hip.hipMemcpy(p, d, n, D2H), n)))
    print("pageable H2D %.1f GB/s  D2H %.1f GB/s" % (rate(lambda: hip.hipMemcpy(d, pa, n, H2D), n), rate(lambda: hip.hipMemcpy(pa, d, n, D2H), n)))
    e = hip.hipHostRegister(pa, n, 0)
    print("hipHostRegister of pageable memory ->", e, "(0 = allowed)")
    if e == 0:
        hip.hipHostUnregister(pa)
    pin = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))
    for th in (1, 2, 4, 8, 16):
        def work(i):
            lo, hi = i * n // th, (i + 1) * n // th
            np.copyto(pag2[lo:hi], pin[lo:hi])
        def run():
            ts = [threading.Thread(target=work, args=(i,)) for i in range(th)]
            [x.start() for x in ts]; [x.join() for x in ts]
        print("memcpy pinned -> pageable, %2d threads: %.1f GB/s" % (th, rate(run, n)))
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))


if __name__ == "__main__":
    main()
