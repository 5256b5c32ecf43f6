#!/usr/bin/env python3
"""Jacobi-PCG at C2 (4096^2) through the device-pointer entry point: (1) time against the iteration count (linear: no
hidden per-solve cost), (2) the same solve on several freshly allocated (x, b) pairs with pads in between -- how much of
an iteration is where the caller's two vectors lie (vectors of 128 MiB against 256 MB of Infinity Cache)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
grid = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "4096,4096,0").split(","))
A = dev.DeviceCSR.poisson(*grid)
n = A.shape[0]
K = dev.DeviceJacobi(A)
aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
ones = dev.DeviceBuffer.from_host(np.ones(n))
info, it, rr = C.c_int(), C.c_int(), C.c_double()


def solve_us(x, b, iters):
    best = 1e9
    for _ in range(3):
        x.zero()
        check(L.psp_synchronize())
        t = time.perf_counter()
        check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
        check(L.psp_synchronize())
        best = min(best, time.perf_counter() - t)
    return best / iters * 1e6


keep = []
for j in range(6):
    keep.append(dev.DeviceBuffer((11 + 53 * j) * (1 << 17) + 512 * j))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    keep.append(dev.DeviceBuffer((7 + 29 * j) * (1 << 17)))
    x = dev.DeviceBuffer(n)
    keep += [b, x]
    print("pair %d  x %s  b %s  %s" % (j, hex(x.ptr), hex(b.ptr), "  ".join("%d its: %.1f us" % (k, solve_us(x, b, k))
                                                                          for k in (100, 400, 1600 if j == 0 else 400))), flush=True)
