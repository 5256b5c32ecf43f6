#!/usr/bin/env python3
"""Fixed cost of ONE solve call: wall time of pcg / minres with maxit = 1 and = 2 through the device-pointer entry points
(vectors resident, nothing copied), 20 calls each: what a caller that solves many short systems pays per call."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
check = _capi.check
for grid in ((100, 100, 0), (300, 300, 0), (1024, 1024, 0), (2048, 2048, 0), (256, 256, 256)):
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    xb.upload(np.ones(n))
    A.matvec_dev(xb.ptr, bb.ptr)
    check(L.psp_synchronize())
    row = {}
    for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
        for kk in (1, 2, 1, 2):
            ts = []
            for _ in range(20):
                xb.zero()
                info, it, rr = C.c_int(), C.c_int(), C.c_double()
                check(L.psp_synchronize())
                t = time.perf_counter()
                check(fn(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
                check(L.psp_synchronize())
                ts.append(time.perf_counter() - t)
            row["%s_maxit%d_us" % (name, kk)] = round(float(np.median(ts)) * 1e6, 1)
    print("x".join(str(g) for g in grid if g), row, flush=True)
