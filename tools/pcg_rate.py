#!/usr/bin/env python3
"""Jacobi-PCG iterations/s at 512^3 (or the grid given as "nx,ny,nz": csr and sss operands) and MINRES, device-resident
vectors: three timed solves of 100 iterations each, best reported (A/B of vector-kernel builds through PSP_LIB_OVERRIDE)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
out = {}
GRID = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(","))
for form, A in (("csr", dev.DeviceCSR.poisson(*GRID)), ("sss", dev.DeviceSSS.poisson(*GRID))):
    n = A.shape[0] if form == "csr" else A.n
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    x = dev.DeviceBuffer(n)
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
        best = 1e9
        for rep in range(4):
            x.zero()
            check(L.psp_synchronize())
            t = time.perf_counter()
            check(fn(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, 100, C.byref(info), C.byref(it), C.byref(rr), None))
            check(L.psp_synchronize())
            if rep:
                best = min(best, time.perf_counter() - t)
        out["%s_%s_iters_per_s" % (form, name)] = round(100 / best, 2)
    del A, K, aop, kop
print(json.dumps(out))
