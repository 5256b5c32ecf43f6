#!/usr/bin/env python3
"""MINRES iterations/s (fixed iteration count, tol = 0) on the Poisson operator, csr_mat and sss_mat forms."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="512,512,512")
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
L = lib()
for form in ("csr", "sss"):
    A = dev.DeviceCSR.poisson(nx, ny, nz) if form == "csr" else dev.DeviceSSS.poisson(nx, ny, nz)
    n = A.shape[0]
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    x = ones
    for pre in (False, True):
        K = dev.DeviceJacobi(A) if pre else None
        aop = dev._Op(A, "matvec")
        kop = dev._Op(K, "precon") if pre else None
        for kk in (2, a.iters):
            x.zero()
            info, it, rr = C.c_int(), C.c_int(), C.c_double()
            check(L.psp_synchronize())
            t = time.perf_counter()
            check(L.psp_minres_dev(aop._h, kop._h if kop else None, n, x.ptr, b.ptr, 0.0, kk, C.byref(info),
                                   C.byref(it), C.byref(rr), None))
            check(L.psp_synchronize())
            dt = time.perf_counter() - t
        print(json.dumps({"grid": [nx, ny, nz], "form": form, "jacobi": pre, "kernel": A.kernel_info()[0],
                          "minres_iters_per_s": a.iters / dt, "check": [info.value, it.value, rr.value]}), flush=True)
    del A, b, ones, x
