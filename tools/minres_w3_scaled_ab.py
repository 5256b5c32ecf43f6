#!/usr/bin/env python3
"""Jacobi-MINRES with a csr_spmv_w3 operator: v = y / beta formed inside the product (x is divided while it is staged in LDS;
round 4) against the separate scale pass (PSP_MINRES_SCALED=0: read once per process, so run this twice).  Operators: the
7-pt Poisson matrix forced to csr_spmv_w3 (what an arbitrary banded csr_mat gets) and the FEM-like stand-in of configs[4]
(renumbered copy).  Prints microseconds per iteration and a checksum of x (the two runs must print the same one)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ["PSP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools import standins  # noqa: E402

L = lib()
cases = []
for grid in ((512, 512, 512), (256, 256, 256), (2048, 2048, 0)):
    A = dev.DeviceCSR.poisson(*grid)
    A.set_variant((1 << 20) + 128 + 64 + 2 + (64 << 8))
    cases.append(("poisson %s as csr_spmv_w3" % "x".join(str(g) for g in grid if g), A))
n0, ind, col, val, diag = standins.fem_sss_arrays(shuffle=512)
cases.append(("fem stand-in (shuffle 512), sss_mat", dev.DeviceSSS.from_arrays(n0, ind, col, val, diag)))
for name, A in cases:
    n = A.shape[0] if hasattr(A, "shape") and not isinstance(A, dev.DeviceSSS) else A.n
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    x = dev.DeviceBuffer(n)
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    iters = 60 if n > (1 << 25) else 300
    best = 1e9
    for rep in range(4):
        x.zero()
        check(L.psp_synchronize())
        t = time.perf_counter()
        check(L.psp_minres_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
        check(L.psp_synchronize())
        if rep:
            best = min(best, time.perf_counter() - t)
    print(json.dumps({"case": name, "kernel": A.kernel_info()[0], "scaled": os.environ.get("PSP_MINRES_SCALED", "1"),
                      "us_per_iter": round(best / iters * 1e6, 2), "iter": it.value, "relres": rr.value,
                      "x_checksum": float(np.abs(x.download()).sum())}), flush=True)
    del A, K, aop, kop, b, x, ones
