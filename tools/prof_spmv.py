#!/usr/bin/env python3
"""Launch the SpMV (and optionally PCG iterations) a few times -- target for rocprofv3."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="512,512,512")
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--pcg", type=int, default=0, help="also run this many Jacobi-PCG iterations")
ap.add_argument("--sss", action="store_true", help="symmetric-skyline operator (sss_spmv_w4 by default)")
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
A = dev.DeviceSSS.poisson(nx, ny, nz) if a.sss else dev.DeviceCSR.poisson(nx, ny, nz)
A.set_variant(a.variant)
n = A.shape[0]
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
for _ in range(a.reps):
    A.matvec_dev(x.ptr, y.ptr)
check(lib().psp_synchronize())
if a.pcg:
    import ctypes as C
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    b = dev.DeviceBuffer.from_host(np.ones(n))
    x.zero()
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(lib().psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, a.pcg, C.byref(info), C.byref(it), C.byref(rr), None))
    print("pcg", info.value, it.value, rr.value)
print("done", n, A.nnz, A.kernel_info())
