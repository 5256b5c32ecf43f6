#!/usr/bin/env python3
"""precon.ssor at scale: schedule build time, one application, and Jacobi- vs SSOR-PCG to 1e-8."""
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
ap.add_argument("--grid", default="256,256,256")
ap.add_argument("--tol", type=float, default=1e-8)
ap.add_argument("--no-pcg", action="store_true")
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
L = lib()
S = dev.DeviceSSS.poisson(nx, ny, nz)
n = S.n
check(L.psp_synchronize())
t = time.perf_counter()
K = dev.DeviceSSOR(S, 1.0, 1)
check(L.psp_synchronize())
t_build = time.perf_counter() - t
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
K.precon_dev(x.ptr, y.ptr)
check(L.psp_synchronize())
t = time.perf_counter()
for _ in range(3):
    K.precon_dev(x.ptr, y.ptr)
check(L.psp_synchronize())
t_apply = (time.perf_counter() - t) / 3
ones = dev.DeviceBuffer.from_host(np.ones(n))
b = dev.DeviceBuffer(n)
S.matvec_dev(ones.ptr, b.ptr)
out = {"grid": [nx, ny, nz], "n": n, "levels": K.levels, "lds_runs": K.lds_runs, "schedule_build_s": t_build,
       "apply_ms": t_apply * 1e3}
for name, P in (() if a.no_pcg else (("jacobi", dev.DeviceJacobi(S)), ("ssor", K))):
    aop, kop = dev._Op(S, "matvec"), dev._Op(P, "precon")
    x.zero()
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(L.psp_synchronize())
    t = time.perf_counter()
    check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, a.tol, 20000, C.byref(info), C.byref(it), C.byref(rr), None))
    check(L.psp_synchronize())
    dt = time.perf_counter() - t
    err = float(np.abs(x.download() - 1.0).max())
    out["pcg_" + name] = {"info": info.value, "iter": it.value, "relres": rr.value, "seconds": dt, "err_inf": err}
print(json.dumps(out))
