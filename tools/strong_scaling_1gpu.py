#!/usr/bin/env python3
"""The single-GPU leg of BASELINE.json's scaling target ("PCG-iter/s 1 -> 8 GPUs at 1024^3"): the whole
1024^3 7-point operator on ONE MI355X (index-free w4 layout, psp_csr_poisson_big), y = A x and
Jacobi-PCG iterations/s.  `bench.py --gpus 8` is the 8-GPU leg of the same problem."""
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
from tools.spmv_sweep import time_launches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="1024,1024,1024")
ap.add_argument("--pcg-iters", type=int, default=32)
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
L = lib()
A = dev.DeviceCSR.poisson_big(nx, ny, nz)
n, nnz = A.shape[0], A.nnz
x = dev.DeviceBuffer(n)
y = dev.DeviceBuffer(n)
chunk = np.random.default_rng(0).standard_normal(1 << 24)
for k in range(0, n, chunk.size):
    m = min(chunk.size, n - k)
    check(L.psp_memcpy_h2d(x.ptr + 8 * k, chunk.ctypes.data, 8 * m))
time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 3)
t_spmv = min(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 10) for _ in range(3))
ones = np.ones(1 << 24)
for k in range(0, n, ones.size):
    m = min(ones.size, n - k)
    check(L.psp_memcpy_h2d(x.ptr + 8 * k, ones.ctypes.data, 8 * m))
b = dev.DeviceBuffer(n)
A.matvec_dev(x.ptr, b.ptr)
K = dev.DeviceJacobi(A)
aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
for kk in (2, a.pcg_iters):
    x.zero()
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(L.psp_synchronize())
    t = time.perf_counter()
    check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
    check(L.psp_synchronize())
    t_pcg = time.perf_counter() - t
print(json.dumps({"grid": [nx, ny, nz], "n": n, "nnz": nnz, "kernel": A.kernel_info()[0], "n_gpus": 1,
                  "spmv_ms": t_spmv, "spmv_GBps_csr_model": (12 * nnz + 20 * n + 4) / t_spmv / 1e6,
                  "pcg_iters_per_s": a.pcg_iters / t_pcg, "pcg_check": [info.value, it.value, rr.value],
                  "device_bytes_matrix": A.device_bytes}))
