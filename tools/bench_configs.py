#!/usr/bin/env python3
"""Numbers for the BASELINE.json configs that are not the bench line (one GPU):
  C2  2-D Poisson 5-pt 4096^2: SpMV GB/s, Jacobi-PCG iterations/s
  C5  irregular symmetric sss_mat stand-in for Emilia_923 (no network: the K1 pattern of
      examples/tendigit.py scaled to n = 923 136, ~1.8e7 strict-lower entries): SpMV GB/s
      against the SSS and CSR byte models, Jacobi-MINRES to convergence."""
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

L = lib()
out = {}

# ---- C2
A = dev.DeviceCSR.poisson(4096, 4096)
n, nnz = A.shape[0], A.nnz
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 10)
t = min(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 50) for _ in range(3))
K = dev.DeviceJacobi(A)
aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
b = dev.DeviceBuffer(n)
ones = dev.DeviceBuffer.from_host(np.ones(n))
A.matvec_dev(ones.ptr, b.ptr)
check(L.psp_synchronize())
info, it, rr = C.c_int(), C.c_int(), C.c_double()
for k in (5, 400):
    x.zero()
    check(L.psp_synchronize())
    t0 = time.perf_counter()
    check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, k, C.byref(info), C.byref(it), C.byref(rr), None))
    check(L.psp_synchronize())
    tp = time.perf_counter() - t0
kern, kinfo = A.kernel_info()
own = 8 * kinfo["nb"] * ((n + 127) // 128 * 128) + 18 * n if kern == "csr_spmv_w4" else 12 * nnz + 20 * n + 4
out["C2_poisson2d_4096"] = {
    "n": n, "nnz": nnz, "kernel": kern, "spmv_ms": t,
    # what the kernel has to move (csr_spmv_w4: 8 B per offset slot + 2 B mask + x + y; x and y, 134 MB each, can
    # live in the 256 MB Infinity Cache at this size, so this is not a pure HBM rate) ...
    "spmv_bytes": own, "spmv_GBps": own / t / 1e6, "spmv_frac_of_hbm_peak": own / t / 1e6 / 8000.0,
    # ... and the same time priced in CSR-model bytes 12 nnz + 20 n + 4 (an equivalent, not a memory rate)
    "spmv_csr_model_equiv_GBps": (12 * nnz + 20 * n + 4) / t / 1e6,
    "pcg_iters_per_s": 400 / tp, "pcg_csr_model_equiv_GBps": (12 * nnz + 108 * n) * 400 / tp / 1e9}
del A, K, aop, kop

# ---- C5 stand-in
n = 923136
offs = []
d = 1
while d < n:
    offs.append(d)
    d *= 2
i = np.arange(n, dtype=np.int64)
cols = [i - o for o in reversed(offs)]  # ascending column within a row
mask = [c >= 0 for c in cols]
lens = np.sum(mask, axis=0).astype(np.int64)
ind = np.zeros(n + 1, dtype=np.int32)
np.cumsum(lens, out=ind[1:])
col = np.concatenate([c[:, None] for c in cols], axis=1)[np.stack(mask, axis=1)].astype(np.int32)
rng = np.random.default_rng(1)
val = np.ones(len(col))
diag = 40.0 + rng.random(n) * 1e3  # irregular, SPD (diagonally dominant)
S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
nl = len(col)
xs = dev.DeviceBuffer.from_host(rng.standard_normal(n))
ys = dev.DeviceBuffer(n)
f = lambda: check(L.psp_sss_matvec_dev(S._h, xs.ptr, ys.ptr))  # noqa: E731
time_launches(f, 10)
t = min(time_launches(f, 50) for _ in range(3))
bh = np.zeros(n)
bh[0] = 1.0
xh = np.zeros(n)
t0 = time.perf_counter()
res = dev.minres(S, bh, xh, 1e-12, 5000, dev.DeviceJacobi(S))
tm = time.perf_counter() - t0
out["C5_irregular_sss_standin"] = {
    "n": n, "nnz_lower": nl, "spmv_ms": t, "spmv_GBps_sss_model": (12 * nl + 28 * n + 4) / t / 1e6,
    "spmv_GBps_csr_model": (12 * (2 * nl + n) + 20 * n + 4) / t / 1e6,
    "minres": {"info": res[0], "iter": res[1], "relres": res[2], "seconds_incl_pcie": tm}}
print(json.dumps(out, indent=1))
