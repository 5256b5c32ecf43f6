#!/usr/bin/env python3
"""Time to first result on the irregular stand-in (config C5 shape): handle creation (upload), first product
(builds the kernel tables; RCM renumbering where it applies), second product, first / second MINRES solve."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools.standins import fem_sss_arrays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shuffle", type=int, default=512)
ap.add_argument("--unsym", action="store_true", help="the stand-in as a csr_mat with every 7th upper entry dropped "
                                                     "(unsymmetric pattern: A + A^T is formed before the numbering)")
ap.add_argument("--others", action="store_true", help="also: generated 512^3 csr / sss, 256^3 csr from host arrays, "
                                                      "the log-spaced stand-in")
a = ap.parse_args()
L = lib()
check(L.psp_set_device(0))
xw = dev.DeviceBuffer(16)  # context creation outside the timings


def first_two(A, n, label, t_create):
    xb, yb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    xb.zero()
    out = {"case": label, "create_s": t_create}
    for k in ("first_matvec_dev_s", "second_matvec_dev_s"):
        check(L.psp_synchronize())
        t = time.perf_counter()
        A.matvec_dev(xb.ptr, yb.ptr)
        check(L.psp_synchronize())
        out[k] = time.perf_counter() - t
    out["kernel"] = A.kernel_info()[0]
    xb.free()
    yb.free()
    print(json.dumps(out), flush=True)


if a.others:
    from pysparse_amd.tools.standins import logspaced_sss_arrays
    t = time.perf_counter()
    A = dev.DeviceCSR.poisson(512, 512, 512)
    check(L.psp_synchronize())
    first_two(A, A.shape[0], "csr poisson 512^3 (generated on the device)", time.perf_counter() - t)
    A.close()
    t = time.perf_counter()
    S = dev.DeviceSSS.poisson(512, 512, 512)
    check(L.psp_synchronize())
    first_two(S, S.shape[0], "sss poisson 512^3 (generated on the device)", time.perf_counter() - t)
    S.close()
    check(L.psp_trim())
    from oracle import oracle as O  # host arrays of a 256^3 operator (tool only)
    H = O.poisson_csr(256, 256, 256)
    t = time.perf_counter()
    A = dev.DeviceCSR.from_arrays(H.shape, H.ind, H.col, H.val)
    check(L.psp_synchronize())
    first_two(A, A.shape[0], "csr 256^3 from host arrays (1.17e8 entries)", time.perf_counter() - t)
    A.close()
    del H
    n, ind, col, val, diag = logspaced_sss_arrays()
    t = time.perf_counter()
    S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
    check(L.psp_synchronize())
    first_two(S, n, "sss log-spaced stand-in (n = %d, %d lower entries)" % (n, len(col)), time.perf_counter() - t)
    S.close()
n, ind, col, val, diag = fem_sss_arrays(68, 68, 67, a.shuffle, 0)
res = {"n": n, "nnz_lower": len(col), "shuffle": a.shuffle}
if a.unsym:
    from pysparse_amd.distributed import sss_rows_expanded
    f_ind, f_col, f_val = sss_rows_expanded(n, ind, col, val, diag)
    rows = np.repeat(np.arange(n), np.diff(f_ind))
    upper = np.nonzero(f_col > rows)[0]
    keep = np.ones(f_col.size, dtype=bool)
    keep[upper[::7]] = False
    f_col, f_val, rows = f_col[keep], f_val[keep], rows[keep]
    f_ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=f_ind[1:])
    res["unsymmetric_csr_nnz"] = int(f_col.size)
t = time.perf_counter()
S = dev.DeviceCSR.from_arrays((n, n), f_ind, f_col, f_val) if a.unsym else dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
check(L.psp_synchronize())
res["create_s"] = time.perf_counter() - t
x = np.random.default_rng(1).standard_normal(n)
y = np.empty(n)
for k in ("first_matvec_s", "second_matvec_s"):
    t = time.perf_counter()
    S.matvec(x, y)
    res[k] = time.perf_counter() - t
res["kernel"] = S.kernel_info()[0]
if a.unsym:
    S.renumbering()
    res["numbered_on"] = S.renumbered_on
    print(json.dumps(res), flush=True)
    sys.exit(0)
b = np.zeros(n)
b[0] = 1.0
K = dev.DeviceJacobi(S)
for k in ("first_minres_s", "second_minres_s"):
    xh = np.zeros(n)
    t = time.perf_counter()
    r = dev.minres(S, b, xh, 1e-10, 2000, K)
    res[k] = time.perf_counter() - t
res["minres"] = list(r)
print(json.dumps(res), flush=True)
