#!/usr/bin/env python3
"""A second stand-in for BASELINE config C5 (Emilia_923: unstructured FEM, 3 dof per node, ~44 nnz/row),
closer to it than the log-spaced K1 pattern: a 68 x 68 x 67 node grid, 3 unknowns per node, every node
coupled to itself, its 6 face and 8 corner neighbours (45 nnz/row, n = 929 424), node numbers shuffled
inside groups of --shuffle consecutive nodes to mimic an unstructured ordering.  sss_mat product
through each applicable kernel + Jacobi-MINRES."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402


def fem_sss(gx, gy, gz, shuffle, seed=0, wild=0):
    from pysparse_amd.tools.standins import fem_sss_arrays
    return fem_sss_arrays(gx, gy, gz, shuffle, seed, wild)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shuffle", type=int, default=32)
    ap.add_argument("--grid", default="68,68,67")
    ap.add_argument("--variants", default="", help="extra kernel variants to time, comma separated")
    ap.add_argument("--wild", type=int, default=0, help="rows with 60 extra couplings to unknowns anywhere below them")
    a = ap.parse_args()
    gx, gy, gz = (int(t) for t in a.grid.split(","))
    L = lib()
    n, ind, col, val, diag = fem_sss(gx, gy, gz, a.shuffle, 0, a.wild)
    nl = len(col)
    S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
    S.prepare(1 << 30)  # round 6: the renumbered copy at first use (the cost rule would start on csr_spmv_w5)
    x = dev.DeviceBuffer.from_host(np.random.default_rng(1).standard_normal(n))
    y = dev.DeviceBuffer(n)
    res = {"n": n, "nnz_lower": nl, "nnz_per_row_full": (2 * nl + n) / n, "shuffle": a.shuffle, "wild": a.wild,
           "w3_outliers": os.environ.get("PSP_SPMV_W3_OUTLIERS", "1"),
           "w3_nb_cap": os.environ.get("PSP_SPMV_W3_NB", "64")}
    ref = None
    extra = [("v%s" % v, int(v)) for v in a.variants.split(",") if v]
    for name, variant in [("default", -1), ("w5", 5259458 + (1 << 27)), ("w2", 16578)] + extra:
        S.set_variant(variant)
        kern, info = S.kernel_info()
        f = lambda: S.matvec_dev(x.ptr, y.ptr)  # noqa: E731
        y.zero()
        time_launches(f, 5)
        yh = y.download()
        if ref is None:
            ref = yh
        assert np.array_equal(ref, yh)
        t = min(time_launches(f, 30) for _ in range(3))
        res[name] = {"kernel": kern, "info": info, "spmv_ms": t,
                     "GBps_sss_model": (12 * nl + 28 * n + 4) / t / 1e6,
                     "GBps_csr_model": (12 * (2 * nl + n) + 20 * n + 4) / t / 1e6}
    S.set_variant(-1)
    b = np.zeros(n)
    b[0] = 1.0
    xh = np.zeros(n)
    t0 = time.perf_counter()
    r = dev.minres(S, b, xh, 1e-10, 2000, dev.DeviceJacobi(S))
    res["minres"] = {"info": r[0], "iter": r[1], "relres": r[2], "seconds_incl_pcie": time.perf_counter() - t0}
    # device-resident: fixed 200 MINRES iterations (tol 0) -> iterations/s, effective CSR-model GB/s of the product
    import ctypes as C
    K = dev.DeviceJacobi(S)
    aop, kop = dev._Op(S, "matvec"), dev._Op(K, "precon")
    bd = dev.DeviceBuffer.from_host(b)
    xd = dev.DeviceBuffer(n)
    for kk in (5, 200):
        xd.zero()
        info, it, rr = C.c_int(), C.c_int(), C.c_double()
        check(L.psp_synchronize())
        t0 = time.perf_counter()
        check(L.psp_minres_dev(aop._h, kop._h, n, xd.ptr, bd.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
        check(L.psp_synchronize())
        dt = time.perf_counter() - t0
    res["minres_200"] = {"iters_per_s": 200 / dt, "us_per_iter": dt / 200 * 1e6,
                         "permuted_space": os.environ.get("PSP_SOLVE_PERMUTED", "1") != "0"}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
