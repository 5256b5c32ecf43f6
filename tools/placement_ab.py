#!/usr/bin/env python3
"""Placement of library-owned vectors, A/B inside one process (psp_place.hip; VERDICT r4 'Next' #3).

One JSON line: at GRID (default 512^3, csr_mat)
  * y = A x on the process' FIRST allocations of x and y (what bench.py's `value` times: the caller owns them),
  * the same launch on the pair psp_place_operands draws for the handle (report: candidates, best / worst per role),
  * Jacobi-PCG and Jacobi-MINRES iterations/s with the solvers' work vectors drawn (placement on, the default) and not
    (psp_set_placement(0); the scratch pool is emptied between the legs so that each leg allocates afresh), alternated,
    with the bits of x compared.
Run it in several fresh processes (tools/make_profiles_r5.sh): the level of a process' first allocations is luck."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
check = _capi.check


def sync():
    check(L.psp_synchronize())


def time_spmv(A, xp, yp, reps=20):
    for _ in range(3):
        A.matvec_dev(xp, yp)
    e0, e1 = C.c_void_p(), C.c_void_p()
    check(L.psp_event_create(C.byref(e0)))
    check(L.psp_event_create(C.byref(e1)))
    check(L.psp_event_record(e0))
    for _ in range(reps):
        A.matvec_dev(xp, yp)
    check(L.psp_event_record(e1))
    ms = C.c_float()
    check(L.psp_event_elapsed_ms(e0, e1, C.byref(ms)))
    L.psp_event_destroy(e0)
    L.psp_event_destroy(e1)
    return ms.value / reps


def solve(fn, aop, kop, n, bb, xb, iters):
    xb.zero()
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    sync()
    t = time.perf_counter()
    check(fn(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
    sync()
    return time.perf_counter() - t, (info.value, it.value, rr.value)


def main():
    grid = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(","))
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    out = {"grid": list(grid), "n": n, "kernel": A.kernel_info()[0], "iters": iters}
    x0, y0 = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    x0.zero()
    y0.zero()
    out["spmv_first_allocation_ms"] = time_spmv(A, x0.ptr, y0.ptr)
    yp, xp = C.c_void_p(), C.c_void_p()
    rep = (C.c_double * 6)()
    check(L.psp_place_operands(A._h, C.byref(yp), C.byref(xp), rep))
    out["spmv_placed_ms"] = time_spmv(A, xp.value, yp.value)
    out["draw"] = {"candidates": int(rep[0]), "y_role_best_ms": rep[1], "y_role_worst_ms": rep[2],
                   "x_role_best_ms": rep[3], "x_role_worst_ms": rep[4], "draw_ms": rep[5]}
    out["spmv_first_allocation_again_ms"] = time_spmv(A, x0.ptr, y0.ptr)
    L.psp_free(yp)
    L.psp_free(xp)
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    ones = np.ones(1 << 24)
    for k in range(0, n, ones.size):
        check(L.psp_memcpy_h2d(xb.ptr + 8 * k, ones.ctypes.data, 8 * min(ones.size, n - k)))
    A.matvec_dev(xb.ptr, bb.ptr)
    sync()
    for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
        rec = {"placed_it_per_s": [], "unplaced_it_per_s": []}
        xs = {}
        for rnd in range(2):
            for on in (0, 1):
                check(L.psp_trim())  # each leg allocates its work vectors afresh
                check(L.psp_set_placement(on))
                solve(fn, aop, kop, n, bb, xb, 2)  # warm-up: pays the draw when placement is on
                dt, res = solve(fn, aop, kop, n, bb, xb, iters)
                rec["placed_it_per_s" if on else "unplaced_it_per_s"].append(iters / dt)
                xs[on] = (res, xb.download())
        rec["same_bits"] = bool(xs[0][0] == xs[1][0] and np.array_equal(xs[0][1], xs[1][1]))
        rec["result"] = list(xs[1][0])
        rec["gain_pct"] = 100.0 * (max(rec["placed_it_per_s"]) / max(rec["unplaced_it_per_s"]) - 1.0)
        out[name] = rec
    en, draws, ms = C.c_int(), C.c_longlong(), C.c_double()
    check(L.psp_placement_info(C.byref(en), C.byref(draws), C.byref(ms)))
    out["draws"] = int(draws.value)
    out["draw_ms_total"] = ms.value
    print(json.dumps(out))


if __name__ == "__main__":
    main()
