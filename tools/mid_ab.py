#!/usr/bin/env python3
"""Single-kernel PCG loop for mid-size systems (psp_mid.hip) against the launch-per-phase loop, in ONE process on the same
operator and vectors, alternated (PSP_MID_MIN is read per solve): microseconds per iteration and the bits of x.
Start with PSP_TUNING=1.  Usage: mid_ab.py [pcg|minres] [nx,ny,nz ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PSP_TUNING") == "1", "start with PSP_TUNING=1"
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
check = _capi.check


def main():
    argv = sys.argv[1:]
    solver = argv.pop(0) if argv and argv[0] in ("pcg", "minres") else "pcg"
    fn = L.psp_pcg_dev if solver == "pcg" else L.psp_minres_dev
    grids = argv or ["1024,1024,0", "724,724,0", "600,600,0", "512,512,0", "80,80,80", "64,64,64"]
    for g in grids:
        nine = g.startswith("9:")  # "9:nx,ny,0": a 9-point operator with random symmetric couplings instead of the 5-point one
        grid = tuple(int(t) for t in g[2 if nine else 0:].split(","))
        if nine:
            import scipy.sparse as sp
            nx, ny = grid[0], grid[1]
            nn = nx * ny
            rg = np.random.default_rng(1)
            ix = np.arange(nn) % nx
            diags, offs = [], []
            for o, ok in ((1, ix[:-1] < nx - 1), (nx - 1, ix[:nn - nx + 1] > 0), (nx, np.ones(nn - nx, bool)), (nx + 1, ix[:nn - nx - 1] < nx - 1)):
                e = -(0.1 + rg.random(nn - o)) * ok
                diags += [e, e]
                offs += [o, -o]
            S = sp.diags(diags, offs, shape=(nn, nn), format="csr")
            S = (S + sp.diags(-np.asarray(S.sum(axis=1)).ravel() + 1e-5)).tocsr()  # nearly singular: thousands of iterations
            S.eliminate_zeros()
            S.sort_indices()
            A = dev.DeviceCSR.from_arrays(S.shape, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
        else:
            A = dev.DeviceCSR.poisson(*grid)
        n = A.shape[0]
        K = dev.DeviceJacobi(A)
        aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
        bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
        xb.upload(np.ones(n))
        A.matvec_dev(xb.ptr, bb.ptr)
        check(L.psp_synchronize())
        # two truncated solves per measurement: the difference cancels the set-up (||b||, r = b - A x, allocations, the
        # cooperative launch itself); both counts lie well before these systems stagnate
        k1, k2 = (50, 250) if grid[2] else ((100, 1100) if n > (1 << 18) else (40, 240))
        rec = {"mid": [], "phase": []}
        xs = {}
        for rnd in range(3):
            for mode in ("phase", "mid"):
                os.environ["PSP_MID_MIN"] = "1" if mode == "mid" else str(1 << 30)
                ts = {}
                for kk in (k1, k1, k2):
                    xb.zero()
                    info, it, rr = C.c_int(), C.c_int(), C.c_double()
                    check(L.psp_synchronize())
                    t = time.perf_counter()
                    check(fn(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
                    check(L.psp_synchronize())
                    ts[kk] = time.perf_counter() - t
                    assert it.value == (kk + 1 if solver == "pcg" else kk), (it.value, info.value)
                rec[mode].append((ts[k2] - ts[k1]) / (k2 - k1) * 1e6)
                xs[mode] = ((info.value, it.value, rr.value), xb.download())
        s, f = C.c_longlong(), C.c_longlong()
        L.psp_debug_mid_count(C.byref(s), C.byref(f))
        out = {"n": n, "us_per_iter_single_kernel": min(rec["mid"]), "us_per_iter_launch_per_phase": min(rec["phase"]),
               "speedup": min(rec["phase"]) / min(rec["mid"]),
               "same_bits": bool(xs["mid"][0] == xs["phase"][0] and np.array_equal(xs["mid"][1], xs["phase"][1])),
               "result": list(xs["mid"][0]), "mid_solves_so_far": s.value, "fallbacks": f.value, "all_us": rec}
        print("x".join(str(v) for v in grid if v), solver, json.dumps(out), flush=True)
        del aop, kop, K
        A.close()
        bb.free()
        xb.free()


if __name__ == "__main__":
    main()
