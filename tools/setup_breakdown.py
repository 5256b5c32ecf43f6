#!/usr/bin/env python3
"""tools/setup_breakdown.py [standin] -- where the time of BASELINE configs[4] goes before the first iteration (round 6,
VERDICT r5 #4a): upload of the sss arrays, the first product (tables, renumbering), the converged Jacobi-MINRES solve with
x back on the host, and the same solve again -- wall times from this script, the library's own stage marks
(PSP_TUNING=1 PSP_SETUP_TRACE=1, stderr) between them.  Run once with PSP_RCM_COOP=0 for the launch-per-level numbering.

    PSP_TUNING=1 PSP_SETUP_TRACE=1 python tools/setup_breakdown.py fem512"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools import standins  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fem512"
    if kind == "logspaced":
        n, ind, col, val, diag = standins.logspaced_sss_arrays(923136)
    else:
        n, ind, col, val, diag = standins.fem_sss_arrays(68, 68, 67, int(kind[3:] or 32))
    L = lib()
    check(L.psp_synchronize())  # the context exists before anything is timed
    warm = dev.DeviceBuffer(1 << 20)
    warm.free()
    # ... and so do the code objects of the translation units this flow launches from (a process pays ~15 ms per unit on
    # its first launch from it, whatever the matrix): the same flow once on a 3 000-row stand-in
    wn, wi, wc, wv, wd = standins.fem_sss_arrays(10, 10, 10, 64)
    Sw = dev.DeviceSSS.from_arrays(wn, wi, wc, wv, wd)
    Kw = dev.DeviceJacobi(Sw)
    dev.minres(Sw, np.ones(wn), np.zeros(wn), 1e-10, 50, Kw)
    Kw.close()
    Sw.close()
    big = np.ones(1 << 22)
    dev.DeviceBuffer.from_host(big).free()
    print("[script] ---- warm-up done", file=sys.stderr, flush=True)
    out0 = {"standin": kind, "n": n, "nnz_lower": int(val.shape[0]), "rcm_coop": os.environ.get("PSP_RCM_COOP", "1")}
    rng = np.random.default_rng(7)
    b = np.zeros(n)
    b[0] = 1.0
    b += 1e-3 * rng.standard_normal(n)
    passes = []
    for pno in (1, 2):  # pass 2: the same flow on a second handle -- what a process pays per matrix once it is warm
        print("[script] ==== pass %d" % pno, file=sys.stderr, flush=True)
        out = dict(out0, **one_pass(L, n, ind, col, val, diag, b, rng))
        out["pass"] = pno
        passes.append(out)
        print(json.dumps(out))


def one_pass(L, n, ind, col, val, diag, b, rng):
    if True:
        out = {}
        t_all = time.perf_counter()
        t0 = time.perf_counter()
        S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
        check(L.psp_synchronize())
        out["upload_ms"] = (time.perf_counter() - t0) * 1e3
        print("[script] ---- upload done", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        K = dev.DeviceJacobi(S)
        out["jacobi_ms"] = (time.perf_counter() - t0) * 1e3
        x = np.zeros(n)
        t0 = time.perf_counter()
        res = dev.minres(S, b, x, 1e-10, 500, K)
        out["first_solve_ms"] = (time.perf_counter() - t0) * 1e3
        out["end_to_end_ms"] = (time.perf_counter() - t_all) * 1e3
        print("[script] ---- first solve done", file=sys.stderr, flush=True)
        out["minres"] = list(res[:3])
        out["kernel"] = S.kernel_info()[0]
        x2 = np.zeros(n)
        t0 = time.perf_counter()
        res2 = dev.minres(S, b, x2, 1e-10, 500, K)
        out["second_solve_ms"] = (time.perf_counter() - t0) * 1e3
        out["same_bits"] = bool(res2[:3] == res[:3] and np.array_equal(x, x2))
        import hashlib
        out["x_sha"] = hashlib.sha256(x.tobytes()).hexdigest()[:16]
        xb, yb = dev.DeviceBuffer.from_host(rng.standard_normal(n)), dev.DeviceBuffer(n)
        S.matvec_dev(xb.ptr, yb.ptr)
        check(L.psp_synchronize())
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(50):
                S.matvec_dev(xb.ptr, yb.ptr)
            check(L.psp_synchronize())
            ts.append((time.perf_counter() - t0) / 50)
        out["product_us"] = min(ts) * 1e6
        out["y_sha"] = hashlib.sha256(yb.download().tobytes()).hexdigest()[:16]
        K.close(); S.close(); xb.free(); yb.free()
        return out


if __name__ == "__main__":
    main()
