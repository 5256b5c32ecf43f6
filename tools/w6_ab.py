#!/usr/bin/env python3
"""csr_spmv_w6 against csr_spmv_w2 (the other kernel that streams the csr_mat's own col / val arrays) and csr_spmv_w3, in ONE
process on the same operator and the same x / y, alternated (VERDICT r4 'Next' #6).  Prints per grid: ms per launch,
fraction of the 8 TB/s peak in CSR-model bytes (12 nnz + 20 n + 4: what w2 and w6 stream), same bits."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
check = _capi.check
W2 = 128 + 64 + 2 + (64 << 8)
W6 = W2 + (1 << 23)
W3 = W2 + (1 << 20)


def time_spmv(A, xp, yp, reps):
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


def main():
    grids = sys.argv[1:] or ["512,512,512", "4096,4096,0", "256,256,256"]
    for g in grids:
        grid = tuple(int(t) for t in g.split(","))
        A = dev.DeviceCSR.poisson(*grid)
        n, nnz = A.shape[0], A.nnz
        x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
        y = dev.DeviceBuffer(n)
        model = 12 * nnz + 20 * n + 4
        reps = max(10, min(200, int(2e10 / model)))
        out = {"grid": list(grid), "n": n, "nnz": nnz}
        ref = None
        for rnd in range(2):
            for name, var in (("w2", W2), ("w6", W6), ("w3", W3)):
                A.set_variant(var)
                kn, info = A.kernel_info()
                ms = time_spmv(A, x.ptr, y.ptr, reps)
                yy = y.download()
                if ref is None:
                    ref = yy
                rec = out.setdefault(name, {"kernel": kn, "info": info, "ms": [], "same_bits": True})
                rec["ms"].append(ms)
                rec["same_bits"] = bool(rec["same_bits"] and np.array_equal(yy, ref))
        for name in ("w2", "w6", "w3"):
            r = out[name]
            r["best_ms"] = min(r["ms"])
            r["csr_model_frac_of_peak"] = model / (r["best_ms"] * 1e-3) / 8e12
        print(json.dumps(out), flush=True)
        A.close()
        x.free()
        y.free()


if __name__ == "__main__":
    main()
