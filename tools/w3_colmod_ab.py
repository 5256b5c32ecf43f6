#!/usr/bin/env python3
"""What would compressing the column stream of csr_spmv_w3 buy?  (VERDICT r3 "Next" #7b: one 16-bit column per dense
r x r block of a 3-dof FEM matrix, 8.7 instead of 10 bytes per entry.)  Measured without building it: PSP_W3_COLMOD=64
makes chunk c read the 16-bit columns of chunk c % 64, so the column stream (2 of the kernel's 10 bytes per entry) comes
out of L2 instead of HBM while every other access, the LDS gathers and the arithmetic stay what they are (wrong results,
timing only).  The time this saves is an UPPER bound for any column compression; block columns save two thirds of it.
One process, the switch alternated between timed batches on the same buffers."""
import json
import os
import sys

import numpy as np

os.environ["PSP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools import standins  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

L = lib()
W3 = (1 << 20) + 128 + 64 + 2 + (64 << 8)


def ab(name, A, n, nnz_lower=None):
    x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    y = dev.DeviceBuffer(n)
    f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
    os.environ["PSP_W3_COLMOD"] = "0"
    time_launches(f, 10)
    kern = A.kernel_info()[0]
    best = {"0": 1e9, "64": 1e9}
    for _ in range(5):
        for m in ("0", "64"):
            os.environ["PSP_W3_COLMOD"] = m
            time_launches(f, 3)
            best[m] = min(best[m], time_launches(f, 30))
    os.environ["PSP_W3_COLMOD"] = "0"
    t0, t1 = best["0"], best["64"]
    out = {"case": name, "kernel": kern, "ms": round(t0, 5), "ms_columns_from_L2": round(t1, 5),
           "upper_bound_gain_pct": round(100 * (t0 / t1 - 1), 2),
           "block_columns_estimate_ms": round(t0 - (2.0 / 3.0) * (t0 - t1), 5)}
    if nnz_lower is not None:
        sss = 12 * nnz_lower + 28 * n + 4
        out["sss_model_TBps"] = round(sss / t0 / 1e9, 3)
        out["sss_model_TBps_block_columns_estimate"] = round(sss / out["block_columns_estimate_ms"] / 1e9, 3)
        out["sss_model_TBps_no_column_bytes_at_all"] = round(sss / t1 / 1e9, 3)
    print(json.dumps(out), flush=True)
    x.free()
    y.free()


for shuffle in (1, 32, 512):
    n, ind, col, val, diag = standins.fem_sss_arrays(68, 68, 67, shuffle)
    S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
    ab("fem stand-in, node ids shuffled in groups of %d" % shuffle, S, n, val.shape[0])
    S.close()
A = dev.DeviceCSR.poisson(512, 512, 512)
A.set_variant(W3)
ab("7-pt Poisson 512^3 through csr_spmv_w3", A, A.shape[0])
A.close()
