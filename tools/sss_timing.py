#!/usr/bin/env python3
"""y = S x for the symmetric-skyline Poisson operator: sss_spmv_w4 (lower triangle only) against the
csr kernels on the mirrored full matrix, one process, interleaved."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="512,512,512")
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
S = dev.DeviceSSS.poisson(nx, ny, nz)
n = S.n
nnz_lower = S.nnz - n
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
B = (1 << 22) + (1 << 20) + 194
variants = {"default": -1, "s32_f0": B + (32 << 8), "s32_ntlow": B + (32 << 8) + (1 << 23),
            "s32_ntshift": B + (32 << 8) + (2 << 23), "s32_both": B + (32 << 8) + (3 << 23),
            "s0": B, "s16": B + (16 << 8), "s64": B + (64 << 8), "s128": B + (128 << 8),
            "mirror_w3": (1 << 20) + 16578, "mirror_w2": 16578}
times = {k: [] for k in variants}
names = {}
ref = None
for k, v in variants.items():
    S.set_variant(v)
    names[k] = S.kernel_info()[0]
    y.zero()
    time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 2)
    yh = y.download()
    if ref is None:
        ref = yh
    assert np.array_equal(yh, ref)
for _ in range(a.rounds):
    for k, v in variants.items():
        S.set_variant(v)
        time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 1)
        times[k].append(time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 10))
b_sss = 12 * nnz_lower + 28 * n + 4          # val + col per stored entry; diag, ind, x, y per row
b_csr = 12 * (2 * nnz_lower + n) + 20 * n + 4  # the mirrored full matrix in the CSR model
for k in variants:
    t = float(np.median(times[k]))
    print(json.dumps({"variant": k, "kernel": names[k], "ms": t, "GBps_sss_model": b_sss / t / 1e6,
                      "GBps_csr_model": b_csr / t / 1e6}), flush=True)
