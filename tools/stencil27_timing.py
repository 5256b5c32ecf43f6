#!/usr/bin/env python3
"""27-point stencil (N^3 grid, all 26 neighbours + diagonal): csr_spmv_w4x against w3 / w2."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=160)
a = ap.parse_args()
N = a.n
n = N ** 3
k = np.arange(n, dtype=np.int64)
i, j, l = k % N, (k // N) % N, k // (N * N)
cols, rows = [], []
for dl in (-1, 0, 1):
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            ok = (i + di >= 0) & (i + di < N) & (j + dj >= 0) & (j + dj < N) & (l + dl >= 0) & (l + dl < N)
            rows.append(k[ok])
            cols.append(k[ok] + di + N * (dj + N * dl))
rows, cols = np.concatenate(rows), np.concatenate(cols)
order = np.lexsort((cols, rows))
rows, cols = rows[order], cols[order]
ind = np.zeros(n + 1, dtype=np.int32)
np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
val = np.where(rows == cols, 26.0, -1.0)
A = dev.DeviceCSR.from_arrays((n, n), ind, cols.astype(np.int32), val)
nnz = len(val)
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
ref = None
for name, variant in (("default", -1), ("w3", (1 << 20) + 16578), ("w2", 16578)):
    A.set_variant(variant)
    kern, info = A.kernel_info()
    f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
    y.zero()
    time_launches(f, 3)
    yh = y.download()
    if ref is None:
        ref = yh
    assert np.array_equal(ref, yh)
    t = min(time_launches(f, 20) for _ in range(3))
    print(json.dumps({"N": N, "n": n, "nnz": nnz, "variant": name, "kernel": kern, "info": info, "ms": t,
                      "GBps_csr_model": (12 * nnz + 20 * n + 4) / t / 1e6}), flush=True)
