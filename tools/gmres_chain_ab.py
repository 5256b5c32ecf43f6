#!/usr/bin/env python3
"""gmres(m): the modified Gram-Schmidt chain of a column enqueued at once (step k's axpy takes h[k] from the device, ONE
scalar read-back per inner iteration; round 4) against one read-back per step (PSP_GMRES_CHAIN=0).  In-process, alternated
on the same system, Jacobi; x must be the same bits."""
import json
import os
import sys
import time

import numpy as np

os.environ["PSP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402

grids = [tuple(int(t) for t in g.split(",")) for g in sys.argv[1:]] or [(100, 100, 0), (1024, 1024, 0), (4096, 4096, 0), (256, 256, 256)]
for grid in grids:
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    row = {"grid": grid}
    sums = {}
    for dim in (20, 5):
        for short, long_ in ((40, 240),):
            best = {"1": {}, "0": {}}
            for rnd in range(3):
                for mode in ("1", "0"):
                    os.environ["PSP_GMRES_CHAIN"] = mode
                    for k in (short, long_):
                        x = np.zeros(n)
                        t = time.perf_counter()
                        r = dev.gmres(A, b, x, 0.0, k, K, dim)
                        dt = time.perf_counter() - t
                        best[mode][k] = min(best[mode].get(k, 1e9), dt)
                        if k == long_:
                            sums[(dim, mode)] = (r[0], r[1], r[2], float(np.abs(x).sum()))
            for mode in ("1", "0"):
                row["gmres(%d) %s us/it" % (dim, "chained" if mode == "1" else "per-step")] = round(
                    (best[mode][long_] - best[mode][short]) / (long_ - short) * 1e6, 2)
            row["gmres(%d) same_bits" % dim] = sums[(dim, "1")] == sums[(dim, "0")]
    print(json.dumps(row), flush=True)
