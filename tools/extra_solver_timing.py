#!/usr/bin/env python3
"""cgs / bicgstab / qmrs / gmres(20) at scale: iterations per second on the 7-pt Poisson operator
with Jacobi, from two runs of different fixed length (tol = 0) so that the host <-> device copies of b and x and the
set-up drop out.  pcg and minres through the same host-pointer entry points beside them."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="512,512,512")
ap.add_argument("--short", type=int, default=5)
ap.add_argument("--long", type=int, default=45)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--only", default="", help="comma separated solver names")
a = ap.parse_args()
nx, ny, nz = (int(t) for t in a.grid.split(","))
A = dev.DeviceCSR.poisson(nx, ny, nz)
n = A.shape[0]
K = dev.DeviceJacobi(A)
b = np.empty(n)
A.matvec(np.ones(n), b)
out = {"grid": [nx, ny, nz], "n": n, "kernel": A.kernel_info()[0]}
for name, fn in (("pcg", dev.pcg), ("minres", dev.minres), ("cgs", dev.cgs), ("bicgstab", dev.bicgstab),
                 ("qmrs", dev.qmrs), ("gmres20", dev.gmres)):
    if a.only and name not in a.only.split(","):
        continue
    ts, res = {}, None
    for k in (a.short, a.long) * a.reps:
        x = np.zeros(n)
        t = time.perf_counter()
        res = fn(A, b, x, 0.0, k, K)
        ts.setdefault(k, []).append(time.perf_counter() - t)
    dt = (min(ts[a.long]) - min(ts[a.short])) / (a.long - a.short)
    out[name] = {"ms_per_iter": dt * 1e3, "iters_per_s": 1.0 / dt, "last": [res[0], res[1], res[2]]}
    print(name, out[name], flush=True)
print(json.dumps(out), flush=True)
