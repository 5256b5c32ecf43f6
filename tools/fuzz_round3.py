#!/usr/bin/env python3
"""One-off differential campaign for the two kernels round 3 added late (not part of the test suite):
  * precon.ssor through the LDS-exchange runs of narrow levels (ssor_run_kernel): random 2-D / 3-D grid operators with
    random extents, missing couplings, omega and step counts -- bit equality with the oracle's sequential sweeps;
  * csr_spmv_w4y: random offset-structured matrices with 33..64 distinct offsets -- bit equality of y = A x, the fused dot
    of the solvers (PCG / MINRES counts and iterates) where the matrix is SPD.
Prints one line per matrix; exits non-zero on the first mismatch."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--count", type=int, default=40)
ap.add_argument("--ssor-only", action="store_true")
ap.add_argument("--bricks", action="store_true", help="3-D grids with levels wider than a run only: the brick sweeps")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)


def grid_sss(nx, ny, nz, keep):
    P = O.poisson_sss(nx, ny, nz)
    sel = rng.random(P.val.size) < keep
    lens = np.add.reduceat(sel.astype(np.int64), P.ind[:-1].astype(np.int64)) * (np.diff(P.ind) > 0)
    ind = np.zeros(P.n + 1, dtype=np.int32)
    np.cumsum(lens, out=ind[1:])
    col = P.col[sel]
    return O.SSS(P.n, -(0.2 + rng.random(col.size)), 6.5 + rng.random(P.n), col, ind)


bad = 0
for t in range(a.count):
    if a.bricks:
        nx, ny, nz = (int(v) for v in rng.integers(70, 230, size=3))
    elif rng.random() < 0.5:
        nx, ny, nz = int(rng.integers(50, 2600)), int(rng.integers(3, 700)), 0
        if nx * ny > 2_000_000:
            ny = max(3, 2_000_000 // nx)
    else:
        nx, ny, nz = (int(v) for v in rng.integers(8, 200, size=3))
    keep = float(rng.choice([1.0, 1.0, 0.9995, 0.99, 0.85]))
    omega = float(rng.choice([1.0, 1.0, 1.3, 0.75]))
    steps = int(rng.choice([1, 1, 2, 3]))
    S = grid_sss(nx, ny, nz, keep)
    D = dev.DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    K = dev.DeviceSSOR(D, omega, steps)
    x = rng.standard_normal(S.n)
    y_ref, y = np.full(S.n, 0.25), np.full(S.n, 0.25)
    O.ssor_apply(S, x, y_ref, omega, steps)
    K.precon(x, y)
    ok = np.array_equal(y, y_ref)
    K.precon(x, y)  # the replayed graph
    ok = ok and np.array_equal(y, y_ref)
    print("ssor grid %s keep %.4f omega %.2f steps %d levels %s lds_runs %s bricks %d %s" % (
        (nx, ny, nz), keep, omega, steps, K.levels, K.lds_runs, K.bricks, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    del K, D
for t in range(0 if a.ssor_only else a.count):
    n = int(rng.integers(600, 60000))
    no = int(rng.integers(33, 65))
    span = int(rng.choice([40, 400, n // 2]))
    half = np.unique(rng.integers(1, max(span, no), size=no // 2 + 8))[: no // 2]
    offs = np.sort(np.concatenate([-half, [0], half]))
    if offs.size < 33:
        offs = np.arange(-20, 21)
    keep = float(rng.choice([1.0, 0.9, 0.6]))
    # symmetric pattern, symmetric values, dominant diagonal: SPD
    rows, cols, vals = [], [], []
    for o in half:
        r = np.arange(o, n)
        sel = rng.random(r.size) < keep
        r = r[sel]
        v = 0.3 * rng.standard_normal(r.size)
        rows += [r, r - o]
        cols += [r - o, r]
        vals += [v, v]
    rows.append(np.arange(n)), cols.append(np.arange(n)), vals.append(np.full(n, 0.0))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    rowsum = np.bincount(r, weights=np.abs(v), minlength=n)
    v[r == c] = 1.0 + rowsum + rng.random(n)
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=ind[1:])
    A = O.CSR((n, n), v, c.astype(np.int32), ind)
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    name, info = D.kernel_info()
    x = rng.standard_normal(n)
    y_ref, y = np.empty(n), np.empty(n)
    A.matvec(x, y_ref)
    D.matvec(x, y)
    ok = np.array_equal(y, y_ref)
    b = rng.standard_normal(n)
    msg = ""
    for solver, osolver in ((dev.pcg, O.pcg), (dev.minres, O.minres)):
        xo, xg = np.zeros(n), np.zeros(n)
        ro = osolver(A, b, xo, 1e-10, 500, O.jacobi_dinv(A.diagonal()))
        rg = solver(D, b, xg, 1e-10, 500, dev.DeviceJacobi(D))
        good = rg[:2] == ro[:2] and np.abs(xg - xo).max() <= 1e-12 * np.abs(xo).max()
        ok = ok and good
        msg += " %s %s" % (solver.__name__, rg[:2])
    print("w4y n %d offsets %d kernel %s nb %s%s %s" % (n, offs.size, name, info.get("nb"), msg, "ok" if ok else "MISMATCH"),
          flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
