#!/usr/bin/env python3
"""One-off differential campaign for the brick form of the single-kernel loops (pcg_brick_kernel / minres_brick_kernel,
psp_mid.hip; not part of the test suite): random SPD 7-offset operators of nx x ny x nz grids -- 1.5e5 .. 2^20 points, grids
that no brick size divides, flat and elongated ones, dropped couplings (row masks), weak to strong diagonal dominance,
constant or varying coefficients -- as csr_mat and as sss_mat, PCG and MINRES with Jacobi and without:
  * against the oracle: equal (info, iter), x within max(1e-12, 32 k sqrt(n) eps / dominance x spread);
  * twice: the same bits;
  * every solve must have run as one kernel (psp_debug_brick_count), none handed back.
Prints one line per matrix; exits non-zero on the first mismatch."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from pysparse_amd import _capi, device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--count", type=int, default=100)
ap.add_argument("--seconds", type=float, default=240.0)
ap.add_argument("--indefinite", action="store_true",
                help="flip the sign of some diagonal entries: the breakdown exits inside the kernels (minres -3 after some "
                     "iterations, runs to maxit); compared with the oracle: (info, iter) (converged runs may cross the "
                     "tolerance a few iterations apart)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
L = _capi.lib()


def count():
    s, f = C.c_longlong(), C.c_longlong()
    L.psp_debug_brick_count(C.byref(s), C.byref(f))
    return s.value, f.value


def build():
    while True:
        nx, ny, nz = (int(v) for v in rng.integers(5, 200, size=3))
        if rng.random() < 0.3:
            nx = ny = nz = int(rng.integers(54, 101))
        n = nx * ny * nz
        if 150000 <= n <= (1 << 20) and nx * ny > 2046:  # (slimmer grids go to the row-block kernels)
            break
    idx = np.arange(n)
    i, j = idx % nx, (idx // nx) % ny
    keep = float(rng.choice([1.0, 1.0, 0.9, 0.6]))
    const = rng.random() < 0.4
    rr, cc, vv = [], [], []
    for o, ok in ((1, i < nx - 1), (nx, j < ny - 1), (nx * ny, idx < n - nx * ny)):
        r = idx[ok & (idx + o < n)]
        r = r[rng.random(r.size) < keep]
        rr.append(r + o)  # lower entry (r + o, r)
        cc.append(r)
        vv.append(-np.ones(r.size) if const else -(0.1 + 0.9 * rng.random(r.size)))
    r, c, v = np.concatenate(rr), np.concatenate(cc), np.concatenate(vv)
    rowsum = np.bincount(np.concatenate([r, c]), weights=np.concatenate([-v, -v]), minlength=n)
    dom = float(rng.choice([1.0, 0.05, 0.002]))
    dg = np.full(n, 6.0 * (1.0 + dom)) if const else rowsum * (1.0 + dom) + 0.01 + rng.random(n) * float(rng.choice([1.0, 0.0]))
    if a.indefinite:
        dg = np.where(rng.random(n) < float(rng.choice([0.5, 0.1, 1e-3, 2.0 / n])), -dg, dg)
    rows = np.concatenate([r, idx, c])
    cols = np.concatenate([c, idx, r])
    vals = np.concatenate([v, dg, v])
    order = np.lexsort((cols, rows))
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
    A = O.CSR((n, n), np.ascontiguousarray(vals[order]), np.ascontiguousarray(cols[order].astype(np.int32)), ind)
    lorder = np.lexsort((c, r))
    lind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=lind[1:])
    S = O.SSS(n, np.ascontiguousarray(v[lorder]), dg, np.ascontiguousarray(c[lorder].astype(np.int32)), lind)
    return dict(grid=(nx, ny, nz), n=n, keep=keep, dom=dom, const=bool(const), spread=float(dg.max() / dg.min())), A, S


t0 = time.time()
done = skipped = 0
declined = set()
soft, exits = [], {}
for it in range(a.count):
    if time.time() - t0 > a.seconds:
        break
    desc, A, S = build()
    n = A.shape[0]
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    Sd = dev.DeviceSSS.from_arrays(n, S.ind, S.col, S.val, S.diag)
    if D.kernel_info()[0] != "csr_spmv_w4":
        print(it, desc, D.kernel_info()[0], "(not offset-structured enough: skipped)", flush=True)
        skipped += 1
        continue
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    b += 0.01 * rng.standard_normal(n)
    dinv = 1.0 / S.diag
    tol, maxit = (1e-10, 150) if n <= 300000 else (0.0, int(rng.choice([1, 2, 7, 20])))
    msg = ""
    for name, sg, so in (("pcg", dev.pcg, O.pcg), ("minres", dev.minres, O.minres)):
        for pre in (True, False):
            xo = np.zeros(n)
            ref = so(A, b, xo, tol, maxit, dinv if pre else None)
            for op in (D, Sd):
                K = dev.DeviceJacobi(op) if pre else None
                c0 = count()
                x1, x2 = np.zeros(n), np.zeros(n)
                r1 = sg(op, b, x1, tol, maxit, K)
                r2 = sg(op, b, x2, tol, maxit, K)
                c1 = count()
                if c1 == c0:
                    declined.add(it)  # no decomposition into <= 256 bricks of <= 4096 points: the launch-per-phase loops ran
                elif c1[0] - c0[0] != 2 or c1[1] != c0[1]:
                    print("NOT A SINGLE KERNEL", name, pre, desc, (c0, c1), flush=True)
                    sys.exit(1)
                if tuple(r1[:3]) != tuple(r2[:3]) or not np.array_equal(x1, x2):
                    print("NOT REPRODUCIBLE", name, pre, desc, r1[:3], r2[:3], flush=True)
                    sys.exit(1)
                if a.indefinite:
                    if r1[0] != ref[0] or abs(r1[1] - ref[1]) > (8 if r1[0] == 0 else 1):
                        print("EXIT DIFFERS FROM THE ORACLE'S", name, pre, desc, tuple(r1[:3]), ref[:3], flush=True)
                        soft.append(it)
                    exits[(name, r1[0])] = exits.get((name, r1[0]), 0) + 1
                    continue
                err = np.abs(x1 - xo).max() / max(np.abs(xo).max(), 1e-300)
                bound = max(1e-12, 32.0 * max(ref[1], 1) * np.sqrt(n) * 2.0 ** -52 / desc["dom"] * (1.0 if pre else desc["spread"]))
                if r1[0] != ref[0] or abs(r1[1] - ref[1]) > (1 if tol > 0 else 0) or err > bound or \
                        abs(r1[2] - ref[2]) > 1e-9 * abs(ref[2]) + bound + (tol if r1[1] != ref[1] else 0.0):
                    print("MISMATCH vs oracle", name, pre, desc, tuple(r1[:3]), ref[:3], err, flush=True)
                    sys.exit(1)
            msg += " %s%s %d %s" % (name, "+jac" if pre else "", ref[1], ("info %d" % ref[0]) if a.indefinite else "%.1e" % err)
    D.close()
    Sd.close()
    done += 1
    print(it, desc, msg, "(plan declined)" if it in declined else "", flush=True)
print("matrices: %d (skipped %d, plan declined for %d), seconds %.0f, single-kernel solves %d, fallbacks %d" % (
    (done, skipped, len(declined), time.time() - t0) + count()))
if a.indefinite:
    print("exits seen (solver, info): count", sorted(exits.items()), "; exits that differ from the oracle's:", len(soft))
