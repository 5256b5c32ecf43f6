#!/usr/bin/env python3
"""One-off differential campaign (not part of the test suite: minutes of CPU oracle time): medium-size irregular
matrices -- banded symmetric patterns, locally / globally shuffled numberings, wild rows, dropped entries
(unsymmetric patterns), several components -- default kernel selection against the oracle: y, A^T x, the fused
dot, Jacobi-PCG / MINRES where the matrix is SPD.  Prints one line per matrix; exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402


def sym_pattern(rng, n, hb, deg, wild):
    """strict-lower (row, col) pairs of a symmetric pattern: ~deg/2 lower entries per row inside a band of hb"""
    rows, cols = [], []
    k = max(1, deg // 2)
    r = np.repeat(np.arange(n), k)
    c = r - rng.integers(1, hb + 1, size=r.size)
    ok = c >= 0
    rows.append(r[ok])
    cols.append(c[ok])
    if wild:
        wr = rng.choice(np.arange(n // 2, n), size=wild, replace=False)
        r = np.repeat(wr, 50)
        c = (rng.random(r.size) * r).astype(np.int64)
        rows.append(r)
        cols.append(c)
    r, c = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(r * n + c)
    return key // n, key % n


def build(rng):
    n = int(rng.choice([3000, 12000, 40000, 90000]))
    hb = int(rng.choice([20, 150, 900]))
    deg = int(rng.choice([14, 30, 56]))
    wild = int(rng.choice([0, 0, 4, 40]))
    shuffle = int(rng.choice([0, 0, 16, 512, -1]))
    drop = float(rng.choice([0.0, 0.0, 0.0, 0.05]))
    comps = int(rng.choice([1, 1, 1, 3]))
    r, c = sym_pattern(rng, n, hb, deg, wild)
    if comps > 1:  # cut the couplings that cross two equal thirds
        third = n // comps
        keep = (r // third) == (c // third)
        r, c = r[keep], c[keep]
    if shuffle:
        ids = np.arange(n)
        if shuffle < 0:
            ids = rng.permutation(n)
        else:
            for a in range(0, n, shuffle):
                b = min(n, a + shuffle)
                ids[a:b] = a + rng.permutation(b - a)
        r, c = ids[r], ids[c]
        lo = c > r
        r[lo], c[lo] = c[lo], r[lo].copy()
    v = -(0.1 + 0.9 * rng.random(r.size))
    # full matrix: lower, diagonal (dominant => SPD), mirrored
    fr = np.concatenate([r, np.arange(n), c])
    fc = np.concatenate([c, np.arange(n), r])
    rowsum = np.bincount(np.concatenate([r, c]), weights=np.concatenate([-v, -v]), minlength=n)
    fv = np.concatenate([v, rowsum + 1.0 + rng.random(n), v])
    spd = True
    if drop:
        keep = (fr == fc) | (rng.random(fr.size) >= drop)
        fr, fc, fv = fr[keep], fc[keep], fv[keep]
        spd = False
    order = np.lexsort((fc, fr))
    fr, fc, fv = fr[order], fc[order], fv[order]
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(fr, minlength=n), out=ind[1:])
    desc = dict(n=n, hb=hb, deg=deg, wild=wild, shuffle=shuffle, drop=drop, comps=comps, nnz=int(fr.size))
    return desc, O.CSR((n, n), fv, fc.astype(np.int32), ind), spd


ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--count", type=int, default=60)
ap.add_argument("--seconds", type=float, default=400.0)
ap.add_argument("--ssor", action="store_true", help="also the sss form + precon.ssor on the SPD cases up to n = 40000")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time()
kinds = {}
for it in range(a.count):
    if time.time() - t0 > a.seconds:
        break
    desc, A, spd = build(rng)
    n = A.shape[0]
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    kern, info = D.kernel_info()
    kinds[kern] = kinds.get(kern, 0) + 1
    x = rng.standard_normal(n)
    y, yo = np.full(n, np.nan), np.empty(n)
    D.matvec(x, y)
    A.matvec(x, yo)
    if not np.array_equal(y, yo):
        print("MISMATCH y", desc, kern, info, int(np.sum(y != yo)), flush=True)
        sys.exit(1)
    yt, yto = np.full(n, np.nan), np.empty(n)
    D.matvec_transp(x, yt)
    A.matvec_transp(x, yto)
    if not np.array_equal(yt, yto):
        print("MISMATCH yT", desc, kern, info, flush=True)
        sys.exit(1)
    msg = ""
    if spd:
        b = np.empty(n)
        A.matvec(np.ones(n), b)
        dinv = 1.0 / A.val[A.ind[:-1] + np.array([np.searchsorted(A.col[A.ind[i]:A.ind[i + 1]], i) for i in range(n)])]
        for name, sg, so in (("pcg", dev.pcg, O.pcg), ("minres", dev.minres, O.minres)):
            xo, xg = np.zeros(n), np.zeros(n)
            ref = so(A, b, xo, 1e-10, 400, dinv)
            got = sg(D, b, xg, 1e-10, 400, dev.DeviceJacobi(D))
            err = np.abs(xg - xo).max() / max(np.abs(xo).max(), 1e-300)
            if got[0] != ref[0] or abs(got[1] - ref[1]) > 1 or err > 1e-10:
                print("MISMATCH", name, desc, kern, got, ref, err, flush=True)
                sys.exit(1)
            msg += " %s %d/%d %.1e" % (name, got[1], ref[1], err)
    if spd and a.ssor and n <= 40000:
        # the same matrix as an sss_mat: product + precon.ssor (level schedule by Kahn's algorithm) against the oracle
        low = A.col < np.repeat(np.arange(n), np.diff(A.ind))
        rows = np.repeat(np.arange(n), np.diff(A.ind))
        lind = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows[low], minlength=n), out=lind[1:])
        dg = A.val[A.col == rows]
        So = O.SSS(n, np.ascontiguousarray(A.val[low]), np.ascontiguousarray(dg), np.ascontiguousarray(A.col[low]), lind)
        Sd = dev.DeviceSSS.from_arrays(n, So.ind, So.col, So.val, So.diag)
        ys, yso = np.full(n, np.nan), np.empty(n)
        Sd.matvec(x, ys)
        So.matvec(x, yso)
        if not np.array_equal(ys, yso):
            print("MISMATCH sss y", desc, Sd.kernel_info(), flush=True)
            sys.exit(1)
        for omega in (1.0, 1.3):
            K = dev.DeviceSSOR(Sd, omega, 1)
            z, zo = np.full(n, np.nan), np.zeros(n)
            K.precon(x, z)
            O.ssor_apply(So, x, zo, omega, 1)
            if not np.array_equal(z, zo):
                print("MISMATCH ssor", desc, omega, K.levels, flush=True)
                sys.exit(1)
        msg += " ssor levels %s" % (K.levels,)
        Sd.close()
    D.close()
    print(it, desc, kern, info["nb"], info["max_blocks"], msg, flush=True)
print("kernels:", kinds, "seconds %.0f" % (time.time() - t0))
