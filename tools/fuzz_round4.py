#!/usr/bin/env python3
"""One-off differential campaign for what round 4 changed under the index-free kernels (not part of the test suite):
random OFFSET-STRUCTURED matrices -- 1 .. 32 distinct offsets (csr_spmv_w4 / w4x), extents that are no multiple of a
block, dropped entries (row masks), with and without a stored diagonal, rectangular shapes, the symmetric ones also as
sss_mat (sss_spmv_w4, <= 8 lower offsets) -- against the oracle:
  * y = A x and A^T x: bit equality;
  * the solvers' fused dot (round 4: the operand's pair is reused from the registers when it IS x seen through a stored
    offset) inside Jacobi-PCG / Jacobi-MINRES on the SPD cases: equal (info, iter), iterates <= max(1e-12, 32 k sqrt(n) eps / dominance) -- through the
    single-kernel loops (n <= 2^18, <= 8 entries per row), the lazy and the eager device loops as the size selects them;
    (beyond n = 70 001: 12 iterations with tol = 0).
Prints one line per matrix; exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--count", type=int, default=80)
ap.add_argument("--seconds", type=float, default=420.0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)


def csr_from_coo(shape, r, c, v):
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    ind = np.zeros(shape[0] + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=shape[0]), out=ind[1:])
    return O.CSR(shape, np.ascontiguousarray(v), np.ascontiguousarray(c.astype(np.int32)), ind)


def build():
    kind = rng.choice(["spd", "spd", "general", "rect"])
    n = int(rng.choice([2, 3, 127, 129, 1000, 4097, 30011, 30011, 70001, 70001, 262144, 262144, 300007, 1200003]))
    hb = int(min(max(1, n - 1), rng.choice([1, 3, 40, 700, 5000])))
    if kind == "spd":
        nlo = int(rng.integers(1, 9 if rng.random() < 0.7 else 17))
        lo = np.unique(-rng.integers(1, hb + 1, size=nlo))
        keep = float(rng.choice([1.0, 1.0, 0.9, 0.5]))
        rr, cc, vv = [], [], []
        for o in lo:
            r = np.arange(-o, n)
            if r.size == 0:
                continue
            sel = rng.random(r.size) < keep
            r = r[sel]
            rr.append(r)
            cc.append(r + o)
            vv.append(-(0.1 + 0.9 * rng.random(r.size)))
        r = np.concatenate(rr) if rr else np.zeros(0, dtype=np.int64)
        c = np.concatenate(cc) if cc else np.zeros(0, dtype=np.int64)
        v = np.concatenate(vv) if vv else np.zeros(0)
        rowsum = np.bincount(np.concatenate([r, c]), weights=np.concatenate([-v, -v]), minlength=n)
        dom = float(rng.choice([1.0, 0.05, 0.002]))  # diagonal dominance: condition number ~ 2 / dom
        dg = rowsum * (1.0 + dom) + 0.01 + rng.random(n) * float(rng.choice([1.0, 0.0]))
        A = csr_from_coo((n, n), np.concatenate([r, np.arange(n), c]), np.concatenate([c, np.arange(n), r]),
                         np.concatenate([v, dg, v]))
        order = np.lexsort((c, r))
        lind = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(r, minlength=n), out=lind[1:])
        S = O.SSS(n, np.ascontiguousarray(v[order]), dg, np.ascontiguousarray(c[order].astype(np.int32)), lind)
        return dict(kind=kind, n=n, offsets=2 * lo.size + 1, keep=keep, dom=dom, nnz=A.nnz), A, S
    m = n if kind == "general" else max(2, int(n * rng.choice([0.5, 0.9, 1.1, 2.0])))
    no = int(rng.integers(1, 13 if rng.random() < 0.7 else 33))
    offs = np.unique(rng.integers(-hb, hb + 1, size=no))
    if rng.random() < 0.7:
        offs = np.unique(np.concatenate([offs, [0]]))
    keep = float(rng.choice([1.0, 0.9, 0.5]))
    rr, cc = [], []
    for o in offs:
        r = np.arange(max(0, -o), min(n, m - o))
        r = r[rng.random(r.size) < keep]
        rr.append(r)
        cc.append(r + o)
    r, c = np.concatenate(rr), np.concatenate(cc)
    A = csr_from_coo((n, m), r, c, rng.standard_normal(r.size))
    return dict(kind=kind, n=n, m=m, offsets=int(offs.size), keep=keep, nnz=A.nnz), A, None


t0 = time.time()
kinds = {}
for it in range(a.count):
    if time.time() - t0 > a.seconds:
        break
    desc, A, S = build()
    n, m = A.shape
    if A.nnz == 0:
        continue
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    kern = D.kernel_info()[0]
    kinds[kern] = kinds.get(kern, 0) + 1
    x = rng.standard_normal(m)
    y, yo = np.full(n, np.nan), np.empty(n)
    D.matvec(x, y)
    A.matvec(x, yo)
    if not np.array_equal(y, yo):
        print("MISMATCH y", desc, kern, int(np.sum(y != yo)), flush=True)
        sys.exit(1)
    xt = rng.standard_normal(n)
    yt, yto = np.full(m, np.nan), np.empty(m)
    D.matvec_transp(xt, yt)
    A.matvec_transp(xt, yto)
    if not np.array_equal(yt, yto):
        print("MISMATCH yT", desc, kern, flush=True)
        sys.exit(1)
    msg = ""
    if S is not None:
        Sd = dev.DeviceSSS.from_arrays(n, S.ind, S.col, S.val, S.diag)
        ys, yso = np.full(n, np.nan), np.empty(n)
        Sd.matvec(x, ys)
        S.matvec(x, yso)
        skern = Sd.kernel_info()[0]
        kinds[skern] = kinds.get(skern, 0) + 1
        if not (np.array_equal(ys, yso) and np.array_equal(ys, yo)):
            print("MISMATCH sss y", desc, skern, flush=True)
            sys.exit(1)
        b = np.empty(n)
        A.matvec(np.ones(n), b)
        dinv = 1.0 / S.diag
        tol, maxit = (1e-10, 300) if n <= 70001 else (0.0, 12)
        for name, sg, so in (("pcg", dev.pcg, O.pcg), ("minres", dev.minres, O.minres)):
            xo = np.zeros(n)
            ref = so(A, b, xo, tol, maxit, dinv)
            xs = []
            for op in (D, Sd):
                xg = np.zeros(n)
                got = sg(op, b, xg, tol, maxit, dev.DeviceJacobi(op))
                err = np.abs(xg - xo).max() / max(np.abs(xo).max(), 1e-300)
                # the dots are summed in another order than the oracle's: 32 k sqrt(n) eps (bench.parity_bound), times the
                # conditioning 1 / dom, for x relative to max |x| and for relres (a norm relative to |b| already) absolutely;
                # a solve that stops on its tolerance may cross it one iteration apart
                bound = max(1e-12, 32.0 * max(ref[1], 1) * np.sqrt(n) * 2.0 ** -52 / desc["dom"])
                if got[0] != ref[0] or abs(got[1] - ref[1]) > (1 if tol > 0 else 0) or err > bound or \
                        abs(got[2] - ref[2]) > 1e-9 * abs(ref[2]) + bound + (tol if got[1] != ref[1] else 0.0):
                    print("MISMATCH", name, desc, kern if op is D else skern, got, ref, err, flush=True)
                    sys.exit(1)
                xs.append(xg)
            # (reported, not required: the two forms' products have the same bits, their dot partials need not)
            msg += " %s %d %.1e%s" % (name, ref[1], err, "" if np.array_equal(xs[0], xs[1]) else " (forms differ %.0e)"
                                      % (np.abs(xs[0] - xs[1]).max() / max(np.abs(xo).max(), 1e-300)))
        Sd.close()
        msg = " " + skern + msg
    D.close()
    print(it, desc, kern, msg, flush=True)
print("kernels:", kinds, "seconds %.0f" % (time.time() - t0))
