#!/usr/bin/env python3
"""One-off differential campaign for round 5's single-kernel loops (psp_mid.hip; not part of the test suite): random
symmetric positive definite OFFSET-STRUCTURED matrices in their range -- 2^16 .. 2^20 rows, 1 .. 4 lower offsets up to 2046
(3 / 5 / 7 / 9 offsets in all), dropped entries (row masks), weak to strong diagonal dominance, constant or varying diagonals,
extents that are no multiple of a span -- as csr_mat and as sss_mat, Jacobi-PCG and Jacobi-MINRES and both without a
preconditioner:
  * against the oracle: equal (info, iter), iterates within max(1e-12, 32 k sqrt(n) eps / dominance);
  * against the launch-per-phase loops in the same process (PSP_MID_MIN, read per solve): info, iter, relres, x BIT FOR BIT;
  * the solves must have run as single kernels (psp_debug_mid_count), none refused.
Start with PSP_TUNING=1.  Prints one line per matrix; exits non-zero on the first mismatch."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PSP_TUNING") == "1", "start with PSP_TUNING=1"
from oracle import oracle as O  # noqa: E402
from pysparse_amd import _capi, device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--count", type=int, default=200)
ap.add_argument("--seconds", type=float, default=420.0)
ap.add_argument("--indefinite", action="store_true",
                help="flip the sign of some diagonal entries (and with them of Jacobi's): the breakdown exits -- "
                     "pcg -2 / -6 / -5, minres -3 / -6 -- inside the kernels; compared: single kernel vs launch per phase "
                     "bit for bit, and (info, iter) with the oracle")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
L = _capi.lib()


def mid_count():
    s, f = C.c_longlong(), C.c_longlong()
    L.psp_debug_mid_count(C.byref(s), C.byref(f))
    return s.value, f.value


def build():
    n = int(rng.choice([65536, 65537, 70001, 131071, 200003, 262144, 300007, 524288, 524800, 777777, 1048575, 1048576,
                        1048577, 1500001, 2097152]))  # (beyond 2^20: blocks of 8192 rows for constant-coefficient PCG, else declined)
    nlo = int(rng.integers(1, 5))  # 3 .. 9 offsets in all
    hb = int(rng.choice([1, 3, 40, 700, 2044]))
    lo = np.unique(-rng.integers(1, hb + 1, size=nlo))
    keep = float(rng.choice([1.0, 1.0, 0.9, 0.5]))
    allconst = rng.random() < 0.25  # one value per offset and a constant diagonal: the constant-coefficient kernel forms
    rr, cc, vv = [], [], []
    for o in lo:
        r = np.arange(-o, n)
        sel = rng.random(r.size) < keep
        r = r[sel]
        rr.append(r)
        cc.append(r + o)
        vv.append(-(0.1 + 0.9 * rng.random(r.size)) if (rng.random() < 0.7 and not allconst) else -np.full(r.size, float(rng.integers(1, 4))))
    r, c, v = np.concatenate(rr), np.concatenate(cc), np.concatenate(vv)
    rowsum = np.bincount(np.concatenate([r, c]), weights=np.concatenate([-v, -v]), minlength=n)
    dom = float(rng.choice([1.0, 0.05, 0.002]))
    if rng.random() < 0.3 or allconst:
        dg = np.full(n, rowsum.max() * (1.0 + dom) + 0.01)  # a constant diagonal: jacobi's dinv is a scalar
    else:
        dg = rowsum * (1.0 + dom) + 0.01 + rng.random(n) * float(rng.choice([1.0, 0.0]))
    if a.indefinite:
        flips = rng.random(n) < float(rng.choice([0.5, 0.1, 1e-3, 2.0 / n]))
        dg = np.where(flips, -dg, dg)
    rows = np.concatenate([r, np.arange(n), c])
    cols = np.concatenate([c, np.arange(n), r])
    vals = np.concatenate([v, dg, v])
    order = np.lexsort((cols, rows))
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
    A = O.CSR((n, n), np.ascontiguousarray(vals[order]), np.ascontiguousarray(cols[order].astype(np.int32)), ind)
    lorder = np.lexsort((c, r))
    lind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=lind[1:])
    S = O.SSS(n, np.ascontiguousarray(v[lorder]), dg, np.ascontiguousarray(c[lorder].astype(np.int32)), lind)
    return dict(n=n, offsets=2 * int(lo.size) + 1, far=int(-lo.min()), keep=keep, dom=dom, const=bool(allconst),
                spread=float(dg.max() / dg.min())), A, S


t0 = time.time()
done = 0
refused = set()
soft, exits = [], {}
for it in range(a.count):
    if time.time() - t0 > a.seconds:
        break
    desc, A, S = build()
    n = A.shape[0]
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    Sd = dev.DeviceSSS.from_arrays(n, S.ind, S.col, S.val, S.diag)
    kern, skern = D.kernel_info()[0], Sd.kernel_info()[0]
    if kern != "csr_spmv_w4":
        print(it, desc, kern, "(not offset-structured enough: skipped)", flush=True)
        continue
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    b += 0.01 * rng.standard_normal(n)
    dinv = 1.0 / S.diag
    tol, maxit = (1e-10, 200) if n <= 131071 else (0.0, int(rng.choice([1, 2, 7, 20])))
    msg = ""
    for name, sg, so in (("pcg", dev.pcg, O.pcg), ("minres", dev.minres, O.minres)):
        for pre in (True, False):
            xo = np.zeros(n)
            ref = so(A, b, xo, tol, maxit, dinv if pre else None)
            for op, opk in ((D, kern), (Sd, skern)):
                K = dev.DeviceJacobi(op) if pre else None
                got = {}
                for mode in ("mid", "phase"):
                    os.environ["PSP_MID_MIN"] = "1" if mode == "mid" else str(1 << 30)
                    os.environ["PSP_COOP"] = "0"
                    c0 = mid_count()
                    xg = np.zeros(n)
                    res = sg(op, b, xg, tol, maxit, K)
                    c1 = mid_count()
                    got[mode] = (tuple(res[:3]), xg, c1[0] - c0[0], c1[1] - c0[1])
                (rm, xm, sm, fm), (rp, xp, sp, fp) = got["mid"], got["phase"]
                if (sm, fm) == (0, 0) and sp == 0:
                    refused.add(it)  # the plan declined (window + vectors beyond the LDS): launch-per-phase loops both times
                elif sm != 1 or fm != 0 or sp != 0:
                    print("NOT A SINGLE KERNEL", name, pre, desc, opk, (sm, fm, sp, fp), flush=True)
                    sys.exit(1)
                if rm != rp or not np.array_equal(xm, xp):
                    print("MISMATCH single kernel vs launch per phase", name, pre, desc, opk, rm, rp, flush=True)
                    sys.exit(1)
                if a.indefinite:  # iterates of a breaking-down recurrence are not comparable digit by digit: the exit is
                    if rm[0] != ref[0] or abs(rm[1] - ref[1]) > (8 if rm[0] == 0 else 1):
                        # (an exit decided by a sign or an exact zero may fall one iteration apart between summation orders;
                        # on an indefinite system the residual of CG / MINRES does not fall monotonically: runs that
                        # converge cross the tolerance a few iterations apart -- 2 .. 5 seen -- in either implementation)
                        print("EXIT DIFFERS FROM THE ORACLE'S", name, pre, desc, opk, rm, ref[:3], flush=True)
                        soft.append((it, name, pre, rm, ref[:3]))
                    exits[(name, rm[0])] = exits.get((name, rm[0]), 0) + 1
                    continue
                err = np.abs(xm - xo).max() / max(np.abs(xo).max(), 1e-300)
                # 32 k sqrt(n) eps (bench.parity_bound) times the conditioning: 1 / dominance, and without the
                # preconditioner also the spread of the diagonal (rows that lost all their neighbours keep 0.01 + random)
                bound = max(1e-12, 32.0 * max(ref[1], 1) * np.sqrt(n) * 2.0 ** -52 / desc["dom"] * (1.0 if pre else desc["spread"]))
                if rm[0] != ref[0] or abs(rm[1] - ref[1]) > (1 if tol > 0 else 0) or err > bound or \
                        abs(rm[2] - ref[2]) > 1e-9 * abs(ref[2]) + bound + (tol if rm[1] != ref[1] else 0.0):
                    print("MISMATCH vs oracle", name, pre, desc, opk, rm, ref[:3], err, flush=True)
                    sys.exit(1)
            msg += " %s%s %d %s" % (name, "+jac" if pre else "", ref[1], ("info %d" % ref[0]) if a.indefinite else "%.1e" % err)
    D.close()
    Sd.close()
    done += 1
    print(it, desc, skern, msg, "(plan declined)" if it in refused else "", flush=True)
print("matrices: %d (plan declined for %d), seconds %.0f, single-kernel solves %d, fallbacks %d" % (
    (done, len(refused), time.time() - t0) + mid_count()))
if a.indefinite:
    print("exits seen (solver, info): count", sorted(exits.items()), "; exits that differ from the oracle's:", len(soft))
