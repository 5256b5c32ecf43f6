#!/usr/bin/env python3
"""One-off: an irregular symmetric matrix beyond 2^24 rows (banded pattern, shuffled in groups of 512 so that the
renumbering runs): device product (default selection, incl. the renumbered copy) against the oracle's, bit for bit.
Exercises the one-wave-per-row set-up kernels at a size where a launch of one block per four rows exceeds 2^32 threads."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pysparse_amd import device as dev
rng = np.random.default_rng(5)
n, hb, k = 20_000_000, 600, 3
r = np.repeat(np.arange(n, dtype=np.int64), k)
c = r - rng.integers(1, hb + 1, size=r.size)
ok = c >= 0
r, c = r[ok], c[ok]
ids = np.arange(n, dtype=np.int64)
for a in range(0, n, 512):
    ids[a:a + 512] = a + rng.permutation(min(512, n - a))
r, c = ids[r], ids[c]
lo = c > r
r[lo], c[lo] = c[lo], r[lo].copy()
key = np.unique(r * n + c)
r, c = key // n, key % n
v = -(0.1 + 0.9 * rng.random(r.size))
fr = np.concatenate([r, np.arange(n), c]); fc = np.concatenate([c, np.arange(n), r])
rowsum = np.bincount(np.concatenate([r, c]), weights=np.concatenate([-v, -v]), minlength=n)
fv = np.concatenate([v, rowsum + 1.0, v])
order = np.argsort(fr * n + fc, kind="stable")
fr, fc, fv = fr[order], fc[order], fv[order]
ind = np.zeros(n + 1, dtype=np.int32); np.cumsum(np.bincount(fr, minlength=n), out=ind[1:])
A = O.CSR((n, n), fv, fc.astype(np.int32), ind)
print("built", n, fr.size, flush=True)
D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
x = rng.standard_normal(n); y = np.full(n, np.nan); yo = np.empty(n)
t = time.time(); D.matvec(x, y); print("first product", round(time.time() - t, 2), "s", D.kernel_info(), flush=True)
A.matvec(x, yo)
print("equal", np.array_equal(y, yo), flush=True)
sys.exit(0 if np.array_equal(y, yo) else 1)
