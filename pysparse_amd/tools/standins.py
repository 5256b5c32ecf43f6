"""Seeded stand-ins for BASELINE.json configs[4] (SuiteSparse Janna/Emilia_923 as an sss_mat + MINRES):
the file cannot be fetched here (no network), so the config is exercised at its scale -- n ~ 9.2e5,
~44 stored entries per row of the full matrix, irregular numbering -- by two synthetic symmetric
matrices, and by the real file when a user supplies it (tests/test_gpu_config5.py, EMILIA_MTX).
Plain NumPy; nothing here computes products."""
import numpy as np

__all__ = ["fem_sss_arrays", "logspaced_sss_arrays"]


def fem_sss_arrays(gx=68, gy=68, gz=67, shuffle=32, seed=0, wild=0):
    """FEM-like: a gx x gy x gz node grid, 3 unknowns per node, every node coupled to itself, its 6 face and
    8 corner neighbours (45 entries per row of the full matrix, n = 3*gx*gy*gz = 929 424 by default), node
    numbers shuffled inside groups of `shuffle` consecutive nodes to mimic an unstructured numbering.
    Returns (n, ind, col, val, diag) of the sss form: strict lower triangle (ascending columns) + diagonal;
    diagonally dominant, hence SPD."""
    rng = np.random.default_rng(seed)
    nn = gx * gy * gz
    ids = np.arange(nn, dtype=np.int64)
    if shuffle > 1:  # local renumbering
        for a in range(0, nn, shuffle):
            b = min(nn, a + shuffle)
            ids[a:b] = a + rng.permutation(b - a)
    i = np.arange(nn) % gx
    j = (np.arange(nn) // gx) % gy
    k = np.arange(nn) // (gx * gy)
    nb = [(0, 0, 0)] + [(s, 0, 0) for s in (-1, 1)] + [(0, s, 0) for s in (-1, 1)] + [(0, 0, s) for s in (-1, 1)]
    nb += [(a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)]
    rows, cols, vals = [], [], []
    for (di, dj, dk) in nb:
        ok = (i + di >= 0) & (i + di < gx) & (j + dj >= 0) & (j + dj < gy) & (k + dk >= 0) & (k + dk < gz)
        p = np.nonzero(ok)[0]
        q = p + di + gx * (dj + gy * dk)
        P, Q = ids[p], ids[q]
        for d in range(3):
            for e in range(3):
                r, c = 3 * P + d, 3 * Q + e
                lower = c < r
                rr, cc = r[lower], c[lower]
                rows.append(rr)
                cols.append(cc)
                vals.append(-(0.05 + 0.01 * ((rr * 7 + cc * 13) % 10)))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    n = 3 * nn
    if wild:  # `wild` rows couple to 60 unknowns anywhere below them (constraint rows, long-range couplings)
        wr = np.random.default_rng(seed + 77).choice(np.arange(n // 2, n), size=wild, replace=False)
        er = np.repeat(wr, 60)
        ec = (np.random.default_rng(seed + 78).random(er.size) * er).astype(np.int64)
        have = set(zip(rows[np.isin(rows, wr)].tolist(), cols[np.isin(rows, wr)].tolist()))
        keep = np.array([(a, b) not in have for a, b in zip(er.tolist(), ec.tolist())])
        er, ec = er[keep], ec[keep]
        key = np.unique(er * n + ec)
        er, ec = key // n, key % n
        rows = np.concatenate([rows, er])
        cols = np.concatenate([cols, ec])
        vals = np.concatenate([vals, np.full(er.size, -0.001)])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
    diag = 10.0 + rng.random(n)
    return n, ind, cols.astype(np.int32), vals, diag


def logspaced_sss_arrays(n=923136, seed=1):
    """The pattern of examples/tendigit.py:26-38 (ones at offsets 1, 2, 4, ... below the diagonal: row degree
    grows with log2 of the row number) scaled to Emilia_923's order, with an irregular, diagonally dominant
    diagonal.  No renumbering can make this one banded: it stays on the gather kernels."""
    offs = []
    d = 1
    while d < n:
        offs.append(d)
        d *= 2
    i = np.arange(n, dtype=np.int64)
    cols = [i - o for o in reversed(offs)]  # ascending column within a row
    mask = [c >= 0 for c in cols]
    lens = np.sum(mask, axis=0).astype(np.int64)
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=ind[1:])
    col = np.concatenate([c[:, None] for c in cols], axis=1)[np.stack(mask, axis=1)].astype(np.int32)
    rng = np.random.default_rng(seed)
    diag = 40.0 + rng.random(n) * 1e3
    return n, ind, col, np.ones(len(col)), diag
