"""Helper of test_gpu_spmv.py::test_device_renumbering_equals_host_renumbering: builds one of the named irregular
matrices, multiplies once (which builds the renumbered copy) and, run as a script, stores the permutation --
the test starts it with PSP_SPMV_REORDER_HOST=1 (host Cuthill-McKee, read once per process) and compares."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def full_csr_from_sss(n, ind, col, val, diag):
    """rows of the full matrix: lower entries (ascending), diagonal, mirrored entries (ascending)"""
    from pysparse_amd.distributed import sss_rows_expanded
    return sss_rows_expanded(n, ind, col, val, diag)


def case_arrays(name):
    from pysparse_amd.tools.standins import fem_sss_arrays
    if name == "fem512":
        n, ind, col, val, diag = fem_sss_arrays(20, 18, 16, 512, 0)
        return (n,) + tuple(full_csr_from_sss(n, ind, col, val, diag))
    if name == "hubs":  # six rows with 60 extra couplings anywhere: left out of the level structure, numbered last
        n, ind, col, val, diag = fem_sss_arrays(20, 18, 16, 512, 0, wild=6)
        return (n,) + tuple(full_csr_from_sss(n, ind, col, val, diag))
    if name == "components":
        # two meshes of different size side by side + isolated (diagonal-only) rows in front, between and behind
        parts = [fem_sss_arrays(12, 11, 10, 64, 1), fem_sss_arrays(9, 14, 8, 512, 2)]
        blocks = []
        for (n, ind, col, val, diag) in parts:
            blocks.append((n,) + tuple(full_csr_from_sss(n, ind, col, val, diag)))
        iso = 5
        n_tot = sum(b[0] for b in blocks) + 3 * iso
        f_ind, f_col, f_val = [0], [], []
        row0 = 0

        def add_iso(k):
            nonlocal row0
            for _ in range(k):
                f_col.append(np.array([row0], dtype=np.int64))
                f_val.append(np.array([2.5]))
                f_ind.append(f_ind[-1] + 1)
                row0 += 1
        add_iso(iso)
        for (n, ind, col, val) in blocks:
            f_col.append(col.astype(np.int64) + row0)
            f_val.append(val)
            f_ind.extend((f_ind[-1] + ind[1:].astype(np.int64)).tolist())
            row0 += n
            add_iso(iso)
        return (n_tot, np.asarray(f_ind, dtype=np.int32), np.concatenate(f_col).astype(np.int32),
                np.concatenate(f_val))
    if name == "unsymmetric":
        n, ind, col, val, diag = fem_sss_arrays(16, 15, 14, 512, 3)
        f_ind, f_col, f_val = full_csr_from_sss(n, ind, col, val, diag)
        # drop every 7th strictly-upper entry: the pattern is no longer symmetric
        rows = np.repeat(np.arange(n), np.diff(f_ind))
        upper = np.nonzero(f_col > rows)[0]
        keep = np.ones(f_col.size, dtype=bool)
        keep[upper[::7]] = False
        f_col, f_val, rows = f_col[keep], f_val[keep], rows[keep]
        f_ind = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=n), out=f_ind[1:])
        return n, f_ind, f_col, f_val
    raise ValueError(name)


def renumbering_of(name):
    from pysparse_amd import device as dev
    n, ind, col, val = case_arrays(name)
    A = dev.DeviceCSR.from_arrays((n, n), ind, col, val)
    A.prepare(1 << 30)  # the copy at first use (the cost rule would wait for 2048 products)
    kern, _ = A.kernel_info()
    x = np.random.default_rng(5).standard_normal(n)
    y = np.empty(n)
    A.matvec(x, y)
    perm = A.renumbering()
    return kern, perm, y, A.renumbered_on


if __name__ == "__main__":
    kern, perm, y, where = renumbering_of(sys.argv[1])
    np.savez(sys.argv[2], kern=np.array(kern), perm=perm if perm is not None else np.zeros(0, dtype=np.int32), y=y,
             where=np.array(str(where)))
