"""MatrixMarket ingest straight into csr_mat / sss_mat, without the ll_mat in between (SURVEY section 8f
rank 1): for files such as SuiteSparse Emilia_923 (4.0e7 entries) where one sorted insertion per entry
(ll_mat_from_mtx, ll_mat.c:3390-3456 -> SpMatrix_LLMatSetItem :250-356) is needlessly slow.  Same result
as spmatrix.ll_mat_from_mtx(path).to_csr() / .to_sss(): entries sorted by (row, column), a symmetric file
expanded to the full matrix for csr (ll_mat.c:1586-1625), strict lower triangle + diagonal for sss
(:1654-1708); tests/test_spmatrix_host.py compares the two routes array by array.

Banner rules are those of mmio / LLMat_from_mtx: "coordinate real general|symmetric" only; a repeated
entry keeps its last value (what repeated ll_mat assignments do)."""
import numpy as np

from ..sparse import spmatrix

__all__ = ["read_mtx", "csr_arrays_from_mtx", "sss_arrays_from_mtx", "csr_from_mtx", "sss_from_mtx"]


def read_mtx(path):
    """-> (nrows, ncols, rows, cols, vals, symmetric); 0-based indices, file order.  The file is parsed by the extension
    module's native reader (spmatrix.mtx_read_coordinate: up to 8 threads, strtod -- correctly rounded like the
    reference's fscanf("%lg")): 5e6 entries in 0.2 s where the pandas parser this function started with took 6 s."""
    m, n, sym, i, j, v = spmatrix.mtx_read_coordinate(path)
    return m, n, i, j, v, bool(sym)


def _read_mtx_python(path):
    """the same in Python (kept as the checker of the native reader in tests/test_spmatrix_host.py)"""
    with open(path, "r") as f:
        banner = f.readline().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket" or banner[1].lower() != "matrix":
            raise spmatrix.error("not a MatrixMarket matrix file")
        fmt, field, sym = (t.lower() for t in banner[2:5])
        if fmt != "coordinate" or field not in ("real", "integer") or sym not in ("general", "symmetric"):
            raise spmatrix.error("matrix type not supported (need coordinate real general|symmetric)")
        line = f.readline()
        while line.startswith("%") or not line.strip():
            line = f.readline()
        m, n, nz = (int(t) for t in line.split())
        i, j, v = [], [], []
        for line in f:
            t = line.split()
            if not t or t[0].startswith("%"):
                continue
            i.append(int(t[0])), j.append(int(t[1])), v.append(float(t[2]))
    i, j, v = np.array(i, dtype=np.int64), np.array(j, dtype=np.int64), np.array(v, dtype=np.float64)
    if i.size != nz:
        raise spmatrix.error("file holds %d entries, the size line promises %d" % (i.size, nz))
    i, j = i - 1, j - 1
    if nz and (i.min() < 0 or j.min() < 0 or i.max() >= m or j.max() >= n):
        raise IndexError("indices out of range")
    return m, n, i, j, v, sym == "symmetric"


def _sorted_unique(rows, cols, vals, nrows):
    """sort by (row, col); a repeated (row, col) keeps its LAST value in file order (native: counting sort by row, the
    short rows by column, spmatrix.coo_sort_unique)"""
    return spmatrix.coo_sort_unique(rows, cols, vals, nrows)


def _sorted_unique_numpy(rows, cols, vals, ncols):
    """the same with NumPy (the checker of the native sort in tests/test_spmatrix_host.py)"""
    key = rows * ncols + cols  # (row, col) as one int64 key: a single stable sort keeps file order inside equal keys
    order = np.argsort(key, kind="stable")
    rows, cols, vals, key = rows[order], cols[order], vals[order], key[order]
    last = np.ones(rows.size, dtype=bool)
    last[:-1] = key[1:] != key[:-1]
    return rows[last], cols[last], vals[last]


def csr_arrays_from_mtx(path):
    """(shape, indptr, indices, data) of ll_mat_from_mtx(path).to_csr(); zero values are dropped like
    ll_mat assignments of 0.0 (ll_mat.c:250-356 with storeZeros off)"""
    m, n, i, j, v, sym = read_mtx(path)
    if sym:
        if m != n:
            raise ValueError("symmetric matrix must be square")
        if np.any(j > i):
            raise IndexError("write operation to upper triangle of symmetric matrix")  # ll_mat.c:256-260
        i, j, v = _sorted_unique(i, j, v, m)
        off = i != j
        i, j, v = np.concatenate([i, j[off]]), np.concatenate([j, i[off]]), np.concatenate([v, v[off]])
    i, j, v = _sorted_unique(i, j, v, m)
    keep = v != 0.0
    i, j, v = i[keep], j[keep], v[keep]
    indptr = np.zeros(m + 1, dtype=np.int32)
    np.cumsum(np.bincount(i, minlength=m), out=indptr[1:])
    return (m, n), indptr, j.astype(np.int32), v


def sss_arrays_from_mtx(path):
    """(n, indptr, indices, data, diag) of ll_mat_from_mtx(path).to_sss(): strict lower triangle + diagonal"""
    m, n, i, j, v, sym = read_mtx(path)
    if m != n:
        raise ValueError("matrix must be square")
    if sym and np.any(j > i):
        raise IndexError("write operation to upper triangle of symmetric matrix")
    i, j, v = _sorted_unique(i, j, v, n)
    keep = v != 0.0
    i, j, v = i[keep], j[keep], v[keep]
    diag = np.zeros(n)
    d = i == j
    diag[i[d]] = v[d]
    low = i > j  # entries above the diagonal of a general file are dropped, like LLMat_to_sss
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(i[low], minlength=n), out=indptr[1:])
    return n, indptr, j[low].astype(np.int32), v[low], diag


def csr_from_mtx(path):
    shape, indptr, indices, data = csr_arrays_from_mtx(path)
    return spmatrix.csr_from_arrays(indptr, indices, data, shape)


def sss_from_mtx(path):
    n, indptr, indices, data, diag = sss_arrays_from_mtx(path)
    return spmatrix.sss_from_arrays(indptr, indices, data, diag)
