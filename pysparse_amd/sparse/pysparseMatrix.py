"""PysparseMatrix -- operator-overloading wrapper around an ll_mat, Python-3 counterpart of
pysparse/sparse/pysparseMatrix.py:61-545 (and its base class sparseMatrix.py:49-165).

`A * x` with a NumPy vector is the matrix-vector product and runs on the GPU (ll_mat.matvec, as in
pysparseMatrix.py:224-272); everything that edits or combines matrices (copy, +, -, scalar *, matrix *
matrix, row / column scaling, take, find, dense export, MatrixMarket export) is host-side sugar over the
ll_mat methods (round 4: copy, scale, row_scale, col_scale, matrixmultiply, export_mtx are the C methods
of pysparse_amd/sparse/src/ll_mat_edit.c; a symmetric right factor of matrix * matrix, which the reference
refuses, goes through scipy.sparse).
"""
import numpy as np

from . import spmatrix

__all__ = ["PysparseMatrix", "PysparseIdentityMatrix", "PysparseSpDiagsMatrix"]


def _triplets(L):
    """(rows, cols, vals) of the entries an ll_mat STORES (lower triangle + diagonal when symmetric)"""
    if L.issym:
        ind, col, val, diag = L.to_sss_arrays()
        rows = np.repeat(np.arange(L.shape[0]), np.diff(ind))
        stored = diag != 0.0  # to_sss_arrays reports 0.0 for an absent diagonal entry
        d = np.nonzero(stored)[0]
        return (np.concatenate([rows, d]), np.concatenate([col.astype(np.int64), d]),
                np.concatenate([val, diag[stored]]))
    ind, col, val = L.to_csr_arrays()
    return np.repeat(np.arange(L.shape[0]), np.diff(ind)), col.astype(np.int64), val


def _new_like(L, size_hint=None):
    n, m = L.shape
    hint = int(size_hint if size_hint is not None else max(L.nnz, 1))
    return spmatrix.ll_mat_sym(n, hint) if L.issym else spmatrix.ll_mat(n, m, hint)


def _general_triplets(L):
    """entries of the full matrix (a symmetric ll_mat mirrored)"""
    r, c, v = _triplets(L)
    if L.issym:
        off = r != c
        return np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, v[off]])
    return r, c, v


class PysparseMatrix:
    """Keywords (pysparseMatrix.py:66-76): nrow, ncol, size, bandwidth, matrix, sizeHint, symmetric,
    storeZeros."""

    __array_ufunc__ = None  # ndarray * A must reach A.__rmul__ with the whole vector (A^T x)

    def __init__(self, **kwargs):
        nrow = kwargs.get("nrow", 0)
        ncol = kwargs.get("ncol", 0)
        bandwidth = kwargs.get("bandwidth", 0)
        matrix = kwargs.get("matrix", None)
        size_hint = kwargs.get("sizeHint", 0)
        store_zeros = bool(kwargs.get("storeZeros", False))
        symmetric = bool(kwargs.get("symmetric", False))
        size = kwargs.get("size", 0)
        if size > 0:
            if (nrow > 0 or ncol > 0) and (size != nrow or size != ncol):
                raise ValueError("size argument was given but does not match nrow and ncol")
            nrow = ncol = size
        if matrix is not None:
            self.matrix = matrix
            return
        if not size_hint:
            size_hint = min(nrow, ncol) if not symmetric else nrow
            if bandwidth > 0:
                size_hint = max(size_hint, bandwidth * (2 * min(nrow, ncol) - bandwidth - 1) // 2)
        size_hint = max(int(size_hint), 1)
        if symmetric and nrow == ncol:
            self.matrix = spmatrix.ll_mat_sym(nrow, size_hint, int(store_zeros))
        else:
            self.matrix = spmatrix.ll_mat(nrow, ncol, size_hint, int(store_zeros))

    # ---- inspection
    def isSymmetric(self):
        return bool(self.matrix.issym)

    def getNnz(self):
        return self.matrix.nnz

    def getMatrix(self):
        return self.matrix

    def getShape(self):
        return self.matrix.shape

    shape = property(getShape)
    nnz = property(getNnz)

    def copy(self):
        return PysparseMatrix(matrix=self.matrix.copy())  # pysparseMatrix.py:98-100

    def __getitem__(self, index):
        m = self.matrix[index]
        return m if isinstance(m, (int, float)) else PysparseMatrix(matrix=m)

    def __setitem__(self, index, value):
        self.matrix[index] = value.matrix if isinstance(value, PysparseMatrix) else value

    def __repr__(self):
        return "<PysparseMatrix %dx%d, %d stored entries%s>" % (self.shape + (self.nnz, ", symmetric" if self.isSymmetric() else ""))

    def __str__(self):
        n, m = self.shape
        if n * m > 400:
            return repr(self)
        a = self.getNumpyArray()
        return "\n".join(" ".join("   ---    " if x == 0.0 else "%9f " % x for x in row) for row in a)

    # ---- arithmetic
    def _combine(self, other, sign):
        if not isinstance(other, PysparseMatrix):
            if other == 0:
                return self.copy()
            raise TypeError("a PysparseMatrix can only be added to another PysparseMatrix (or 0)")
        if self.shape != other.shape:
            raise TypeError("cannot add matrices of shapes %s and %s" % (self.shape, other.shape))
        both_sym = self.isSymmetric() and other.isSymmetric()
        n, m = self.shape
        L = spmatrix.ll_mat_sym(n, self.nnz + other.nnz) if both_sym else spmatrix.ll_mat(n, m, self.nnz + other.nnz)
        get = _triplets if both_sym else _general_triplets
        r, c, v = get(self.matrix)
        if v.size:
            L.put(v, r, c)
        r, c, v = get(other.matrix)
        if v.size:
            L.update_add_at(sign * v, r, c)
        return PysparseMatrix(matrix=L)

    def __add__(self, other):
        return self._combine(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._combine(other, -1.0)

    def __rsub__(self, other):
        return (-self)._combine(other, 1.0)

    def __iadd__(self, other):
        self.matrix = self._combine(other, 1.0).matrix
        return self

    def __isub__(self, other):
        self.matrix = self._combine(other, -1.0).matrix
        return self

    def __neg__(self):
        return self * -1.0

    def __pos__(self):
        return self

    def __mul__(self, other):
        """matrix * scalar, matrix * vector (on the GPU) or matrix * matrix"""
        if isinstance(other, PysparseMatrix):
            if self.shape[1] != other.shape[0]:
                raise TypeError("matrix dimensions do not match for multiplication")
            if not other.isSymmetric():  # what spmatrix.matrixmultiply offers (ll_mat.c:3461-3660), in its summation order
                return PysparseMatrix(matrix=spmatrix.matrixmultiply(self.matrix, other.matrix))
            import scipy.sparse as sp  # a symmetric right factor: NotImplementedError in the reference
            a = sp.csr_matrix((_general_triplets(self.matrix)[2], _general_triplets(self.matrix)[:2]), shape=self.shape)
            b = sp.csr_matrix((_general_triplets(other.matrix)[2], _general_triplets(other.matrix)[:2]), shape=other.shape)
            p = (a @ b).tocoo()
            L = spmatrix.ll_mat(self.shape[0], other.shape[1], max(p.nnz, 1))
            if p.nnz:
                L.put(p.data, p.row.astype(np.int64), p.col.astype(np.int64))
            return PysparseMatrix(matrix=L)
        if isinstance(other, np.ndarray) or isinstance(other, (list, tuple)):
            x = np.ascontiguousarray(other, dtype=np.float64)
            if x.ndim != 1 or x.shape[0] != self.shape[1]:
                raise TypeError("matrix and vector dimensions do not match")
            y = np.empty(self.shape[0])
            self.matrix.matvec(x, y)
            return y
        L = self.matrix.copy()
        L.scale(float(other))
        return PysparseMatrix(matrix=L)

    def __rmul__(self, other):
        """scalar * matrix, or vector * matrix = A^T x"""
        if isinstance(other, np.ndarray):
            x = np.ascontiguousarray(other, dtype=np.float64)
            if x.ndim != 1 or x.shape[0] != self.shape[0]:
                raise TypeError("vector and matrix dimensions do not match")
            y = np.empty(self.shape[1])
            self.matrix.matvec_transp(x, y)
            return y
        return self * other

    def __imul__(self, other):
        self.matrix = (self * float(other)).matrix
        return self

    def matvec(self, x):
        return self * x

    # ---- scaling
    def col_scale(self, v):
        """A := A diag(v)"""
        if self.isSymmetric():
            raise TypeError("row / column scaling of a symmetric ll_mat is not supported")
        self.matrix.col_scale(np.ascontiguousarray(v, dtype=np.float64))

    def row_scale(self, v):
        """A := diag(v) A"""
        if self.isSymmetric():
            raise TypeError("row / column scaling of a symmetric ll_mat is not supported")
        self.matrix.row_scale(np.ascontiguousarray(v, dtype=np.float64))

    # ---- bulk access
    def find(self):
        """(values, rows, cols) of the stored entries"""
        r, c, v = _triplets(self.matrix)
        return v, r, c

    @staticmethod
    def _ids(n, id1, id2):
        if id1 is None:
            id1 = np.arange(n)
        id1 = np.asarray(id1, dtype=np.int64).ravel()
        id2 = id1 if id2 is None else np.asarray(id2, dtype=np.int64).ravel()
        return id1, id2

    def put(self, value, id1=None, id2=None):
        """A[id1[k], id2[k]] = value[k] (scalars broadcast; id2 defaults to id1)"""
        id1, id2 = self._ids(min(self.shape), id1, id2)
        vals = np.broadcast_to(np.asarray(value, dtype=np.float64), id1.shape).copy()
        if vals.size:
            self.matrix.put(vals, id1, id2)

    def putDiagonal(self, vector):
        v = np.atleast_1d(np.asarray(vector, dtype=np.float64))
        k = np.arange(v.size if v.size > 1 else min(self.shape))
        self.put(vector, k, k)

    def take(self, id1=None, id2=None):
        id1, id2 = self._ids(min(self.shape), id1, id2)
        sym = self.isSymmetric()
        out = np.empty(id1.size)
        for k, (i, j) in enumerate(zip(id1.tolist(), id2.tolist())):
            out[k] = self.matrix[(j, i) if sym and i < j else (i, j)]
        return out

    def takeDiagonal(self):
        k = np.arange(min(self.shape))
        return self.take(k, k)

    def addAt(self, vector, id1, id2):
        """A[id1[k], id2[k]] += vector[k]"""
        id1, id2 = self._ids(min(self.shape), id1, id2)
        vals = np.broadcast_to(np.asarray(vector, dtype=np.float64), id1.shape).copy()
        if vals.size:
            self.matrix.update_add_at(vals, id1, id2)

    def addAtDiagonal(self, vector):
        v = np.atleast_1d(np.asarray(vector, dtype=np.float64))
        k = np.arange(v.size if v.size > 1 else min(self.shape))
        self.addAt(vector, k, k)

    def getNumpyArray(self):
        a = np.zeros(self.shape)
        r, c, v = _general_triplets(self.matrix)
        a[r, c] = v
        return a

    def exportMmf(self, filename):
        """MatrixMarket coordinate file: ll_mat.export_mtx (pysparseMatrix.py:470-478; 17 digits: the same doubles come back)"""
        self.matrix.export_mtx(filename, 17)


class PysparseIdentityMatrix(PysparseMatrix):
    def __init__(self, size):
        PysparseMatrix.__init__(self, nrow=size, ncol=size, bandwidth=1, symmetric=True, sizeHint=size)
        k = np.arange(size)
        self.put(np.ones(size), k, k)


class PysparseSpDiagsMatrix(PysparseMatrix):
    """spdiags: vals[k] on the diagonal pos[k] (0 main, > 0 above, < 0 below); vals may be one 1-D array
    per diagonal or a 2-D array with one row per diagonal (truncated like pysparseMatrix.py:521-543)."""

    def __init__(self, size, vals, pos, **kwargs):
        pos = list(pos)
        sym = bool(kwargs.get("symmetric", False))
        PysparseMatrix.__init__(self, size=size, symmetric=sym, sizeHint=max(size * len(pos), 1))
        for d, v in zip(pos, vals):
            if sym and d > 0:
                continue
            m = size - abs(d)
            if m <= 0:
                continue
            v = np.broadcast_to(np.asarray(v, dtype=np.float64), (size,))[:m] if np.ndim(v) == 0 else np.asarray(v, dtype=np.float64)[:m]
            k = np.arange(m)
            self.put(v, k - min(d, 0), k + max(d, 0))
