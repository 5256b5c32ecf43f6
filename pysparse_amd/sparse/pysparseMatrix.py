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

__all__ = ["PysparseMatrix", "PysparseIdentityMatrix", "PysparseSpDiagsMatrix", "PysparseMatrix4Scipy"]


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
        return PysparseMatrix(matrix=self.matrix.copy())  # pysparseMatrix.py:126-128

    def __getitem__(self, index):  # :142-147: a number for [i, j], a (general) sub-matrix otherwise
        m = self.matrix[index]
        return m if isinstance(m, (int, float)) else PysparseMatrix(matrix=m)

    def __setitem__(self, index, value):  # :149-154
        self.matrix[index] = value.matrix if isinstance(value, PysparseMatrix) else value

    def __repr__(self):  # sparseMatrix.py:101-102
        return repr(self.matrix)

    def __str__(self):
        """sparseMatrix.py:78-99: cells of width 11, '---' for zeros; beyond 10 000 cells the one-line repr"""
        n, m = self.shape
        if n * m > 10000:
            return repr(self)
        rows = []
        for row in self.getNumpyArray():
            cells = []
            for v in row:
                if v == 0:
                    cells.append("---".center(11))
                elif abs(np.log(abs(v))) <= 4:
                    cells.append(("%9.6f" % v).ljust(11))
                else:
                    cells.append(("%9.2e" % v).ljust(11))
            rows.append("".join(cells))
        return "\n".join(rows)

    # ---- arithmetic (pysparseMatrix.py:156-297)
    def _check_same_shape(self, other, what):
        if self.getShape() != other.getShape():
            raise TypeError("Only sparse matrices of the same size may be %s" % what)

    def _shifted(self, sign, other):
        """self + sign * other as a new ll_mat: copy, generalize when a general matrix joins a symmetric one, shift"""
        L = self.matrix.copy()
        if self.isSymmetric() and not other.isSymmetric():
            L.generalize()
        L.shift(sign, other.getMatrix())
        return L

    def __add__(self, other):
        if isinstance(other, PysparseMatrix):
            self._check_same_shape(other, "added")
            return PysparseMatrix(matrix=self._shifted(1, other))
        if isinstance(other, (int, float, np.integer, np.floating)):
            if other == 0:
                return self
            L = self.copy()  # a number is added to the entries of the nonzero pattern (:183-188)
            val, irow, jcol = L.find()
            L.matrix.update_add_at(other * np.ones(val.shape), irow, jcol)
            return L
        return NotImplemented

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, PysparseMatrix):
            self._check_same_shape(other, "subtracted")
            return PysparseMatrix(matrix=self._shifted(-1, other))
        if isinstance(other, (int, float, np.integer, np.floating)):
            return self.__add__(-other)
        return NotImplemented

    def __rsub__(self, other):
        return (-self).__add__(other)

    def _iadd(self, other, sign):
        if not isinstance(other, PysparseMatrix):
            raise TypeError("in-place addition is with sparse matrices only")
        self._check_same_shape(other, "added")
        if self.isSymmetric() and not other.isSymmetric():
            self.matrix.generalize()
        self.matrix.shift(sign, other.getMatrix())
        return self

    def __iadd__(self, other):
        return self._iadd(other, 1)

    def __isub__(self, other):
        return self._iadd(other, -1)

    def __neg__(self):
        return self * -1

    def __pos__(self):
        return self

    def __mul__(self, other):
        """matrix * matrix (spmatrix.matrixmultiply), matrix * scalar, matrix * vector (ll_mat.matvec: on the GPU)"""
        nrow, ncol = self.getShape()
        if isinstance(other, PysparseMatrix):
            if ncol != other.getShape()[0]:
                raise TypeError("Matrices dimensions do not match for product")
            if not other.isSymmetric():  # what spmatrix.matrixmultiply offers (ll_mat.c:3461-3660), in its summation order
                return PysparseMatrix(matrix=spmatrix.matrixmultiply(self.matrix, other.matrix))
            import scipy.sparse as sp  # a symmetric right factor: NotImplementedError in the reference
            a = sp.csr_matrix((_general_triplets(self.matrix)[2], _general_triplets(self.matrix)[:2]), shape=self.shape)
            b = sp.csr_matrix((_general_triplets(other.matrix)[2], _general_triplets(other.matrix)[:2]), shape=other.shape)
            prod = (a @ b).tocoo()
            L = spmatrix.ll_mat(nrow, other.shape[1], max(prod.nnz, 1))
            if prod.nnz:
                L.put(prod.data, prod.row.astype(np.int64), prod.col.astype(np.int64))
            return PysparseMatrix(matrix=L)
        shape = np.shape(other)
        if shape == ():
            L = self.matrix.copy()
            L.scale(float(other))
            return PysparseMatrix(matrix=L)
        if shape == (ncol,):
            y = np.empty(nrow)
            self.matrix.matvec(np.ascontiguousarray(other, dtype=np.float64), y)
            return y
        raise TypeError("Cannot multiply objects")

    def __rmul__(self, other):
        """scalar * matrix, or vector * matrix = A^T x"""
        if isinstance(other, np.ndarray) and other.ndim == 1:
            if other.shape[0] != self.shape[0]:
                raise TypeError("Cannot multiply objects")
            y = np.empty(self.shape[1])
            self.matrix.matvec_transp(np.ascontiguousarray(other, dtype=np.float64), y)
            return y
        return self * other

    def __imul__(self, other):
        if not isinstance(other, (int, float, np.integer, np.floating)):
            raise TypeError("In-place multiplication is with scalars only")
        self.matrix.scale(float(other))
        return self

    def matvec(self, x):
        return self * x

    # ---- scaling (:299-311)
    def col_scale(self, v):
        """A[:, i] *= v[i] (stored entries; a symmetric matrix would lose its symmetry: TypeError here)"""
        if self.isSymmetric():
            raise TypeError("row / column scaling of a symmetric ll_mat is not supported")
        self.matrix.col_scale(np.ascontiguousarray(v, dtype=np.float64))

    def row_scale(self, v):
        """A[i, :] *= v[i]"""
        if self.isSymmetric():
            raise TypeError("row / column scaling of a symmetric ll_mat is not supported")
        self.matrix.row_scale(np.ascontiguousarray(v, dtype=np.float64))

    # ---- bulk access (:313-478)
    def find(self):
        """(val, irow, jcol) of the stored entries, row by row"""
        return self.matrix.find()

    def _ids(self, value, id1, id2):
        """:358-370: id2 defaults to id1; both default to 0 .. len(value)-1 (0 .. nrow-1 / ncol-1 for a scalar)"""
        nrow, ncol = self.getShape()
        scalar = np.ndim(value) == 0
        if id2 is None and id1 is not None:
            id2 = id1
        if id1 is None:
            id1 = np.arange(nrow if scalar else len(value))
        if id2 is None:
            id2 = np.arange(ncol if scalar else len(value))
        return np.asarray(id1, dtype=np.int64).ravel(), np.asarray(id2, dtype=np.int64).ravel()

    def put(self, value, id1=None, id2=None):
        """A[id1[k], id2[k]] = value[k] (a scalar goes to every position)"""
        id1, id2 = self._ids(value, id1, id2)
        vals = np.broadcast_to(np.asarray(value, dtype=np.float64), id1.shape).copy()
        if vals.size:
            self.matrix.put(vals, id1, id2)

    def putDiagonal(self, vector):
        if np.ndim(vector) == 0:
            k = np.arange(min(self.shape))
            self.put(vector, k, k)
        else:
            v = np.asarray(vector, dtype=np.float64).ravel()
            k = np.arange(v.size)
            self.put(v, k, k)

    def take(self, id1=None, id2=None):
        """val[k] = A[id1[k], id2[k]]"""
        if id1 is None and id2 is None:
            id1 = np.arange(min(self.shape))
        id1, id2 = self._ids(np.empty(0) if id1 is None else id1, id1, id2)
        out = np.zeros(id1.size)
        self.matrix.take(out, id1, id2)
        return out

    def takeDiagonal(self):
        k = np.arange(min(self.shape))
        return self.take(k, k)

    def addAt(self, vector, id1, id2):
        """A[id1[k], id2[k]] += vector[k]"""
        id1 = np.asarray(id1, dtype=np.int64).ravel()
        id2 = np.asarray(id2, dtype=np.int64).ravel()
        vals = np.broadcast_to(np.asarray(vector, dtype=np.float64), id1.shape).copy()
        if vals.size:
            self.matrix.update_add_at(vals, id1, id2)

    def addAtDiagonal(self, vector):
        if np.ndim(vector) == 0:
            k = np.arange(min(self.shape))
            self.addAt(np.full(k.size, float(vector)), k, k)
        else:
            v = np.asarray(vector, dtype=np.float64).ravel()
            k = np.arange(v.size)
            self.addAt(v, k, k)

    def getNumpyArray(self):
        a = np.zeros(self.shape)
        val, irow, jcol = self.matrix.find()
        a[irow, jcol] = val
        if self.isSymmetric():
            a[jcol, irow] = val
        return a

    def exportMmf(self, filename):
        """MatrixMarket coordinate file: ll_mat.export_mtx (:470-478; 17 digits here, so the same doubles come back)"""
        self.matrix.export_mtx(filename, 17)


class PysparseMatrix4Scipy(PysparseMatrix):
    """:545-556: matvec(x, y) with the two-argument signature some SciPy-style callers expect"""

    def matvec(self, x, y):
        return self.matrix.matvec(x, y)


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
