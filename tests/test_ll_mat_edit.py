"""The editing half of ll_mat (pysparse_amd/sparse/src/ll_mat_edit.c) -- host code, no GPU needed.

First part: the reference's own unit tests, test/test_spmatrix.py, transcribed to Python 3 / pytest case by case (same
matrices, same assertions; `llmat_isEqual` is the reference's helper, :10-15) -- everything there except the
matrix-multiply case, which goes through ll_mat.matvec and is the GPU test at the end.
Second part: every method against dense NumPy (the oracle for this file: the definitions in the reference's doc strings,
ll_mat.c, cited per test), symmetric and general storage, the argument errors the reference raises."""
import random

import numpy as np
import pytest

from pysparse_amd.sparse import spmatrix
from pysparse_amd.tools import poisson


def llmat_isEqual(aMat, bMat):  # test/test_spmatrix.py:10-15
    if aMat.issym and not bMat.issym:
        aMat, bMat = bMat, aMat
    zMat = aMat.copy()
    zMat.shift(-1.0, bMat)
    return zMat.nnz == 0


def ll_mat_rand(n, m, density, rng):  # pysparse/tools/spmatrix_util.py:44-56
    nnz = int(density * n * m)
    A = spmatrix.ll_mat(n, m, max(nnz, 1))
    for _ in range(nnz):
        A[rng.randrange(n), rng.randrange(m)] = rng.random()
    return A


def dense(A):
    """the full matrix an ll_mat stands for"""
    val, irow, jcol = A.find()
    D = np.zeros(A.shape)
    D[irow, jcol] = val
    if A.issym:
        D[jcol, irow] = val
    return D


def from_dense(D, sym=False, store_zeros=0):
    n, m = D.shape
    A = spmatrix.ll_mat_sym(n, 10, store_zeros) if sym else spmatrix.ll_mat(n, m, 10, store_zeros)
    for i in range(n):
        for j in range(i + 1 if sym else m):
            if D[i, j] != 0.0:
                A[i, j] = D[i, j]
    return A


# ------------------------------------------------------------------ test/test_spmatrix.py, transcribed

def test_ref_create_and_entry():  # :17-65
    n = 10
    A, S = spmatrix.ll_mat(n, n), spmatrix.ll_mat_sym(n)
    assert A.shape == (n, n) and A.nnz == 0 and not A.issym
    assert S.shape == (n, n) and S.nnz == 0 and S.issym
    A[0, 0] = 1.0
    S[0, 0] = 1.0
    assert A[0, 0] == 1.0 and A.nnz == 1 and S[0, 0] == 1.0 and S.nnz == 1
    with pytest.raises(IndexError):
        S[0, 1] = 1.0
    A[0, 0] += 1.0
    assert A[0, 0] == 2.0 and A.nnz == 1
    A[0, 0] -= 2.0
    assert A[0, 0] == 0.0 and A.nnz == 0
    for key in ((10, 0), (0, -11), (0, 10)):
        with pytest.raises(IndexError):
            S[key] = 1.0
    Q = spmatrix.ll_mat(10, 10, 100)
    for i in range(10):
        for j in range(10):
            Q[i, j] = 10 * i + j
    for i in range(-10, 0):
        for j in range(-10, 0):
            assert Q[i, j] == Q[10 + i, 10 + j]


def test_ref_poisson_basic():  # :68-82
    n = 20
    A, S, B = poisson.poisson2d(n), poisson.poisson2d_sym(n), poisson.poisson2d_sym_blk(n)
    assert S.nnz == n * (3 * n - 2) and A.nnz == n * (5 * n - 4)
    assert llmat_isEqual(A, A) and llmat_isEqual(S, S) and llmat_isEqual(A, S) and llmat_isEqual(A, B)


def test_ref_poisson_blocks_by_submatrix_assignment():
    """the reference's own builder of poisson2d_sym_blk (pysparse/tools/poisson.py:52-66): blocks assigned through slices"""
    n = 12
    L = spmatrix.ll_mat_sym(n * n, 3 * n * n - 2 * n)
    Id = spmatrix.ll_mat_sym(n, n)
    P = spmatrix.ll_mat_sym(n, 2 * n - 1)
    for i in range(n):
        Id[i, i] = -1
        P[i, i] = 4
        if i > 0:
            P[i, i - 1] = -1
    for i in range(0, n * n, n):
        L[i:i + n, i:i + n] = P
        if i > 0:
            L[i:i + n, i - n:i] = Id
    assert llmat_isEqual(L, poisson.poisson2d_sym(n)) and L.nnz == n * (3 * n - 2)


def test_ref_submatrix():  # :84-120
    n = 20
    rng = random.Random(3)
    A = poisson.poisson2d(n)
    Psym, P = poisson.poisson1d_sym(n), poisson.poisson1d(n)
    for i in range(n):
        P[i, i] = 4.0
        Psym[i, i] = 4.0
    for i in range(n):
        blk = A[n * i:n * (i + 1), n * i:n * (i + 1)]
        assert llmat_isEqual(blk, P) and llmat_isEqual(blk, Psym)
    R = ll_mat_rand(n * n, n * n, 0.01, rng)
    for i in range(n):
        s = slice(n * i, n * (i + 1))
        R[s, s] = P
        assert llmat_isEqual(R[s, s], P)
        R[s, s] = Psym
        assert llmat_isEqual(R[s, s], Psym)
    for i in range(n - 1):
        s, t = slice(n * i, n * (i + 1)), slice(n * (i + 1), n * (i + 2))
        R[s, t] = P
        assert llmat_isEqual(R[s, t], P)
        R[s, t] = Psym
        assert llmat_isEqual(R[s, t], Psym)
    R = spmatrix.ll_mat_sym(n * n)
    for i in range(n):
        s = slice(n * i, n * (i + 1))
        R[s, s] = Psym
        assert llmat_isEqual(R[s, s], Psym)
    for i in range(n - 1):
        s, t = slice(n * (i + 1), n * (i + 2)), slice(n * i, n * (i + 1))
        R[s, t] = P
        assert llmat_isEqual(R[s, t], P)
        R[s, t] = Psym
        assert llmat_isEqual(R[s, t], Psym)


@pytest.fixture
def delete_setup():  # :122-139
    n = 30
    P = poisson.poisson1d(n)
    for i in range(n):
        P[i, i] = 4.0
    Id = spmatrix.ll_mat_sym(n)
    for i in range(n):
        Id[i, i] = -1.0
    mask = np.zeros(n ** 2, "l")
    mask[n // 2 * n:(n // 2 + 1) * n] = 1
    mask1 = np.zeros(n ** 2, "l")
    mask1[(n // 2 + 1) * n:(n // 2 + 2) * n] = 1
    return n, P, poisson.poisson2d(n), poisson.poisson2d_sym(n), Id, mask, mask1


def test_ref_delete_rowcols(delete_setup):  # :141-158
    n, P, A, S, Id, mask, mask1 = delete_setup
    S.delete_rowcols(mask)
    assert S.shape == (n, n) and llmat_isEqual(S, P)
    A2 = A.copy()
    A2.delete_rowcols(mask)
    assert llmat_isEqual(A2, P)
    A3 = A.copy()
    A3.delete_rows(mask)
    assert A3.shape == (n, n * n)
    A3.delete_cols(mask)
    assert llmat_isEqual(A3, P)
    A.delete_rows(mask)
    A.delete_cols(mask1)
    assert llmat_isEqual(A, Id)


def test_ref_compress(delete_setup):  # :160-178
    n, P, A, S, Id, mask, mask1 = delete_setup
    A.delete_rows(mask)
    A.delete_cols(mask1)
    norm1 = A.norm("fro")
    freed = A.compress()
    assert freed > 0 and A.norm("fro") == norm1 and A.compress() == 0 and llmat_isEqual(A, Id)
    rng = random.Random(5)
    n = 20
    A = spmatrix.ll_mat(n, n)
    D = np.zeros((n, n))
    for _ in range(20):
        for v in (1.0, 0.0):
            for _ in range(n * n // 2):
                i, j = rng.randrange(n), rng.randrange(n)
                A[i, j] = v
                D[i, j] = v
        A.compress()
        assert np.array_equal(dense(A), D) and A.nnz == int(D.sum())


def test_ref_norms():  # :180-202
    n = 30
    A = poisson.poisson2d(n)
    assert A.norm("1") == 8 and A.norm("inf") == 8
    assert poisson.poisson1d(3).norm("fro") == 4
    S = spmatrix.ll_mat_sym(4)
    S[0, 0], S[1, 1], S[2, 2], S[3, 3] = 1, 2, 3, 4
    S[1, 0], S[2, 0], S[3, 0] = 3, 2, 2
    assert S.norm("fro") == 8
    Ssym = poisson.poisson2d_sym(n)
    for p in ("1", "inf"):
        with pytest.raises(NotImplementedError):
            Ssym.norm(p)
    with pytest.raises(ValueError, match="unknown norm type"):
        A.norm("2")


# ------------------------------------------------------------------ every method against dense NumPy

@pytest.fixture
def mats():
    rng = np.random.default_rng(7)
    G = rng.standard_normal((9, 7)) * (rng.random((9, 7)) < 0.4)
    Sq = rng.standard_normal((8, 8)) * (rng.random((8, 8)) < 0.4)
    Sy = np.tril(Sq) + np.tril(Sq, -1).T
    return rng, G, Sq, Sy


def test_copy_scale_shift_generalize(mats):  # ll_mat.c:1817-1839, :2174-2189, :1984-2031, :1713-1735
    rng, G, Sq, Sy = mats
    A, S = from_dense(G), from_dense(Sy, sym=True)
    C = A.copy()
    C[0, 0] = 99.0
    assert np.array_equal(dense(A), G) and not np.array_equal(dense(C), G) and S.copy().issym
    A.scale(2.5)
    assert np.array_equal(dense(A), G * 2.5)
    B = from_dense(Sq)
    B.shift(0.5, S)  # general += sigma * symmetric: both triangles
    assert np.array_equal(dense(B), Sq + 0.5 * Sy)
    S2 = S.copy()
    S2.shift(-1.0, S)
    assert S2.nnz == 0 and S2.issym
    T = S.copy()
    T.shift(2.0, T)  # the matrix itself
    assert np.array_equal(dense(T), 3.0 * Sy)
    with pytest.raises(NotImplementedError):
        S.shift(1.0, from_dense(Sq))
    with pytest.raises(ValueError, match="matrix shapes do not match"):
        from_dense(G).shift(1.0, from_dense(Sq))
    S.generalize()
    assert not S.issym and np.array_equal(dense(S), Sy) and S.nnz == int((Sy != 0).sum())
    S[0, 7] = 5.0  # above the diagonal is writable now


def test_row_and_col_scale(mats):  # ll_mat.c:1473-1571
    rng, G, Sq, Sy = mats
    A = from_dense(G)
    r, c = rng.standard_normal(9), rng.standard_normal(7)
    A.row_scale(r)
    assert np.array_equal(dense(A), G * r[:, None])
    A.col_scale(c)
    assert np.array_equal(dense(A), (G * r[:, None]) * c[None, :])
    with pytest.raises(spmatrix.error):
        A.row_scale(c)
    with pytest.raises(spmatrix.error):
        A.col_scale(r)


def test_keys_values_items_find_take(mats):  # ll_mat.c:2038-2168, :2999-3035, :2398-2487
    rng, G, Sq, Sy = mats
    A, S = from_dense(G), from_dense(Sy, sym=True)
    rows, cols = np.nonzero(G)
    assert A.keys() == list(zip(rows.tolist(), cols.tolist()))  # row by row, ascending column
    assert A.values() == G[rows, cols].tolist()
    assert A.items() == [((i, j), G[i, j]) for i, j in zip(rows.tolist(), cols.tolist())]
    val, irow, jcol = A.find()
    assert irow.dtype == np.int32 and np.array_equal(irow, rows) and np.array_equal(jcol, cols)
    assert np.array_equal(val, G[rows, cols])
    with pytest.raises(NotImplementedError):
        S.keys()
    with pytest.raises(NotImplementedError):
        S.values()
    lr, lc = np.nonzero(np.tril(Sy))
    assert S.items() == [((i, j), Sy[i, j]) for i, j in zip(lr.tolist(), lc.tolist())]
    b = np.empty(5)
    A.take(b, [0, 3, 8, 2, 2], [1, 3, 6, 0, 5])
    assert np.array_equal(b, G[[0, 3, 8, 2, 2], [1, 3, 6, 0, 5]])
    d = np.empty(7)
    A.take(d)  # the diagonal
    assert np.array_equal(d, np.diag(G)[:7])
    e = np.empty(4)
    S.take(e, [0, 1, 7, 3], [7, 0, 2, 3])  # both triangles of a symmetric matrix
    assert np.array_equal(e, Sy[[0, 1, 7, 3], [7, 0, 2, 3]])
    with pytest.raises(IndexError):
        A.take(b, [0, 1], [0, 1])
    with pytest.raises(TypeError):
        A.take([0.0, 0.0], [0, 1], [0, 1])  # a list cannot receive the values


def test_export_mtx_round_trip(mats, tmp_path):  # ll_mat.c:1757-1810 and :3390-3456
    rng, G, Sq, Sy = mats
    for M, sym in ((G, False), (Sy, True)):
        A = from_dense(M, sym=sym)
        f = str(tmp_path / "a.mtx")
        A.export_mtx(f)
        lines = open(f).read().splitlines()
        assert lines[0] == "%%MatrixMarket matrix coordinate real " + ("symmetric" if sym else "general")
        assert lines[1] == "% file created by pysparse module"
        assert lines[2] == "%d %d %d" % (M.shape[0], M.shape[1], A.nnz) and len(lines) == 3 + A.nnz
        B = spmatrix.ll_mat_from_mtx(f)
        assert B.issym == A.issym and np.allclose(dense(B), M, rtol=1e-15, atol=0.0)  # default: 16 significant digits
        A.export_mtx(f, 17)
        assert np.array_equal(dense(spmatrix.ll_mat_from_mtx(f)), M)  # 17 digits give the same doubles back
        A.export_mtx(f, 3)
        assert np.allclose(dense(spmatrix.ll_mat_from_mtx(f)), M, rtol=1e-2)
    with pytest.raises(IOError):
        A.export_mtx(str(tmp_path / "no" / "such" / "dir.mtx"))


def test_submatrix_read_every_index_kind(mats):  # ll_mat.c:632-884
    rng, G, Sq, Sy = mats
    A, S = from_dense(G), from_dense(Sy, sym=True)
    cases = [(slice(1, 8), slice(0, 7)), (slice(None, None, 2), slice(1, None, 3)), (slice(8, 2, -2), slice(6, None, -1)),
             ([0, 3, 3, 8], [6, 0, 2]), ([2, 5], [1, 1, 4]), (np.array([1, 4, 7]), np.array([0, 6])),
             (3, slice(None)), (slice(None), 2), (-1, [0, 6]), ([1, 2], -2), (slice(2, 2), slice(0, 3))]
    for ri, cj in cases:
        sub = A[ri, cj]
        want = G[np.ix_(np.atleast_1d(np.arange(9)[ri]), np.atleast_1d(np.arange(7)[cj]))]
        assert not sub.issym and sub.shape == want.shape and np.array_equal(dense(sub), want), (ri, cj)
    for ri, cj in ((slice(2, 7), slice(0, 5)), ([0, 7, 3], [7, 0, 5, 5]), (slice(None, None, 3), [1, 6]), (4, slice(None))):
        sub = S[ri, cj]  # a symmetric matrix is read through both triangles; the result is general
        want = Sy[np.ix_(np.atleast_1d(np.arange(8)[ri]), np.atleast_1d(np.arange(8)[cj]))]
        assert not sub.issym and np.array_equal(dense(sub), want), (ri, cj)
    assert isinstance(A[2, 3], float) and A[-1, -1] == G[8, 6]
    for bad, exc in (((0, 7), IndexError), (([0, 9], [0]), IndexError), (([0], [-1]), IndexError), ((0,), IndexError),
                     ((0, 1, 2), IndexError), (3, IndexError), ((1.5, 0), TypeError), ((["a"], [0]), ValueError)):
        with pytest.raises(exc):
            A[bad]


def test_submatrix_write_every_kind(mats):  # ll_mat.c:927-1255
    rng, G, Sq, Sy = mats
    Bd = rng.standard_normal((3, 4)) * (rng.random((3, 4)) < 0.6)
    B = from_dense(Bd)
    # two slices and a matrix: afterwards the block IS the matrix (test/test_spmatrix.py:97-103)
    A, D = from_dense(G), G.copy()
    A[2:5, 1:5] = B
    D[2:5, 1:5] = Bd
    assert np.array_equal(dense(A), D)
    A[8:2:-2, 6:2:-1] = B  # negative steps
    D[8:2:-2, 6:2:-1] = Bd
    assert np.array_equal(dense(A), D) and A.nnz == int((D != 0).sum())
    # lists / arrays / mixtures: every element of the block is written (zeros delete)
    for ri, cj in (([0, 4, 8], [6, 0, 2, 3]), (np.array([1, 2, 3]), np.array([0, 1, 2, 3])), (slice(0, 3), [5, 1, 0, 6])):
        A[ri, cj] = B
        D[np.ix_(np.arange(9)[ri], np.arange(7)[cj])] = Bd
        assert np.array_equal(dense(A), D), (ri, cj)
    # a number fills the block
    A[1:3, [0, 6]] = 7.5
    D[1:3, [0, 6]] = 7.5
    A[:, 3] = 0.0  # ... and zero empties it
    D[:, 3] = 0.0
    assert np.array_equal(dense(A), D) and A.nnz == int((D != 0).sum())
    A[4, :] = 2
    D[4, :] = 2
    assert np.array_equal(dense(A), D)
    with pytest.raises(ValueError, match="Matrix shapes are different"):
        A[0:2, 0:2] = B
    with pytest.raises(ValueError):
        A[0:2, 0:2] = "x"
    with pytest.raises(ValueError, match="Value must be double"):
        A[0, 0] = B
    # a symmetric right-hand side carries both of its triangles into a general matrix, whatever the block
    Sd = Sy[:4, :4]
    Ssm = from_dense(Sd, sym=True)
    A2, D2 = from_dense(Sq), Sq.copy()
    A2[0:4, 4:8] = Ssm
    D2[0:4, 4:8] = Sd
    assert np.array_equal(dense(A2), D2)
    A2[[7, 1, 0, 2], [3, 4, 5, 6]] = Ssm
    D2[np.ix_([7, 1, 0, 2], [3, 4, 5, 6])] = Sd
    assert np.array_equal(dense(A2), D2)
    # a symmetric target: lower triangle only, unless the value is symmetric itself (then its mirror image coincides)
    T, Dt = from_dense(Sy, sym=True), Sy.copy()
    T[4:8, 4:8] = Ssm  # diagonal block
    Dt[4:8, 4:8] = Sd
    assert T.issym and np.array_equal(dense(T), Dt)
    T[4:8, 0:3] = from_dense(Bd.T.copy())  # a block below the diagonal takes a general matrix
    Dt[4:8, 0:3] = Bd.T
    Dt[0:3, 4:8] = Bd
    assert np.array_equal(dense(T), Dt)
    T[0:4, 4:8] = Ssm  # a symmetric value above the diagonal lands mirrored below it
    Dt[0:4, 4:8] = Sd
    Dt[4:8, 0:4] = Sd.T
    assert np.array_equal(dense(T), Dt)
    before = dense(T)
    with pytest.raises(IndexError, match="upper triangle"):
        T[0:4, 4:8] = from_dense(np.ones((4, 4)))
    assert np.array_equal(dense(T), before)  # refused before anything changed
    with pytest.raises(IndexError, match="upper triangle"):
        T[0:2, 0:2] = 1.0
    R = from_dense(Sq)
    R[0:8, 0:8] = R  # the matrix itself
    assert np.array_equal(dense(R), Sq)


def test_delete_argument_checks(mats):  # ll_mat.c:2771-2783, :2826-2836, :2907-2914
    rng, G, Sq, Sy = mats
    A, S = from_dense(G), from_dense(Sy, sym=True)
    keep_r = np.array([1, 0, 1, 1, 0, 0, 1, 1, 1], "l")
    keep_c = np.array([0, 1, 1, 0, 1, 1, 1], dtype=bool)  # any integer / bool dtype
    B = A.copy()
    B.delete_rows(keep_r)
    assert np.array_equal(dense(B), G[keep_r != 0])
    B.delete_cols(keep_c)
    assert np.array_equal(dense(B), G[keep_r != 0][:, keep_c]) and B.nnz == int((dense(B) != 0).sum())
    B[0, 0] = 3.25  # the free list is intact: entries can be added again
    assert B[0, 0] == 3.25
    for bad in (np.zeros(3, "l"), np.zeros(9), np.zeros((9, 1), "l"), [1] * 9):
        with pytest.raises(ValueError, match="mask must be a 1D integer NumPy array"):
            A.delete_rows(bad)
    with pytest.raises(spmatrix.error):
        S.delete_rows(np.ones(8, "l"))
    with pytest.raises(spmatrix.error):
        S.delete_cols(np.ones(8, "l"))
    with pytest.raises(spmatrix.error, match="square"):
        A.delete_rowcols(np.ones(9, "l"))
    m = np.array([1, 1, 0, 1, 0, 1, 1, 0], "l")
    S.delete_rowcols(m)
    assert S.issym and np.array_equal(dense(S), Sy[np.ix_(m != 0, m != 0)])


def test_update_add_mask_and_its_symmetric_twin(mats):  # ll_mat.c:2202-2391
    rng, G, Sq, Sy = mats
    A, D = from_dense(Sq), Sq.copy()
    b = rng.standard_normal((3, 3))
    ind0, ind1 = np.array([0, 5, -1]), np.array([2, 2, 7])
    m0, m1 = np.array([1, 0, 1]), np.array([1, 1, 1])
    A.update_add_mask(b, ind0, ind1, m0, m1)
    for i in range(3):
        for j in range(3):
            if m0[i] and m1[j]:
                D[ind0[i], ind1[j]] += b.ravel()[i + 3 * j]  # the reference reads b at i + len0 * j (ll_mat.c:2262)
    assert np.array_equal(dense(A), D)
    with pytest.raises(spmatrix.error):
        from_dense(Sy, sym=True).update_add_mask(b, ind0, ind1, m0, m1)
    with pytest.raises(IndexError, match="arg 2"):
        A.update_add_mask(b, np.array([0, 8, 1]), ind1, np.ones(3, "l"), m1)
    with pytest.raises(ValueError, match="index and mask"):
        A.update_add_mask(b, ind0, ind1, m0[:2], m1)
    with pytest.raises(ValueError, match="input matrix"):
        A.update_add_mask(b[:2], ind0, ind1, m0, m1)
    bs = b + b.T
    ind, mask = np.array([6, 1, 4]), np.array([1, 1, 1])
    S, Ds = from_dense(Sy, sym=True), Sy.copy()
    Gm, Dg = from_dense(Sq), Sq.copy()
    S.update_add_mask_sym(bs, ind, mask)
    Gm.update_add_mask_sym(bs, ind, mask)
    for i in range(3):
        for j in range(i + 1):
            v = bs.ravel()[i + 3 * j]
            Ds[ind[i], ind[j]] += v
            Dg[ind[i], ind[j]] += v
            if ind[i] != ind[j]:
                Ds[ind[j], ind[i]] += v
                Dg[ind[j], ind[i]] += v
    assert S.issym and np.array_equal(dense(S), Ds) and np.array_equal(dense(Gm), Dg)


def test_products_of_two_matrices(mats):  # ll_mat.c:3461-3796
    rng, G, Sq, Sy = mats
    A, B, S = from_dense(G), from_dense(Sq[:7, :]), from_dense(Sy, sym=True)
    C = spmatrix.matrixmultiply(A, B)

    def rowsum_order(L, R):  # C[i, c] accumulated over L's row in ascending column order: the reference's order
        out = np.zeros((L.shape[0], R.shape[1]))
        for i in range(L.shape[0]):
            for j in np.nonzero(L[i])[0]:
                for c in np.nonzero(R[j])[0]:
                    out[i, c] = L[i, j] * R[j, c] if out[i, c] == 0.0 and not _touched[i, c] else out[i, c] + L[i, j] * R[j, c]
                    _touched[i, c] = True
        return out

    _touched = np.zeros((9, 8), dtype=bool)
    assert not C.issym and C.shape == (9, 8) and np.array_equal(dense(C), rowsum_order(G, Sq[:7, :]))
    assert np.allclose(dense(C), G @ Sq[:7, :], rtol=1e-14, atol=1e-14)
    C2 = spmatrix.matrixmultiply(S, from_dense(Sq))  # symmetric * general
    assert np.allclose(dense(C2), Sy @ Sq, rtol=1e-13, atol=1e-13) and not C2.issym
    with pytest.raises(NotImplementedError):
        spmatrix.matrixmultiply(from_dense(Sq), S)
    with pytest.raises(NotImplementedError):
        spmatrix.matrixmultiply(S, S)
    with pytest.raises(ValueError, match="matrix dimensions must agree"):
        spmatrix.matrixmultiply(A, A)
    Dt = spmatrix.dot(A, from_dense(G))  # A^T * A as a general matrix
    assert Dt.shape == (7, 7) and np.allclose(dense(Dt), G.T @ G, rtol=1e-13, atol=1e-13)
    Sd = spmatrix.symdot(A)
    assert Sd.issym and np.allclose(dense(Sd), G.T @ G, rtol=1e-13, atol=1e-13)
    d = rng.standard_normal(9)
    Sdd = spmatrix.symdot(A, d)
    assert Sdd.issym and np.allclose(dense(Sdd), G.T @ np.diag(d) @ G, rtol=1e-13, atol=1e-13)
    for f in (lambda: spmatrix.dot(S, S), lambda: spmatrix.symdot(S)):
        with pytest.raises(NotImplementedError):
            f()
    with pytest.raises(spmatrix.error):
        spmatrix.symdot(A, np.ones(3))


def test_store_zeros_semantics():  # ll_mat.c:250-356, :362-460: explicit zeros stay when storeZeros is set
    A = spmatrix.ll_mat(3, 3, 4, 1)
    A[0, 0] = 0.0
    A.update_add_at(np.array([0.0]), np.array([1]), np.array([2]))
    assert A.nnz == 2 and A.keys() == [(0, 0), (1, 2)] and A.copy().nnz == 2
    Z = spmatrix.ll_mat(3, 3)
    Z[0, 0] = 0.0
    Z.update_add_at(np.array([0.0]), np.array([1]), np.array([2]))
    assert Z.nnz == 0


def test_empty_and_degenerate_shapes():
    """0 x 0 and n x 0 matrices, empty index sets, everything deleted, negative-step slices over a symmetric matrix"""
    E = spmatrix.ll_mat(0, 0)
    assert E.shape == (0, 0) and E.nnz == 0 and E.norm("fro") == 0.0 and E.keys() == [] and E.find()[0].size == 0
    assert str(E) == "ll_mat(general, [0,0]):\n" and E.compress() == 999 and E.compress() == 0
    Z = spmatrix.ll_mat(3, 0)
    assert Z[0:3, 0:0].shape == (3, 0) and str(Z) == "ll_mat(general, [3,0]):\n\n\n\n"
    A = spmatrix.ll_mat(4, 5, 1)
    A[0:0, :] = 1.0
    assert A.nnz == 0 and A[[], [0, 1]].shape == (0, 2)
    A[[], []] = spmatrix.ll_mat(0, 0)
    A[:, :] = 2.0
    assert A.nnz == 20 and A.norm("1") == 8.0 and A.norm("inf") == 10.0
    A.delete_rows(np.zeros(4, "l"))
    assert A.shape == (0, 5) and A.nnz == 0
    A.delete_cols(np.zeros(5, "l"))
    assert A.shape == (0, 0) and A.compress() > 0
    S = spmatrix.ll_mat_sym(3)
    S[2, 0] = 1
    S[1, 1] = 2
    val, irow, jcol = S[::-1, ::-1].find()  # the full matrix, both axes reversed
    assert val.tolist() == [1.0, 2.0, 1.0] and irow.tolist() == [0, 1, 2] and jcol.tolist() == [2, 1, 0]
    S.delete_rowcols(np.array([0, 1, 0]))
    assert S.shape == (1, 1) and S.items() == [((0, 0), 2.0)]
    assert spmatrix.matrixmultiply(spmatrix.ll_mat(2, 0), spmatrix.ll_mat(0, 3)).shape == (2, 3)
    D = spmatrix.symdot(spmatrix.ll_mat(0, 3))
    assert D.shape == (3, 3) and D.issym and D.nnz == 0
    big = spmatrix.ll_mat(10, 10)
    big[2:8:2, 1::3] = 5
    assert big.nnz == 9 and big[2:8:2, 1::3].nnz == 9 and big[-1:-11:-1, :].shape == (10, 10)
    big.scale(0.0)  # ll_mat.c:2174-2189 multiplies the stored values; it removes nothing
    assert big.nnz == 9 and set(big.values()) == {0.0}
    with pytest.raises(TypeError):
        spmatrix.matrixmultiply(big, 3)
    assert len(spmatrix.ll_mat(70000, 70000)) == 4900000000


def test_random_edit_sequences_against_dense():
    """2000 random edits (element set / update-add, block reads and writes with slices / lists, scaling, shift, row and
    column deletion, compress, generalize) on general and symmetric matrices, the dense NumPy image checked after each:
    the free list, the sorted rows and nnz survive any order of operations (run under ASan by tools/sanitize_host.sh)"""
    rng = np.random.default_rng(17)
    for sym in (False, True):
        n = m = 12
        A = spmatrix.ll_mat_sym(n) if sym else spmatrix.ll_mat(n, m, 3)
        D = np.zeros((n, m))

        def put_dense(i, j, v, add=False):
            D[i, j] = D[i, j] + v if add else v
            if A.issym and i != j:
                D[j, i] = D[i, j]
        for step in range(1000):
            n, m = A.shape
            if n < 4 or m < 4:  # start again from a fresh matrix once deletions have eaten it
                n = m = 12
                A = spmatrix.ll_mat_sym(n) if sym else spmatrix.ll_mat(n, m, 3)
                D = np.zeros((n, m))
            op = rng.integers(0, 11)
            i, j = int(rng.integers(0, n)), int(rng.integers(0, m))
            if A.issym and i < j:
                i, j = j, i
            v = float(rng.choice([0.0, 1.5, -2.25, 3.0]))
            if op <= 2:
                A[i, j] = v
                put_dense(i, j, v)
            elif op == 3:
                A.update_add_at(np.array([v]), np.array([i]), np.array([j]))
                put_dense(i, j, v, add=True)
            elif op == 4 and not A.issym:
                r0, c0 = int(rng.integers(0, n - 2)), int(rng.integers(0, m - 2))
                Bd = rng.integers(-1, 2, (2, 3 if c0 + 3 <= m else 2)).astype(float)
                A[r0:r0 + 2, c0:c0 + Bd.shape[1]] = from_dense(Bd)
                D[r0:r0 + 2, c0:c0 + Bd.shape[1]] = Bd
            elif op == 5:
                ri = sorted(set(rng.integers(0, n, 3).tolist()))
                cj = sorted(set(rng.integers(0, m, 3).tolist()))
                assert np.array_equal(dense(A[ri, cj]), D[np.ix_(ri, cj)])
                assert np.array_equal(dense(A[1:n:2, ::-1]), D[1:n:2, ::-1])
            elif op == 6:
                A.scale(-1.0)
                D *= -1.0
            elif op == 7:
                A.shift(0.5, A.copy())
                D *= 1.5
            elif op == 8 and not A.issym and n > 5:
                mask = (rng.random(n) < 0.85).astype("l")
                A.delete_rows(mask)
                D = D[mask != 0]
            elif op == 9 and n > 5 and n == m:
                mask = (rng.random(n) < 0.85).astype("l")
                A.delete_rowcols(mask)
                D = D[np.ix_(mask != 0, mask != 0)]
            elif op == 10:
                if rng.random() < 0.3 and A.issym:
                    A.generalize()
                else:
                    A.compress()
            assert A.shape == D.shape and np.array_equal(dense(A), D), (sym, step, op)
            nnz_full = int((D != 0).sum())
            assert A.nnz == (int((np.tril(D) != 0).sum()) if A.issym else nnz_full), (sym, step, op)


def test_str_len_and_attributes():  # ll_mat.c:3085-3151 (the text tp_print writes), :3154-3163, :3193-3197
    A = poisson.poisson1d(4)
    A[0, 3] = 12345.678
    A[3, 0] = 1e-7
    assert str(A) == ("ll_mat(general, [4,4]):\n"
                      " 2.000000 -1.000000  --------  12345.68 \n"
                      "-1.000000  2.000000 -1.000000  -------- \n"
                      " -------- -1.000000  2.000000 -1.000000 \n"
                      "  1.0e-07  -------- -1.000000  2.000000 \n")
    assert str(poisson.poisson1d_sym(3)) == ("ll_mat(symmetric, [3,3]):\n 2.000000 \n-1.000000  2.000000 \n"
                                             " -------- -1.000000  2.000000 \n")
    B = spmatrix.ll_mat(600, 30)
    assert str(B) == "ll_mat(general, [600,30])"
    B[1, 2] = 3.5
    B[599, 0] = -1
    assert str(B) == "ll_mat(general, [600,30], [(1,2): 3.5, (599,0): -1])"
    assert len(A) == 16 and len(B) == 18000 and A.storeZeros == 0 and spmatrix.ll_mat(2, 2, 4, 1).storeZeros == 1
    assert repr(A).startswith("<ll_mat object")


@pytest.mark.gpu
def test_ref_matrixmultiply_through_matvec():  # test/test_spmatrix.py:204-221
    eps = 2.2204460492503131e-16
    n, m, k = 30, 60, 30
    rng = random.Random(11)
    nrng = np.random.default_rng(11)
    for _ in range(20):
        A = ll_mat_rand(n, k, 0.9, rng)
        B = ll_mat_rand(k, m, 0.4, rng)
        C = spmatrix.matrixmultiply(A, B)
        t, y1, y2 = np.zeros(k), np.zeros(n), np.zeros(n)
        for _ in range(3):
            x = nrng.random(m)
            C.matvec(x, y1)
            B.matvec(x, t)
            A.matvec(t, y2)
            assert np.sqrt(np.dot(y1 - y2, y1 - y2)) < eps * n * m * k
