"""CPU: pin the oracle (oracle/pysparse_oracle.c) against
  - the golden vectors produced by the COMPILED REFERENCE (tests/golden/, oracle/make_golden.py),
  - the compiled reference itself when oracle/_ref/ is built (this container; the GPU box
    receives the prebuilt files),
  - analytic invariants the reference's own tests assert, and the ten-digit known answer."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "ref_pcg.json")) as f:
        return json.load(f), np.load(os.path.join(golden_dir, "ref_iterates.npz"))


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def setup_case(oracle, name):
    if name.startswith(("G4", "G5")):
        N = 32 if name.startswith("G4") else 64
        A = oracle.poisson_csr(N, N, N)
    else:
        A = oracle.poisson_sss(100, 100) if name.endswith("_sss") else oracle.poisson_csr(100, 100)
    return A


@pytest.mark.parametrize("name", ["G1_none", "G1_jacobi", "G2", "G2_sss", "G3", "G3_sss", "G4_none", "G4_jacobi",
                                  "G5_jacobi", "fixed_1", "fixed_2", "fixed_10", "fixed_50", "fixed_jacobi_10",
                                  "zero_rhs"])
def test_oracle_pcg_matches_reference_golden(oracle, gold, name):
    cases, its = gold
    g = cases[name]
    A = setup_case(oracle, name)
    n = A.shape[0]
    C = A if isinstance(A, oracle.CSR) else oracle.poisson_csr(100, 100)
    if g.get("b", "") == "A*ones":
        b = np.empty(n)
        C.matvec(np.ones(n), b)
    elif name == "zero_rhs":
        b = np.zeros(n)
    else:
        b = np.ones(n)
    dinv = oracle.jacobi_dinv(C.diagonal()) if "jacobi" in name else None
    x = np.zeros(n)
    info, it, relres = oracle.pcg(A, b, x, g.get("tol", 1e-8), g.get("maxit", 10), dinv)
    assert (info, it) == (g["info"], g["iter"])
    if g["relres"] > 0:
        # BLAS-1 summation order (OpenBLAS in the reference build vs sequential here) moves the
        # recurred residual norm by rounding; near the exit threshold that is ~1e-3 relative
        assert abs(relres - g["relres"]) <= 5e-3 * g["relres"]
    else:
        assert relres == 0.0
    for i, v in zip(g["x"]["idx"], g["x"]["val"]):
        assert abs(x[i] - v) <= 1e-12 * max(abs(v), 1e-300) or v == 0.0
    if name in its.files:
        assert relerr(x, its[name]) < 1e-12


def test_oracle_matches_compiled_reference_live(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(4)
    for grid in ((30, 20, 0), (12, 13, 14)):
        A = oracle.poisson_csr(*grid)
        n = A.shape[0]
        b = rng.standard_normal(n)
        for dinv in (None, oracle.jacobi_dinv(A.diagonal())):
            for tol, maxit in ((1e-10, 1000), (0.0, 7)):
                x1, x2 = np.zeros(n), np.zeros(n)
                r1 = oracle.pcg(A, b, x1, tol, maxit, dinv)
                r2 = oracle.ref_pcg(A, b, x2, tol, maxit, dinv)
                assert r1[:2] == r2[:2]
                assert relerr(x1, x2) < 1e-12
    S = oracle.poisson_sss(25, 31)
    n = S.n
    b = rng.standard_normal(n)
    x1, x2 = np.zeros(n), np.zeros(n)
    assert oracle.pcg(S, b, x1, 1e-11, 2000)[:2] == oracle.ref_pcg(S, b, x2, 1e-11, 2000)[:2]
    assert relerr(x1, x2) < 1e-12


def test_standalone_program_golden(golden_dir, oracle):
    with open(os.path.join(golden_dir, "ref_standalone.json")) as f:
        g = json.load(f)
    S = oracle.poisson_sss(100, 100)
    x = np.zeros(S.n)
    info, it, relres = oracle.pcg(S, np.ones(S.n), x, 1e-12, 2000)
    assert (info, it) == (0, g["iter"]) == (0, 225)
    assert abs(relres - g["relres_printed"]) < 0.06e-13 + 5e-3 * relres


def test_sss_and_csr_matvec_agree_and_structure(oracle, golden_dir):
    with open(os.path.join(golden_dir, "structure.json")) as f:
        gold = json.load(f)
    for n in (3, 4, 5, 6):
        g = gold["poisson2d_%d" % n]
        A, S = oracle.poisson_csr(n, n), oracle.poisson_sss(n, n)
        assert A.ind.tolist() == g["csr"]["ind"] and A.col.tolist() == g["csr"]["col"]
        assert A.val.tolist() == g["csr"]["val"]
        assert S.ind.tolist() == g["sss"]["ind"] and S.col.tolist() == g["sss"]["col"]
        assert A.nnz == n * (5 * n - 4) and S.nnz_lower + n * n == n * (3 * n - 2)
        assert g["norm1"] == g["norminf"] == (8.0 if n >= 3 else g["norm1"])
    rng = np.random.default_rng(1)
    T = oracle.tendigit_sss(3000)
    C = oracle.sss_to_csr(T)
    x = rng.standard_normal(3000)
    y1, y2 = np.full(3000, 9.9), np.empty(3000)
    T.matvec(x, y1)
    C.matvec(x, y2)
    assert np.array_equal(y1, y2)  # same per-row summation order (sss_mat.c:45-55)
    yt = np.empty(3000)
    C.matvec_transp(x, yt)
    assert np.allclose(yt, y2, rtol=1e-13, atol=1e-13)


def test_row_parallel_matvec_has_the_one_thread_bits(oracle):
    """bench.py's labelled "all host cores" line (NOT the reference, which is one thread): the rows go to POSIX threads in
    contiguous ranges, each row summed exactly as csr_mat.c:49-54 -- same bits for any thread count, ragged rows, m < threads."""
    rng = np.random.default_rng(5)
    C = oracle.sss_to_csr(oracle.tendigit_sss(3000))  # 1 .. 25 entries per row
    x = rng.standard_normal(3000)
    y1 = np.empty(3000)
    C.matvec(x, y1)
    for t in (1, 2, 3, 7, 16, 300):
        y2 = np.full(3000, np.nan)
        assert C.matvec_threads(x, y2, t) == min(t, 256)
        assert np.array_equal(y1, y2), t
    A = oracle.poisson_csr(2, 2)
    xs, ya, yb = np.arange(4.0), np.empty(4), np.empty(4)
    A.matvec(xs, ya)
    assert A.matvec_threads(xs, yb, 64) == 4 and np.array_equal(ya, yb)


def test_tendigit_known_answer(oracle, golden_dir):
    with open(os.path.join(golden_dir, "tendigit.json")) as f:
        g = json.load(f)
    T = oracle.tendigit_sss(g["n"])
    assert T.nnz_lower == g["nnz_lower"] == 267233
    b = np.zeros(T.n)
    b[0] = 1.0
    x = np.zeros(T.n)
    info, it, relres = oracle.minres(T, b, x, 1e-16, T.n, oracle.jacobi_dinv(T.diag))
    assert info == 0 and abs(x[0] - g["x0_exact"]) < 5e-15
    x = np.zeros(T.n)
    info, it, relres = oracle.pcg(T, b, x, 1e-15, T.n, oracle.jacobi_dinv(T.diag))
    assert info == 0 and abs(x[0] - g["x0_exact"]) < 5e-15


def test_minres_exits(oracle):
    A = oracle.poisson_csr(20, 20)
    n = A.shape[0]
    b = np.ones(n)
    assert oracle.minres(A, b, np.zeros(n), 1e-14, 5)[:2] == (-1, 5)  # it_max reached (minres.c:114)
    neg = -np.ones(n)  # K not SPD -> -3 (minres.c:79-80)
    assert oracle.minres(A, b, np.zeros(n), 1e-8, 50, neg)[0] == -3
    info, it, rr = oracle.minres(A, b, np.zeros(n), 1e-8, 500)
    xp = np.zeros(n)
    assert info == 0 and oracle.pcg(A, b, xp, 1e-10, 500)[0] == 0


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.0, 3), (1.3, 1), (0.7, 2)])
def test_oracle_ssor_identity_against_triangular_solves(oracle, omega, steps):
    """orc_symgs / orc_ssor (preconmodule.c:95-193) have no compilable reference and no golden vector:
    pinned by the iteration they implement, x <- x + w (D + w L)^-1 (b - A x) forward then the same
    with L^T backward, written with SciPy triangular solves (agreement to rounding, not bits)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    S = oracle.poisson_sss(9, 8, 7)
    n = S.n
    A = oracle.sss_to_csr(S)
    M = sp.csr_matrix((A.val, A.col, A.ind), shape=(n, n))
    Lo = sp.tril(M, -1).tocsr()
    D = sp.diags(M.diagonal())
    b = np.random.default_rng(3).standard_normal(n)
    y = np.full(n, 7.0)
    oracle.ssor_apply(S, b, y, omega, steps)
    x = np.zeros(n)
    for _ in range(steps):
        x = x + spl.spsolve_triangular((D / omega + Lo).tocsr(), b - M @ x, lower=True)
        x = x + spl.spsolve_triangular((D / omega + Lo.T).tocsr(), b - M @ x, lower=False)
    assert np.abs(y - x).max() <= 1e-13 * np.abs(x).max()
    # steps = 0 leaves y untouched (for omega == 1 the kernel still does nothing to x)
    y0 = np.full(n, 5.0)
    oracle.ssor_apply(S, b, y0, omega, 0)
    assert np.all(y0 == 5.0)
