"""cgs / bicgstab / qmrs / gmres.

CPU part: the oracle restatements agree with PCG / a direct solve (their parity against the
reference's own compiled kernels is pinned in tests/test_oracle_krylov_golden.py).
GPU part: the device loops reproduce the oracle's info / iteration counts and iterates."""
import numpy as np
import pytest

SOLVERS = ("cgs", "bicgstab", "qmrs", "gmres")


def nonsym_csr(oracle, n, seed):
    """diagonally dominant non-symmetric tridiagonal-plus-band matrix"""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(n):
        ent = {i: 8.0 + rng.random()}
        for off in (-7, -1, 1, 5):
            j = i + off
            if 0 <= j < n:
                ent[j] = rng.standard_normal()
        for j in sorted(ent):
            rows.append(i), cols.append(j), vals.append(ent[j])
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
    return oracle.CSR((n, n), np.array(vals), np.array(cols, dtype=np.int32), ind)


@pytest.mark.parametrize("solver", SOLVERS)
def test_oracle_more_solvers_reach_the_pcg_solution(oracle, solver):
    A = oracle.poisson_csr(24, 20)
    n = A.shape[0]
    b = np.random.default_rng(2).standard_normal(n)
    xp = np.zeros(n)
    assert oracle.pcg(A, b, xp, 1e-12, 4000)[0] == 0
    for dinv in (None, oracle.jacobi_dinv(A.diagonal())):
        x = np.full(n, 0.5) if solver != "qmrs" else np.zeros(n)
        info, it, rr = oracle.krylov_more(solver, A, b, x, 1e-11, 4000, dinv, dim=25)
        assert info == 0 and 0 < it < 4000
        assert np.abs(x - xp).max() / np.abs(xp).max() < 1e-8


@pytest.mark.parametrize("solver", ("cgs", "bicgstab", "gmres"))
def test_oracle_more_solvers_nonsymmetric(oracle, solver):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A = nonsym_csr(oracle, 400, 3)
    b = np.ones(400)
    xs = spla.spsolve(sp.csr_matrix((A.val, A.col, A.ind), shape=A.shape).tocsc(), b)
    x = np.zeros(400)
    info, it, rr = oracle.krylov_more(solver, A, b, x, 1e-12, 2000, oracle.jacobi_dinv(A.diagonal()), dim=30)
    assert info == 0 and np.abs(x - xs).max() / np.abs(xs).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("solver", SOLVERS)
def test_gpu_more_solvers_match_oracle(oracle, solver):
    from pysparse_amd import device as dev
    fn = getattr(dev, solver)
    cases = [(oracle.poisson_csr(40, 32), True), (oracle.poisson_csr(12, 11, 10), True)]
    if solver != "qmrs":
        cases.append((nonsym_csr(oracle, 3000, 5), False))
    for A, sym in cases:
        n = A.shape[0]
        D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
        b = np.random.default_rng(7).standard_normal(n)
        for use_k in (False, True):
            dinv = oracle.jacobi_dinv(A.diagonal()) if use_k else None
            K = dev.DeviceJacobi(D) if use_k else None
            kw = {"dim": 15} if solver == "gmres" else {}
            # (1) a few iterations from the same start: same counts, iterates to rounding.
            # These recurrences (bicgstab and cgs above all) amplify the reduction-order
            # rounding of the dot products, so identical iteration counts at convergence are
            # not a property even of two CPU BLAS libraries; parity is checked early on ...
            xo = np.full(n, 0.25)
            ro = oracle.krylov_more(solver, A, b, xo, 1e-30, 6, dinv, **kw)
            x = np.full(n, 0.25)
            r = fn(D, b, x, 1e-30, 6, K, **kw)
            assert r[:2] == ro[:2], (solver, r, ro)
            assert np.abs(x - xo).max() / np.abs(xo).max() < 1e-11
            # (2) ... and at convergence by the solution and a comparable iteration count
            xo = np.full(n, 0.25)
            ro = oracle.krylov_more(solver, A, b, xo, 1e-10, 3000, dinv, **kw)
            x = np.full(n, 0.25)
            r = fn(D, b, x, 1e-10, 3000, K, **kw)
            assert r[0] == ro[0] == 0
            assert abs(r[1] - ro[1]) <= max(3, 0.25 * ro[1]), (solver, r, ro)
            assert np.abs(x - xo).max() / np.abs(xo).max() < 1e-7
    # maxit exhausted
    A = oracle.poisson_csr(40, 32)
    n = A.shape[0]
    D = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    b = np.ones(n)
    xo, x = np.zeros(n), np.zeros(n)
    kw = {"dim": 4} if solver == "gmres" else {}
    ro = oracle.krylov_more(solver, A, b, xo, 1e-14, 6, None, **kw)
    r = fn(D, b, x, 1e-14, 6, None, **kw)
    assert r[:2] == ro[:2] and np.abs(x - xo).max() <= 1e-12 * np.abs(xo).max()


@pytest.mark.gpu
def test_gpu_more_solvers_through_the_module(oracle):
    """pysparse.itsolvers.krylov.{cgs,bicgstab,qmrs,gmres} and the ItSolver wrappers
    (itsolvers_util.py:154-178 runs all six on poisson2d_sym(100))."""
    from pysparse.itsolvers import krylov, Qmrs, Cgs, Bicgstab, Gmres
    from pysparse.sparse import spmatrix
    from pysparse.precon import precon
    A = spmatrix.poisson_csr(50, 50)
    O = oracle.poisson_csr(50, 50)
    n = 2500
    b = np.ones(n)
    K = precon.jacobi(A)
    dinv = oracle.jacobi_dinv(O.diagonal())
    for name in SOLVERS:
        xo = np.zeros(n)
        ro = oracle.krylov_more(name, O, b, xo, 1e-9, 2000, dinv, dim=20)
        x = np.zeros(n)
        r = getattr(krylov, name)(A, b, x, 1e-9, 2000, K)
        assert r[0] == ro[0] == 0 and abs(r[1] - ro[1]) <= max(3, 0.25 * ro[1])
        assert np.abs(x - xo).max() / np.abs(xo).max() < 1e-7
    x = np.zeros(n)
    r = krylov.gmres(A, b, x, 1e-9, 2000, None, 35)
    xo = np.zeros(n)
    ro = oracle.krylov_more("gmres", O, b, xo, 1e-9, 2000, None, dim=35)
    assert r[0] == ro[0] == 0 and abs(r[1] - ro[1]) <= 3
    for cls in (Qmrs, Cgs, Bicgstab, Gmres):
        s = cls(A)
        x = np.ones(n)
        s.solve(b, x, 1e-8, 2000, K)
        assert s.lastInfo == 0 and s.nofCalled == 1
        r = np.empty(n)
        A.matvec(x, r)
        assert np.linalg.norm(b - r) < 1e-6 * np.linalg.norm(b)


@pytest.mark.gpu
def test_gpu_more_solvers_unfused_paths_still_match_oracle():
    """cgs / bicgstab / qmrs / gmres run fused vector passes for a native matrix with None / jacobi(1); every other
    operator pair (ssor, jacobi with steps > 1, duck-typed Python operators) keeps the one-kernel-per-BLAS-call loops.
    The switches that select those loops are read once per process, so the oracle comparison above is repeated in a
    fresh interpreter with all four switched off."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PSP_TUNING="1", PSP_CGS_FUSED="0", PSP_BICGSTAB_FUSED="0", PSP_QMRS_FUSED="0", PSP_GMRES_FUSED="0")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", here, "-k",
                        "test_gpu_more_solvers_match_oracle"], env=env, cwd=os.path.dirname(os.path.dirname(here)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.gpu
def test_gpu_device_resident_scalars_give_the_host_scalar_loops_bits():
    """cgs / bicgstab / qmrs (round 5): the recurrences of cgs.c / bicgstab.c / qmrs.c run in the thread that finishes each
    reduction and the host reads the state once per batch of iterations; PSP_KRY_DEVSCAL=0 keeps the loops that read
    every reduced value back.  The same expressions on the same reduced values: info, iteration count, residual and x
    must agree BIT FOR BIT -- converged runs, every small truncation point (the batch boundary at 8 included), with and
    without Jacobi, on a stencil operator (csr_spmv_w4), a general banded one (csr_spmv_w3) and an sss_mat, and for the
    breakdown exits (zero right-hand side / exact start)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);\n"
        "from pysparse_amd import device as dev\n"
        "out = []\n"
        "ops = [dev.DeviceCSR.poisson(60, 50), dev.DeviceSSS.poisson(20, 18, 16)]\n"
        "W = dev.DeviceCSR.poisson(70, 40); W.set_variant(1065154); assert W.kernel_info()[0] == 'csr_spmv_w3'; ops.append(W)\n"
        "for M in ops:\n"
        "    n = M.shape[0] if hasattr(M, 'shape') else M.n\n"
        "    b = np.random.default_rng(3).standard_normal(n)\n"
        "    for K in (None, dev.DeviceJacobi(M)):\n"
        "        for name in ('cgs', 'bicgstab', 'qmrs'):\n"
        "            for tol, mx in [(1e-10, 3000)] + [(0.0, k) for k in (0, 1, 2, 3, 7, 8, 9, 16, 17, 40)]:\n"
        "                x = np.zeros(n); r = getattr(dev, name)(M, b, x, tol, mx, K)\n"
        "                out.append([name, r[0], r[1], float(r[2]).hex(), float(np.abs(x).sum()).hex(), x.tobytes().hex()[:128]])\n"
        "    for name in ('cgs', 'bicgstab', 'qmrs'):\n"  # zero right-hand side; a start that already solves the system
        "        x = np.ones(n); r = getattr(dev, name)(M, np.zeros(n), x, 1e-8, 50, None); out.append([name, 'b=0', r[0], r[1], float(np.abs(x).sum()).hex()])\n"
        "        y = np.empty(n); M.matvec(np.ones(n), y); x = np.ones(n); r = getattr(dev, name)(M, y, x, 1e-8, 50, None); out.append([name, 'exact', r[0], r[1], float(np.abs(x).sum()).hex()])\n"
        "print(json.dumps(out))"
    ) % root
    outs = []
    for env in ({}, {"PSP_KRY_DEVSCAL": "0"}):
        e = dict(os.environ, PSP_TUNING="1")
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert len(outs[0]) == len(outs[1]) > 200
    for a, b in zip(outs[0], outs[1]):
        assert a == b, (a[:4], b[:4])
    assert any(r[1] == 0 for r in outs[0] if isinstance(r[1], int))  # the converged runs converged


@pytest.mark.gpu
def test_gpu_gmres_gram_schmidt_chain_variants_give_the_same_bits():
    """gmres.c:110-116 (modified Gram-Schmidt) three ways: one read-back per step (PSP_GMRES_CHAIN=0), the chain with a
    finishing launch per step (1), the finishing reduction folded into the next step's kernel (2, the default up to 2^20
    rows; round 5).  The same sums in the same order: info, iteration count, residual and x agree BIT FOR BIT -- restarts,
    truncated runs, with and without Jacobi, odd sizes (the V = 1 mapping), a size with several groups of partial sums."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);\n"
        "from pysparse_amd import device as dev\n"
        "out = []\n"
        "for grid in ((60, 50, 0), (37, 31, 0), (20, 18, 17), (640, 401, 0)):\n"
        "    M = dev.DeviceCSR.poisson(*grid); n = M.shape[0]\n"
        "    b = np.random.default_rng(3).standard_normal(n)\n"
        "    for K in (None, dev.DeviceJacobi(M)):\n"
        "        for dim in (5, 20):\n"
        "            for tol, mx in [(1e-9, 400)] + [(0.0, k) for k in (1, 2, dim - 1, dim, dim + 1, 2 * dim + 3)]:\n"
        "                x = np.zeros(n); r = dev.gmres(M, b, x, tol, mx, K, dim)\n"
        "                out.append([grid[0], dim, r[0], r[1], float(r[2]).hex(), float(np.abs(x).sum()).hex(), x.tobytes().hex()[:128]])\n"
        "print(json.dumps(out))"
    ) % root
    outs = []
    for env in ({}, {"PSP_GMRES_CHAIN": "1"}, {"PSP_GMRES_CHAIN": "0"}):
        e = dict(os.environ, PSP_TUNING="1")
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert len(outs[0]) == len(outs[1]) == len(outs[2]) > 100
    for a, b, c in zip(*outs):
        assert a == b == c, (a[:5], b[:5], c[:5])
