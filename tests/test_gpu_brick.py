"""GPU: the single-kernel PCG and MINRES loops for 3-D grid operators, the points dealt out in bricks
(pysparse_amd/csrc/psp_mid.hip, pcg_brick_kernel / minres_brick_kernel; pcg.c:91-166, minres.c:96-193).  A contiguous block of rows of a 3-D grid has a whole grid plane as its halo, so the
row-block kernels decline such operators; here a workgroup owns a brick of <= 4096 grid points and exchanges its surface.

Its reductions add in another order than the launch-per-phase loops' (the rows of a span lie in several bricks), so -- as for
psp_coop.hip -- the comparison is with the ORACLE: equal (info, iter), relres to 1e-6 relative, x and the residual history
to 1e-12 / 1e-5, converged runs and truncated ones, with and without Jacobi, cubes and grids whose last bricks are cut,
constant and varying coefficients, csr_mat and sss_mat; runs are bitwise reproducible; a refused launch falls back."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _counts(L):
    s, f = C.c_longlong(), C.c_longlong()
    L.psp_debug_brick_count(C.byref(s), C.byref(f))
    return s.value, f.value


def _varying(oracle, grid, seed):
    """7-point operator with random symmetric couplings and a dominant, varying diagonal on an nx x ny x nz grid"""
    nx, ny, nz = grid
    n = nx * ny * nz
    g = np.random.default_rng(seed)
    import scipy.sparse as sp
    idx = np.arange(n)
    i, j = idx % nx, (idx // nx) % ny
    e1 = -(0.1 + g.random(n - 1)) * (i[:-1] < nx - 1)
    e2 = -(0.1 + g.random(n - nx)) * (j[:n - nx] < ny - 1)
    e3 = -(0.1 + g.random(n - nx * ny))
    S = sp.diags([e3, e2, e1, e1, e2, e3], [-nx * ny, -nx, -1, 1, nx, nx * ny], shape=(n, n), format="csr")
    S = (S + sp.diags(-np.asarray(S.sum(axis=1)).ravel() + 0.05 + g.random(n))).tocsr()
    S.eliminate_zeros()
    S.sort_indices()
    return oracle.CSR(S.shape, S.data, S.indices.astype(np.int32), S.indptr.astype(np.int32))


CASES = [("poisson", (80, 80, 80)), ("poisson", (50, 60, 70)), ("poisson", (47, 101, 33)), ("varying", (56, 56, 56))]


@pytest.mark.parametrize("kind,grid", CASES)
def test_brick_loop_matches_the_oracle(oracle, kind, grid):
    from pysparse_amd import device as dev
    from pysparse_amd._capi import lib
    L = lib()
    O = oracle.poisson_csr(*grid) if kind == "poisson" else _varying(oracle, grid, 7)
    n = O.shape[0]
    D = dev.DeviceCSR.from_arrays(O.shape, O.ind, O.col, O.val) if kind == "varying" else dev.DeviceCSR.poisson(*grid)
    assert D.kernel_info()[0] == "csr_spmv_w4"
    b = np.random.default_rng(1).standard_normal(n)
    dinv = oracle.jacobi_dinv(O.diagonal())
    s0 = _counts(L)
    runs = 0
    for K, dg in ((None, None), (dev.DeviceJacobi(D), dinv)):
        for tol, maxit in ((1e-10, 3000), (0.0, 9), (0.0, 1)):
            for solver, osolver in ((dev.pcg, oracle.pcg), (dev.minres, oracle.minres)):
                xo, xg = np.full(n, 0.5), np.full(n, 0.5)
                ro = osolver(O, b, xo, tol, maxit, dg, hist=True)
                rg = solver(D, b, xg, tol, maxit, K, hist=True)
                runs += 1
                assert rg[:2] == ro[:2], (solver.__name__, tol, rg[:3], ro[:3])
                assert abs(rg[2] - ro[2]) <= 1e-6 * ro[2]
                assert relerr(xg, xo) < 1e-12
                m = np.isfinite(ro[3])
                assert np.array_equal(m, np.isfinite(rg[3]))
                assert np.allclose(rg[3][m], ro[3][m], rtol=1e-5, atol=0)
                xg2 = np.full(n, 0.5)
                rg2 = solver(D, b, xg2, tol, maxit, K, hist=True)
                runs += 1
                assert rg2[:3] == rg[:3] and np.array_equal(xg, xg2)  # fixed reduction order: the same bits every run
    s1 = _counts(L)
    assert s1[0] - s0[0] == runs and s1[1] == s0[1]  # every solve ran as one kernel, none was handed back


def test_sss_form_and_the_launch_per_phase_loops(oracle):
    """the same operator as an sss_mat goes the same way; PSP_BRICK=0 and a refused launch give the launch-per-phase
    loops' result, which the brick loop's agrees with to rounding"""
    code = (
        "import sys, json, ctypes as C, numpy as np; sys.path.insert(0, %r);\n"
        "from pysparse_amd import device as dev, _capi\n"
        "L = _capi.lib(); out = []\n"
        "for form in ('csr', 'sss'):\n"
        "    A = (dev.DeviceCSR if form == 'csr' else dev.DeviceSSS).poisson(50, 54, 60); n = A.shape[0]\n"
        "    b = np.random.default_rng(5).standard_normal(n)\n"
        "    for K in (None, dev.DeviceJacobi(A)):\n"
        "        for tol, mx in ((1e-9, 4000), (0.0, 11)):\n"
        "          for s in (dev.pcg, dev.minres):\n"
        "            x = np.zeros(n); r = s(A, b, x, tol, mx, K)\n"
        "            out.append([r[0], r[1], r[2], float(np.abs(x).max()), x[::max(1, n // 97)].tolist()])\n"
        "s, f = C.c_longlong(), C.c_longlong(); L.psp_debug_brick_count(C.byref(s), C.byref(f)); out.append([s.value, f.value])\n"
        "print(json.dumps(out))"
    ) % ROOT
    res = []
    for env in ({}, {"PSP_TUNING": "1", "PSP_BRICK": "0", "PSP_COOP": "0"}, {"PSP_TUNING": "1", "PSP_COOP_FAIL": "1"}):
        e = dict(os.environ)
        e.pop("PSP_TUNING", None)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert res[0][-1] == [16, 0] and res[1][-1] == [0, 0] and res[2][-1] == [0, 16]
    for a, b in zip(res[0][:-1], res[1][:-1]):
        assert a[:2] == b[:2], (a[:3], b[:3])
        assert abs(a[2] - b[2]) <= 1e-6 * b[2]
        assert np.abs(np.array(a[4]) - np.array(b[4])).max() <= 1e-12 * b[3]
    assert res[2][:-1] == res[1][:-1]  # a refused launch IS the launch-per-phase loop from the same vectors


def test_operators_the_bricks_do_not_take(oracle):
    """a 7-offset operator that couples across the ends of grid lines (a banded matrix, not a grid) and a 2-D grid keep the
    other loops"""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import lib
    import scipy.sparse as sp
    L = lib()
    n, s1, s2 = 60 * 50 * 40, 60, 3000
    g = np.random.default_rng(3)
    offs = [-s2, -s1, -1, 1, s1, s2]
    es = [-(0.1 + g.random(n - abs(o))) for o in (s2, s1, 1)]
    S = sp.diags([es[0], es[1], es[2], es[2], es[1], es[0]], offs, shape=(n, n), format="csr")  # wraps at every line end
    S = (S + sp.diags(-np.asarray(S.sum(axis=1)).ravel() + 0.5)).tocsr()
    S.sort_indices()
    O = oracle.CSR(S.shape, S.data, S.indices.astype(np.int32), S.indptr.astype(np.int32))
    D = dev.DeviceCSR.from_arrays(O.shape, O.ind, O.col, O.val)
    b = g.standard_normal(n)
    s0 = _counts(L)
    xo, xg = np.zeros(n), np.zeros(n)
    ro = oracle.pcg(O, b, xo, 1e-10, 2000, oracle.jacobi_dinv(O.diagonal()))
    rg = dev.pcg(D, b, xg, 1e-10, 2000, dev.DeviceJacobi(D))
    assert rg[:2] == ro[:2] and relerr(xg, xo) < 1e-12
    A2 = dev.DeviceCSR.poisson(300, 300)
    x2 = np.zeros(90000)
    dev.pcg(A2, np.ones(90000), x2, 1e-8, 50, None)
    assert _counts(L) == s0
