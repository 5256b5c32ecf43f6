"""GPU: the single-kernel PCG / MINRES loops for small systems (pysparse_amd/csrc/psp_coop.hip; pcg.c:91-166,
minres.c:96-193): up to 2^18 rows (2^17 until round 4) with at most 8 entries each, native matrix, K = None or jacobi(1).

  * against the oracle: identical info / iteration counts, iterates <= 1e-12, residual histories <= 1e-5 (at the rounding floor), on sizes that
    need 1, a few and many workgroups (every barrier path), 2-D / 3-D stencils and an irregular matrix;
  * against the launch-per-phase loops (PSP_COOP=0 under PSP_TUNING=1, child process): the same answers to rounding;
  * bitwise reproducible from run to run; matrices the kernel does not take (longer rows, more rows) are unaffected.
The exits (-2 / -5 / -6, maxit, zero right-hand side, badly scaled systems) are exercised by tests/test_gpu_solvers.py and
tests/test_gpu_krylov_golden.py, whose small cases all run through these kernels."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import krylov_cases as KC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def small_irregular(oracle, n, seed):
    """symmetric, diagonally dominant, 3..8 entries per row, long-range couplings (gathers across workgroups)"""
    rng = np.random.default_rng(seed)
    ent = [dict() for _ in range(n)]
    for i in range(n):
        for j in rng.choice(n, size=3, replace=False):
            j = int(j)
            if j != i and len(ent[i]) < 6 and len(ent[j]) < 6:
                v = float(rng.standard_normal())
                ent[i][j] = v
                ent[j][i] = v
    rows, cols, vals = [], [], []
    for i in range(n):
        ent[i][i] = 8.0 + sum(abs(v) for v in ent[i].values())
        for j in sorted(ent[i]):
            rows.append(i), cols.append(j), vals.append(ent[i][j])
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=ind[1:])
    return oracle.CSR((n, n), np.array(vals), np.array(cols, dtype=np.int32), ind)


CASES = [("p2d", (31, 29, 0)), ("p2d", (100, 100, 0)), ("p2d", (300, 300, 0)), ("p3d", (20, 19, 18)), ("p3d", (48, 47, 46)),
         ("irr", 5000), ("irr", 70000), ("p2d", (512, 512, 0)), ("p3d", (64, 63, 62))]  # (the last two: 245 / 256 workgroups)


@pytest.mark.parametrize("kind,arg", CASES)
def test_single_kernel_loops_match_the_oracle(oracle, kind, arg):
    from pysparse_amd import device as dev
    O = oracle.poisson_csr(*arg) if kind != "irr" else small_irregular(oracle, arg, 3)
    n = O.shape[0]
    assert n <= (1 << 18) and np.diff(O.ind).max() <= 8
    D = dev.DeviceCSR.from_arrays(O.shape, O.ind, O.col, O.val)
    b = np.random.default_rng(1).standard_normal(n)
    dinv = oracle.jacobi_dinv(O.diagonal())
    for K, dg in ((None, None), (dev.DeviceJacobi(D), dinv)):
        for tol, maxit in ((1e-10, 3000), (0.0, 9)):
            if n > 200000 and K is None and tol > 0.0:
                continue  # the two largest cases converge with the Jacobi operand only (the CPU side of these was 15 s)
            for solver, osolver in ((dev.pcg, oracle.pcg), (dev.minres, oracle.minres)):
                xo, xg = np.full(n, 0.5), np.full(n, 0.5)
                ro = osolver(O, b, xo, tol, maxit, dg, hist=True)
                rg = solver(D, b, xg, tol, maxit, K, hist=True)
                assert rg[:2] == ro[:2], (solver.__name__, tol, rg[:3], ro[:3])
                assert abs(rg[2] - ro[2]) <= 1e-6 * ro[2]
                assert relerr(xg, xo) < 1e-12
                m = np.isfinite(ro[3])
                assert np.array_equal(m, np.isfinite(rg[3]))
                # (a recurred PCG residual ten orders below ||r0|| carries the rounding of the dot sums in its last digits)
                assert np.allclose(rg[3][m], ro[3][m], rtol=1e-5, atol=0)
                xg2 = np.full(n, 0.5)
                rg2 = solver(D, b, xg2, tol, maxit, K, hist=True)
                assert rg2[:3] == rg[:3] and np.array_equal(xg, xg2)  # fixed reduction order: the same bits every run


def test_single_kernel_loops_against_the_launch_per_phase_loops():
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd import device as dev;"
        "out = [];\n"
        "for grid in ((100, 100, 0), (24, 23, 22), (256, 256, 0)):\n"
        "    A = dev.DeviceCSR.poisson(*grid); n = A.shape[0]; b = np.random.default_rng(5).standard_normal(n)\n"
        "    for K in (None, dev.DeviceJacobi(A)):\n"
        "        for s in (dev.pcg, dev.minres):\n"
        "            x = np.zeros(n); r = s(A, b, x, 1e-9, 4000, K)\n"
        "            out.append([r[0], r[1], r[2], float(np.abs(x).max()), x[::max(1, n // 97)].tolist()])\n"
        "print(json.dumps(out))"
    ) % ROOT
    res = []
    # {}: the single-kernel loops; PSP_COOP=0: the launch-per-phase loops; PSP_COOP_FAIL=1: a grid barrier "gave up" (the
    # kernel is not launched, the fall-back path restores r / y and runs the launch-per-phase loops); PSP_COOP_CAPACITY=4:
    # the device "holds" four workgroups at once, so only systems of <= 4096 rows are taken
    for env in ({}, {"PSP_TUNING": "1", "PSP_COOP": "0"}, {"PSP_TUNING": "1", "PSP_COOP_FAIL": "1"},
                {"PSP_TUNING": "1", "PSP_COOP_CAPACITY": "4"}):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for a, b in zip(res[0], res[1]):
        assert a[:2] == b[:2], (a[:3], b[:3])
        assert abs(a[2] - b[2]) <= 1e-6 * b[2]
        assert np.abs(np.array(a[4]) - np.array(b[4])).max() <= 1e-12 * b[3]
    # a refused / failed single-kernel loop IS the launch-per-phase loop from the same vectors: the same bits
    assert res[2] == res[1]
    # capacity 4: every system here has more than 4096 rows -> launch-per-phase loops throughout
    assert res[3] == res[1]


def test_reference_goldens_at_config0_size(oracle, golden_dir):
    """poisson2d(100) -- BASELINE.json configs[0] -- through the single-kernel loops against the vectors of the compiled
    reference kernels (csr and sss operands, None / jacobi)"""
    from test_gpu_krylov_golden import run_gpu
    from pysparse_amd import device as dev
    with open(os.path.join(golden_dir, "ref_krylov.json")) as f:
        cases = json.load(f)["cases"]
    its = np.load(os.path.join(golden_dir, "ref_krylov_iterates.npz"))
    for name in ("pcg_G1_none", "pcg_G1_jacobi", "pcg_G2", "pcg_G3_sss", "pcg_fixed_50", "minres_csr_1e-08_none",
                 "minres_sss_1e-12_jacobi", "minres_fixed_50", "minres_fixed_jacobi_10"):
        got = run_gpu(dev, oracle, KC.CASES[name])
        KC.check_against_golden(name, got, cases[name]["expect"], its, relres_unset_ok=True)


def test_matrices_the_kernel_does_not_take(oracle):
    """rows longer than 8 entries (27-point stencil) and systems beyond 2^18 rows run on the other loops, same answers"""
    from pysparse_amd import device as dev
    A = oracle.poisson_csr(9, 8, 7)
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.ind))
    dense_rows = []
    n = A.shape[0]
    rng = np.random.default_rng(2)
    ind = [0]
    cols, vals = [], []
    for i in range(n):
        c = sorted(set([i] + [int(j) for j in rng.choice(n, size=11, replace=False)]))
        for j in c:
            cols.append(j)
            vals.append(30.0 if j == i else float(rng.standard_normal()) * 0.5)
        ind.append(len(cols))
    M = oracle.CSR((n, n), np.array(vals), np.array(cols, dtype=np.int32), np.array(ind, dtype=np.int32))
    S = oracle.CSR((n, n), *_symmetrize(M))
    D = dev.DeviceCSR.from_arrays(S.shape, S.ind, S.col, S.val)
    assert np.diff(S.ind).max() > 8
    b = rng.standard_normal(n)
    xo, xg = np.zeros(n), np.zeros(n)
    ro = oracle.pcg(S, b, xo, 1e-11, 500, oracle.jacobi_dinv(S.diagonal()))
    rg = dev.pcg(D, b, xg, 1e-11, 500, dev.DeviceJacobi(D))
    assert rg[:2] == ro[:2] and relerr(xg, xo) < 1e-12


def _symmetrize(M):
    import scipy.sparse as sp
    A = sp.csr_matrix((M.val, M.col, M.ind), shape=M.shape)
    A = ((A + A.T) * 0.5).tocsr()
    A.sort_indices()
    return A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32)
