"""GPU: the multi-device operator behind the C ABI and the drop-in modules (psp_csr_poisson_multi / psp_csr_create_multi,
pysparse_amd/csrc/psp_multi.hip; SURVEY.md section 8b "multi-GPU variants taking a device list", 8e).

The pool's boxes have ONE GPU, so the device lists here repeat device 0: N ranks with their own streams, row blocks, ghost
copies and fold-kernel reductions -- everything of the N > 1 path except peer copies between different devices and
RCCL between different devices (RCCL itself is exercised with one rank).  Reference loops: pcg.c:91-163,
minres.c:96-193, csr_mat.c:49-54."""
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
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("grid", [(24, 20, 0), (12, 11, 10), (64, 64, 0)])
def test_one_rank_is_the_single_gpu_solver_bit_for_bit(grid):
    """devices=[0]: same kernels, same reductions, same state machine as psp_pcg / psp_minres.  The single-GPU operand is
    the index-free operator (psp_csr_poisson_big, what a rank's slab is): a small csr_mat with CSR arrays would take the
    single-kernel loop of psp_coop.hip, whose reductions are ordered differently (agreement to rounding, tested there)."""
    from pysparse_amd import device as dev
    A1 = dev.DeviceCSR.poisson_big(*grid)
    AM = dev.DeviceCSR.poisson_multi(*grid, devices=[0])
    assert AM.multi_info() == (1, 1, False) and A1.multi_info() == (0, 0, False)
    assert AM.shape == A1.shape and AM.nnz == A1.nnz
    n = A1.shape[0]
    x = np.random.default_rng(0).standard_normal(n)
    y1, ym = np.empty(n), np.empty(n)
    A1.matvec(x, y1)
    AM.matvec(x, ym)
    assert np.array_equal(y1, ym)
    b = np.random.default_rng(1).standard_normal(n)
    for solver in (dev.pcg, dev.minres):
        for K1, KM in ((None, None), (dev.DeviceJacobi(A1), dev.DeviceJacobi(AM))):
            for tol, maxit in ((1e-10, 2000), (0.0, 7)):
                x1, xm = np.zeros(n), np.zeros(n)
                r1 = solver(A1, b, x1, tol, maxit, K1, hist=True)
                rm = solver(AM, b, xm, tol, maxit, KM, hist=True)
                assert r1[:3] == rm[:3], (solver.__name__, r1[:3], rm[:3])
                assert np.array_equal(x1, xm)
                assert np.array_equal(r1[3], rm[3], equal_nan=True)


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_poisson_slabs_on_ranks_sharing_the_gpu(oracle, ranks):
    from pysparse_amd import device as dev
    for grid in ((16, 9, 0), (6, 5, 8), (12, 11, 10)):
        O = oracle.poisson_csr(*grid)
        AM = dev.DeviceCSR.poisson_multi(*grid, devices=[0] * ranks)
        assert AM.multi_info() == (ranks, 1, False)
        assert "multi[%d ranks" % ranks in AM.kernel_info()[0]
        n = O.shape[0]
        x = np.random.default_rng(0).standard_normal(n)
        y, yo = np.empty(n), np.empty(n)
        AM.matvec(x, y)
        O.matvec(x, yo)
        assert np.array_equal(y, yo)  # a row's products are added in the reference's order on whatever rank owns it
        xs = np.random.default_rng(1).standard_normal(2 * n)[::2]  # strided views, like csr_mat.matvec (csr_mat.c:141-163)
        ys = np.zeros(3 * n)[::3]
        AM.matvec(xs, ys)
        O.matvec(np.ascontiguousarray(xs), yo)
        assert np.array_equal(ys, yo)
        assert np.array_equal(AM.diagonal(), O.diagonal())
        b = np.empty(n)
        O.matvec(np.ones(n), b)
        dinv = oracle.jacobi_dinv(O.diagonal())
        KM = dev.DeviceJacobi(AM)
        for K, dg in ((None, None), (KM, dinv)):
            xo, xm = np.zeros(n), np.zeros(n)
            ro = oracle.pcg(O, b, xo, 1e-10, 2000, dg, hist=True)
            rm = dev.pcg(AM, b, xm, 1e-10, 2000, K, hist=True)
            assert rm[:2] == ro[:2] and relerr(xm, xo) < 1e-12
            k = ro[1] + 1
            assert np.allclose(rm[3][:k], ro[3][:k], rtol=1e-5, atol=0)  # the last entries sit at the rounding floor of r
            xo, xm = np.zeros(n), np.zeros(n)
            ro = oracle.minres(O, b, xo, 1e-10, 2000, dg)
            rm = dev.minres(AM, b, xm, 1e-10, 2000, K)
            assert rm[:2] == ro[:2] and abs(rm[2] - ro[2]) <= 1e-8 * ro[2] and relerr(xm, xo) < 1e-12
        # exits: maxit exhausted (iter = maxit + 1, pcg.c:165), zero right-hand side, exact start, minres maxit 0
        assert dev.pcg(AM, b, np.zeros(n), 1e-30, 3)[:2] == (-1, 4)
        xz = np.ones(n)
        assert dev.pcg(AM, np.zeros(n), xz, 1e-9, 10) == (0, 0, 0.0) and not xz.any()
        assert dev.pcg(AM, b, np.ones(n), 1e-8, 10)[:2] == (0, 0)
        assert dev.minres(AM, b, np.ones(n), 1e-9, 0)[:2] == oracle.minres(O, b, np.ones(n), 1e-9, 0)[:2]


def test_two_jacobi_handles_of_one_multi_device_matrix_keep_their_own_omega(oracle):
    """the Jacobi factors of a multi-device matrix live with its row blocks; a second precon.jacobi(A, omega2) must not
    change what the first handle applies (preconmodule.c:352-412: dinv belongs to the jacobi object)"""
    from pysparse_amd import device as dev
    O = oracle.poisson_csr(24, 20, 0)
    n = O.shape[0]
    AM = dev.DeviceCSR.poisson_multi(24, 20, devices=[0, 0, 0])
    K1 = dev.DeviceJacobi(AM, 1.0)
    K2 = dev.DeviceJacobi(AM, 0.5)   # replaces the factors on the row blocks ...
    x = np.random.default_rng(2).standard_normal(n)
    for K, omega in ((K1, 1.0), (K2, 0.5), (K1, 1.0)):   # ... and every use re-establishes the handle's own
        y = np.empty(n)
        K.precon(x, y)
        assert np.array_equal(y, x * oracle.jacobi_dinv(O.diagonal(), omega))
    b = np.empty(n)
    O.matvec(np.ones(n), b)
    for K, omega in ((K2, 0.5), (K1, 1.0), (K2, 0.5)):
        for solver, osolver in ((dev.pcg, oracle.pcg), (dev.minres, oracle.minres)):
            xo, xm = np.zeros(n), np.zeros(n)
            ro = osolver(O, b, xo, 1e-30, 7, oracle.jacobi_dinv(O.diagonal(), omega))
            rm = solver(AM, b, xm, 1e-30, 7, K)
            assert rm[:2] == ro[:2]
            assert relerr(xm, xo) < 1e-12


@pytest.mark.parametrize("ranks", [2, 4])
def test_general_csr_row_blocks(oracle, ranks):
    """irregular coupling across the partition (log-spaced bands: scattered ghost sets, gathered send lists) and a
    non-symmetric band matrix (contiguous ghost ranges)"""
    from pysparse_amd import device as dev
    rng = np.random.default_rng(5)
    S = oracle.tendigit_sss(3000)
    S.diag[:] = 40.0 + rng.random(S.n)
    S.val[:] = rng.standard_normal(S.val.size)
    G = oracle.sss_to_csr(S)
    N = KC.nonsym_csr(oracle, 2500, 3)
    for O, sym in ((G, True), (N, False)):
        n = O.shape[0]
        AM = dev.DeviceCSR.from_arrays_multi(O.shape, O.ind, O.col, O.val, devices=[0] * ranks)
        x = rng.standard_normal(n)
        y, yo = np.empty(n), np.empty(n)
        AM.matvec(x, y)
        O.matvec(x, yo)
        assert np.array_equal(y, yo)
        assert np.array_equal(AM.diagonal(), O.diagonal())
        if not sym:
            continue
        b = np.empty(n)
        O.matvec(np.ones(n), b)
        dinv = oracle.jacobi_dinv(O.diagonal())
        K = dev.DeviceJacobi(AM)
        for solver, osolver in ((dev.pcg, oracle.pcg), (dev.minres, oracle.minres)):
            for tol, maxit in ((1e-11, 500), (1e-30, 6)):
                xo, xm = np.zeros(n), np.zeros(n)
                ro = osolver(O, b, xo, tol, maxit, dinv)
                rm = solver(AM, b, xm, tol, maxit, K)
                assert rm[:2] == ro[:2], (solver.__name__, rm, ro)
                assert relerr(xm, xo) < 1e-12


def test_reference_goldens_through_two_ranks(oracle, golden_dir):
    """G1 (demo_pcg.py: iter 160) and the compiled reference's MINRES vector through a two-rank operator"""
    from pysparse_amd import device as dev
    with open(os.path.join(golden_dir, "ref_krylov.json")) as f:
        cases = json.load(f)["cases"]
    its = np.load(os.path.join(golden_dir, "ref_krylov_iterates.npz"))
    AM = dev.DeviceCSR.poisson_multi(100, 100, devices=[0, 0])
    K = dev.DeviceJacobi(AM)
    for name in ("pcg_G1_jacobi", "pcg_G2", "minres_csr_1e-08_jacobi", "minres_fixed_10", "pcg_fixed_50"):
        case = KC.CASES[name]
        A, b, x, _ = KC.build(oracle, case)
        r = getattr(dev, case["solver"])(AM, b, x, case["tol"], case["maxit"], K if case["K"] else None)
        KC.check_against_golden(name, (r[0], r[1], r[2], x), cases[name]["expect"], its)


def test_drop_in_modules_with_a_device_list(oracle):
    """spmatrix.poisson_csr(..., devices=[...]) -> csr_mat; precon.jacobi(A); krylov.pcg / minres; what a multi-device
    matrix does not offer raises ValueError"""
    from pysparse.sparse import spmatrix
    from pysparse.itsolvers import krylov
    from pysparse.precon import precon
    A = spmatrix.poisson_csr(40, 32, devices=[0, 0, 0])
    A1 = spmatrix.poisson_csr(40, 32)
    assert A.shape == A1.shape == (1280, 1280) and A.nnz == A1.nnz
    n = 1280
    b = np.ones(n)
    K, K1 = precon.jacobi(A), precon.jacobi(A1)
    x, x1 = np.zeros(n), np.zeros(n)
    assert krylov.pcg(A, b, x, 1e-10, 2000, K)[:2] == krylov.pcg(A1, b, x1, 1e-10, 2000, K1)[:2]
    assert relerr(x, x1) < 1e-12
    x, x1 = np.zeros(n), np.zeros(n)
    assert krylov.minres(A, b, x, 1e-10, 2000)[:2] == krylov.minres(A1, b, x1, 1e-10, 2000)[:2]
    assert relerr(x, x1) < 1e-12
    y, y1 = np.empty(n), np.empty(n)
    A.matvec(x, y)
    A1.matvec(x, y1)
    assert np.array_equal(y, y1)
    z, z1 = np.empty(n), np.empty(n)
    K.precon(b, z)
    K1.precon(b, z1)
    assert np.array_equal(z, z1)
    G = oracle.poisson_csr(12, 11, 10)
    B = spmatrix.csr_from_arrays(G.ind, G.col, G.val, G.shape, devices=[0, 0])
    x = np.zeros(G.shape[0])
    xo = np.zeros(G.shape[0])
    bb = np.random.default_rng(2).standard_normal(G.shape[0])
    assert krylov.pcg(B, bb, x, 1e-10, 500, precon.jacobi(B))[:2] == oracle.pcg(G, bb, xo, 1e-10, 500, oracle.jacobi_dinv(G.diagonal()))[:2]
    assert relerr(x, xo) < 1e-12
    # ll_mat.to_csr(devices=[...]): the reference's own construction route (ll_mat.c:1577-1648) onto a device list
    Lm = spmatrix.ll_mat_sym(G.shape[0], 10 * G.shape[0])
    for i in range(G.shape[0]):
        for k in range(G.ind[i], G.ind[i + 1]):
            if G.col[k] <= i:
                Lm[i, int(G.col[k])] = float(G.val[k])
    Cm = Lm.to_csr(devices=[0, 0, 0])
    xm = np.zeros(G.shape[0])
    assert krylov.pcg(Cm, bb, xm, 1e-10, 500, precon.jacobi(Cm))[:2] == krylov.pcg(B, bb, np.zeros(G.shape[0]), 1e-10, 500, precon.jacobi(B))[:2]
    assert relerr(xm, xo) < 1e-12
    for bad in (lambda: A.matvec_transp(b, y), lambda: A.to_arrays(), lambda: krylov.cgs(A, b, x1, 1e-8, 10),
                lambda: krylov.pcg(A, b, x1, 1e-8, 10, K1), lambda: krylov.pcg(A1, b, x1, 1e-8, 10, K),
                lambda: precon.jacobi(A, 1.0, 2), lambda: spmatrix.poisson_csr(8, 8, devices=[7]),
                lambda: spmatrix.poisson_csr(8, 2, devices=[0, 0, 0])):
        with pytest.raises((ValueError, RuntimeError)):
            bad()


def test_rccl_reductions_with_one_rank():
    """the RCCL leg (dlopen of librccl, ncclCommInitAll, grouped ncclAllReduce in stream order) needs one device per rank:
    on this box that is one rank.  PSP_MULTI_REDUCE=rccl (under PSP_TUNING=1) forces it; the iterates must not change."""
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd import device as dev;"
        "A = dev.DeviceCSR.poisson_multi(100, 100, devices=[0]); n = 10000; x = np.zeros(n);"
        "r = dev.pcg(A, np.ones(n), x, 1e-8, 2000, dev.DeviceJacobi(A));"
        "xm = np.zeros(n); rm = dev.minres(A, np.ones(n), xm, 1e-8, 2000);"
        "print(json.dumps([list(A.multi_info()), r[0], r[1], r[2], float(x[0]), float(x[n // 2]), rm[0], rm[1], float(xm[7])]))"
    ) % ROOT
    outs = []
    for env in ({}, {"PSP_TUNING": "1", "PSP_MULTI_REDUCE": "rccl"}):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0][0] == [1, 1, False] and outs[1][0] == [1, 1, True]
    assert outs[0][1:] == outs[1][1:] and outs[0][1:3] == [0, 187]
