"""GPU parity: pcg / minres / jacobi through the C ABI vs the CPU oracle and the golden
vectors produced by the compiled reference (tests/golden/, oracle/make_golden.py).

Bar (BASELINE.json north_star): identical info and iteration counts; fp64 iterates within
1e-12 relative of the reference C path.  Reductions on the GPU are tree-ordered, so x is
compared in the max-norm relative to ||x||_inf, and at FIXED iteration counts as well as
at convergence."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_X = 1e-12


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "ref_pcg.json")) as f:
        cases = json.load(f)
    its = np.load(os.path.join(golden_dir, "ref_iterates.npz"))
    return cases, its


@pytest.fixture(scope="module")
def p2d(oracle):
    from pysparse_amd.device import DeviceCSR, DeviceSSS
    A = oracle.poisson_csr(100, 100)
    return A, DeviceCSR.poisson(100, 100), oracle.poisson_sss(100, 100), DeviceSSS.poisson(100, 100)


def test_demo_pcg_plumbing_G1(oracle, golden, p2d):
    """examples/demo_pcg.py:47-98 on poisson2d(100): b = A*e, tol 1e-6, maxit 2n, None and Jacobi."""
    from pysparse_amd.device import DeviceJacobi, pcg
    cases, its = golden
    A, D, _, _ = p2d
    n = A.shape[0]
    e = np.ones(n)
    b = np.empty(n)
    D.matvec(e, b)
    b_ref = np.empty(n)
    A.matvec(e, b_ref)
    assert np.array_equal(b, b_ref)
    for name, K in (("G1_none", None), ("G1_jacobi", DeviceJacobi(D, 1.0, 1))):
        g = cases[name]
        x = np.zeros(n)
        info, it, relres = pcg(D, b, x, 1e-6, 2 * n, K)
        assert (info, it) == (g["info"], g["iter"]) == (0, 160)
        assert abs(relres - g["relres"]) <= 1e-9 * g["relres"]
        assert relerr(x, its[name]) < RTOL_X
        assert abs(np.abs(x - e).max() - g["err_inf"]) < 1e-12


@pytest.mark.parametrize("name,tol", [("G2", 1e-8), ("G3", 1e-12)])
def test_poisson_test_G2_G3_csr_and_sss(golden, p2d, name, tol):
    """examples/poisson_test.py:50-124 with x0 = 0: CSR and SSS operators, b = ones."""
    from pysparse_amd.device import pcg
    cases, its = golden
    A, D, S, DS = p2d
    n = A.shape[0]
    for M, key in ((D, name), (DS, name + "_sss")):
        g = cases[key]
        x = np.zeros(n)
        info, it, relres = pcg(M, np.ones(n), x, tol, 2000)
        assert (info, it) == (g["info"], g["iter"])
        assert abs(relres - g["relres"]) <= 2e-2 * g["relres"]  # ||r|| at the noise floor of the recurrence
        assert relerr(x, its[name]) < RTOL_X


# SpMV kernel behind the solve: default (csr_spmv_w4 on the Poisson operators), w3, w2
SPMV_KERNELS = {"w4": (-1, "csr_spmv_w4"), "w3": ((1 << 20) + 16578, "csr_spmv_w3"), "w2": (16578, "csr_spmv_w2")}


@pytest.mark.parametrize("kern", sorted(SPMV_KERNELS))
@pytest.mark.parametrize("k", [1, 2, 10, 50])
def test_fixed_iteration_counts(golden, p2d, k, kern):
    """tol = 0 never converges: exactly k iterations, info -1 and iter = k+1 (pcg.c:165)."""
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg
    cases, its = golden
    A = p2d[0]
    D = DeviceCSR.poisson(100, 100)
    D.set_variant(SPMV_KERNELS[kern][0])
    assert D.kernel_info()[0] == SPMV_KERNELS[kern][1]
    n = A.shape[0]
    g = cases["fixed_%d" % k]
    x = np.zeros(n)
    info, it, relres = pcg(D, np.ones(n), x, 0.0, k)
    assert (info, it) == (g["info"], g["iter"]) == (-1, k + 1)
    assert abs(relres - g["relres"]) <= 1e-11 * g["relres"]
    assert relerr(x, its["fixed_%d" % k]) < RTOL_X
    gj = cases["fixed_jacobi_%d" % k]
    b = np.empty(n)
    D.matvec(np.ones(n), b)
    x = np.zeros(n)
    info, it, relres = pcg(D, b, x, 0.0, k, DeviceJacobi(D))
    assert (info, it) == (gj["info"], gj["iter"])
    assert abs(relres - gj["relres"]) <= 1e-11 * gj["relres"]
    assert abs(np.linalg.norm(x) - gj["x"]["norm2"]) <= 1e-12 * gj["x"]["norm2"]


@pytest.mark.parametrize("kern", sorted(SPMV_KERNELS))
@pytest.mark.parametrize("N,name", [(32, "G4"), (64, "G5")])
def test_poisson3d_G4_G5(oracle, golden, N, name, kern):
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg
    cases, _ = golden
    D = DeviceCSR.poisson(N, N, N)
    D.set_variant(SPMV_KERNELS[kern][0])
    assert D.kernel_info()[0] == SPMV_KERNELS[kern][1]
    n = D.shape[0]
    b = np.empty(n)
    D.matvec(np.ones(n), b)
    for suffix, K in (("_none", None), ("_jacobi", DeviceJacobi(D))):
        g = cases[name + suffix]
        x = np.zeros(n)
        info, it, relres, hist = pcg(D, b, x, 1e-8, 2000, K, hist=True)
        assert (info, it) == (g["info"], g["iter"])
        assert abs(relres - g["relres"]) <= 1e-8 * g["relres"]
        assert abs(np.abs(x - 1).max() - g["err_inf"]) < 1e-12
        for i, v in zip(g["x"]["idx"], g["x"]["val"]):
            assert abs(x[i] - v) <= RTOL_X * abs(v)
    # residual history against the oracle, every iteration
    A = oracle.poisson_csr(N, N, N)
    xo = np.zeros(n)
    info_o, it_o, rr_o, hist_o = oracle.pcg(A, b, xo, 1e-8, 2000, oracle.jacobi_dinv(A.diagonal()), hist=True)
    assert (info_o, it_o) == (info, it)
    assert np.allclose(hist[:it + 1], hist_o[:it + 1], rtol=1e-9, atol=0)
    assert relerr(x, xo) < RTOL_X


def test_special_exits(oracle, golden, p2d):
    from pysparse_amd.device import pcg
    cases, _ = golden
    A, D, _, _ = p2d
    n = A.shape[0]
    x = np.full(n, 3.0)
    info, it, relres = pcg(D, np.zeros(n), x, 1e-8, 10)  # b == 0: x := 0, info 0 (pcg.c:58-67)
    g = cases["zero_rhs"]
    assert (info, it, relres) == (g["info"], g["iter"], g["relres"]) == (0, 0, 0.0)
    assert np.all(x == 0)
    b = np.empty(n)
    D.matvec(np.ones(n), b)
    x = np.ones(n)
    info, it, relres = pcg(D, b, x, 1e-8, 10)  # exact initial guess (pcg.c:77-84)
    assert (info, it, relres) == (0, 0, 0.0)
    # maxit exhausted: iter == maxit + 1
    x = np.zeros(n)
    info, it, relres = pcg(D, b, x, 1e-30, 7)
    xo = np.zeros(n)
    assert (info, it) == (-1, 8) == oracle.pcg(A, b, xo, 1e-30, 7)[:2]
    assert relerr(x, xo) < RTOL_X
    # x given as int array: solved on a converted copy, caller's array untouched (itsolversmodule.c:70-76)
    xi = np.zeros(n, dtype=np.int64)
    info, it, _ = pcg(D, b, xi, 1e-6, 1000)
    assert info == 0 and np.all(xi == 0)
    with pytest.raises(ValueError):
        pcg(D, b[:-1], np.zeros(n - 1), 1e-6, 10)


def test_stagnation_and_breakdown_codes(oracle):
    """-5 (stagnation) and -6/-2 (zero scalars) against the oracle on crafted inputs."""
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg
    n = 64
    ind = np.arange(n + 1, dtype=np.int32)
    col = np.arange(n, dtype=np.int32)
    # indefinite diagonal operator with p.Ap == 0 -> breakdown -6
    val = np.ones(n)
    val[n // 2:] = -1.0
    A = oracle.CSR((n, n), val, col, ind)
    D = DeviceCSR.from_arrays((n, n), ind, col, val)
    b = np.ones(n)
    xo, xd = np.zeros(n), np.zeros(n)
    ro = oracle.pcg(A, b, xo, 1e-10, 50)
    rd = pcg(D, b, xd, 1e-10, 50)
    assert ro[:2] == rd[:2] and ro[0] == -6
    # alpha == 0 -> stag = 1 -> -5 (pcg.c:124-125,159-162): q = A p overflows to inf, so
    # pq = inf and alpha = rho/inf = 0 while rho stays finite thanks to a tiny preconditioner
    val = np.full(n, 1e300)
    A = oracle.CSR((n, n), val, col, ind)
    D = DeviceCSR.from_arrays((n, n), ind, col, val)
    b = np.full(n, 1e150)

    class HugeDiag:
        shape = (n, n)

        def __getitem__(self, ij):
            return 1e100

    xo, xd = np.zeros(n), np.zeros(n)
    with np.errstate(all="ignore"):
        ro = oracle.pcg(A, b, xo, 1e-10, 50, np.full(n, 1e-100))
    rd = pcg(D, b, xd, 1e-10, 50, DeviceJacobi(HugeDiag()))
    assert ro[:2] == rd[:2] == (-5, 1)
    assert rd[2] == ro[2] == 1.0  # daxpy quick return for alpha == 0 leaves r untouched
    # rho == 0 -> -2 (pcg.c:101-104): indefinite preconditioner with r.z == 0
    val = np.ones(n)
    A = oracle.CSR((n, n), val, col, ind)
    D = DeviceCSR.from_arrays((n, n), ind, col, val)
    dinv = np.ones(n)
    dinv[n // 2:] = -1.0

    class SignDiag:
        shape = (n, n)

        def __getitem__(self, ij):
            return 1.0 if ij[0] < n // 2 else -1.0

    b = np.ones(n)
    xo, xd = np.zeros(n), np.zeros(n)
    ro = oracle.pcg(A, b, xo, 1e-10, 50, dinv)
    rd = pcg(D, b, xd, 1e-10, 50, DeviceJacobi(SignDiag()))
    assert ro[:2] == rd[:2] == (-2, 1)


def test_stagnation_scan_kernel():
    """pcg.c:127-139 on crafted vectors through psp_k_xr_update: out[2] != 0 <=> 1 + dmax != 1."""
    import ctypes as C
    from pysparse_amd.device import DeviceBuffer
    from pysparse_amd._capi import check, lib

    def scan(alpha, p, x):
        dmax = 0.0
        for pi, xi in zip(p, x):
            if xi != 0.0:
                d = abs(alpha * pi / xi)
                if d > dmax:
                    dmax = d
            elif pi != 0.0:
                dmax = 1.0
        return 1.0 + dmax != 1.0

    n = 1000
    rng = np.random.default_rng(3)
    cases = []
    x = rng.standard_normal(n) + 3.0
    cases.append((1.0, x * 1e-17, x))                      # every update below eps/2: stagnated
    p = x * 1e-17
    p[777] = x[777] * 1e-15
    cases.append((1.0, p, x))                              # one component still moves
    xz = x.copy()
    xz[5] = 0.0
    pz = x * 1e-18
    cases.append((1.0, pz, xz))                            # x_i == 0 with p_i != 0 -> dmax = 1
    pz2 = pz.copy()
    pz2[5] = 0.0
    cases.append((1.0, pz2, xz))                           # x_i == 0 and p_i == 0: ignored
    cases.append((2.0 ** -53, x.copy(), x))                # |alpha p/x| == 2^-53 exactly: tie rounds to 1
    cases.append((2.0 ** -53 * (1 + 2.0 ** -52), x.copy(), x))
    for alpha, p, x0 in cases:
        q = rng.standard_normal(n)
        r = rng.standard_normal(n)
        dp, dq = DeviceBuffer.from_host(p), DeviceBuffer.from_host(q)
        dx, dr = DeviceBuffer.from_host(x0), DeviceBuffer.from_host(r)
        out = DeviceBuffer(4)
        check(lib().psp_k_xr_update(n, alpha, dp.ptr, dq.ptr, None, dx.ptr, dr.ptr, out.ptr))
        o = out.download()
        assert (o[2] != 0.0) == scan(alpha, p, x0)
        assert np.array_equal(dx.download(), x0 + alpha * p)
        r_new = r + (-alpha) * q
        assert np.array_equal(dr.download(), r_new)
        assert abs(o[0] - np.dot(r_new, r_new)) <= 1e-13 * o[0] and o[1] == o[0]


def test_jacobi_object(oracle, p2d):
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, DeviceSSS
    A, D, S, DS = p2d
    n = A.shape[0]
    x = np.random.default_rng(1).standard_normal(n)
    for omega, steps in ((1.0, 1), (0.7, 1), (1.0, 3), (0.8, 2)):
        dinv = oracle.jacobi_dinv(A.diagonal(), omega)
        # oracle: preconmodule.c:35-54
        y_ref = x * dinv
        tmp = np.empty(n)
        for _ in range(1, steps):
            t = y_ref.copy()
            A.matvec(t, tmp)
            y_ref = (x - tmp) * dinv + t
        for M in (D, DS):
            K = DeviceJacobi(M, omega, steps)
            assert K.shape == (n, n)
            y = np.full(n, np.nan)
            K.precon(x, y)
            assert np.array_equal(y, y_ref)
    with pytest.raises(ValueError):
        DeviceJacobi(D).precon(x[::2], np.zeros(n // 2))  # precon needs contiguous args (spmatrix.h:18-36)
    # singular diagonal -> ValueError("diagonal element close to zero") (preconmodule.c:395-397)
    ind = np.arange(4, dtype=np.int32)
    Z = DeviceCSR.from_arrays((3, 3), ind, np.arange(3, dtype=np.int32), np.array([1.0, 1e-20, 2.0]))
    with pytest.raises(ValueError, match="close to zero"):
        DeviceJacobi(Z)


def test_pcg_multistep_jacobi_and_callbacks(oracle, p2d):
    """generic loop: jacobi(steps=2) and duck-typed Python operators (spmatrixmodule.c:169-248)."""
    from pysparse_amd.device import DeviceJacobi, pcg
    A, D, _, _ = p2d
    n = A.shape[0]
    b = np.ones(n)
    dinv = oracle.jacobi_dinv(A.diagonal(), 0.9)
    xo = np.zeros(n)
    ro = oracle.pcg(A, b, xo, 1e-9, 2000, dinv, steps=2)
    x = np.zeros(n)
    r = pcg(D, b, x, 1e-9, 2000, DeviceJacobi(D, 0.9, 2))
    assert r[:2] == ro[:2] and relerr(x, xo) < RTOL_X

    class PyOp:  # examples/fixme/pysparse_test.py:143-151 style user operator
        shape = (n, n)

        def matvec(self, xx, yy):
            A.matvec(np.ascontiguousarray(xx), yy)

    class PyDiag:
        shape = (n, n)

        def precon(self, xx, yy):
            yy[:] = xx * 0.25

    xo = np.zeros(n)
    ro = oracle.pcg(A, b, xo, 1e-9, 2000, np.full(n, 0.25))
    x = np.zeros(n)
    r = pcg(PyOp(), b, x, 1e-9, 2000, PyDiag())
    assert r[:2] == ro[:2] and relerr(x, xo) < RTOL_X

    class Bad:
        shape = (n, n)

        def matvec(self, xx, yy):
            raise KeyError("boom")

    with pytest.raises(KeyError):
        pcg(Bad(), b, np.zeros(n), 1e-9, 10)

    class NotSquare:
        shape = (n, n + 1)

    with pytest.raises(ValueError, match="not square"):
        pcg(NotSquare(), b, np.zeros(n), 1e-9, 10)


def test_minres_tendigit_known_answer(oracle, golden_dir):
    """K1: examples/tendigit.py -- sss_mat + MINRES on an irregular-degree matrix."""
    from pysparse_amd.device import DeviceJacobi, DeviceSSS, minres
    with open(os.path.join(golden_dir, "tendigit.json")) as f:
        g = json.load(f)
    S = oracle.tendigit_sss(g["n"])
    assert S.nnz_lower == g["nnz_lower"]
    D = DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    b = np.zeros(S.n)
    b[0] = 1.0
    x = np.zeros(S.n)
    info, it, relres = minres(D, b, x, 1e-16, S.n, DeviceJacobi(D))
    xo = np.zeros(S.n)
    io, ito, rro = oracle.minres(S, b, xo, 1e-16, S.n, oracle.jacobi_dinv(S.diag))
    assert (info, it) == (io, ito) and info == 0
    assert abs(x[0] - g["x0_exact"]) < 5e-15
    assert relerr(x, xo) < RTOL_X
    # unpreconditioned: thousands of iterations, still the oracle's count
    x = np.zeros(S.n)
    info, it, relres, hist = minres(D, b, x, 1e-10, 5000, hist=True)
    xo = np.zeros(S.n)
    io, ito, rro, hist_o = oracle.minres(S, b, xo, 1e-10, 5000, hist=True)
    assert (info, it) == (io, ito)
    assert np.allclose(hist[:it + 1], hist_o[:it + 1], rtol=1e-8)
    assert relerr(x, xo) < 1e-10  # long Lanczos recurrences amplify reduction-order rounding


def test_minres_poisson_vs_oracle(oracle, p2d):
    from pysparse_amd.device import DeviceJacobi, minres
    A, D, S, DS = p2d
    n = A.shape[0]
    b = np.ones(n)
    for M, K, dinv in ((D, None, None), (DS, None, None), (D, DeviceJacobi(D), oracle.jacobi_dinv(A.diagonal()))):
        x = np.zeros(n)
        info, it, relres, hist = minres(M, b, x, 1e-8, 2000, K, hist=True)
        xo = np.zeros(n)
        io, ito, rro, hist_o = oracle.minres(A, b, xo, 1e-8, 2000, dinv, hist=True)
        assert (info, it) == (io, ito) and info == 0
        assert abs(relres - rro) <= 1e-9 * rro
        assert relerr(x, xo) < RTOL_X
    # it_max hit: info -1, iter == it_max (minres.c:114)
    x = np.zeros(n)
    info, it, relres = minres(D, b, x, 1e-14, 5)
    assert (info, it) == (-1, 5) == oracle.minres(A, b, np.zeros(n), 1e-14, 5)[:2]


def test_pcg_loop_variants_agree(golden, p2d, monkeypatch):
    """host-scalar loop (PSP_PCG_ASYNC=0), device-scalar loop (default), its hipGraph replay
    (PSP_PCG_GRAPH=1), the p-update folded into the SpMV (PSP_PCG_PFUSED=1, both loops), the dinv
    stream kept (PSP_DINV_CONST=0) and the lazy x update (PSP_PCG_LAZYX=2 forces it at this size; it is
    the default from 2^25 unknowns on: x update folded into the next p-update pass) are the same algorithm: identical info / iteration counts / iterates."""
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg;"
        "D = DeviceCSR.poisson(100, 100); n = 10000; x = np.zeros(n);"
        "r = pcg(D, np.ones(n), x, 1e-8, 2000, DeviceJacobi(D), hist=True);"
        "print(json.dumps([r[0], r[1], r[2], float(x[0]), float(x[n // 2]), float(np.nansum(r[3]))]))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({"PSP_PCG_ASYNC": "0"}, {}, {"PSP_PCG_GRAPH": "1"}, {"PSP_PCG_PFUSED": "1"},
                {"PSP_PCG_PFUSED": "1", "PSP_PCG_ASYNC": "0"}, {"PSP_DINV_CONST": "0"},
                {"PSP_DINV_CONST": "0", "PSP_PCG_PFUSED": "1"}, {"PSP_PCG_LAZYX": "2"},
                {"PSP_PCG_LAZYX": "2", "PSP_DINV_CONST": "0"},
                # round 5: the lazy loop with p update AND pending x update folded into the product (csr_spmv_w4_pf<XU>)
                {"PSP_PCG_LAZYX": "2", "PSP_PCG_LAZYPF": "1"}, {"PSP_PCG_LAZYX": "2", "PSP_PCG_LAZYPF": "0"},
                {"PSP_PCG_LAZYX": "2", "PSP_PCG_LAZYPF": "1", "PSP_DINV_CONST": "0"}):
        # PSP_TUNING: the master switch that makes the library read its A/B variables; PSP_COOP=0: these tests compare
        # the launch-per-phase loops with each other (at this size the default is the single-kernel loop, psp_coop.hip)
        e = dict(os.environ, PSP_TUNING="1", PSP_COOP="0")
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout
        outs.append(json.loads(out.strip().splitlines()[-1]))
    cases, _ = golden
    for o in outs:
        assert (o[0], o[1]) == (0, cases["G2"]["iter"])
    assert all(o == outs[0] for o in outs)  # bitwise: same products, same reduction order


def test_minres_loop_variants_agree(oracle):
    """MINRES with device-resident scalars (default: Lanczos / Givens recurrences evaluated by the thread that
    finishes each reduction, 16 iterations enqueued per read of the state) against the host-scalar loop
    (PSP_MINRES_ASYNC=0): the same algorithm -- identical info / iteration counts / residual histories /
    iterates, on the scaled index-free path (stencil csr and sss), without it (PSP_MINRES_SCALED=0), with csr_spmv_w3
    operators (a stencil forced to it, a banded matrix without stencil structure) and on a general CSR matrix
    (csr_spmv_w2, variant 16578); with and without Jacobi; converged, truncated at every small maxit, and maxit = 0."""
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd.device import DeviceCSR, DeviceSSS, DeviceJacobi, minres;"
        "out = [];\n"
        "for M in (DeviceCSR.poisson(60, 50), DeviceSSS.poisson(20, 18, 16), 'w2', 'w3', 'w3band'):\n"
        "    if M == 'w2':\n"
        "        M = DeviceCSR.poisson(60, 50); M.set_variant(16578)\n"
        "    if M == 'w3':\n"  # csr_spmv_w3 (v = y / beta stays a pass of its own there: profiles/r4_minres_w3_scaled_ab.txt)
        "        M = DeviceCSR.poisson(60, 50); M.set_variant(1065154); assert M.kernel_info()[0] == 'csr_spmv_w3'\n"
        "    if M == 'w3band':\n"  # a banded SPD matrix with no stencil structure, a few far couplings (outlier chunks)
        "        import scipy.sparse as sp; g = np.random.default_rng(5); nn = 6000\n"
        "        r = np.repeat(np.arange(nn), 6); c = r - g.integers(1, 40, r.size); k = c >= 0; r, c = r[k], c[k]\n"
        "        far = g.integers(3000, nn, 40); r = np.concatenate([r, far]); c = np.concatenate([c, far - g.integers(2000, 3000, 40)])\n"
        "        L = sp.coo_matrix((-(0.1 + g.random(r.size)), (r, c)), shape=(nn, nn)).tocsr(); L.sum_duplicates()\n"
        "        S = (L + L.T).tocsr(); S = (S + sp.diags(1.0 - np.asarray(S.sum(axis=1)).ravel())).tocsr(); S.sort_indices()\n"
        "        M = DeviceCSR.from_arrays(S.shape, S.indptr, S.indices, S.data); assert M.kernel_info()[0].startswith('csr_spmv_w3')\n"
        "    n = M.shape[0]; b = np.random.default_rng(3).standard_normal(n)\n"
        "    for K in (None, DeviceJacobi(M)):\n"
        "        for tol, mx in [(1e-9, 4000)] + [(0.0, k) for k in range(0, 40, 3)]:\n"
        "            x = np.zeros(n); r = minres(M, b, x, tol, mx, K, hist=True)\n"
        "            out.append([r[0], r[1], r[2], float(x[0]), float(x[n // 2]), float(np.abs(x).sum()),"
        " [float(t) for t in r[3] if t == t]])\n"
        "print(json.dumps(out))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({"PSP_MINRES_ASYNC": "0"}, {}, {"PSP_MINRES_SCALED": "0"},
                {"PSP_MINRES_SCALED": "0", "PSP_MINRES_ASYNC": "0"}):
        # PSP_TUNING: the master switch that makes the library read its A/B variables; PSP_COOP=0: these tests compare
        # the launch-per-phase loops with each other (at this size the default is the single-kernel loop, psp_coop.hip)
        e = dict(os.environ, PSP_TUNING="1", PSP_COOP="0")
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout
        outs.append(json.loads(out.strip().splitlines()[-1]))
    assert outs[0][0][0] == 0 and outs[0][0][1] > 50  # the first case converges
    assert all(o == outs[0] for o in outs)  # bitwise: same products, same reduction order


def test_scalar_readback_routes_agree():
    """the host-scalar loops (cgs, bicgstab, qmrs, gmres, and pcg / minres with PSP_*_ASYNC=0) read their scalars back through
    mapped host memory that the host polls (round 4, psp_runtime.hip fetch_scalars / finish_partials_fetch); the copy +
    stream synchronisation of rounds 1-3 (PSP_FETCH_POLL=0) must give the same numbers -- sizes on both sides of the
    one-block / group-fold boundary of the reductions"""
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd import device as dev;"
        "out = [];\n"
        "for grid in ((60, 50, 0), (1500, 1500, 0)):\n"
        "    A = dev.DeviceCSR.poisson(*grid); n = A.shape[0]; K = dev.DeviceJacobi(A)\n"
        "    b = np.random.default_rng(2).standard_normal(n)\n"
        "    for f in (dev.pcg, dev.minres, dev.cgs, dev.bicgstab, dev.qmrs, dev.gmres):\n"
        "        x = np.zeros(n); r = f(A, b, x, 0.0, 12, K)\n"
        "        out.append([r[0], r[1], r[2], float(x[0]), float(x[n // 2]), float(np.abs(x).sum())])\n"
        "print(json.dumps(out))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"PSP_FETCH_POLL": "0"}, {"PSP_PCG_ASYNC": "0", "PSP_MINRES_ASYNC": "0"},
                {"PSP_PCG_ASYNC": "0", "PSP_MINRES_ASYNC": "0", "PSP_FETCH_POLL": "0"}):
        e = dict(os.environ, PSP_TUNING="1", PSP_COOP="0")
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout
        outs.append(json.loads(out.strip().splitlines()[-1]))
    assert len(outs[0]) == 12 and all(o == outs[0] for o in outs)


def test_minres_async_special_exits(oracle):
    """exits of the device-resident MINRES loop: -3 (indefinite preconditioner: beta^2 < 0 inside the loop),
    a zero right-hand side (norm_r0 = 0: NaN relres like the reference), maxit = 0"""
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, minres
    A = oracle.poisson_csr(12, 10)
    n = A.shape[0]
    D = DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)

    class Indef:  # preconditioner that is not positive definite -> generic path, -3 at setup or in the loop
        shape = (n, n)

        def precon(self, x, y):
            y[:] = -x
    b = np.random.default_rng(0).standard_normal(n)
    x = np.zeros(n)
    assert minres(D, b, x, 1e-9, 50, Indef())[0] == -3
    # variable-sign diagonal through the fused (device-scalar) path: dinv with negative entries
    val = A.val.copy()
    rows = np.repeat(np.arange(n), np.diff(A.ind))
    val[(A.col == rows) & (rows % 2 == 1)] *= -1.0
    Dn = DeviceCSR.from_arrays(A.shape, A.ind, A.col, val)
    An = oracle.CSR(A.shape, val, A.col, A.ind)
    xo, xg = np.zeros(n), np.zeros(n)
    dinv = oracle.jacobi_dinv(An.diagonal())
    ref = oracle.minres(An, b, xo, 1e-9, 200, dinv)
    got = minres(Dn, b, xg, 1e-9, 200, DeviceJacobi(Dn))
    assert got[:2] == ref[:2] and (ref[0] in (-3, -6) or np.abs(xg - xo).max() <= 1e-10 * np.abs(xo).max())
    # b = 0, x0 = 0: norm_r0 = 0, the strict test 0 < tol*0 never holds and the reference iterates on NaNs
    x = np.zeros(n)
    r = minres(D, np.zeros(n), x, 1e-9, 10)
    ro = oracle.minres(A, np.zeros(n), np.zeros(n), 1e-9, 10)
    assert r[:2] == ro[:2] == (-1, 10) and np.isnan(r[2]) and np.isnan(x).all()
    assert minres(D, b, np.zeros(n), 1e-9, 0)[:2] == (-1, 0)


def test_pcg_jacobi_constant_and_variable_diagonal(oracle):
    """precon.jacobi registers a dinv that holds one value everywhere (constant-diagonal operator) so
    the vector kernels form z = r*c without streaming dinv; a variable diagonal takes the array path.
    Same products either way: iteration counts identical to the oracle, iterates <= 1e-12."""
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg
    A = oracle.poisson_csr(40, 30, 20)
    n = A.shape[0]
    for variable in (False, True):
        val = A.val.copy()
        if variable:  # bump the diagonal entries by a smooth, row-dependent amount (stays SPD)
            rows = np.repeat(np.arange(n), np.diff(A.ind))
            dmask = A.col == rows
            val[dmask] += 0.5 + 0.4 * np.sin(np.arange(n) * 0.01)
        B = oracle.CSR(A.shape, val, A.col, A.ind)
        D = DeviceCSR.from_arrays(B.shape, B.ind, B.col, B.val)
        b = np.empty(n)
        B.matvec(np.ones(n), b)
        xo, xs = np.zeros(n), np.zeros(n)
        ref = oracle.pcg(B, b, xo, 1e-10, 2000, oracle.jacobi_dinv(B.diagonal()), hist=True)
        res = pcg(D, b, xs, 1e-10, 2000, DeviceJacobi(D), hist=True)
        assert res[:2] == ref[:2] and res[0] == 0
        # the recurred residual has dropped ten orders below ||r0|| at the end: its last digits are the rounding of the dot
        # sums (whose order differs between the oracle's sequential loops and any device reduction)
        assert np.allclose(res[3][:res[1] + 1], ref[3][:ref[1] + 1], rtol=1e-7, atol=0)
        assert np.allclose(res[3][:res[1] // 2], ref[3][:ref[1] // 2], rtol=1e-10, atol=0)
        assert np.abs(xs - xo).max() <= 1e-12 * np.abs(xo).max()


def test_hint_constant_abi():
    """psp_k_hint_constant on caller-owned vectors (the multi-GPU driver's dinv slice): a constant vector
    and a non-constant one give the same residual reductions as the unhinted call"""
    import ctypes as C
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    n = 100003
    rng = np.random.default_rng(3)
    bh, rh = rng.standard_normal(n), rng.standard_normal(n)
    for dh in (np.full(n, 1.0 / 6.0), 1.0 / (4.0 + rng.random(n))):
        b, d = dev.DeviceBuffer.from_host(bh), dev.DeviceBuffer.from_host(dh)
        outs = []
        for hinted in (False, True):
            r = dev.DeviceBuffer.from_host(rh)
            out = dev.DeviceBuffer(4)
            if hinted:
                check(L.psp_k_hint_constant(d.ptr, n))
            check(L.psp_k_residual(n, b.ptr, r.ptr, d.ptr, out.ptr))
            outs.append((out.download()[:2].copy(), r.download()))
            if hinted:
                check(L.psp_k_unhint(d.ptr))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        t = bh - rh
        assert abs(outs[0][0][1] - np.dot(t, t * dh)) <= 1e-12 * abs(np.dot(t, t * dh))


def test_pcg_lazy_x_update_exit_semantics_match_eager_loop():
    """The lazy x-update loop learns about the stagnation of iteration k only inside iteration k+1 and
    when the loop runs out; every truncation point must still give the eager loop's (info, iter,
    relres, x) bit for bit: tol = 0 on a small SPD system for maxit = 1 .. beyond the iteration where
    the updates stagnate (-5), plus a converging run and the variable-diagonal Jacobi path."""
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r);"
        "from pysparse_amd.device import DeviceCSR, DeviceJacobi, pcg;"
        "D = DeviceCSR.poisson(12, 12); n = 144; out = [];\n"
        "b = np.random.default_rng(0).standard_normal(n)\n"
        "for K in (None, DeviceJacobi(D)):\n"
        "    for maxit in list(range(1, 140)) + [1000]:\n"
        "        x = np.zeros(n); r = pcg(D, b, x, 0.0, maxit, K)\n"
        "        out.append([r[0], r[1], r[2].hex(), x.tobytes().hex()[:64], float(np.abs(x).sum()).hex()])\n"
        "x = np.zeros(n); r = pcg(D, b, x, 1e-9, 500, DeviceJacobi(D)); out.append([r[0], r[1], r[2].hex(), float(x.sum()).hex()])\n"
        "print(json.dumps(out))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # the last one (round 5): lazy loop with the p update and the pending x update folded into the product -- the
    # direction vector ping-pongs between two buffers there, so every truncation point (both parities) matters
    for env in ({"PSP_PCG_LAZYX": "0"}, {"PSP_PCG_LAZYX": "2", "PSP_PCG_LAZYPF": "0"}, {"PSP_PCG_ASYNC": "0"},
                {"PSP_PCG_LAZYX": "2", "PSP_PCG_LAZYPF": "1"}):
        # PSP_TUNING: the master switch that makes the library read its A/B variables; PSP_COOP=0: these tests compare
        # the launch-per-phase loops with each other (at this size the default is the single-kernel loop, psp_coop.hip)
        e = dict(os.environ, PSP_TUNING="1", PSP_COOP="0")
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout
        outs.append(json.loads(out.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2] == outs[3]
    infos = {o[0] for o in outs[1][:-1]}
    assert -5 in infos and -1 in infos  # the sweep really crosses the stagnation point
    k5 = min(o[1] for o in outs[1][:-1] if o[0] == -5)
    assert k5 > 3  # stagnation after several iterations, not a crafted first-iteration case


@pytest.mark.parametrize("n", [1000, 4097, 1 << 16])
def test_reductions_do_not_depend_on_alignment(n):
    """the vector kernels use one 16-byte access per lane when every pointer is 16-byte aligned and n is even,
    two 8-byte accesses otherwise -- with the SAME elements per thread, so a reduction has the same bits at
    either alignment (the owned slice of an extended vector starts at an odd offset as often as not)"""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    bx, by = dev.DeviceBuffer(n + 2), dev.DeviceBuffer(n + 2)
    out = dev.DeviceBuffer(4)
    res = []
    for off in (0, 1):
        check(L.psp_memcpy_h2d(bx.ptr + 8 * off, x.ctypes.data, 8 * n))
        check(L.psp_memcpy_h2d(by.ptr + 8 * off, y.ctypes.data, 8 * n))
        check(L.psp_k_dot(n, bx.ptr + 8 * off, by.ptr + 8 * off, out.ptr))
        d = out.download()[0]
        r = dev.DeviceBuffer(n + 2)
        check(L.psp_memcpy_h2d(r.ptr + 8 * off, y.ctypes.data, 8 * n))
        check(L.psp_k_residual(n, bx.ptr + 8 * off, r.ptr + 8 * off, None, out.ptr))
        res.append((d,) + tuple(out.download()[:2]))
    assert res[0] == res[1]


@pytest.mark.parametrize("scale", [1e-170, 1e160, 2.0 ** -600, 1e-120, 1e140])
def test_badly_scaled_right_hand_side(oracle, scale):
    """|b| ~ 1e-170 or 1e+160: sum b_i^2 underflows to 0 / overflows to inf, where the reference's dnrm2 (scaled
    form, pcg.c:57,75; minres.c:71) still gives ||b||.  The setup norms are formed at a power-of-two scale then,
    so the solvers do NOT mistake b for a zero right-hand side (x := 0, info 0); from there on the reference's own
    ddot reductions underflow / overflow exactly like the ones here and both leave through the same exit
    (rho == 0 -> -2 at iteration 1, NaN iterates -> -5 / -1).  At 1e-120 / 1e140 the iteration itself stays in range and the
    solves converge.  Bar: the oracle's (info, iter), relres equal or both NaN, iterates equal to 1e-12 or NaN
    in the same places."""
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, minres, pcg
    A = oracle.poisson_csr(30, 20)
    n = A.shape[0]
    D = DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
    b1 = np.empty(n)
    A.matvec(np.ones(n), b1)
    b = b1 * scale
    dinv = oracle.jacobi_dinv(A.diagonal())
    for sg, so in ((pcg, oracle.pcg), (minres, oracle.minres)):
        for K, dv in ((None, None), (DeviceJacobi(D), dinv)):
            xo, xg = np.zeros(n), np.zeros(n)
            ref = so(A, b, xo, 1e-9, 60, dv)
            got = sg(D, b, xg, 1e-9, 60, K)
            assert got[:2] == ref[:2], (sg.__name__, scale, ref, got)
            assert (np.isnan(ref[2]) and np.isnan(got[2])) or abs(got[2] - ref[2]) <= 1e-6 * abs(ref[2])
            assert np.array_equal(np.isnan(xg), np.isnan(xo))
            ok = ~np.isnan(xo)
            if ok.any() and np.abs(xo[ok]).max() > 0:
                assert np.abs(xg[ok] - xo[ok]).max() <= 1e-12 * np.abs(xo[ok]).max()
    # a right-hand side that IS zero still takes pcg.c:58-67
    x = np.ones(n)
    assert pcg(D, np.zeros(n), x, 1e-9, 10) == (0, 0, 0.0) and not x.any()


def test_concurrent_solves_from_python_threads(oracle):
    """The extension modules and ctypes release the GIL around device work, and the library's reduction workspace
    and stream are process-global: every compute entry point takes one library-wide lock, so solves issued from
    several Python threads at once serialise (as they do in the reference, which holds the GIL) instead of reading
    each other's partial sums.  Eight threads x mixed pcg / minres / matvec on different handles: every result
    equals the one obtained alone."""
    import threading
    from pysparse_amd.device import DeviceCSR, DeviceJacobi, DeviceSSS, minres, pcg
    work = []
    for k, grid in enumerate([(40, 30, 0), (24, 20, 12), (64, 64, 0), (18, 17, 16)]):
        A = DeviceCSR.poisson(*grid)
        S = DeviceSSS.poisson(*grid)
        n = A.shape[0]
        b = np.random.default_rng(k).standard_normal(n)
        work.append((pcg, A, b, DeviceJacobi(A)))
        work.append((minres, S, b, DeviceJacobi(S)))
    def solve(item):
        fn, M, b, K = item
        x = np.zeros(len(b))
        r = fn(M, b, x, 1e-9, 3000, K)
        y = np.empty(len(b))
        M.matvec(x, y)
        return r, x, y
    alone = [solve(w) for w in work]
    for trial in range(3):
        got = [None] * len(work)
        def run(i):
            got[i] = solve(work[i])
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(work))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for (r0, x0, y0), (r1, x1, y1) in zip(alone, got):
            assert r0 == r1 and np.array_equal(x0, x1) and np.array_equal(y0, y1)
