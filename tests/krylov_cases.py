"""The problems behind tests/golden/ref_krylov.json, shared by the generator (oracle/make_golden.py, which runs the
COMPILED REFERENCE kernels on them) and by the tests (which run the oracle restatement and the HIP solvers on them).

A case is data: generator parameters for the matrix, the right-hand side, the start vector and the preconditioner,
plus (solver, tol, maxit, dim).  `build(O, case)` turns it into oracle-side objects; nothing here computes a result."""
import numpy as np


def nonsym_csr(O, n, seed):
    """diagonally dominant non-symmetric band matrix (offsets -7, -1, 1, 5)"""
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
    return O.CSR((n, n), np.array(vals), np.array(cols, dtype=np.int32), ind)


def diag_csr(O, vals):
    n = len(vals)
    return O.CSR((n, n), np.asarray(vals, dtype=np.float64), np.arange(n, dtype=np.int32),
                 np.arange(n + 1, dtype=np.int32))


def signed_poisson(O, nx, ny):
    """poisson2d with the diagonal of every odd row negated: symmetric indefinite, Jacobi not SPD (minres -3 inside
    the loop or a converging indefinite run, depending on b)"""
    A = O.poisson_csr(nx, ny)
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.ind))
    val = A.val.copy()
    val[(A.col == rows) & (rows % 2 == 1)] *= -1.0
    return O.CSR(A.shape, val, A.col, A.ind)


def matrix(O, spec):
    kind = spec[0]
    if kind == "poisson":
        form, dims = spec[1], spec[2:]
        dims = tuple(dims) + (0,) * (3 - len(dims))
        return O.poisson_csr(*dims) if form == "csr" else O.poisson_sss(*dims)
    if kind == "nonsym":
        return nonsym_csr(O, spec[1], spec[2])
    if kind == "diag":
        if spec[1] == "half_negative":
            v = np.ones(spec[2])
            v[spec[2] // 2:] = -1.0
            return diag_csr(O, v)
        if spec[1] == "const":
            return diag_csr(O, np.full(spec[2], float(spec[3])))
    if kind == "signed_poisson":
        return signed_poisson(O, spec[1], spec[2])
    if kind == "tendigit":
        return O.tendigit_sss(spec[1])
    raise ValueError(spec)


def csr_of(O, A):
    return A if isinstance(A, O.CSR) else O.sss_to_csr(A)


def rhs(O, A, spec):
    n = A.shape[0]
    if spec == "ones":
        return np.ones(n)
    if spec == "zeros":
        return np.zeros(n)
    if spec == "e0":
        b = np.zeros(n)
        b[0] = 1.0
        return b
    if spec == "A*ones":
        b = np.empty(n)
        A.matvec(np.ones(n), b)
        return b
    if spec.startswith("randn:"):
        return np.random.default_rng(int(spec[6:])).standard_normal(n)
    if spec.startswith("const:"):
        return np.full(n, float(spec[6:]))
    raise ValueError(spec)


def precon(O, A, spec):
    """None | ("jacobi", dinv, steps) | ("ssor", omega, steps) -- the K argument of oracle.ref_krylov"""
    if spec is None:
        return None
    n = A.shape[0]
    if spec[0] == "jacobi":
        d = A.diag if isinstance(A, O.SSS) else A.diagonal()
        omega = spec[2] if len(spec) > 2 else 1.0
        return ("jacobi", O.jacobi_dinv(d, omega), spec[1])
    if spec[0] == "dinv_const":
        return ("jacobi", np.full(n, float(spec[1])), 1)
    if spec[0] == "dinv_half_negative":
        d = np.ones(n)
        d[n // 2:] = -1.0
        return ("jacobi", d, 1)
    if spec[0] == "ssor":
        return ("ssor", float(spec[1]), int(spec[2]))
    raise ValueError(spec)


def build(O, case):
    A = matrix(O, case["matrix"])
    b = rhs(O, A, case["b"])
    x0 = np.full(A.shape[0], float(case.get("x0", 0.0)))
    K = precon(O, A, case.get("K"))
    return A, b, x0, K


def run_oracle(O, case):
    """(info, iter, relres, x) from the restatement in oracle/pysparse_oracle.c"""
    A, b, x, K = build(O, case)
    with np.errstate(all="ignore"):
        info, it, rr, _ = O.solve(case["solver"], A, b, x, case["tol"], case["maxit"], K, dim=case.get("dim", 20))
    return info, it, rr, x


def run_reference(O, case):
    """(info, iter, relres, x, rc) from the compiled reference kernel (oracle/_ref/libref_krylov.so)"""
    A, b, x, K = build(O, case)
    with np.errstate(all="ignore"):
        info, it, rr, rc = O.ref_krylov(case["solver"], A, b, x, case["tol"], case["maxit"], K, dim=case.get("dim", 20))
    return info, it, rr, x, rc


P100 = ("poisson", "csr", 100, 100)
S100 = ("poisson", "sss", 100, 100)
P40 = ("poisson", "csr", 40, 32)
P3D = ("poisson", "csr", 12, 11, 10)
S3D = ("poisson", "sss", 12, 11, 10)
NS = ("nonsym", 3000, 5)
J = ("jacobi", 1)


def _cases():
    c = {}

    def add(name, solver, matrix, b, tol, maxit, K=None, kind="converge", **kw):
        assert name not in c, name
        c[name] = dict(solver=solver, matrix=list(matrix), b=b, tol=tol, maxit=maxit, K=None if K is None else list(K),
                       kind=kind, **kw)

    # ---- PCG through pysparse/itsolvers/src/pcg.c (the module's kernel, flags -1/-2/-5/-6 as is)
    add("pcg_G1_none", "pcg", P100, "A*ones", 1e-6, 20000)
    add("pcg_G1_jacobi", "pcg", P100, "A*ones", 1e-6, 20000, J)
    add("pcg_G2", "pcg", P100, "ones", 1e-8, 2000)
    add("pcg_G3", "pcg", P100, "ones", 1e-12, 2000)
    add("pcg_G3_sss", "pcg", S100, "ones", 1e-12, 2000)
    add("pcg_3d_jacobi2", "pcg", P3D, "randn:7", 1e-10, 2000, ("jacobi", 2))
    add("pcg_3d_ssor", "pcg", S3D, "randn:7", 1e-10, 2000, ("ssor", 1.0, 1))
    add("pcg_3d_ssor_omega", "pcg", S3D, "randn:7", 1e-10, 2000, ("ssor", 1.3, 2))
    for k in (1, 2, 10, 50):
        add("pcg_fixed_%d" % k, "pcg", P100, "ones", 0.0, k, kind="fixed")
    add("pcg_fixed_jacobi_10", "pcg", P100, "A*ones", 0.0, 10, J, kind="fixed")
    add("pcg_zero_rhs", "pcg", P100, "zeros", 1e-8, 10, kind="exit")
    add("pcg_exact_guess", "pcg", P100, "A*ones", 1e-8, 10, x0=1.0, kind="exit")
    add("pcg_flag_m6_pq", "pcg", ("diag", "half_negative", 64), "ones", 1e-10, 50, kind="exit")  # p.Ap == 0
    add("pcg_flag_m2_rho", "pcg", ("diag", "const", 64, 1.0), "ones", 1e-10, 50, ("dinv_half_negative",), kind="exit")
    add("pcg_flag_m5_stag", "pcg", ("diag", "const", 64, 1e300), "const:1e150", 1e-10, 50, ("dinv_const", 1e-100),
        kind="exit")  # q overflows -> alpha == 0 -> stagnation
    add("pcg_maxit", "pcg", P40, "ones", 1e-14, 6, kind="fixed")

    # ---- MINRES (minres.c:43-200)
    for form, M in (("csr", P100), ("sss", S100)):
        for tol in (1e-8, 1e-12):
            for kn, K in (("none", None), ("jacobi", J)):
                add("minres_%s_%g_%s" % (form, tol, kn), "minres", M, "ones", tol, 2000, K)
    for k in (1, 2, 10, 50):
        add("minres_fixed_%d" % k, "minres", P100, "ones", 0.0, k, kind="fixed")
    add("minres_fixed_jacobi_10", "minres", P100, "A*ones", 0.0, 10, J, kind="fixed")
    add("minres_fixed_sss_10", "minres", S100, "ones", 0.0, 10, kind="fixed")
    add("minres_3d_randn", "minres", P3D, "randn:7", 1e-10, 2000, J, x0=0.25)
    add("minres_3d_jacobi2", "minres", P3D, "randn:7", 1e-10, 2000, ("jacobi", 2))
    add("minres_3d_ssor", "minres", S3D, "randn:7", 1e-10, 2000, ("ssor", 1.0, 1))
    add("minres_itmax", "minres", P40, "ones", 1e-14, 5, kind="fixed")
    add("minres_maxit0", "minres", P40, "ones", 1e-9, 0, kind="exit")
    add("minres_m3_setup", "minres", P40, "ones", 1e-8, 50, ("dinv_const", -1.0), kind="exit")
    add("minres_signed_jacobi", "minres", ("signed_poisson", 12, 10), "randn:0", 1e-9, 200, J, kind="exit")
    add("minres_indefinite", "minres", ("diag", "half_negative", 6), "ones", 1e-10, 20, kind="exit")
    add("minres_m6_zero_operator", "minres", ("diag", "const", 6, 0.0), "ones", 1e-10, 20, kind="exit")
    add("minres_zero_rhs", "minres", ("poisson", "csr", 12, 10), "zeros", 1e-9, 10, kind="nan")
    add("minres_tendigit", "minres", ("tendigit", 20000), "e0", 1e-16, 20000, J, kind="known")

    # ---- cgs / bicgstab / qmrs / gmres
    for s in ("cgs", "bicgstab", "qmrs", "gmres"):
        mats = [("p40", P40), ("p3d", P3D)] + ([] if s == "qmrs" else [("ns", NS)])
        kw = {"dim": 15} if s == "gmres" else {}
        for mn, M in mats:
            for kn, K in (("none", None), ("jacobi", J)):
                add("%s_%s_%s_k6" % (s, mn, kn), s, M, "randn:7", 1e-30, 6, K, kind="fixed", x0=0.25, **kw)
                add("%s_%s_%s_conv" % (s, mn, kn), s, M, "randn:7", 1e-10, 3000, K, kind="amplified", x0=0.25, **kw)
        add("%s_maxit" % s, s, P40, "ones", 1e-14, 6, kind="fixed", **({"dim": 4} if s == "gmres" else {}))
        add("%s_p100_jacobi" % s, s, P100, "ones", 1e-8, 2000, J, kind="amplified", **({"dim": 20} if s == "gmres" else {}))
    add("gmres_dim5_k12", "gmres", P40, "randn:3", 1e-30, 12, J, kind="fixed", dim=5)
    add("gmres_dim20_k25", "gmres", P40, "randn:3", 1e-30, 25, None, kind="fixed", dim=20)
    add("gmres_dim5_conv", "gmres", P3D, "randn:3", 1e-9, 3000, J, kind="amplified", dim=5)
    add("gmres_zero_rhs", "gmres", P40, "zeros", 1e-9, 10, x0=1.0, kind="exit")
    add("bicgstab_zero_rhs", "bicgstab", P40, "zeros", 1e-9, 10, x0=1.0, kind="exit")
    add("cgs_good_start", "cgs", P40, "A*ones", 1e-9, 10, x0=1.0, kind="exit")
    add("qmrs_m2_delta", "qmrs", ("diag", "const", 64, 1.0), "ones", 1e-10, 50, ("dinv_half_negative",), kind="exit")
    add("qmrs_m6_eps", "qmrs", ("diag", "half_negative", 64), "ones", 1e-10, 50, kind="exit")
    return c


CASES = _cases()


# ------------------------------------------------------------------ comparison with the goldens

def _isnan_spec(v):
    return v == "nan" or (isinstance(v, float) and v != v)


def check_against_golden(name, got, gold, iterates, relres_unset_ok=False):
    """got = (info, iter, relres, x) of some implementation; gold = the "expect" record the compiled reference kernel
    produced for the case; iterates = the npz with whole reference x vectors.  Bars by case kind:
      fixed      same info / iter, x <= 1e-12 relative (north_star's fp64 bar), relres 1e-9 (+1e-13 absolute)
      converge   same info / iter (pcg, minres, qmrs, gmres), x <= 1e-11, relres 2 % (the recurred norm at the threshold)
      amplified  same info, iter within max(3, 40 %), x <= 1e-6: cgs / bicgstab amplify the summation order of their
                 dot products, so counts at convergence differ between two CPU BLAS libraries already
      exit       same info / iter / relres (NaN = never written), x <= 1e-12 when finite
      nan        same info / iter, NaN relres and NaN x (b = 0 in minres: 0 < tol*0 never holds, minres.c:114)
      known      info 0, same iter, x[0] = the ten-digit answer to 5e-15"""
    case, kind = CASES[name], CASES[name]["kind"]
    info, it, rr, x = got
    assert info == gold["info"], (name, got[:3], gold)
    if kind == "amplified":
        assert abs(it - gold["iter"]) <= max(3, 0.4 * gold["iter"]), (name, got[:3], gold)
    else:
        assert it == gold["iter"], (name, got[:3], gold)
    g_rr = gold["relres"]
    if _isnan_spec(g_rr):
        assert relres_unset_ok or rr != rr, (name, rr)
    elif kind in ("converge", "amplified", "known"):
        if kind == "converge":
            assert abs(rr - g_rr) <= 2e-2 * g_rr, (name, rr, g_rr)
        else:
            assert rr <= max(case["tol"], 1e-15) * (1 + 1e-9) or info != 0, (name, rr)
    else:
        assert abs(rr - g_rr) <= 1e-9 * abs(g_rr) + 1e-13, (name, rr, g_rr)  # relres is relative to the start: O(1) scale
    gx = gold["x"]
    if kind == "nan":
        assert gx == "nan" and np.isnan(x).all(), name
        return
    if kind == "known":
        assert abs(x[0] - 0.7250783462684011674686877133) < 5e-15, (name, x[0])
        return
    if isinstance(gx, str):
        assert not np.isfinite(x).all(), name
        return
    tol = {"fixed": 1e-12, "converge": 1e-11, "amplified": 1e-6, "exit": 1e-12}[kind]
    scale = max(np.abs(x).max(), 1e-300)
    for i, v in zip(gx["idx"], gx["val"]):
        assert abs(x[i] - v) <= tol * scale, (name, i, x[i], v)
    assert abs(np.linalg.norm(x) - gx["norm2"]) <= 10 * tol * max(gx["norm2"], 1e-300), name
    if name in iterates.files:
        assert np.abs(x - iterates[name]).max() <= tol * scale, (name, np.abs(x - iterates[name]).max() / scale)
