#!/usr/bin/env python3
"""Generate tests/golden/ from the COMPILED REFERENCE (oracle/_ref) in this container.

Run once here (needs /root/reference for `make -C oracle ref`); the outputs are data
only -- inputs are described by their generator parameters, expected outputs are
numbers -- and are committed.  No reference source text is stored.

What is pinned, and by what:
  ref_pcg.json       G1..G5 of BASELINE.md + fixed-iteration-count iterates: produced by
                     the reference's own examples/poisson_test/pcg.c (unmodified, built
                     by oracle/Makefile) driven through the bound CSR/SSS/Jacobi shims.
  ref_standalone.json  the unmodified standalone program examples/poisson_test/
                     poisson_test.c run on a generated matrices/poi2d_100.mtx
                     (iteration count and printed relative residual).
  ref_iterates.npz   full x vectors of the 100x100 cases from the compiled reference.
  structure.json     CSR/SSS triples of poisson2d(3..6) through the ll_mat restatement
                     (sorted insertion, to_csr general/symmetric, to_sss) and the
                     analytic invariants the reference's tests assert
                     (test/test_spmatrix.py:77-78,186-187).
  tendigit.json      K1: examples/tendigit.py known answer (Trefethen challenge #7).
  ref_published_table.json  the reference's only published benchmark for this path (doc/pysparse/source/itsolvers.rst:
                     120-130,189-199: n = 100 / 300 / 500, sss_mat, no preconditioner, tol 1e-12, maxit 2000): the
                     UNMODIFIED standalone program run on generated .mtx files of those sizes (iteration count, printed
                     relres) and the compiled pcg.c driven by the SSS product (info / iter / relres / strided x);
                     ref_published_table.npz holds x[::97] of each.  (round 6; `--published-only` regenerates it alone)
  ref_krylov.json    the module's OWN six kernels -- pysparse/itsolvers/src/{pcg,minres,cgs,bicgstab,
                     qmrs,gmres}.c compiled unmodified into oracle/_ref/libref_krylov.so -- on the
                     cases of tests/krylov_cases.py: info / iter / relres / kernel return value and
                     samples of x; ref_krylov_iterates.npz holds whole x vectors.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import krylov_cases as KC  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def samples(x):
    n = len(x)
    idx = [0, 1, n // 3, n // 2, n - 2, n - 1]
    return {"idx": idx, "val": [float(x[i]) for i in idx], "norm2": float(np.linalg.norm(x)),
            "sum": float(x.sum())}


def run_ref(A, b, tol, maxit, dinv=None):
    x = np.zeros(A.shape[0])
    info, it, relres = O.ref_pcg(A, b, x, tol, maxit, dinv)
    return x, {"info": info, "iter": it, "relres": relres, "x": samples(x)}


def _num(v):
    """JSON has no NaN: spell it"""
    return "nan" if v != v else float(v)


def krylov_goldens():
    """tests/golden/ref_krylov.json + ref_krylov_iterates.npz from oracle/_ref/libref_krylov.so"""
    out, iterates = {}, {}
    whole_big = {"minres_csr_1e-08_none", "minres_sss_1e-12_jacobi", "minres_fixed_10", "minres_fixed_50",
                 "minres_fixed_jacobi_10", "pcg_fixed_50"}
    for name, case in KC.CASES.items():
        info, it, rr, x, rc = KC.run_reference(O, case)
        e = {"info": info, "iter": it, "relres": _num(rr), "rc": rc}
        if np.isfinite(x).all():
            e["x"] = samples(x)
            if len(x) <= 3000 or name in whole_big:
                iterates[name] = x
        else:
            e["x"] = "nan" if np.isnan(x).all() else "nonfinite"
        out[name] = {"case": case, "expect": e}
    with open(os.path.join(OUT, "ref_krylov.json"), "w") as f:
        json.dump({"source": "pysparse/itsolvers/src/{pcg,minres,cgs,bicgstab,qmrs,gmres}.c compiled unmodified "
                             "(oracle/Makefile: _ref/libref_krylov.so, OpenBLAS BLAS-1, one thread)",
                   "cases": out}, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "ref_krylov_iterates.npz"), **iterates)


def write_sym_mtx(path, S):
    """poisson2d_sym(n).to_sss() as a symmetric coordinate file (lower triangle + diagonal, 1-based)"""
    n = S.n
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%d %d %d\n" % (n, n, S.nnz_lower + n))
        for i in range(n):
            for k in range(S.ind[i], S.ind[i + 1]):
                f.write("%d %d %.17g\n" % (i + 1, S.col[k] + 1, S.val[k]))
            f.write("%d %d %.17g\n" % (i + 1, i + 1, S.diag[i]))


def run_standalone(S):
    """the unmodified examples/poisson_test/poisson_test.c (oracle/_ref/poisson_test) on S written to the one file name
    the program opens (matrices/poi2d_100.mtx -- the name is fixed in its source, the size is not): (iter, printed
    relres, last stdout line, wall seconds of the whole program = its read + convert + solve)"""
    import time
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "matrices"))
        write_sym_mtx(os.path.join(td, "matrices", "poi2d_100.mtx"), S)
        t0 = time.perf_counter()
        out = subprocess.run([O.REF_BIN_PATH], cwd=td, capture_output=True, text=True).stdout
        wall = time.perf_counter() - t0
    m = re.search(r"converged at iteration (\d+) to a solution with relative residual ([0-9.eE+-]+)", out)
    assert m, out
    return int(m.group(1)), float(m.group(2)), out.strip().splitlines()[-1], wall


def published_table_goldens():
    """tests/golden/ref_published_table.json (+ .npz): itsolvers.rst:120-130,189-199 -- n = 100 / 300 / 500"""
    rows, strided = [], {}
    for nn in (100, 300, 500):
        S = O.poisson_sss(nn, nn)
        n = S.n
        it, rr, tail, wall = run_standalone(S)
        x, r = run_ref(S, np.ones(n), 1e-12, 2000)
        assert r["info"] == 0 and r["iter"] == it, (nn, r, it)
        strided["x_%d" % nn] = x[::97].copy()
        rows.append({"n": nn, "rows": n, "nnz_lower": int(S.nnz_lower),
                     "standalone": {"iter": it, "relres_printed": rr, "stdout_tail": tail,
                                    "wall_s_in_the_build_container": wall},
                     "compiled_pcg_with_sss_product": r})
    with open(os.path.join(OUT, "ref_published_table.json"), "w") as f:
        json.dump({"what": "L x = 1, L = poisson2d_sym(n).to_sss(), x0 = 0, pcg(S, b, x, 1e-12, 2000), no preconditioner",
                   "published_in": "doc/pysparse/source/itsolvers.rst:120-130 (script), :189-199 (table)",
                   "published_seconds_assembly_solve_total": {
                       "Python": {"100": [0.03, 1.12, 1.15], "300": [0.21, 49.65, 49.86], "500": [0.62, 299.39, 300.01]},
                       "Native C": {"100": [0.30, 0.96, 1.26], "300": [3.14, 48.38, 51.52], "500": [10.86, 288.67, 299.53]},
                       "Matlab": {"100": [0.21, 8.85, 9.06], "300": [2.05, 387.26, 389.31], "500": [6.23, 1905.67, 1911.8]},
                       "note": "machine unknown (the doc says so itself): context, not a target"},
                   "program": "examples/poisson_test/poisson_test.c + pcg.c + mmio.c compiled unmodified (oracle/Makefile); "
                              "the program opens matrices/poi2d_100.mtx whatever the size of the matrix inside",
                   "rows": rows}, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "ref_published_table.npz"), **strided)


def main():
    O.build(ref=True)
    os.makedirs(OUT, exist_ok=True)
    if "--published-only" in sys.argv:
        published_table_goldens()
        return
    krylov_goldens()
    if "--krylov-only" in sys.argv:
        return
    published_table_goldens()
    cases = {}
    iterates = {}

    A = O.poisson_csr(100, 100)
    S = O.poisson_sss(100, 100)
    n = A.shape[0]
    e = np.ones(n)
    b_Ae = np.empty(n)
    A.matvec(e, b_Ae)
    dinv = O.jacobi_dinv(A.diagonal())

    for name, K in (("G1_none", None), ("G1_jacobi", dinv)):
        x, r = run_ref(A, b_Ae, 1e-6, 2 * n, K)
        r.update(problem="poisson2d 100x100 CSR", b="A*ones", tol=1e-6, maxit=2 * n,
                 err_inf=float(np.abs(x - 1).max()))
        cases[name] = r
        iterates[name] = x
    for name, tol in (("G2", 1e-8), ("G3", 1e-12)):
        x, r = run_ref(A, np.ones(n), tol, 2000)
        r.update(problem="poisson2d 100x100 CSR", b="ones", tol=tol, maxit=2000)
        cases[name] = r
        iterates[name] = x
    x, r = run_ref(S, np.ones(n), 1e-12, 2000)
    r.update(problem="poisson2d 100x100 SSS", b="ones", tol=1e-12, maxit=2000)
    cases["G3_sss"] = r
    iterates["G3_sss"] = x
    x, r = run_ref(S, np.ones(n), 1e-8, 2000)
    r.update(problem="poisson2d 100x100 SSS", b="ones", tol=1e-8, maxit=2000)
    cases["G2_sss"] = r

    # fixed iteration counts: tol = 0 never converges -> exactly maxit iterations,
    # flag -1 and iter = maxit + 1 (pcg.c:165)
    for k in (1, 2, 10, 50):
        x, r = run_ref(A, np.ones(n), 0.0, k)
        r.update(problem="poisson2d 100x100 CSR", b="ones", tol=0.0, maxit=k)
        cases["fixed_%d" % k] = r
        iterates["fixed_%d" % k] = x
        x, r = run_ref(A, b_Ae, 0.0, k, dinv)
        r.update(problem="poisson2d 100x100 CSR jacobi", b="A*ones", tol=0.0, maxit=k)
        cases["fixed_jacobi_%d" % k] = r

    for N, name in ((32, "G4"), (64, "G5")):
        A3 = O.poisson_csr(N, N, N)
        n3 = A3.shape[0]
        b3 = np.empty(n3)
        A3.matvec(np.ones(n3), b3)
        d3 = O.jacobi_dinv(A3.diagonal())
        for suffix, K in (("_none", None), ("_jacobi", d3)):
            x, r = run_ref(A3, b3, 1e-8, 2000, K)
            r.update(problem="poisson3d %d^3 CSR" % N, b="A*ones", tol=1e-8, maxit=2000,
                     err_inf=float(np.abs(x - 1).max()))
            cases[name + suffix] = r

    # special exits of the reference kernel
    x, r = run_ref(A, np.zeros(n), 1e-8, 10)  # b == 0 -> flag 0, iter 0, relres 0
    cases["zero_rhs"] = r
    xg = np.ones(n)
    info, it, relres = O.ref_pcg(A, b_Ae, xg, 1e-8, 10)  # exact initial guess
    cases["exact_guess"] = {"info": info, "iter": it, "relres": relres}

    with open(os.path.join(OUT, "ref_pcg.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "ref_iterates.npz"), **iterates)

    # ---- the unmodified standalone program on a generated .mtx
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "matrices"))
        with open(os.path.join(td, "matrices", "poi2d_100.mtx"), "w") as f:
            f.write("%%MatrixMarket matrix coordinate real symmetric\n")
            f.write("%d %d %d\n" % (n, n, S.nnz_lower + n))
            for i in range(n):
                for k in range(S.ind[i], S.ind[i + 1]):
                    f.write("%d %d %.17g\n" % (i + 1, S.col[k] + 1, S.val[k]))
                f.write("%d %d %.17g\n" % (i + 1, i + 1, S.diag[i]))
        out = subprocess.run([O.REF_BIN_PATH], cwd=td, capture_output=True, text=True).stdout
    m = re.search(r"converged at iteration (\d+) to a solution with relative residual ([0-9.eE+-]+)", out)
    assert m, out
    with open(os.path.join(OUT, "ref_standalone.json"), "w") as f:
        json.dump({"program": "examples/poisson_test/poisson_test.c (unmodified)",
                   "input": "poi2d_100.mtx generated from poisson2d_sym(100), b = ones, tol 1e-12",
                   "iter": int(m.group(1)), "relres_printed": float(m.group(2)),
                   "stdout_tail": out.strip().splitlines()[-1]}, f, indent=1)

    # ---- structure fixtures through the ll_mat restatement
    struct = {}
    for nn in (3, 4, 5, 6):
        L = O.poisson2d_ll(nn)
        Ls = O.poisson2d_ll(nn, sym=True)
        Ac, As, Ss = L.to_csr(), Ls.to_csr(), Ls.to_sss()
        assert (Ac.col == As.col).all() and (Ac.ind == As.ind).all() and (Ac.val == As.val).all()
        rows = np.repeat(np.arange(nn * nn), np.diff(Ac.ind))
        norm1 = float(np.bincount(Ac.col, weights=np.abs(Ac.val)).max())
        norminf = float(np.bincount(rows, weights=np.abs(Ac.val)).max())
        struct["poisson2d_%d" % nn] = {
            "n": nn, "ll_nnz": L.nnz, "ll_sym_nnz": Ls.nnz,
            "csr": {"val": Ac.val.tolist(), "col": Ac.col.tolist(), "ind": Ac.ind.tolist()},
            "sss": {"val": Ss.val.tolist(), "col": Ss.col.tolist(), "ind": Ss.ind.tolist(),
                    "diag": Ss.diag.tolist()},
            "norm1": norm1, "norminf": norminf,
        }
    with open(os.path.join(OUT, "structure.json"), "w") as f:
        json.dump(struct, f, indent=1, sort_keys=True)

    # ---- K1 ten-digit problem (examples/tendigit.py); answer from the literature:
    # Bornemann et al., "The SIAM 100-Digit Challenge", problem 7.
    T = O.tendigit_sss(20000)
    with open(os.path.join(OUT, "tendigit.json"), "w") as f:
        json.dump({"n": 20000, "nnz_lower": T.nnz_lower,
                   "x0_exact": 0.7250783462684011674686877133,
                   "source": "examples/tendigit.py:26-49; exact value = 100-digit challenge #7"},
                  f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
