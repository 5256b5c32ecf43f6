"""Child of tests/test_cpu_mode.py: started with PSP_DEVICE=cpu, runs the drop-in modules on the library's host loops
(pysparse_amd/csrc/psp_cpu.hip) and checks them against the oracle and the compiled reference's goldens.  Prints one
JSON line; any assertion failure is the child's non-zero exit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("PSP_DEVICE") == "cpu"

from oracle import oracle as O  # noqa: E402
from pysparse.itsolvers import krylov  # noqa: E402
from pysparse.precon import precon  # noqa: E402
from pysparse.sparse import spmatrix  # noqa: E402
from pysparse_amd import _capi  # noqa: E402
import krylov_cases as KC  # noqa: E402

out = {"version": _capi.lib().psp_version().decode()}
assert "PSP_DEVICE=cpu" in out["version"]

# ---- containers and products: bit for bit the oracle's loops (same order of operations)
L = spmatrix.ll_mat_sym(49, 200)
for i in range(7):
    for j in range(7):
        k = i + 7 * j
        L[k, k] = 4
        if i > 0:
            L[k, k - 1] = -1
        if j > 0:
            L[k, k - 7] = -1
A, S = L.to_csr(), L.to_sss()
Oc, Os = O.poisson_csr(7, 7), O.poisson_sss(7, 7)
ind, col, val = A.to_arrays()
assert np.array_equal(ind, Oc.ind) and np.array_equal(col, Oc.col) and np.array_equal(val, Oc.val)
rng = np.random.default_rng(0)
for (M, Om, n) in ((spmatrix.poisson_csr(30, 20), O.poisson_csr(30, 20), 600), (spmatrix.poisson_sss(9, 8, 7), O.poisson_sss(9, 8, 7), 504),
                   (A, Oc, 49), (S, Os, 49), (L, Oc, 49)):
    x = rng.standard_normal(n)
    y, yo = np.full(n, 7.0), np.full(n, 7.0)
    M.matvec(x, y)
    Om.matvec(x, yo)
    assert np.array_equal(y, yo)
    xs, ys = rng.standard_normal(2 * n)[::2], np.zeros(3 * n)[::3]
    M.matvec(xs, ys)
    Om.matvec(np.ascontiguousarray(xs), yo)
    assert np.array_equal(ys, yo)
yt, yto = np.empty(600), np.empty(600)
x = rng.standard_normal(600)
spmatrix.poisson_csr(30, 20).matvec_transp(x, yt)
O.poisson_csr(30, 20).matvec_transp(x, yto)
assert np.array_equal(yt, yto)
assert S[3, 10] == S[10, 3] == -1.0 and S[5, 5] == 4.0 and S[0, 30] == 0.0
assert "PSP_DEVICE=cpu" in repr(A) and "on the GPU" not in repr(A) and "PSP_DEVICE=cpu" in repr(S)  # host-mode objects say where they live
# print(A): the text of the reference's tp_print slots (csr_mat.c:186-207, sss_mat.c:127-147)
T3 = spmatrix.ll_mat_sym(3, 5)
for i in range(3):
    T3[i, i] = 2
    if i:
        T3[i, i - 1] = -1
assert str(T3.to_csr()) == "csr_mat([3,3], [(0,0): 2, (0,1): -1, (1,0): -1, (1,1): 2, (1,2): -1, (2,1): -1, (2,2): 2])"
assert str(T3.to_sss()) == "sss_mat([3,3], [(0,0): 2, (1,0): -1, (1,1): 2, (2,1): -1, (2,2): 2])"
assert str(spmatrix.poisson_csr(60, 60)).startswith("<csr_mat object")  # beyond 10 000 entries: the one-line repr

# ---- solvers against the goldens of the compiled reference kernels and, bit for bit, the oracle's sequential loops
with open(os.path.join(ROOT, "tests", "golden", "ref_krylov.json")) as f:
    gold = json.load(f)["cases"]
its = np.load(os.path.join(ROOT, "tests", "golden", "ref_krylov_iterates.npz"))


class Diag:
    def __init__(self, dinv):
        self.d = 1.0 / np.asarray(dinv)
        self.shape = (len(self.d), len(self.d))

    def __getitem__(self, ij):
        return float(self.d[ij[0]])


done = []
for name, case in KC.CASES.items():
    if case["solver"] not in ("pcg", "minres") or name == "minres_tendigit":
        continue
    Ao, b, x0, Ko = KC.build(O, case)
    if isinstance(Ao, O.SSS):
        M = spmatrix.sss_from_arrays(Ao.ind, Ao.col, Ao.val, Ao.diag)
    else:
        M = spmatrix.csr_from_arrays(Ao.ind, Ao.col, Ao.val, Ao.shape)
    spec = case.get("K")
    if spec is None:
        K = None
    elif spec[0] == "jacobi":
        K = precon.jacobi(M, spec[2] if len(spec) > 2 else 1.0, spec[1])
    elif spec[0] == "ssor":
        K = precon.ssor(M, float(spec[1]), int(spec[2]))
    else:
        K = precon.jacobi(Diag(Ko[1]))
    x = x0.copy()
    with np.errstate(all="ignore"):
        r = getattr(krylov, case["solver"])(M, b, x, case["tol"], case["maxit"], K)
    KC.check_against_golden(name, (r[0], r[1], r[2], x), gold[name]["expect"], its, relres_unset_ok=True)
    ro = KC.run_oracle(O, case)
    assert (r[0], r[1]) == (ro[0], ro[1]), (name, r, ro[:3])
    if np.isfinite(ro[3]).all():
        assert np.array_equal(x, ro[3]), name  # two sequential restatements of the same loops: the same bits
    done.append(name)
out["cases"] = len(done)

# ---- the reference's script flow (examples/demo_pcg.py:23-100) on poisson2d(100): G1
n = 10000
A = spmatrix.poisson_csr(100, 100)
e = np.ones(n)
b = np.empty(n)
A.matvec(e, b)
x = np.zeros(n)
info, it, relres = krylov.pcg(A, b, x, 1e-6, 2 * n, precon.jacobi(A, 1.0, 1))
out["G1"] = [info, it, relres, float(np.abs(x - 1).max())]

# ---- a duck-typed Python operator and preconditioner (the reference's protocol) on the host loops
class PyOp:
    shape = (n, n)

    def matvec(self, x, y):
        A.matvec(x, y)

    def precon(self, x, y):
        y[:] = 0.25 * x


x2 = np.zeros(n)
r2 = krylov.pcg(PyOp(), b, x2, 1e-6, 2 * n, PyOp())
assert r2[:2] == (info, it) and np.array_equal(x, x2)

# ---- what the host mode does not cover says so
for bad in (lambda: krylov.cgs(A, b, np.zeros(n), 1e-6, 10), lambda: spmatrix.poisson_csr(8, 8, devices=[0])):
    try:
        bad()
        raise SystemExit("expected a RuntimeError naming the host mode")
    except RuntimeError as err:
        assert "PSP_DEVICE=cpu" in str(err) or "no HIP device" in str(err), str(err)
print(json.dumps(out))
