"""GPU: general (non-stencil) csr_mat beyond the reference's C int (csr_mat.h:6-13 -- `int nnz`, `int *ind`):
64-bit row offsets at the C ABI (psp_csr_create64), rows cut into parts of < 2^30 nonzeros on the device.
  * small matrices with the part size lowered (PSP_PART_NNZ, child process): every path of a partitioned
    handle -- product, fused dot, diagonal, Jacobi-PCG / MINRES, row download -- against the oracle, bit for bit
    where the single-part handle is;
  * a 2.4e9-nonzero banded matrix generated on the device (psp_csr_random_banded): sampled row blocks,
    including the part boundaries, multiplied by the oracle loop (csr_mat.c:49-54) from rows rebuilt with the
    generator's formula AND from the rows downloaded from the device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = (1 << 64) - 1


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def banded_rows(r0, r1, ncols, m, stride, seed):
    """the formula of psp_csr.hip: random_banded_kernel, restated (uint64 arithmetic wraps like C's)"""
    with np.errstate(over="ignore"):
        r = np.arange(r0, r1, dtype=np.uint64)[:, None]
        j = np.arange(m, dtype=np.uint64)[None, :]
        h = splitmix64(np.uint64(seed) + r * np.uint64(0x100000001B3) + j * np.uint64(0xD6E8FEB86659FD93))
        off = (j.astype(np.int64) - m // 2) * stride + (h % np.uint64(stride)).astype(np.int64)
        c = (r.astype(np.int64) + off) % ncols
        h2 = splitmix64(h)
        v = (h2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0
    ind = np.arange(0, (r1 - r0) * m + 1, m, dtype=np.int32)
    return ind, c.astype(np.int32).ravel(), v.ravel()


CHILD = r"""
import json, sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
from pysparse_amd import device as dev
from pysparse_amd._capi import check, lib
L = lib()
out = {}
A = O.poisson_csr(30, 20, 10)
n = A.shape[0]
rng = np.random.default_rng(0)
val = A.val * (1.0 + 0.1 * rng.random(A.nnz))          # general values, SPD stays (diagonally dominant)
rows = np.repeat(np.arange(n), np.diff(A.ind))
val[A.col == rows] = 7.0 + rng.random(n)
A = O.CSR(A.shape, val, A.col, A.ind)
P = dev.DeviceCSR.from_arrays64(A.shape, A.ind.astype(np.int64), A.col, A.val)   # cut into parts (PSP_PART_NNZ)
S = dev.DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)                      # one handle
out["nnz"] = [P.nnz, S.nnz, A.nnz]
x = rng.standard_normal(n)
yp, ys, yo = np.empty(n), np.empty(n), np.empty(n)
P.matvec(x, yp); S.matvec(x, ys); A.matvec(x, yo)
out["spmv"] = bool(np.array_equal(yp, yo) and np.array_equal(ys, yo))
out["diag"] = bool(np.array_equal(P.diagonal(), A.diagonal()))
i64, c, v = P.download_rows(0, n)
out["download"] = bool(np.array_equal(i64, A.ind) and np.array_equal(c, A.col) and np.array_equal(v, A.val))
i64, c, v = P.download_rows(n // 3, 2 * n // 3)
a, b_ = A.ind[n // 3], A.ind[2 * n // 3]
out["download_mid"] = bool(np.array_equal(i64, A.ind[n // 3:2 * n // 3 + 1] - a) and np.array_equal(c, A.col[a:b_]))
xd = dev.DeviceBuffer.from_host(x); yd = dev.DeviceBuffer(n); od = dev.DeviceBuffer(1)
check(L.psp_k_csr_matvec_dot(P._h, xd.ptr, 0, yd.ptr, od.ptr))
d = float(od.download()[0])
out["dot"] = bool(np.array_equal(yd.download(), yo) and abs(d - float(np.dot(x, yo))) <= 1e-12 * abs(d))
b = np.empty(n); A.matvec(np.ones(n), b)
dinv = O.jacobi_dinv(A.diagonal())
res = {}
for name, sg, so in (("pcg", dev.pcg, O.pcg), ("minres", dev.minres, O.minres)):
    xo, xg = np.zeros(n), np.zeros(n)
    ref = so(A, b, xo, 1e-10, 500, dinv)
    got = sg(P, b, xg, 1e-10, 500, dev.DeviceJacobi(P))
    res[name] = [list(ref[:2]), list(got[:2]), float(np.abs(xg - xo).max() / np.abs(xo).max())]
out["solvers"] = res
try:
    P.download()
    out["download_whole_refused"] = False
except Exception:
    out["download_whole_refused"] = True
print(json.dumps(out))
"""


@pytest.mark.parametrize("part_nnz", [1000, 7777])
def test_partitioned_csr_small(part_nnz):
    env = dict(os.environ, PSP_TUNING="1", PSP_PART_NNZ=str(part_nnz))
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["nnz"][0] == out["nnz"][1] == out["nnz"][2]
    for k in ("spmv", "diag", "download", "download_mid", "dot"):
        assert out[k], k
    for name, (ref, got, err) in out["solvers"].items():
        assert ref == got and err <= 1e-12, (name, ref, got, err)


def test_from_arrays64_small_is_an_ordinary_handle(oracle):
    from pysparse_amd import device as dev
    A = oracle.poisson_csr(12, 9)
    D = dev.DeviceCSR.from_arrays64(A.shape, A.ind.astype(np.int64), A.col, A.val)
    ind, col, val = D.download()
    assert np.array_equal(ind, A.ind) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
    assert D.kernel_info()[0] == "csr_spmv_w4"


def test_random_banded_generator_matches_its_formula(oracle):
    from pysparse_amd import device as dev
    n, m, stride, seed = 50000, 9, 32, 12345
    D = dev.DeviceCSR.random_banded(n, n, m, stride, seed)
    ind, col, val = D.download()
    ri, rc, rv = banded_rows(0, n, n, m, stride, seed)
    assert np.array_equal(ind, ri) and np.array_equal(col, rc) and np.array_equal(val, rv)


def test_csr_beyond_2_31_nonzeros_bit_exact(oracle):
    """n = 2^28 rows x 9 entries = 2 415 919 104 nonzeros (> 2^31): three parts on the device.  The product is
    checked on sampled row blocks against the oracle loop, with the rows taken (a) from the generator's formula
    and (b) from psp_csr_download_rows; x lives on the host once (2 GiB)."""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    n, m, stride, seed = 1 << 28, 9, 32, 99
    D = dev.DeviceCSR.random_banded(n, n, m, stride, seed)
    assert D.nnz == n * m > 2 ** 31 and D.shape == (n, n)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    xd = dev.DeviceBuffer(n)
    yd = dev.DeviceBuffer(n)
    step = 1 << 24
    for k in range(0, n, step):
        check(L.psp_memcpy_h2d(xd.ptr + 8 * k, x[k:k + step].ctypes.data, 8 * step))
    D.matvec_dev(xd.ptr, yd.ptr)
    check(L.psp_synchronize())
    rows_per_part = (1 << 30) // m
    starts = [0, n - 4096, rows_per_part - 2048, 2 * rows_per_part - 2048] + [int(s) for s in rng.integers(0, n - 4096, 12)]
    for r0 in starts:
        r1 = r0 + 4096
        y = np.empty(r1 - r0)
        check(L.psp_memcpy_d2h(y.ctypes.data, yd.ptr + 8 * r0, 8 * (r1 - r0)))
        ind, col, val = banded_rows(r0, r1, n, m, stride, seed)
        B = oracle.CSR((r1 - r0, n), val, col, ind)
        yo = np.empty(r1 - r0)
        B.matvec(x, yo)
        assert np.array_equal(y, yo), r0
        i64, c2, v2 = D.download_rows(r0, r1)
        assert np.array_equal(i64, ind) and np.array_equal(c2, col) and np.array_equal(v2, val), r0
    # fused dot of the solver path over all parts: against the same sum formed from y on the device
    od = dev.DeviceBuffer(2)
    y2 = dev.DeviceBuffer(n)
    check(L.psp_k_csr_matvec_dot(D._h, xd.ptr, 0, y2.ptr, od.ptr))
    d_fused = float(od.download()[0])
    check(L.psp_k_dot(n, xd.ptr, yd.ptr, od.ptr))
    d_sep = float(od.download()[0])
    assert abs(d_fused - d_sep) <= 1e-10 * max(abs(d_sep), 1.0)
