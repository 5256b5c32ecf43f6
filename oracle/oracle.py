"""ctypes front end of the CPU oracle (oracle/pysparse_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under pysparse_amd/ may import this module.

Parity status: pinned (see the header of pysparse_oracle.c and tests/golden/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_LIB_PATH = os.path.join(REF_DIR, "libref_pcg.so")
REF_BIN_PATH = os.path.join(REF_DIR, "poisson_test")
REF_KRYLOV_PATH = os.path.join(REF_DIR, "libref_krylov.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, ndim=1, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, ndim=1, flags="C_CONTIGUOUS")


def build(ref=None):
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if ref is None:
        ref = os.path.isdir("/root/reference/examples/poisson_test")
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # the reference is single-threaded; so is its BLAS here

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _declare(L):
    L.orc_ddot.restype = C.c_double
    L.orc_ddot.argtypes = [C.c_int, _dp, _dp]
    L.orc_dnrm2.restype = C.c_double
    L.orc_dnrm2.argtypes = [C.c_int, _dp]
    L.orc_csr_matvec.restype = None
    L.orc_csr_matvec.argtypes = [C.c_int, _dp, _dp, _dp, _ip, _ip]
    L.orc_csr_matvec_threads.restype = C.c_int
    L.orc_csr_matvec_threads.argtypes = [C.c_int, _dp, _dp, _dp, _ip, _ip, C.c_int]
    L.orc_csr_matvec_stride.restype = None
    L.orc_csr_matvec_stride.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, _dp, _ip, _ip]
    L.orc_csr_matvec_transp.restype = None
    L.orc_csr_matvec_transp.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _ip, _ip]
    L.orc_sss_matvec.restype = None
    L.orc_sss_matvec.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _ip, _ip]
    L.orc_sss_getitem.restype = C.c_double
    L.orc_sss_getitem.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip, _ip]
    L.orc_jacobi_setup.restype = C.c_int
    L.orc_jacobi_setup.argtypes = [C.c_int, _dp, C.c_double, _dp]
    L.orc_poisson_csr.restype = C.c_long
    L.orc_poisson_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_poisson_sss.restype = C.c_long
    L.orc_poisson_sss.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ipt, dpt = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.orc_pcg_csr.restype = C.c_int
    L.orc_pcg_csr.argtypes = [C.c_int, _dp, _ip, _ip, C.c_void_p, C.c_int, _dp, _dp, C.c_double,
                              C.c_int, ipt, dpt, ipt, C.c_void_p]
    L.orc_pcg_sss.restype = C.c_int
    L.orc_pcg_sss.argtypes = [C.c_int, _dp, _dp, _ip, _ip, C.c_void_p, C.c_int, _dp, _dp, C.c_double,
                              C.c_int, ipt, dpt, ipt, C.c_void_p]
    L.orc_minres_csr.restype = C.c_int
    L.orc_minres_csr.argtypes = [C.c_int, _dp, _ip, _ip, C.c_void_p, C.c_int, _dp, _dp, C.c_double,
                                 C.c_int, ipt, dpt, C.c_void_p]
    L.orc_minres_sss.restype = C.c_int
    L.orc_minres_sss.argtypes = [C.c_int, _dp, _dp, _ip, _ip, C.c_void_p, C.c_int, _dp, _dp, C.c_double,
                                 C.c_int, ipt, dpt, C.c_void_p]
    L.orc_symgs.restype = None
    L.orc_symgs.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]
    L.orc_ssor.restype = None
    L.orc_ssor.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_double, C.c_int]
    L.orc_pcg_sss_ssor.restype = C.c_int
    L.orc_pcg_sss_ssor.argtypes = [C.c_int, _dp, _dp, _ip, _ip, C.c_double, C.c_int, _dp, _dp, C.c_double,
                                   C.c_int, ipt, dpt, ipt, C.c_void_p]
    L.orc_krylov_more.restype = C.c_int
    L.orc_krylov_more.argtypes = [C.c_int, C.c_int, _dp, C.c_void_p, _ip, _ip, C.c_void_p, _dp, _dp, C.c_double,
                                  C.c_int, C.c_int, ipt, dpt]
    # ll_mat feeder restatement
    L.orc_ll_new.restype = C.c_void_p
    L.orc_ll_new.argtypes = [C.c_int] * 5
    L.orc_ll_free.restype = None
    L.orc_ll_free.argtypes = [C.c_void_p]
    L.orc_ll_nnz.restype = C.c_int
    L.orc_ll_nnz.argtypes = [C.c_void_p]
    L.orc_ll_set.restype = C.c_int
    L.orc_ll_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    L.orc_ll_get.restype = C.c_double
    L.orc_ll_get.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_ll_csr_nnz.restype = C.c_int
    L.orc_ll_csr_nnz.argtypes = [C.c_void_p]
    L.orc_ll_to_csr.restype = None
    L.orc_ll_to_csr.argtypes = [C.c_void_p, _dp, _ip, _ip]
    L.orc_ll_sss_nnz.restype = C.c_int
    L.orc_ll_sss_nnz.argtypes = [C.c_void_p]
    L.orc_ll_to_sss.restype = None
    L.orc_ll_to_sss.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip]
    L.orc_ll_matvec.restype = None
    L.orc_ll_matvec.argtypes = [C.c_void_p, _dp, _dp]
    # bound shims for the compiled reference
    L.orc_bind_csr.restype = None
    L.orc_bind_csr.argtypes = [C.c_int, C.c_int, _dp, _ip, _ip]
    L.orc_bind_sss.restype = None
    L.orc_bind_sss.argtypes = [C.c_int, _dp, _dp, _ip, _ip]
    L.orc_bind_dinv.restype = None
    L.orc_bind_dinv.argtypes = [C.c_int, _dp]
    L.orc_solve_cb.restype = C.c_int
    L.orc_solve_cb.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, ipt, dpt, ipt, _dp,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


# ----------------------------------------------------------------------------- containers

class CSR:
    """Host CSR triple with the reference's field names (csr_mat.h:6-13)."""

    def __init__(self, shape, val, col, ind):
        self.shape = (int(shape[0]), int(shape[1]))
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.ind = np.ascontiguousarray(ind, dtype=np.int32)
        self.nnz = int(self.val.shape[0])

    def matvec(self, x, y):
        lib().orc_csr_matvec(self.shape[0], x, y, self.val, self.col, self.ind)

    def matvec_threads(self, x, y, nthreads):
        """NOT the reference (single-threaded): the same rows over POSIX threads, same bits; -> threads that ran"""
        return lib().orc_csr_matvec_threads(self.shape[0], x, y, self.val, self.col, self.ind, nthreads)

    def matvec_transp(self, x, y):
        lib().orc_csr_matvec_transp(self.shape[0], self.shape[1], x, y, self.val, self.col, self.ind)

    def diagonal(self):
        d = np.zeros(self.shape[0])
        for i in range(self.shape[0]):
            for k in range(self.ind[i], self.ind[i + 1]):
                if self.col[k] == i:
                    d[i] = self.val[k]
        return d


class SSS:
    """Host symmetric-skyline storage (sss_mat.h:6-14)."""

    def __init__(self, n, val, diag, col, ind):
        self.n = int(n)
        self.shape = (self.n, self.n)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.diag = np.ascontiguousarray(diag, dtype=np.float64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.ind = np.ascontiguousarray(ind, dtype=np.int32)
        self.nnz_lower = int(self.val.shape[0])
        self.nnz = self.nnz_lower + self.n  # sss_mat.c:155

    def matvec(self, x, y):
        lib().orc_sss_matvec(self.n, x, y, self.val, self.diag, self.col, self.ind)

    def getitem(self, i, j):
        return lib().orc_sss_getitem(i, j, self.val, self.diag, self.col, self.ind)


class LL:
    """The ll_mat feeder restatement (sorted insertion + conversions)."""

    def __init__(self, m, n, size_hint=1000, sym=False, store_zeros=False):
        self._h = lib().orc_ll_new(m, n, size_hint, int(sym), int(store_zeros))
        self.shape = (m, n)
        self.issym = bool(sym)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ll_free(self._h)
            self._h = None

    @property
    def nnz(self):
        return lib().orc_ll_nnz(self._h)

    def __setitem__(self, ij, v):
        rc = lib().orc_ll_set(self._h, int(ij[0]), int(ij[1]), float(v))
        if rc:
            raise IndexError("write operation to upper triangle of symmetric matrix" if rc == -1
                             else "indices out of range")

    def __getitem__(self, ij):
        return lib().orc_ll_get(self._h, int(ij[0]), int(ij[1]))

    def to_csr(self):
        nnz = lib().orc_ll_csr_nnz(self._h)
        val = np.empty(nnz)
        col = np.empty(nnz, dtype=np.int32)
        ind = np.empty(self.shape[0] + 1, dtype=np.int32)
        lib().orc_ll_to_csr(self._h, val, col, ind)
        return CSR(self.shape, val, col, ind)

    def to_sss(self):
        assert self.shape[0] == self.shape[1]
        nnz = lib().orc_ll_sss_nnz(self._h)
        val = np.empty(nnz)
        col = np.empty(nnz, dtype=np.int32)
        diag = np.empty(self.shape[0])
        ind = np.empty(self.shape[0] + 1, dtype=np.int32)
        lib().orc_ll_to_sss(self._h, val, diag, col, ind)
        return SSS(self.shape[0], val, diag, col, ind)

    def matvec(self, x, y):
        lib().orc_ll_matvec(self._h, x, y)


# ----------------------------------------------------------------------------- generators

def poisson_csr(nx, ny, nz=0):
    """Direct CSR of the 5-/7-point operator in tools/poisson.py ordering (nz=0: 2-D)."""
    L = lib()
    nnz = L.orc_poisson_csr(nx, ny, nz, None, None, None)
    n = nx * ny * max(nz, 1)
    val = np.empty(nnz)
    col = np.empty(nnz, dtype=np.int32)
    ind = np.empty(n + 1, dtype=np.int32)
    L.orc_poisson_csr(nx, ny, nz, val.ctypes.data, col.ctypes.data, ind.ctypes.data)
    return CSR((n, n), val, col, ind)


def poisson_sss(nx, ny, nz=0):
    L = lib()
    nnz = L.orc_poisson_sss(nx, ny, nz, None, None, None, None)
    n = nx * ny * max(nz, 1)
    val = np.empty(nnz)
    col = np.empty(nnz, dtype=np.int32)
    diag = np.empty(n)
    ind = np.empty(n + 1, dtype=np.int32)
    L.orc_poisson_sss(nx, ny, nz, val.ctypes.data, diag.ctypes.data, col.ctypes.data, ind.ctypes.data)
    return SSS(n, val, diag, col, ind)


def poisson2d_ll(n, sym=False):
    """tools/poisson.py:22-37 / :39-50 through the ll_mat restatement (element-wise)."""
    A = LL(n * n, n * n, (3 * n * n - 2 * n) if sym else (5 * n * n - 4 * n), sym=sym)
    for i in range(n):
        for j in range(n):
            k = i + n * j
            A[k, k] = 4
            if i > 0:
                A[k, k - 1] = -1
            if not sym and i < n - 1:
                A[k, k + 1] = -1
            if j > 0:
                A[k, k - n] = -1
            if not sym and j < n - 1:
                A[k, k + n] = -1
    return A


def tendigit_sss(n=20000):
    """examples/tendigit.py:26-38: diag = first n primes, ones at offsets 2^k below it."""
    sieve = np.ones(max(16, int(n * (np.log(n) + np.log(np.log(n)) + 2))), dtype=bool)
    sieve[:2] = False
    for p in range(2, int(len(sieve) ** 0.5) + 1):
        if sieve[p]:
            sieve[p * p::p] = False
    primes = np.flatnonzero(sieve)[:n].astype(np.float64)
    assert len(primes) == n
    offs = []
    d = 1
    while d < n:
        offs.append(d)
        d *= 2
    ind = np.zeros(n + 1, dtype=np.int32)
    cols = []
    for i in range(n):
        c = sorted(i - d for d in offs if i - d >= 0)
        cols.extend(c)
        ind[i + 1] = len(cols)
    col = np.array(cols, dtype=np.int32)
    return SSS(n, np.ones(len(col)), primes, col, ind)


def sss_to_csr(S):
    """Full CSR (sorted columns) of an SSS matrix == ll_mat_sym.to_csr() layout."""
    n = S.n
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(S.ind))
    r = np.concatenate([rows, S.col.astype(np.int64), np.arange(n, dtype=np.int64)])
    c = np.concatenate([S.col.astype(np.int64), rows, np.arange(n, dtype=np.int64)])
    v = np.concatenate([S.val, S.val, S.diag])
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=ind[1:])
    return CSR((n, n), v, c.astype(np.int32), ind)


# ----------------------------------------------------------------------------- solvers

def jacobi_dinv(diag, omega=1.0):
    dinv = np.empty_like(diag)
    rc = lib().orc_jacobi_setup(len(diag), np.ascontiguousarray(diag), omega, dinv)
    if rc:
        raise ValueError("diagonal element close to zero")
    return dinv


def ssor_apply(S, x, y, omega=1.0, steps=1):
    """y = K x for precon.ssor(S, omega, steps) (preconmodule.c:95-223); y is also read when steps == 0."""
    n = S.n
    L = lib()
    if omega == 1.0:
        L.orc_symgs(n, np.ascontiguousarray(x), y, np.empty(n), S.val, S.diag, S.col, S.ind, steps)
    else:
        L.orc_ssor(n, np.ascontiguousarray(x), y, np.empty(n), np.empty(n), S.val, S.diag, S.col, S.ind,
                   omega, steps)


def pcg_ssor(S, b, x, tol, maxit, omega=1.0, steps=1, hist=False):
    """pcg(S, b, x, tol, maxit, precon.ssor(S, omega, steps)) on the CPU."""
    it, fl, rr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    h = np.full(maxit + 2, np.nan) if hist else None
    rc = lib().orc_pcg_sss_ssor(S.n, S.val, S.diag, S.col, S.ind, omega, steps, x, b, tol, maxit,
                                C.byref(it), C.byref(rr), C.byref(fl), _opt(h))
    assert rc == 0
    return (fl.value, it.value, rr.value, h) if hist else (fl.value, it.value, rr.value)


def pcg(A, b, x, tol, maxit, dinv=None, steps=1, hist=False):
    """info, iter, relres[, hist] = pcg(...) with the reference's semantics (pcg.c:22-171)."""
    n = A.shape[0]
    it, fl, rr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    h = np.full(maxit + 2, np.nan) if hist else None
    L = lib()
    if isinstance(A, CSR):
        rc = L.orc_pcg_csr(n, A.val, A.col, A.ind, _opt(dinv), steps, x, b, tol, maxit,
                           C.byref(it), C.byref(rr), C.byref(fl), _opt(h))
    else:
        rc = L.orc_pcg_sss(n, A.val, A.diag, A.col, A.ind, _opt(dinv), steps, x, b, tol, maxit,
                           C.byref(it), C.byref(rr), C.byref(fl), _opt(h))
    assert rc == 0
    return (fl.value, it.value, rr.value, h) if hist else (fl.value, it.value, rr.value)


def minres(A, b, x, tol, maxit, dinv=None, steps=1, hist=False):
    n = A.shape[0]
    it, rr = C.c_int(0), C.c_double(np.nan)
    h = np.full(maxit + 2, np.nan) if hist else None
    L = lib()
    if isinstance(A, CSR):
        info = L.orc_minres_csr(n, A.val, A.col, A.ind, _opt(dinv), steps, x, b, tol, maxit,
                                C.byref(it), C.byref(rr), _opt(h))
    else:
        info = L.orc_minres_sss(n, A.val, A.diag, A.col, A.ind, _opt(dinv), steps, x, b, tol, maxit,
                                C.byref(it), C.byref(rr), _opt(h))
    return (info, it.value, rr.value, h) if hist else (info, it.value, rr.value)


_MORE = {"cgs": 0, "bicgstab": 1, "qmrs": 2, "gmres": 3}


def krylov_more(solver, A, b, x, tol, maxit, dinv=None, dim=20):
    """cgs / bicgstab / qmrs / gmres restatements (pinned by oracle/_ref/libref_krylov.so, see pysparse_oracle.c)."""
    n = A.shape[0]
    it, rr = C.c_int(0), C.c_double(0.0)
    info = lib().orc_krylov_more(_MORE[solver], n, A.val, _opt(A.diag) if isinstance(A, SSS) else None, A.col, A.ind,
                                 _opt(dinv), x, b, tol, maxit, dim, C.byref(it), C.byref(rr))
    return info, it.value, rr.value


# ----------------------------------------------------------------------------- compiled reference

_ref = None


def have_ref():
    return os.path.exists(REF_LIB_PATH)


def ref_lib():
    """oracle/_ref/libref_pcg.so = the reference's examples/poisson_test/pcg.c, unmodified."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_LIB_PATH)
        fn = C.c_void_p
        _ref.pcg.restype = None
        _ref.pcg.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int),
                             C.POINTER(C.c_double), C.POINTER(C.c_int), _dp, fn, fn]
    return _ref


# the standalone program numbers its flags 0/1/2/3/4; the extension module -1/-2/-5/-6
# (examples/poisson_test/pcg.c header vs pysparse/itsolvers/src/pcg.c:70,102,110,118,160)
REF_FLAG_TO_INFO = {0: 0, 1: -1, 2: -2, 3: -5, 4: -6}


def ref_pcg(A, b, x, tol, maxit, dinv=None):
    """Run the COMPILED REFERENCE PCG on operator A (CSR or SSS) with K=None or Jacobi(1)."""
    L, R = lib(), ref_lib()
    n = A.shape[0]
    if isinstance(A, CSR):
        L.orc_bind_csr(A.shape[0], A.shape[1], A.val, A.col, A.ind)
        mv = C.cast(L.orc_bound_csr_matvec, C.c_void_p)
    else:
        L.orc_bind_sss(A.n, A.val, A.diag, A.col, A.ind)
        mv = C.cast(L.orc_bound_sss_matvec, C.c_void_p)
    pc = None
    if dinv is not None:
        L.orc_bind_dinv(n, dinv)
        pc = C.cast(L.orc_bound_jacobi, C.c_void_p)
    it, fl, rr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    work = np.zeros(4 * n)
    R.pcg(n, x, b, tol, maxit, 0, C.byref(it), C.byref(rr), C.byref(fl), work, mv, pc)
    return REF_FLAG_TO_INFO[fl.value], it.value, rr.value


# ----------------------------------------------------------------------------- compiled reference, module kernels

class _CsrCtx(C.Structure):  # orc_csr_t
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("va", C.c_void_p), ("ja", C.c_void_p), ("ia", C.c_void_p)]


class _SssCtx(C.Structure):  # orc_sss_t
    _fields_ = [("n", C.c_int), ("va", C.c_void_p), ("da", C.c_void_p), ("ja", C.c_void_p), ("ia", C.c_void_p)]


class _CsrMtCtx(C.Structure):  # orc_csr_mt_t
    _fields_ = [("A", _CsrCtx), ("nthreads", C.c_int)]


class _JacobiMtCtx(C.Structure):  # orc_jacobi_mt_t
    _fields_ = [("n", C.c_int), ("dinv", C.c_void_p), ("nthreads", C.c_int)]


class _JacobiCtx(C.Structure):  # orc_jacobi_t
    _fields_ = [("n", C.c_int), ("dinv", C.c_void_p), ("steps", C.c_int), ("temp", C.c_void_p),
                ("matvec", C.c_void_p), ("mctx", C.c_void_p)]


class _SsorCtx(C.Structure):  # orc_ssor_t
    _fields_ = [("n", C.c_int), ("va", C.c_void_p), ("da", C.c_void_p), ("ja", C.c_void_p), ("ia", C.c_void_p),
                ("omega", C.c_double), ("steps", C.c_int), ("temp", C.c_void_p), ("temp2", C.c_void_p)]


REF_SOLVERS = {"pcg": 0, "minres": 1, "cgs": 2, "bicgstab": 3, "qmrs": 4, "gmres": 5}
_refk = None


def have_ref_krylov():
    return os.path.exists(REF_KRYLOV_PATH)


def ref_krylov_lib():
    """oracle/_ref/libref_krylov.so = pysparse/itsolvers/src/{pcg,minres,cgs,bicgstab,qmrs,gmres}.c
    compiled unmodified + oracle/ref_krylov_harness.c (the itsolvers_spmatrix table)."""
    global _refk
    if _refk is None:
        _refk = C.CDLL(REF_KRYLOV_PATH)
        ipt, dpt = C.POINTER(C.c_int), C.POINTER(C.c_double)
        _refk.refk_solve.restype = C.c_int
        _refk.refk_solve.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, ipt, dpt, ipt, _dp,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return _refk


def _operator_ctx(A, threads=1):
    """(callback address, context struct, keep-alive list) of the oracle's matvec for A.  threads > 1 (CSR only): the
    row-parallel callback -- the same bits per row; the kernel that calls it stays what it is"""
    L = lib()
    if isinstance(A, CSR):
        ctx = _CsrCtx(A.shape[0], A.shape[1], A.val.ctypes.data, A.col.ctypes.data, A.ind.ctypes.data)
        if threads > 1:
            return C.cast(L.orc_csr_matvec_threads_cb, C.c_void_p), _CsrMtCtx(ctx, int(threads))
        return C.cast(L.orc_csr_matvec_cb, C.c_void_p), ctx
    ctx = _SssCtx(A.n, A.val.ctypes.data, A.diag.ctypes.data, A.col.ctypes.data, A.ind.ctypes.data)
    return C.cast(L.orc_sss_matvec_cb, C.c_void_p), ctx


def _precon_ctx(A, K, mv, mctx, threads=1):
    """(callback address, context struct, keep-alive list) for K = None | ("jacobi", dinv[, steps]) |
    ("ssor", omega, steps) -- the oracle's restatements of preconmodule.c"""
    L = lib()
    n = A.shape[0]
    if K is None:
        return None, None, []
    if K[0] == "jacobi":
        dinv = np.ascontiguousarray(K[1], dtype=np.float64)
        steps = K[2] if len(K) > 2 else 1
        if threads > 1 and steps == 1:
            return C.cast(L.orc_jacobi_threads_cb, C.c_void_p), _JacobiMtCtx(n, dinv.ctypes.data, int(threads)), [dinv]
        temp = np.zeros(max(n, 1))
        ctx = _JacobiCtx(n, dinv.ctypes.data, steps, temp.ctypes.data, mv, C.addressof(mctx))
        return C.cast(L.orc_jacobi_apply, C.c_void_p), ctx, [dinv, temp]
    if K[0] == "ssor":
        assert isinstance(A, SSS)
        t1, t2 = np.zeros(max(n, 1)), np.zeros(max(n, 1))
        ctx = _SsorCtx(A.n, A.val.ctypes.data, A.diag.ctypes.data, A.col.ctypes.data, A.ind.ctypes.data,
                       float(K[1]), int(K[2]), t1.ctypes.data, t2.ctypes.data)
        return C.cast(L.orc_ssor_apply, C.c_void_p), ctx, [t1, t2]
    raise ValueError(K[0])


def _solve_cb(fn, has_fail, solver, A, b, x, tol, maxit, K, dim, fails=(), threads=1):
    n = A.shape[0]
    mv, mctx = _operator_ctx(A, threads)
    pc, pctx, keep = _precon_ctx(A, K, mv, mctx, threads)
    it, info, rr = C.c_int(0), C.c_int(0), C.c_double(np.nan)
    work = np.zeros(8 * max(n, 1))
    bb = np.ascontiguousarray(b, dtype=np.float64).copy()  # the reference kernels take a non-const b
    args = [REF_SOLVERS[solver], n, x, bb, tol, maxit, dim, C.byref(it), C.byref(rr), C.byref(info), work,
            mv, C.addressof(mctx), pc, C.addressof(pctx) if pctx is not None else None]
    rc = fn(*(args + list(fails))) if has_fail else fn(*args)
    del keep
    return info.value, it.value, rr.value, rc


def ref_krylov(solver, A, b, x, tol, maxit, K=None, dim=20, mv_fail_after=-1, pc_fail_after=-1, threads=1):
    """info, iter, relres, rc = the COMPILED REFERENCE kernel `solver` (module sources, unmodified) on
    operator A (CSR or SSS; the matvec is the oracle's restatement -- csr_mat.c / sss_mat.c do not
    compile here) with K = None, ("jacobi", dinv[, steps]) or ("ssor", omega, steps) on an SSS matrix.
    relres comes back NaN when the kernel never wrote it (minres on -3 / -6; minres.c:79-80,158-159).
    rc is the kernel's own return value (-1 also means "a callback raised", e.g. pcg.c:8-11).
    threads > 1 (CSR, K = None or jacobi with one step): the operator / preconditioner callbacks are row-parallel (same
    bits per row and element); the kernel -- the reference's compiled code -- is unchanged."""
    return _solve_cb(ref_krylov_lib().refk_solve, True, solver, A, b, x, tol, maxit, K, dim,
                     (mv_fail_after, pc_fail_after), threads)


def solve(solver, A, b, x, tol, maxit, K=None, dim=20):
    """info, iter, relres, rc = the oracle's RESTATEMENT of `solver`, on the same operator / preconditioner
    contexts and with the same calling convention as ref_krylov."""
    return _solve_cb(lib().orc_solve_cb, False, solver, A, b, x, tol, maxit, K, dim)
