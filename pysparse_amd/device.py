"""Python handles over the C ABI (include/pysparse_hip.h) for NumPy callers.

This is the thin layer bench.py, the GPU parity tests and the multi-GPU driver use; it
mirrors the reference's operator protocol (objects with `shape` + `matvec(x, y)` /
`precon(x, y)`; `info, iter, relres = pcg(A, b, x, tol, maxit, K)`), see
doc/pysparse/source/itsolvers.rst:26-76 and precon.rst:15-24 of the reference.
All arithmetic happens in libpysparse_hip.so on the GPU; nothing here computes.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib


def _f64(a, n, what):
    if not isinstance(a, np.ndarray) or a.ndim != 1 or a.dtype != np.float64 or a.shape[0] != n:
        raise ValueError("%s must be a 1-dimensional double array of appropriate size." % what)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """n doubles (or raw bytes) of HBM."""

    def __init__(self, n, dtype=np.float64):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.nbytes = self.n * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().psp_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.size, a.dtype)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.n
        check(lib().psp_memcpy_h2d(self.ptr, _ptr(a), self.nbytes))

    def download(self):
        out = np.empty(self.n, dtype=self.dtype)
        check(lib().psp_memcpy_d2h(_ptr(out), self.ptr, self.nbytes))
        return out

    def zero(self):
        check(lib().psp_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if getattr(self, "ptr", None):
            lib().psp_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceCSR:
    """csr_mat on the GPU (csr_mat.h:6-13): shape, nnz, matvec, matvec_transp."""

    def __init__(self, handle):
        self._h = handle
        nr, nc, nz = C.c_int(), C.c_int(), C.c_int()
        check(lib().psp_csr_shape(handle, C.byref(nr), C.byref(nc), C.byref(nz)))
        self.shape = (nr.value, nc.value)
        self.nnz = nz.value if nz.value >= 0 else int(lib().psp_csr_nnz64(handle))  # > 2^31: the 64-bit count

    @classmethod
    def from_arrays64(cls, shape, ind, col, val):
        """CSR triple with 64-bit row offsets (nnz may exceed the reference's C int, csr_mat.h:6-13)"""
        ind = np.ascontiguousarray(ind, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if ind.shape[0] != shape[0] + 1 or col.shape[0] != val.shape[0]:
            raise ValueError("inconsistent CSR arrays")
        h = C.c_void_p()
        check(lib().psp_csr_create64(shape[0], shape[1], val.shape[0], _ptr(ind), _ptr(col), _ptr(val), C.byref(h)))
        return cls(h)

    @classmethod
    def random_banded(cls, nrows, ncols, m, stride, seed=0):
        """synthetic general CSR generated on the device (nrows*m may exceed 2^31), psp_csr_random_banded"""
        h = C.c_void_p()
        check(lib().psp_csr_random_banded(nrows, ncols, m, stride, seed, C.byref(h)))
        return cls(h)

    def download_rows(self, row_lo, row_hi):
        """(ind64 relative to row_lo, col, val) of rows [row_lo, row_hi)"""
        ind = np.empty(row_hi - row_lo + 1, dtype=np.int64)
        check(lib().psp_csr_download_rows(self._h, row_lo, row_hi, _ptr(ind), None, None))
        col = np.empty(int(ind[-1]), dtype=np.int32)
        val = np.empty(int(ind[-1]), dtype=np.float64)
        check(lib().psp_csr_download_rows(self._h, row_lo, row_hi, _ptr(ind), _ptr(col), _ptr(val)))
        return ind, col, val

    @classmethod
    def from_arrays(cls, shape, ind, col, val):
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if ind.shape[0] != shape[0] + 1 or col.shape[0] != val.shape[0]:
            raise ValueError("inconsistent CSR arrays")
        h = C.c_void_p()
        check(lib().psp_csr_create(shape[0], shape[1], val.shape[0], _ptr(ind), _ptr(col), _ptr(val),
                                   C.byref(h)))
        return cls(h)

    @classmethod
    def poisson(cls, nx, ny, nz=0):
        h = C.c_void_p()
        check(lib().psp_csr_poisson(nx, ny, nz, C.byref(h)))
        return cls(h)

    @classmethod
    def poisson_multi(cls, nx, ny, nz=0, devices=(0,)):
        """the Poisson operator as row slabs on a LIST of devices, one process (psp_csr_poisson_multi): matvec,
        DeviceJacobi(A), pcg and minres work on it; a device may be listed more than once"""
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(lib().psp_csr_poisson_multi(nx, ny, nz, _ptr(dv), len(dv), C.byref(h)))
        A = cls(h)
        A.nnz = int(lib().psp_csr_nnz64(h))
        return A

    @classmethod
    def from_arrays_multi(cls, shape, ind, col, val, devices=(0,)):
        """any square CSR matrix as row blocks on a list of devices (psp_csr_create_multi)"""
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if ind.shape[0] != shape[0] + 1 or col.shape[0] != val.shape[0]:
            raise ValueError("inconsistent CSR arrays")
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(lib().psp_csr_create_multi(shape[0], shape[1], val.shape[0], _ptr(ind), _ptr(col), _ptr(val),
                                         _ptr(dv), len(dv), C.byref(h)))
        return cls(h)

    def multi_info(self):
        """(ranks, distinct devices, reductions through RCCL) -- (0, 0, False) for a single-device matrix"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().psp_csr_multi_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, bool(c.value)

    @classmethod
    def poisson_big(cls, nx, ny, nz=0):
        """the Poisson operator in the index-free w4 layout only (nnz may exceed 32 bits: 1024^3)"""
        h = C.c_void_p()
        check(lib().psp_csr_poisson_big(nx, ny, nz, C.byref(h)))
        A = cls(h)
        A.nnz = int(lib().psp_csr_nnz64(h))
        return A

    @classmethod
    def poisson_big_slab(cls, nx, ny, nz, row_lo, row_hi, col_shift, ncols_local):
        """row slab of the index-free operator (per-rank nnz may exceed 32 bits: 1024^3 over 2 GPUs)"""
        h = C.c_void_p()
        check(lib().psp_csr_poisson_big_slab(nx, ny, nz, row_lo, row_hi, col_shift, ncols_local, C.byref(h)))
        A = cls(h)
        A.nnz = int(lib().psp_csr_nnz64(h))
        return A

    @classmethod
    def poisson_slab(cls, nx, ny, nz, row_lo, row_hi, col_shift, ncols_local):
        h = C.c_void_p()
        check(lib().psp_csr_poisson_slab(nx, ny, nz, row_lo, row_hi, col_shift, ncols_local, C.byref(h)))
        return cls(h)

    def matvec(self, x, y):
        _f64(x, self.shape[1], "arg 1")
        _f64(y, self.shape[0], "arg 2")
        es = x.itemsize
        check(lib().psp_csr_matvec_stride(self._h, _ptr(x), x.strides[0] // es, _ptr(y), y.strides[0] // es))

    def matvec_transp(self, x, y):
        _f64(x, self.shape[0], "arg 1")
        _f64(y, self.shape[1], "arg 2")
        es = x.itemsize
        check(lib().psp_csr_matvec_transp_stride(self._h, _ptr(x), x.strides[0] // es, _ptr(y),
                                                 y.strides[0] // es))

    def matvec_dev(self, x_ptr, y_ptr):
        check(lib().psp_csr_matvec_dev(self._h, x_ptr, y_ptr))

    def download(self):
        ind = np.empty(self.shape[0] + 1, dtype=np.int32)
        col = np.empty(self.nnz, dtype=np.int32)
        val = np.empty(self.nnz, dtype=np.float64)
        check(lib().psp_csr_download(self._h, _ptr(ind), _ptr(col), _ptr(val)))
        return ind, col, val

    def diagonal(self):
        d = np.empty(self.shape[0])
        check(lib().psp_csr_diagonal(self._h, _ptr(d)))
        return d

    def set_variant(self, v):
        check(lib().psp_csr_set_variant(self._h, int(v)))

    def set_schedule(self, strip_rows):
        check(lib().psp_csr_set_schedule(self._h, int(strip_rows)))

    def kernel_info(self):
        """(kernel name, {nb, max_blocks, scheduled, half_band}) of the product y = A x."""
        name = C.create_string_buffer(160)
        info = (C.c_int * 4)()
        check(lib().psp_csr_kernel_info(self._h, name, 160, info))
        return name.value.decode(), {"nb": info[0], "max_blocks": info[1], "scheduled": bool(info[2]),
                                     "half_band": info[3], "setup_ms": self.setup_info()["reorder_ms"]}

    def release_arrays(self):
        """psp_csr_release_arrays: an offset-structured operator keeps its index-free tables only (no download, no
        variants afterwards): 1.65 x -> 1.0 x device memory"""
        check(lib().psp_csr_release_arrays(self._h))

    def prepare(self, expected_products):
        """psp_csr_prepare: the caller expects about this many products / solver iterations with this handle -- whatever
        pays for itself within them (the renumbered copy of an irregular numbering: from 2048 on) is built at the next
        product instead of after that many have been counted"""
        check(lib().psp_csr_prepare(self._h, int(expected_products)))

    def setup_info(self):
        v = (C.c_double * 4)()
        check(lib().psp_csr_setup_info(self._h, v))
        return {"reorder_ms": v[0], "products_counted": int(v[1]), "reorder_after": int(v[2]), "reorder_state": int(v[3])}

    def renumbering(self):
        """perm[new] = old row of the renumbered copy behind csr_spmv_w3_rcm, or None when the handle has none."""
        perm = np.empty(self.shape[0], dtype=np.int32)
        have = C.c_int()
        check(lib().psp_csr_renumbering(self._h, perm.ctypes.data, C.byref(have)))
        self.renumbered_on = {0: None, 1: "host", 2: "device"}[have.value]
        return perm if have.value else None

    @property
    def device_bytes(self):
        return lib().psp_csr_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().psp_csr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSSS:
    """sss_mat on the GPU (sss_mat.h:6-14); nnz = strict-lower count + n (sss_mat.c:155)."""

    def __init__(self, handle):
        self._h = handle
        n, nz = C.c_int(), C.c_int()
        check(lib().psp_sss_shape(handle, C.byref(n), C.byref(nz)))
        self.n = n.value
        self.shape = (self.n, self.n)
        self.nnz = nz.value

    @classmethod
    def from_arrays(cls, n, ind, col, val, diag):
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        diag = np.ascontiguousarray(diag, dtype=np.float64)
        h = C.c_void_p()
        check(lib().psp_sss_create(n, val.shape[0], _ptr(ind), _ptr(col), _ptr(val), _ptr(diag), C.byref(h)))
        return cls(h)

    @classmethod
    def poisson(cls, nx, ny, nz=0):
        h = C.c_void_p()
        check(lib().psp_sss_poisson(nx, ny, nz, C.byref(h)))
        return cls(h)

    def matvec(self, x, y):
        _f64(x, self.n, "arg 1")
        _f64(y, self.n, "arg 2")
        es = x.itemsize
        check(lib().psp_sss_matvec_stride(self._h, _ptr(x), x.strides[0] // es, _ptr(y), y.strides[0] // es))

    matvec_transp = matvec  # sss_mat.c:108

    def matvec_dev(self, x_ptr, y_ptr):
        check(lib().psp_sss_matvec_dev(self._h, x_ptr, y_ptr))

    def set_variant(self, v):
        check(lib().psp_sss_set_variant(self._h, int(v)))

    def prepare(self, expected_products):
        """psp_sss_prepare (see DeviceCSR.prepare)"""
        check(lib().psp_sss_prepare(self._h, int(expected_products)))

    def setup_info(self):
        v = (C.c_double * 4)()
        check(lib().psp_sss_setup_info(self._h, v))
        return {"reorder_ms": v[0], "products_counted": int(v[1]), "reorder_after": int(v[2]), "reorder_state": int(v[3])}

    def kernel_info(self):
        name = C.create_string_buffer(64)
        info = (C.c_int * 4)()
        check(lib().psp_sss_kernel_info(self._h, name, 64, info))
        return name.value.decode(), {"nb": info[0], "max_blocks": info[1], "scheduled": bool(info[2]),
                                     "half_band": info[3], "setup_ms": self.setup_info()["reorder_ms"]}

    def __getitem__(self, ij):
        if not (isinstance(ij, tuple) and len(ij) == 2 and all(isinstance(t, (int, np.integer)) for t in ij)):
            raise IndexError("slices not supported")
        i, j = int(ij[0]), int(ij[1])
        if i < 0:
            i += self.n
        if j < 0:
            j += self.n
        if not (0 <= i < self.n and 0 <= j < self.n):
            raise IndexError("indices out of range")
        v = C.c_double()
        check(lib().psp_sss_getitem(self._h, i, j, C.byref(v)))
        return v.value

    def download(self):
        nl = self.nnz - self.n
        ind = np.empty(self.n + 1, dtype=np.int32)
        col = np.empty(nl, dtype=np.int32)
        val = np.empty(nl, dtype=np.float64)
        diag = np.empty(self.n, dtype=np.float64)
        check(lib().psp_sss_download(self._h, _ptr(ind), _ptr(col), _ptr(val), _ptr(diag)))
        return ind, col, val, diag

    def close(self):
        if getattr(self, "_h", None):
            lib().psp_sss_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Op:
    """psp_op_t wrapper; keeps the Python object and the ctypes callback alive."""

    def __init__(self, obj, method):
        self.obj = obj
        self.exc = None
        h = C.c_void_p()
        if isinstance(obj, DeviceCSR):
            check(lib().psp_op_from_csr(obj._h, C.byref(h)))
        elif isinstance(obj, DeviceSSS):
            check(lib().psp_op_from_sss(obj._h, C.byref(h)))
        elif isinstance(obj, DeviceJacobi):
            check(lib().psp_op_from_jacobi(obj._h, C.byref(h)))
        elif isinstance(obj, DeviceSSOR):
            check(lib().psp_op_from_ssor(obj._h, C.byref(h)))
        else:
            # duck-typed operator: shape + matvec/precon (spmatrixmodule.c:86-132, :169-248)
            shape = obj.shape
            if len(shape) != 2:
                raise ValueError("invalid matrix shape")
            if int(shape[0]) != int(shape[1]):
                raise ValueError("matrix is not square")
            n = int(shape[0])
            fn = getattr(obj, method)

            def trampoline(ctx, nn, xp, yp):
                try:
                    x = np.ctypeslib.as_array(xp, shape=(nn,))
                    y = np.ctypeslib.as_array(yp, shape=(nn,))
                    fn(x, y)
                    return 0
                except BaseException as e:  # noqa: BLE001 - re-raised by the caller
                    self.exc = e
                    return 1

            self._cb = _capi.HOST_APPLY_FN(trampoline)
            check(lib().psp_op_from_callback(n, self._cb, None, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().psp_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceJacobi:
    """precon.jacobi(A, omega=1.0, steps=1) (preconmodule.c:352-412, :470-485)."""

    def __init__(self, A, omega=1.0, steps=1):
        h = C.c_void_p()
        self._A = A
        self._aop = None
        if isinstance(A, DeviceCSR):
            check(self._map(lib().psp_jacobi_create_csr(A._h, omega, steps, C.byref(h))))
            n = A.shape[0]
        elif isinstance(A, DeviceSSS):
            check(self._map(lib().psp_jacobi_create_sss(A._h, omega, steps, C.byref(h))))
            n = A.n
        else:
            shape = A.shape
            if int(shape[0]) != int(shape[1]):
                raise ValueError("matrix is not square")
            n = int(shape[0])
            diag = np.array([float(A[i, i]) for i in range(n)])  # preconmodule.c:389-392
            if steps > 1:
                self._aop = _Op(A, "matvec")
            check(self._map(lib().psp_jacobi_create_diag(n, _ptr(diag), omega, steps,
                                                         self._aop._h if self._aop else None, C.byref(h))))
        self._h = h
        self.shape = (n, n)

    @staticmethod
    def _map(rc):
        if rc == -4:  # PSP_ESINGULAR -> the reference's ValueError (preconmodule.c:396)
            raise ValueError("diagonal element close to zero")
        return rc

    def precon(self, x, y):
        n = self.shape[0]
        for k, a in ((1, x), (2, y)):
            if (not isinstance(a, np.ndarray) or a.ndim != 1 or a.dtype != np.float64 or a.shape[0] != n
                    or not a.flags.c_contiguous):
                raise ValueError("arg %d must be a contiguous 1-dimensional double array of appropriate size." % k)
        check(lib().psp_jacobi_precon(self._h, _ptr(x), _ptr(y)))

    def close(self):
        if getattr(self, "_h", None):
            lib().psp_jacobi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSSOR:
    """precon.ssor(A, omega=1.0, steps=1) on an sss_mat (preconmodule.c:414-459, :95-223)."""

    def __init__(self, A, omega=1.0, steps=1):
        if not isinstance(A, DeviceSSS):
            raise TypeError("ssor() argument 1 must be sss_mat")  # "O!" with SSSMatType, preconmodule.c:499
        h = C.c_void_p()
        check(lib().psp_ssor_create(A._h, float(omega), int(steps), C.byref(h)))
        self._A = A  # the handle borrows the matrix
        self._h = h
        self.shape = (A.n, A.n)
        lf, lb = C.c_int(), C.c_int()
        check(lib().psp_ssor_info(h, None, C.byref(lf), C.byref(lb)))
        self.levels = (lf.value, lb.value)
        rf, rb, lv, sl = C.c_int(), C.c_int(), C.c_long(), C.c_long()
        check(lib().psp_ssor_run_info(h, C.byref(rf), C.byref(rb), C.byref(lv), C.byref(sl)))
        # (runs forward, runs backward, levels covered, slots covered) of the LDS-exchange runs of narrow levels
        self.lds_runs = (rf.value, rb.value, lv.value, sl.value)
        nbr, edge = C.c_int(), C.c_int()
        check(lib().psp_ssor_brick_info(h, C.byref(nbr), C.byref(edge)))
        self.bricks = nbr.value  # 3-D grid operators with wide levels: bricks of edge^3 points, 0 otherwise

    def precon(self, x, y):
        n = self.shape[0]
        for k, a in ((1, x), (2, y)):
            if (not isinstance(a, np.ndarray) or a.ndim != 1 or a.dtype != np.float64 or a.shape[0] != n
                    or not a.flags.c_contiguous):
                raise ValueError("arg %d must be a contiguous 1-dimensional double array of appropriate size." % k)
        check(lib().psp_ssor_precon(self._h, _ptr(x), _ptr(y)))

    def precon_dev(self, x_ptr, y_ptr):
        check(lib().psp_ssor_precon_dev(self._h, x_ptr, y_ptr))

    def close(self):
        if getattr(self, "_h", None):
            lib().psp_ssor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _solve(fn, A, b, x, tol, maxit, K, hist):
    aop = _Op(A, "matvec")
    kop = _Op(K, "precon") if K is not None else None
    n = int(A.shape[0])
    # ItSolvers_pcg (itsolversmodule.c:70-88): x is updated in place only when it already is
    # a contiguous float64 array; anything else is converted, solved and discarded
    xw = x if (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous
               and x.ndim == 1) else np.ascontiguousarray(x, dtype=np.float64)
    bw = np.ascontiguousarray(b, dtype=np.float64)
    if xw.ndim != 1 or bw.ndim != 1 or xw.shape[0] != bw.shape[0] or xw.shape[0] != n:
        raise ValueError("incompatible operand shapes")
    info, it, rr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    h = np.full(maxit + 2, np.nan) if hist else None
    rc = fn(aop._h, kop._h if kop else None, n, _ptr(xw), _ptr(bw), float(tol), int(maxit),
            C.byref(info), C.byref(it), C.byref(rr), _ptr(h) if hist else None)
    for op in (aop, kop):
        if op is not None and op.exc is not None:
            raise op.exc
    check(rc)
    res = (info.value, it.value, rr.value)
    return res + (h,) if hist else res


def pcg(A, b, x, tol, maxit, K=None, hist=False):
    """info, iter, relres = pcg(A, b, x, tol, maxit[, K]) -- itsolversmodule.c:32-118."""
    return _solve(lib().psp_pcg, A, b, x, tol, maxit, K, hist)


def minres(A, b, x, tol, maxit, K=None, hist=False):
    """info, iter, relres = minres(A, b, x, tol, maxit[, K]) -- itsolversmodule.c:217-305."""
    return _solve(lib().psp_minres, A, b, x, tol, maxit, K, hist)


def _solve_more(name, A, b, x, tol, maxit, K, dim=None):
    aop = _Op(A, "matvec")
    kop = _Op(K, "precon") if K is not None else None
    n = int(A.shape[0])
    xw = x if (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous
               and x.ndim == 1) else np.ascontiguousarray(x, dtype=np.float64)
    bw = np.ascontiguousarray(b, dtype=np.float64)
    if xw.ndim != 1 or bw.ndim != 1 or xw.shape[0] != bw.shape[0] or xw.shape[0] != n:
        raise ValueError("incompatible operand shapes")
    info, it, rr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    fn = getattr(lib(), "psp_" + name)
    args = [aop._h, kop._h if kop else None, n, _ptr(xw), _ptr(bw), float(tol), int(maxit)]
    if dim is not None:
        args.append(int(dim))
    rc = fn(*args, C.byref(info), C.byref(it), C.byref(rr))
    for op in (aop, kop):
        if op is not None and op.exc is not None:
            raise op.exc
    check(rc)
    return info.value, it.value, rr.value


def cgs(A, b, x, tol, maxit, K=None):
    """info, iter, relres = cgs(A, b, x, tol, maxit[, K]) -- itsolversmodule.c:503-586."""
    return _solve_more("cgs", A, b, x, tol, maxit, K)


def bicgstab(A, b, x, tol, maxit, K=None):
    """info, iter, relres = bicgstab(A, b, x, tol, maxit[, K]) -- itsolversmodule.c:125-216."""
    return _solve_more("bicgstab", A, b, x, tol, maxit, K)


def qmrs(A, b, x, tol, maxit, K=None):
    """info, iter, relres = qmrs(A, b, x, tol, maxit[, K]) -- itsolversmodule.c:410-496."""
    return _solve_more("qmrs", A, b, x, tol, maxit, K)


def gmres(A, b, x, tol, maxit, K=None, dim=20):
    """info, iter, relres = gmres(A, b, x, tol, maxit[, K[, dim]]) -- itsolversmodule.c:313-403."""
    return _solve_more("gmres", A, b, x, tol, maxit, K, dim)


def last_solve_info():
    """(loop name, {launches, vec_bytes_per_row, dinv_streamed, single_kernel_fallbacks}) of the calling thread's last
    pcg / minres (psp_last_solve_info): which of the library's loops ran and what it moves per row beside its product."""
    name = C.create_string_buffer(64)
    info = (C.c_int * 4)()
    check(lib().psp_last_solve_info(name, 64, info))
    return name.value.decode(), {"launches": info[0], "vec_bytes_per_row": info[1], "dinv_streamed": bool(info[2]),
                                 "single_kernel_fallbacks": info[3]}


def set_single_kernel_loops(on):
    """psp_set_single_kernel_loops: False keeps every pcg / minres of this process on the launch-per-phase loops"""
    check(lib().psp_set_single_kernel_loops(1 if on else 0))


def device_count():
    return lib().psp_device_count()


def device_info():
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(), C.c_int64()
    check(lib().psp_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return name.value.decode(), cu.value, mem.value
