"""ctypes binding of libpysparse_hip.so (include/pysparse_hip.h).

Process rule: when PyTorch is used in the same process, `import torch` BEFORE the first call
of lib() -- the library then binds to the HIP runtime bundled with torch (same SONAME); the
other order loads two HIP runtimes and the second one to initialise sees no device.

Used by bench.py, the GPU tests and the multi-GPU driver; the drop-in extension modules
(pysparse_amd.sparse.spmatrix, .itsolvers.krylov, .precon.precon) link the same library
directly from C.  There is no CPU fallback: every compute call raises when no GPU is
present or the library is missing.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSP_LIB_OVERRIDE") or os.path.join(HERE, "libpysparse_hip.so")  # override: A/B builds

# every symbol include/pysparse_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = """
psp_last_error psp_version psp_device_count psp_set_device psp_thread_info psp_debug_hold_handles psp_debug_mid_count psp_debug_brick_count psp_debug_shake psp_debug_shake_count psp_debug_spin psp_peer_access psp_set_stream psp_synchronize
psp_set_placement psp_placement_info psp_place_operands psp_device_info psp_mem_info psp_malloc psp_free psp_memcpy_h2d psp_memcpy_d2h psp_memset psp_trim
psp_event_create psp_event_destroy psp_event_record psp_event_elapsed_ms psp_stream_probe psp_build_id psp_last_solve_info psp_set_single_kernel_loops
psp_csr_create psp_csr_poisson psp_csr_poisson_slab psp_csr_poisson_big psp_csr_poisson_big_slab psp_csr_nnz64 psp_csr_create64 psp_csr_random_banded psp_csr_download_rows psp_csr_destroy psp_csr_shape
psp_csr_download psp_csr_diagonal psp_csr_matvec psp_csr_matvec_stride psp_csr_matvec_transp
psp_csr_matvec_transp_stride psp_csr_matvec_dev psp_csr_matvec_transp_dev psp_csr_set_variant
psp_csr_set_schedule psp_csr_kernel_info psp_csr_prepare psp_csr_setup_info psp_csr_release_arrays psp_csr_renumbering psp_csr_device_bytes
psp_csr_poisson_multi psp_csr_create_multi psp_csr_multi_info psp_csr_multi_spmv_time psp_csr_multi_phase_time psp_multi_plan
psp_sss_create psp_sss_poisson psp_sss_destroy psp_sss_shape psp_sss_download psp_sss_getitem
psp_sss_matvec psp_sss_matvec_stride psp_sss_matvec_dev psp_sss_device_bytes
psp_sss_kernel_info psp_sss_set_variant psp_sss_prepare psp_sss_setup_info
psp_jacobi_create_csr psp_jacobi_create_sss psp_jacobi_create_diag psp_jacobi_destroy
psp_jacobi_shape psp_jacobi_precon psp_jacobi_precon_dev
psp_ssor_create psp_ssor_destroy psp_ssor_info psp_ssor_run_info psp_ssor_brick_info psp_ssor_precon psp_ssor_precon_dev
psp_op_from_csr psp_op_from_sss psp_op_from_jacobi psp_op_from_ssor psp_op_from_callback psp_op_destroy
psp_pcg psp_pcg_dev psp_minres psp_minres_dev psp_cgs psp_bicgstab psp_qmrs psp_gmres
psp_k_dot psp_k_residual psp_k_pupdate psp_k_csr_matvec_dot psp_k_xr_update psp_k_gather
psp_k_csr_matvec_overlap psp_k_hint_constant psp_k_unhint psp_k_px_update psp_k_r_update psp_k_x_update
psp_k_jacobi psp_pcgstate_create psp_pcgstate_destroy psp_pcgstate_init psp_pcgstate_fetch psp_pcgstate_hist
psp_kd_px_update psp_kd_csr_matvec_overlap psp_kd_pcg_scalar_xpq psp_kd_r_update psp_kd_pcg_scalar_r
psp_minresstate_create psp_minresstate_destroy psp_minresstate_init psp_minresstate_fetch psp_minresstate_hist
psp_kd_minres_scale psp_kd_minres_matvec psp_kd_minres_lanczos psp_kd_minres_scalar psp_kd_minres_wx
""".split()


class PcgStatus(C.Structure):
    """psp_pcg_status_t"""
    _fields_ = [("status", C.c_int), ("info", C.c_int), ("iter", C.c_int), ("it", C.c_int), ("xpend", C.c_int),
                ("stag0", C.c_int), ("pend_maxit", C.c_int), ("relres", C.c_double), ("normr", C.c_double),
                ("n2b", C.c_double), ("alpha_x", C.c_double)]


class MinresStatus(C.Structure):
    """psp_minres_status_t"""
    _fields_ = [("status", C.c_int), ("stop", C.c_int), ("info", C.c_int), ("iter", C.c_int),
                ("relres", C.c_double), ("norm_rmr", C.c_double)]

WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
HOST_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))


class PspError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libpysparse_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the library (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise PspError(rc, lib().psp_last_error().decode())
    return rc


def _declare(L):
    vp, i, d, i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
    pvp, pi, pd = C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_double)
    sz, pt = C.c_size_t, C.c_ssize_t
    L.psp_last_error.restype = C.c_char_p
    L.psp_last_error.argtypes = []
    L.psp_version.restype = C.c_char_p
    L.psp_version.argtypes = []
    L.psp_device_count.restype = i
    L.psp_device_count.argtypes = []
    sig = {
        "psp_set_device": [i], "psp_thread_info": [pi, pi, pvp], "psp_debug_hold_handles": [vp, vp, i], "psp_debug_shake": [C.c_longlong, i, i, C.c_uint, C.c_uint, i], "psp_debug_shake_count": [C.POINTER(C.c_longlong)], "psp_debug_spin": [i], "psp_debug_mid_count": [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], "psp_debug_brick_count": [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], "psp_peer_access": [i, i, pi], "psp_set_stream": [vp], "psp_synchronize": [], "psp_trim": [],
        "psp_set_placement": [i], "psp_placement_info": [pi, C.POINTER(C.c_longlong), pd], "psp_place_operands": [vp, pvp, pvp, pd],
        "psp_device_info": [C.c_char_p, i, pi, C.POINTER(i64)],
        "psp_mem_info": [C.POINTER(i64), C.POINTER(i64)],
        "psp_malloc": [pvp, sz], "psp_free": [vp], "psp_memcpy_h2d": [vp, vp, sz],
        "psp_memcpy_d2h": [vp, vp, sz], "psp_memset": [vp, i, sz],
        "psp_event_create": [pvp], "psp_event_destroy": [vp], "psp_event_record": [vp],
        "psp_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
        "psp_stream_probe": [i, i, sz, i, C.POINTER(C.c_float), C.POINTER(C.c_float)],
        "psp_csr_create": [i, i, i, vp, vp, vp, pvp],
        "psp_csr_poisson": [i, i, i, pvp], "psp_csr_poisson_big": [i, i, i, pvp],
        "psp_csr_poisson_slab": [i, i, i, i64, i64, i64, i, pvp],
        "psp_csr_poisson_big_slab": [i, i, i, i64, i64, i64, i, pvp],
        "psp_csr_create64": [i, i, i64, vp, vp, vp, pvp],
        "psp_csr_poisson_multi": [i, i, i, vp, i, pvp], "psp_csr_create_multi": [i, i, i, vp, vp, vp, vp, i, pvp],
        "psp_csr_multi_info": [vp, pi, pi, pi], "psp_csr_multi_spmv_time": [vp, i, i, pd], "psp_csr_multi_phase_time": [vp, i, i, i, pd],
        "psp_multi_plan": [i, i, vp, vp, i, i, C.POINTER(i64), vp, vp, i, vp, i, vp],
        "psp_csr_random_banded": [i, i, i, i, C.c_uint64, pvp],
        "psp_csr_download_rows": [vp, i, i, vp, vp, vp],
        "psp_csr_destroy": [vp], "psp_csr_shape": [vp, pi, pi, pi],
        "psp_csr_download": [vp, vp, vp, vp], "psp_csr_diagonal": [vp, vp],
        "psp_csr_matvec": [vp, vp, vp], "psp_csr_matvec_stride": [vp, vp, pt, vp, pt],
        "psp_csr_matvec_transp": [vp, vp, vp], "psp_csr_matvec_transp_stride": [vp, vp, pt, vp, pt],
        "psp_csr_matvec_dev": [vp, vp, vp], "psp_csr_matvec_transp_dev": [vp, vp, vp],
        "psp_csr_set_variant": [vp, i], "psp_csr_set_schedule": [vp, i],
        "psp_csr_kernel_info": [vp, C.c_char_p, i, pi],
        "psp_csr_renumbering": [vp, vp, pi],
        "psp_sss_create": [i, i, vp, vp, vp, vp, pvp], "psp_sss_poisson": [i, i, i, pvp],
        "psp_sss_destroy": [vp], "psp_sss_shape": [vp, pi, pi],
        "psp_sss_download": [vp, vp, vp, vp, vp], "psp_sss_getitem": [vp, i, i, pd],
        "psp_sss_matvec": [vp, vp, vp], "psp_sss_matvec_stride": [vp, vp, pt, vp, pt],
        "psp_sss_matvec_dev": [vp, vp, vp],
        "psp_sss_kernel_info": [vp, C.c_char_p, i, pi], "psp_sss_set_variant": [vp, i],
        "psp_jacobi_create_csr": [vp, d, i, pvp], "psp_jacobi_create_sss": [vp, d, i, pvp],
        "psp_jacobi_create_diag": [i, vp, d, i, vp, pvp], "psp_jacobi_destroy": [vp],
        "psp_jacobi_shape": [vp, pi], "psp_jacobi_precon": [vp, vp, vp],
        "psp_jacobi_precon_dev": [vp, vp, vp],
        "psp_ssor_create": [vp, d, i, pvp], "psp_ssor_destroy": [vp], "psp_ssor_info": [vp, pi, pi, pi], "psp_ssor_run_info": [vp, pi, pi, vp, vp], "psp_ssor_brick_info": [vp, pi, pi],
        "psp_ssor_precon": [vp, vp, vp], "psp_ssor_precon_dev": [vp, vp, vp], "psp_op_from_ssor": [vp, pvp],
        "psp_op_from_csr": [vp, pvp], "psp_op_from_sss": [vp, pvp], "psp_op_from_jacobi": [vp, pvp],
        "psp_op_from_callback": [i, HOST_APPLY_FN, vp, pvp], "psp_op_destroy": [vp],
        "psp_pcg": [vp, vp, i, vp, vp, d, i, pi, pi, pd, vp],
        "psp_pcg_dev": [vp, vp, i, vp, vp, d, i, pi, pi, pd, vp],
        "psp_minres": [vp, vp, i, vp, vp, d, i, pi, pi, pd, vp],
        "psp_minres_dev": [vp, vp, i, vp, vp, d, i, pi, pi, pd, vp],
        "psp_cgs": [vp, vp, i, vp, vp, d, i, pi, pi, pd], "psp_bicgstab": [vp, vp, i, vp, vp, d, i, pi, pi, pd],
        "psp_qmrs": [vp, vp, i, vp, vp, d, i, pi, pi, pd], "psp_gmres": [vp, vp, i, vp, vp, d, i, i, pi, pi, pd],
        "psp_k_dot": [i, vp, vp, vp], "psp_k_residual": [i, vp, vp, vp, vp],
        "psp_k_pupdate": [i, vp, vp, d, i, vp], "psp_k_csr_matvec_dot": [vp, vp, i, vp, vp],
        "psp_k_xr_update": [i, d, vp, vp, vp, vp, vp, vp], "psp_k_gather": [i, vp, vp, vp],
        "psp_k_csr_matvec_overlap": [vp, vp, i, vp, i, i, WAIT_FN, vp, vp],
        "psp_k_hint_constant": [vp, i], "psp_k_unhint": [vp],
        "psp_k_px_update": [i, vp, vp, d, i, d, i, vp, vp, vp], "psp_k_r_update": [i, d, vp, vp, vp, vp],
        "psp_k_x_update": [i, d, vp, vp, vp],
        "psp_k_jacobi": [i, vp, vp, vp],
        "psp_pcgstate_create": [pvp], "psp_pcgstate_destroy": [vp], "psp_pcgstate_init": [vp, d, d, d, d, i, i],
        "psp_pcgstate_fetch": [vp, C.POINTER(PcgStatus)], "psp_pcgstate_hist": [vp, i, i, vp],
        "psp_kd_px_update": [vp, i, vp, vp, vp, vp, vp],
        "psp_kd_csr_matvec_overlap": [vp, vp, vp, i, vp, i, i, WAIT_FN, vp, vp],
        "psp_kd_pcg_scalar_xpq": [vp, vp], "psp_kd_r_update": [vp, i, vp, vp, vp, vp],
        "psp_kd_pcg_scalar_r": [vp, vp],
        "psp_minresstate_create": [pvp], "psp_minresstate_destroy": [vp],
        "psp_minresstate_init": [vp, d, d, d, i, i], "psp_minresstate_fetch": [vp, C.POINTER(MinresStatus)],
        "psp_minresstate_hist": [vp, i, i, vp],
        "psp_kd_minres_scale": [vp, i, vp, vp],
        "psp_kd_minres_matvec": [vp, vp, vp, i, vp, i, i, WAIT_FN, vp, vp],
        "psp_kd_minres_lanczos": [vp, i, vp, vp, vp, vp, vp, vp], "psp_kd_minres_scalar": [vp, i, vp],
        "psp_kd_minres_wx": [vp, i, vp, vp, vp, vp],
        "psp_last_solve_info": [C.c_char_p, i, C.POINTER(i)], "psp_set_single_kernel_loops": [i],
        "psp_csr_release_arrays": [vp], "psp_csr_prepare": [vp, C.c_longlong], "psp_sss_prepare": [vp, C.c_longlong],
        "psp_csr_setup_info": [vp, C.POINTER(d)], "psp_sss_setup_info": [vp, C.POINTER(d)],
    }
    for name, argtypes in sig.items():
        f = getattr(L, name)
        f.restype = i
        f.argtypes = argtypes
    L.psp_build_id.restype = C.c_char_p
    L.psp_build_id.argtypes = []
    L.psp_csr_device_bytes.restype = i64
    L.psp_csr_device_bytes.argtypes = [vp]
    L.psp_csr_nnz64.restype = i64
    L.psp_csr_nnz64.argtypes = [vp]
    L.psp_sss_device_bytes.restype = i64
    L.psp_sss_device_bytes.argtypes = [vp]
