"""Row-range partitioned SpMV + Jacobi-PCG over the GPUs of one node.

The reference is single-process; this is the MI355X-native extension of its PCG
(pysparse/itsolvers/src/pcg.c:22-171) to N ranks, one process per GPU:

  * rows of A and the matching slices of x, b, r, q, dinv are split into contiguous row
    ranges; each rank stores its row block with column indices renumbered into a local
    "extended" vector  [ghost_lo | owned | ghost_hi]  (ghosts sorted by global index, so a
    banded operator keeps its global ordering and the SpMV kernel needs no change);
  * before every q = A p the ghost entries of p are exchanged with the neighbouring ranks
    (RCCL send/recv through torch.distributed, point-to-point over xGMI);
  * the three sum reductions and the stagnation test of one PCG iteration are packed into
    TWO all-reduces:  #1 {p.q, number of non-stagnated workgroups}  and  #2 {r.r, r.z};
  * alpha / beta / the exit tests live in a small DEVICE-side state (psp_pcgstate_*, psp_kd_*):
    the all-reduces are issued in stream order between "finish the local partial sums" and "take
    the reference's branches on the reduced values", the host enqueues 16 iterations at a time and
    reads the state once per batch (dist_pcg, dist_minres) -- no host round trip per reduction.

All arithmetic on vectors is done by a *backend*: HipBackend calls the HIP kernels of
libpysparse_hip.so through the C ABI on torch CUDA tensors (the product path); the CPU
gloo tests inject an oracle-backed stand-in (tests/test_distributed_cpu.py) so that the
partition, halo and reduction logic of this file is exercised at world_size 2 without a GPU.
"""
import ctypes as C

import os

import numpy as np

try:  # torch is plumbing here: device memory, streams and torch.distributed
    import torch
    import torch.distributed as dist
except ImportError:  # pragma: no cover
    torch = None
    dist = None


def row_range(n_global, world, rank):
    """Contiguous, balanced row ranges: rank r owns [lo, hi)."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slab_range(n_planes, plane_rows, world, rank):
    """Row range made of whole grid planes (z-slabs for 3-D, y-lines for 2-D)."""
    lo, hi = row_range(n_planes, world, rank)
    return lo * plane_rows, hi * plane_rows


class HaloPlan:
    """Who sends which entries to whom.  `recv[q]` = slice of the extended vector filled
    by rank q; `send[q]` = owned-local indices (or a slice) this rank sends to q."""

    def __init__(self, n_owned, ghost_lo, ghost_hi):
        self.n_owned = n_owned
        self.ghost_lo = ghost_lo  # number of ghost entries below the owned block
        self.ghost_hi = ghost_hi
        self.p_offset = ghost_lo
        self.n_ext = ghost_lo + n_owned + ghost_hi
        self.recv = {}  # rank -> (start, stop) in the extended vector
        self.send = {}  # rank -> (start, stop) in OWNED coordinates, or an index tensor
        # owned rows [interior[0], interior[1]) reference no ghost entry: they can be
        # multiplied while the halo exchange is still in flight
        self.interior = (0, n_owned)


def poisson_halo_plan(nx, ny, nz, world, rank):
    """Slab partition of the 5-/7-point operator: ghosts are the `reach` rows just below
    and above the owned block (reach = one grid plane / line), clipped at the boundary."""
    three_d = nz > 0
    planes = nz if three_d else ny
    plane_rows = nx * ny if three_d else nx
    if planes < world:
        raise ValueError("fewer grid planes than ranks")
    lo, hi = slab_range(planes, plane_rows, world, rank)
    n = plane_rows * planes
    reach = plane_rows
    g_lo = min(reach, lo)
    g_hi = min(reach, n - hi)
    plan = HaloPlan(hi - lo, g_lo, g_hi)
    plan.row_lo, plan.row_hi, plan.n_global = lo, hi, n
    if g_lo:  # lower neighbour owns [lo - reach, lo)
        plan.recv[rank - 1] = (0, g_lo)
        plan.send[rank - 1] = (0, reach)  # my first plane is its ghost_hi
    if g_hi:
        plan.recv[rank + 1] = (g_lo + plan.n_owned, plan.n_ext)
        plan.send[rank + 1] = (plan.n_owned - reach, plan.n_owned)
    plan.interior = (reach if g_lo else 0, plan.n_owned - reach if g_hi else plan.n_owned)
    return plan


def general_halo_plan(n_global, row_lo, row_hi, col_global, world, rank, all_gather_object, ind=None):
    """Partition of an arbitrary CSR row block (host arrays, small/medium problems): the
    ghost set is the sorted set of referenced off-rank columns; send lists are exchanged
    once at setup.  `ind` (local row pointers) lets the plan find the ghost-free row range
    used to overlap the exchange with the SpMV.  Returns (plan, col_local)."""
    col_global = np.asarray(col_global, dtype=np.int64)
    needed = np.unique(col_global)
    g_lo_ids = needed[needed < row_lo]
    g_hi_ids = needed[needed >= row_hi]
    plan = HaloPlan(row_hi - row_lo, len(g_lo_ids), len(g_hi_ids))
    plan.row_lo, plan.row_hi, plan.n_global = row_lo, row_hi, n_global
    ext_ids = np.concatenate([g_lo_ids, np.arange(row_lo, row_hi, dtype=np.int64), g_hi_ids])
    col_local = np.searchsorted(ext_ids, col_global).astype(np.int32)
    ranges = all_gather_object((row_lo, row_hi))
    ghosts = np.concatenate([g_lo_ids, g_hi_ids])
    wanted = {}  # owner rank -> ids I need from it
    for q, (qlo, qhi) in enumerate(ranges):
        if q == rank:
            continue
        ids = ghosts[(ghosts >= qlo) & (ghosts < qhi)]
        if len(ids):
            wanted[q] = ids
            start = int(np.searchsorted(ext_ids, ids[0]))
            plan.recv[q] = (start, start + len(ids))
    # widest run of rows whose columns are all owned (overlaps the exchange with the SpMV)
    if ind is not None and len(col_global):
        ind_local = np.asarray(ind, dtype=np.int64)
        is_ghost = ((col_global < row_lo) | (col_global >= row_hi)).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(is_ghost)])
        touches = csum[ind_local[1:]] - csum[ind_local[:-1]]
        bad = np.flatnonzero(touches > 0)
        if len(bad):
            edges = np.concatenate([[-1], bad, [row_hi - row_lo]])
            k = int(np.argmax(np.diff(edges)))
            plan.interior = (int(edges[k] + 1), int(edges[k + 1]))
    everyone = all_gather_object(wanted)
    for q, w in enumerate(everyone):
        if q != rank and rank in w:
            plan.send[q] = (w[rank] - row_lo).astype(np.int32)  # owned-local indices
    return plan, col_local


def sss_rows_expanded(n, ind, col, val, diag):
    """Full CSR of an SSS matrix with every row in sss_matvec's summation order (sss_mat.c:45-55): stored lower
    entries (ascending column), the diagonal, mirrored entries (j, i), j > i, by ascending j.  Host arrays."""
    ind = np.asarray(ind, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float64)
    nl = np.diff(ind)
    rows = np.repeat(np.arange(n, dtype=np.int64), nl)
    up_cnt = np.bincount(col, minlength=n)                  # mirrored entries land in row col[k]
    f_ind = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nl + 1 + up_cnt, out=f_ind[1:])
    f_col = np.empty(f_ind[-1], dtype=np.int64)
    f_val = np.empty(f_ind[-1], dtype=np.float64)
    within = np.arange(len(col), dtype=np.int64) - ind[rows]
    f_col[f_ind[rows] + within] = col                       # lower part, stored order
    f_val[f_ind[rows] + within] = val
    dpos = f_ind[:-1] + nl
    f_col[dpos] = np.arange(n)
    f_val[dpos] = diag
    order = np.argsort(col, kind="stable")                  # by target row, then by source row (ascending)
    tgt = col[order]
    start = np.concatenate([[0], np.cumsum(up_cnt)])[tgt]
    k = np.arange(len(col), dtype=np.int64) - start
    f_col[dpos[tgt] + 1 + k] = rows[order]
    f_val[dpos[tgt] + 1 + k] = val[order]
    if f_ind[-1] > 2 ** 31 - 1:
        raise ValueError("expanded matrix exceeds 32-bit offsets: partition before expanding")
    return f_ind.astype(np.int32), f_col.astype(np.int32), f_val


# ----------------------------------------------------------------------------- backends

class HipBackend:
    """Vectors = torch CUDA float64 tensors; kernels = libpysparse_hip.so via the C ABI."""

    def __init__(self, device_index):
        from . import _capi
        self._capi = _capi
        self.L = _capi.lib()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        _capi.check(self.L.psp_set_device(device_index))
        # enqueue our kernels on torch's current stream so that collectives order with them
        self.bind_current_stream()
        self._scal = torch.zeros(8, dtype=torch.float64, device=self.device)

    def bind_current_stream(self):
        """(Re)pin the library's stream to the torch stream that is current NOW.  The device-scalar drivers rely on
        stream order alone between the psp_kd_* kernels and the NCCL collectives / P2P batches, which synchronise
        with torch.cuda.current_stream() at call time: dist_pcg / dist_minres call this on entry so that a solve
        started under another `torch.cuda.stream(...)` context does not race."""
        self._capi.check(self.L.psp_set_stream(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def index_tensor(self, idx):
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(self.device)

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def dot(self, x, y):
        out = self._scal[:1]
        self._capi.check(self.L.psp_k_dot(x.numel(), self._p(x), self._p(y), self._p(out)))
        return out

    def residual(self, b, r, dinv):
        out = self._scal[:2]
        self._capi.check(self.L.psp_k_residual(r.numel(), self._p(b), self._p(r),
                                               self._p(dinv) if dinv is not None else None, self._p(out)))
        return out

    def pupdate(self, r, dinv, beta, first, p_owned):
        self._capi.check(self.L.psp_k_pupdate(r.numel(), self._p(r), self._p(dinv) if dinv is not None else None,
                                              float(beta), int(first), self._p(p_owned)))

    def matvec_dot(self, A, p_ext, p_offset, q):
        out = self._scal[:1]
        self._capi.check(self.L.psp_k_csr_matvec_dot(A._h, self._p(p_ext), int(p_offset), self._p(q), self._p(out)))
        return out

    def matvec(self, A, p_ext, q):
        self._capi.check(self.L.psp_csr_matvec_dev(A._h, self._p(p_ext), self._p(q)))

    def matvec_overlap(self, A, p_ext, p_offset, q, interior, wait, want_dot):
        """q = A p with the rows of `interior` launched before wait() (the halo exchange)
        returns; optionally the fused p.q."""
        out = self._scal[:1] if want_dot else None
        err = []

        def _wait(ctx):
            try:
                wait()
                return 0
            except BaseException as e:  # noqa: BLE001 - re-raised below
                err.append(e)
                return 1

        cb = self._capi.WAIT_FN(_wait)
        rc = self.L.psp_k_csr_matvec_overlap(A._h, self._p(p_ext), int(p_offset), self._p(q), int(interior[0]),
                                             int(interior[1]), cb, None, self._p(out) if want_dot else None)
        if err:
            raise err[0]
        self._capi.check(rc)
        return out

    def xr_update(self, alpha, p_owned, q, dinv, x, r):
        out = self._scal[:3]
        self._capi.check(self.L.psp_k_xr_update(x.numel(), float(alpha), self._p(p_owned), self._p(q),
                                                self._p(dinv) if dinv is not None else None, self._p(x),
                                                self._p(r), self._p(out)))
        return out

    def gather(self, idx, v, out):
        self._capi.check(self.L.psp_k_gather(idx.numel(), self._p(idx), self._p(v), self._p(out)))

    # ---- lazy-x arrangement: the matvec's dot lands in _scal[0], the scan of px / x in _scal[1], so
    # that {p.q, nonstag} travel in ONE all-reduce
    def px_update(self, r, dinv, beta, first, alpha_x, xpend, p_owned, x):
        out = self._scal[1:2]
        self._capi.check(self.L.psp_k_px_update(r.numel(), self._p(r), self._p(dinv) if dinv is not None else None,
                                                float(beta), int(first), float(alpha_x), int(xpend),
                                                self._p(p_owned), self._p(x), self._p(out)))
        return out

    def r_update(self, alpha, q, dinv, r):
        out = self._scal[2:4]
        self._capi.check(self.L.psp_k_r_update(r.numel(), float(alpha), self._p(q),
                                               self._p(dinv) if dinv is not None else None, self._p(r), self._p(out)))
        return out

    def x_update(self, alpha, p_owned, x):
        out = self._scal[1:2]
        self._capi.check(self.L.psp_k_x_update(x.numel(), float(alpha), self._p(p_owned), self._p(x), self._p(out)))
        return out

    # ---- device-resident scalars (psp_pcgstate_* / psp_minresstate_*, include/pysparse_hip.h): results of
    # the local reductions land in self.scal, the caller all-reduces slices of it in stream order
    @property
    def scal(self):
        return self._scal

    def pcg_state(self, n2b, tolb, normr0, rho0, maxit, want_hist):
        return _HipPcgState(self, n2b, tolb, normr0, rho0, maxit, want_hist)

    def kd_px_update(self, st, r, dinv, p_owned, x):
        self._capi.check(self.L.psp_kd_px_update(st._h, r.numel(), self._p(r), self._p(dinv) if dinv is not None
                                                 else None, self._p(p_owned), self._p(x), self._p(self._scal[1:2])))

    def _wait_cb(self, wait, err):
        def _wait(ctx):
            try:
                if wait is not None:
                    wait()
                return 0
            except BaseException as e:  # noqa: BLE001 - re-raised by the caller
                err.append(e)
                return 1
        return self._capi.WAIT_FN(_wait)

    def kd_matvec_overlap(self, st, A, p_ext, p_offset, q, interior, wait):
        err = []
        cb = self._wait_cb(wait, err)
        rc = self.L.psp_kd_csr_matvec_overlap(st._h, A._h, self._p(p_ext), int(p_offset), self._p(q),
                                              int(interior[0]), int(interior[1]), cb, None, self._p(self._scal[0:1]))
        if err:
            raise err[0]
        self._capi.check(rc)

    def kd_pcg_scalar_xpq(self, st):
        self._capi.check(self.L.psp_kd_pcg_scalar_xpq(st._h, self._p(self._scal[0:2])))

    def kd_r_update(self, st, q, dinv, r):
        self._capi.check(self.L.psp_kd_r_update(st._h, r.numel(), self._p(q), self._p(dinv) if dinv is not None
                                                else None, self._p(r), self._p(self._scal[2:4])))

    def kd_pcg_scalar_r(self, st):
        self._capi.check(self.L.psp_kd_pcg_scalar_r(st._h, self._p(self._scal[2:4])))

    def minres_state(self, norm_r0, beta0, tol, maxit, want_hist):
        return _HipMinresState(self, norm_r0, beta0, tol, maxit, want_hist)

    def jacobi(self, x, dinv, y):
        """y = x .* dinv (preconmodule.c:41-42)"""
        self._capi.check(self.L.psp_k_jacobi(x.numel(), self._p(x), self._p(dinv), self._p(y)))

    def kd_minres_scale(self, st, y, v_owned):
        self._capi.check(self.L.psp_kd_minres_scale(st._h, y.numel(), self._p(y), self._p(v_owned)))

    def kd_minres_matvec(self, st, A, v_ext, v_offset, av, interior, wait):
        err = []
        cb = self._wait_cb(wait, err)
        rc = self.L.psp_kd_minres_matvec(st._h, A._h, self._p(v_ext), int(v_offset), self._p(av), int(interior[0]),
                                         int(interior[1]), cb, None, self._p(self._scal[0:1]))
        if err:
            raise err[0]
        self._capi.check(rc)

    def kd_minres_lanczos(self, st, av, v_hat, v_hat_old, dinv, y):
        self._capi.check(self.L.psp_kd_minres_lanczos(st._h, av.numel(), self._p(av), self._p(v_hat),
                                                      self._p(v_hat_old), self._p(dinv) if dinv is not None else None,
                                                      self._p(y) if y is not None else None, self._p(self._scal[4:5])))

    def kd_minres_scalar(self, st, which):
        off = 0 if which == 0 else 4
        self._capi.check(self.L.psp_kd_minres_scalar(st._h, int(which), self._p(self._scal[off:off + 1])))

    def kd_minres_wx(self, st, v_owned, w, w_old, x):
        self._capi.check(self.L.psp_kd_minres_wx(st._h, x.numel(), self._p(v_owned), self._p(w), self._p(w_old),
                                                 self._p(x)))

    def hint_constant(self, v):
        self._capi.check(self.L.psp_k_hint_constant(self._p(v), v.numel()))

    def unhint(self, v):
        self._capi.check(self.L.psp_k_unhint(self._p(v)))

    def synchronize(self):
        torch.cuda.synchronize(self.device)



class _HipPcgState:
    """psp_pcgstate_t handle"""

    def __init__(self, be, n2b, tolb, normr0, rho0, maxit, want_hist):
        self.be = be
        self._h = C.c_void_p()
        be._capi.check(be.L.psp_pcgstate_create(C.byref(self._h)))
        be._capi.check(be.L.psp_pcgstate_init(self._h, float(n2b), float(tolb), float(normr0), float(rho0),
                                              int(maxit), 1 if want_hist else 0))

    def fetch(self):
        s = self.be._capi.PcgStatus()
        self.be._capi.check(self.be.L.psp_pcgstate_fetch(self._h, C.byref(s)))
        return s

    def hist(self, first, count):
        out = np.empty(count)
        self.be._capi.check(self.be.L.psp_pcgstate_hist(self._h, int(first), int(count),
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.be.L.psp_pcgstate_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _HipMinresState:
    """psp_minresstate_t handle"""

    def __init__(self, be, norm_r0, beta0, tol, maxit, want_hist):
        self.be = be
        self._h = C.c_void_p()
        be._capi.check(be.L.psp_minresstate_create(C.byref(self._h)))
        be._capi.check(be.L.psp_minresstate_init(self._h, float(norm_r0), float(beta0), float(tol), int(maxit),
                                                 1 if want_hist else 0))

    def fetch(self):
        s = self.be._capi.MinresStatus()
        self.be._capi.check(self.be.L.psp_minresstate_fetch(self._h, C.byref(s)))
        return s

    def hist(self, first, count):
        out = np.empty(count)
        self.be._capi.check(self.be.L.psp_minresstate_hist(self._h, int(first), int(count),
                                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.be.L.psp_minresstate_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostPcgState:
    """Python mirror of the device state machine (psp_solvers.hip: pcg_lazy_scalar_x / _pq / _r) for
    backends that keep scalars on the host (the CPU test backends); same fields, same branch order."""

    def __init__(self, n2b, tolb, normr0, rho0, maxit, want_hist):
        self.rho, self.rho1, self.alpha, self.beta = rho0, 1.0, 0.0, 0.0
        self.normr, self.tolb, self.n2b, self.relres = normr0, tolb, n2b, 0.0
        self.status = self.info = self.iter = self.stag = 0
        self.it, self.maxit = 1, maxit
        self.xpend = self.stag0 = self.head_rho0 = self.head_beta0 = self.pend_maxit = 0
        self.alpha_x = 0.0
        self._hist = {} if want_hist else None

    def _finish(self, code, it):
        self.status, self.info, self.iter = 1, code, it
        self.relres = self.normr / self.n2b  # pcg.c:166

    def scalar_x(self, nonstag):
        if self.status:
            return
        if self.xpend:  # iteration it-1: pcg.c:159-162
            self.xpend = 0
            if self.stag0 or nonstag == 0.0:
                self.stag = 1
                self._finish(-5, self.it - 1)
                return
        if self.head_rho0:  # pcg.c:101-104 of iteration it
            self._finish(-2, self.it)
        elif self.head_beta0:  # pcg.c:109-112
            self._finish(-6, self.it)

    def scalar_pq(self, pq):
        if self.status:
            return
        if pq == 0.0:  # pcg.c:118-120
            self._finish(-6, self.it)
            return
        self.alpha = self.alpha_x = self.rho / pq
        self.stag0 = 1 if self.alpha == 0.0 else 0
        self.xpend = 1

    def scalar_r(self, rr, rz):
        if self.status:
            return
        it = self.it
        self.normr = float(np.sqrt(rr))
        if self._hist is not None:
            self._hist[it] = self.normr
        if self.normr <= self.tolb:
            self._finish(0, it)  # x update of iteration it still pending: final pass
        elif it == self.maxit:
            self.pend_maxit = 1  # -5 or -1: decided by the scan of the final pass
            self.status = 1
        else:
            self.rho1, self.rho = self.rho, rz
            self.it = it + 1
            self.head_rho0 = 1 if rz == 0.0 else 0
            self.head_beta0 = 0
            if rz != 0.0:
                self.beta = self.rho / self.rho1
                self.head_beta0 = 1 if self.beta == 0.0 else 0

    def fetch(self):
        return self

    def hist(self, first, count):
        return np.array([self._hist.get(i, np.nan) for i in range(first, first + count)])

    def close(self):
        pass


class HostMinresState:
    """Python mirror of psp_solvers.hip's minres_scalar_alpha / minres_scalar_beta (minres.c:129-192)."""

    def __init__(self, norm_r0, beta0, tol, maxit, want_hist):
        self.beta, self.beta_old, self.alpha = beta0, 1.0, 0.0
        self.c = self.c_old = 1.0
        self.s = self.s_old = 0.0
        self.eta, self.norm_rmr, self.norm_r0, self.errtol, self.relres = beta0, norm_r0, norm_r0, tol, 0.0
        self.c1 = self.c2 = self.r1 = self.r2 = self.r3 = self.c_eta = 0.0
        self.status = self.stop = self.skip = 0
        self.info, self.iter, self.it_max = -1, 1, maxit
        self._hist = {} if want_hist else None

    def scalar_alpha(self, alpha):
        if self.status:
            return
        if self.stop:
            self.status = 1
            return
        self.alpha = alpha
        self.c1 = alpha / self.beta
        self.c2 = self.beta / self.beta_old

    def scalar_beta(self, b2):
        if self.status:
            return
        alpha, beta_old = self.alpha, self.beta
        self.beta_old = beta_old
        if b2 < 0.0:  # minres.c:144-146
            self.status = self.skip = 1
            self.info = -3
            return
        beta = float(np.sqrt(b2))
        self.beta = beta
        c_oold, c_old, s_oold, s_old = self.c_old, self.c, self.s_old, self.s
        self.c_old, self.s_old = c_old, s_old
        r1_hat = c_old * alpha - c_oold * s_old * beta_old
        r1 = float(np.sqrt(r1_hat * r1_hat + beta * beta))
        r2 = s_old * alpha + c_oold * c_old * beta_old
        r3 = s_oold * beta_old
        if r1 == 0.0:  # minres.c:160-162
            self.status = self.skip = 1
            self.info = -6
            return
        self.c, self.s = r1_hat / r1, beta / r1
        self.r1, self.r2, self.r3 = r1, r2, r3
        self.c_eta = self.c * self.eta
        self.eta = -self.s * self.eta
        self.norm_rmr = self.norm_rmr * abs(self.s)
        if self._hist is not None:
            self._hist[self.iter] = self.norm_rmr
        conv = self.norm_rmr < self.errtol * self.norm_r0
        if self.iter >= self.it_max or conv:
            self.stop = self.skip = 1
            self.relres = self.norm_rmr / self.norm_r0
            self.info = 0 if conv else -1
        else:
            self.iter += 1

    def fetch(self):
        return self

    def hist(self, first, count):
        return np.array([self._hist.get(i, np.nan) for i in range(first, first + count)])

    def close(self):
        pass


class HostStateOps:
    """The state-driven phase ops (kd_*) expressed with a backend's host-scalar ops + the Python state
    mirrors above: what a CPU test backend inherits so that dist_pcg / dist_minres run the same driver
    code as HipBackend.  `scal` is an 8-double tensor like HipBackend's."""

    @property
    def scal(self):
        if getattr(self, "_scal8", None) is None:
            self._scal8 = torch.zeros(8, dtype=torch.float64)
        return self._scal8

    def pcg_state(self, n2b, tolb, normr0, rho0, maxit, want_hist):
        return HostPcgState(n2b, tolb, normr0, rho0, maxit, want_hist)

    def kd_px_update(self, st, r, dinv, p_owned, x):
        if not st.status:
            self.scal[1] = self.px_update(r, dinv, st.beta, st.it == 1, st.alpha_x, bool(st.xpend), p_owned, x)[0]

    def kd_matvec_overlap(self, st, A, p_ext, p_offset, q, interior, wait):
        if st.status:
            if wait is not None:
                wait()
            return
        if wait is None:
            self.scal[0] = self.matvec_dot(A, p_ext, p_offset, q)[0]
        else:
            self.scal[0] = self.matvec_overlap(A, p_ext, p_offset, q, interior, wait, True)[0]

    def kd_pcg_scalar_xpq(self, st):
        st.scalar_x(float(self.scal[1]))
        st.scalar_pq(float(self.scal[0]))

    def kd_r_update(self, st, q, dinv, r):
        if not st.status:
            self.scal[2:4] = self.r_update(st.alpha, q, dinv, r)

    def kd_pcg_scalar_r(self, st):
        st.scalar_r(float(self.scal[2]), float(self.scal[3]))

    def minres_state(self, norm_r0, beta0, tol, maxit, want_hist):
        return HostMinresState(norm_r0, beta0, tol, maxit, want_hist)

    def kd_minres_scale(self, st, y, v_owned):
        if not st.skip:
            self.scale_div(y, st.beta, v_owned)

    def kd_minres_matvec(self, st, A, v_ext, v_offset, av, interior, wait):
        if st.skip:
            if wait is not None:
                wait()
            return
        if wait is None:
            self.scal[0] = self.matvec_dot(A, v_ext, v_offset, av)[0]
        else:
            self.scal[0] = self.matvec_overlap(A, v_ext, v_offset, av, interior, wait, True)[0]

    def kd_minres_lanczos(self, st, av, v_hat, v_hat_old, dinv, y):
        if not st.skip:
            self.scal[4] = self.lanczos(av, st.c1, st.c2, v_hat, v_hat_old, dinv, y)[0]

    def kd_minres_scalar(self, st, which):
        if which == 0:
            st.scalar_alpha(float(self.scal[0]))
        else:
            st.scalar_beta(float(self.scal[4]))

    def kd_minres_wx(self, st, v_owned, w, w_old, x):
        if not st.status:
            self.minres_wx(v_owned, st.r1, st.r2, st.r3, st.c_eta, w, w_old, x)


class Comm:
    """torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).

    RCCL work is stream-ordered: a collective / P2P batch waits for the kernels enqueued before it on the current
    stream and `wait()` makes the current stream wait for it -- nothing blocks the host.  gloo moves device tensors
    too on this image (the one-GPU rehearsals, tests/test_gpu_distributed.py), but through host staging on its own
    streams: there the device is synchronised around every transfer (`_dev_sync`), which only costs time."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stream_ordered = dist.get_backend(group) == "nccl"

    def _dev_sync(self, tensors):
        if not self.stream_ordered and any(t.is_cuda for t in tensors):
            torch.cuda.synchronize()

    def allreduce_sum(self, t):
        if self.world > 1:
            self._dev_sync([t])
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self._dev_sync([t])
        return t

    def all_gather_object(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def exchange(self, sends, recvs):
        """sends/recvs: lists of (peer, tensor).  One grouped RCCL send/recv batch."""
        self.exchange_start(sends, recvs)()

    def exchange_start(self, sends, recvs):
        """Post the grouped send/recv batch; returns wait() that completes it (stream order)."""
        if not sends and not recvs:
            return lambda: None
        tensors = [t for _, t in sends] + [t for _, t in recvs]
        self._dev_sync(tensors)
        ops = [dist.P2POp(dist.irecv, t, peer, self.group) for peer, t in recvs]
        ops += [dist.P2POp(dist.isend, t, peer, self.group) for peer, t in sends]
        works = dist.batch_isend_irecv(ops)

        def wait():
            for w in works:
                w.wait()
            self._dev_sync(tensors)
        return wait

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


class SingleComm:
    """world_size 1 without a process group."""
    rank, world = 0, 1

    def allreduce_sum(self, t):
        return t

    def all_gather_object(self, obj):
        return [obj]

    def exchange(self, sends, recvs):
        assert not sends and not recvs

    def exchange_start(self, sends, recvs):
        assert not sends and not recvs
        return lambda: None

    def barrier(self):
        pass


# ----------------------------------------------------------------------------- operator

class DistCSR:
    """Row block of a CSR matrix + its halo plan.  `A_local` is a device csr handle whose
    column space is the extended vector."""

    def __init__(self, A_local, plan, comm, backend):
        self.A, self.plan, self.comm, self.be = A_local, plan, comm, backend
        self.n_local = plan.n_owned
        self.n_global = plan.n_global
        self._send_idx = {}
        self._send_buf = {}
        for q, s in plan.send.items():
            if not isinstance(s, tuple):
                self._send_idx[q] = backend.index_tensor(s)
                self._send_buf[q] = backend.zeros(len(s))

    @classmethod
    def poisson(cls, nx, ny, nz, comm, backend, make_local):
        """make_local(nx, ny, nz, row_lo, row_hi, col_shift, ncols_local) -> csr handle."""
        plan = poisson_halo_plan(nx, ny, nz, comm.world, comm.rank)
        A = make_local(nx, ny, nz, plan.row_lo, plan.row_hi, plan.row_lo - plan.ghost_lo, plan.n_ext)
        return cls(A, plan, comm, backend)

    @classmethod
    def from_global_csr(cls, n, ind, col, val, comm, backend, make_local):
        """Row block [lo, hi) of a global CSR triple held on the host by every rank (small / medium problems:
        MatrixMarket files, the irregular configs).  make_local(shape, ind, col, val) -> csr handle
        (HipBackend: DeviceCSR.from_arrays)."""
        lo, hi = row_range(n, comm.world, comm.rank)
        a, b = int(ind[lo]), int(ind[hi])
        ind_loc = (np.asarray(ind[lo:hi + 1], dtype=np.int64) - a).astype(np.int32)
        plan, col_local = general_halo_plan(n, lo, hi, col[a:b], comm.world, comm.rank, comm.all_gather_object,
                                            ind=ind_loc)
        A = make_local((hi - lo, plan.n_ext), ind_loc, col_local, np.ascontiguousarray(val[a:b], dtype=np.float64))
        return cls(A, plan, comm, backend)

    @classmethod
    def from_global_sss(cls, n, ind, col, val, diag, comm, backend, make_local):
        """An sss_mat (strict lower triangle + diagonal, sss_mat.h:6-14) on row blocks: every rank expands the
        GLOBAL matrix (it is handed the global arrays, as from_global_csr is) into full rows in the order
        sss_matvec adds them (sss_mat.c:45-55: lower entries by ascending column,
        the diagonal, then the mirrored entries by ascending row), so the distributed product has the bits of the
        single-GPU sss_mat.matvec."""
        f_ind, f_col, f_val = sss_rows_expanded(n, ind, col, val, diag)
        return cls.from_global_csr(n, f_ind, f_col, f_val, comm, backend, make_local)

    def new_ext(self):
        return self.be.zeros(self.plan.n_ext)

    def owned(self, v_ext):
        o = self.plan.p_offset
        return v_ext[o:o + self.n_local]

    def _halo_ops(self, v_ext):
        own = self.owned(v_ext)
        sends, recvs = [], []
        for q, s in sorted(self.plan.send.items()):
            if isinstance(s, tuple):
                sends.append((q, own[s[0]:s[1]]))
            else:
                self.be.gather(self._send_idx[q], own, self._send_buf[q])
                sends.append((q, self._send_buf[q]))
        for q, (a, b) in sorted(self.plan.recv.items()):
            recvs.append((q, v_ext[a:b]))
        return sends, recvs

    def halo_exchange(self, v_ext):
        """Fill the ghost entries of v_ext from the neighbours' owned entries."""
        if self.comm.world == 1:
            return
        sends, recvs = self._halo_ops(v_ext)
        self.comm.exchange(sends, recvs)

    def halo_time(self, v_ext, reps=10, warmup=2):
        """milliseconds per ghost exchange on its own (posted, completed, nothing else in flight): what the interior
        rows have to hide.  Collective: every rank calls it."""
        import time
        cuda = bool(getattr(v_ext, "is_cuda", False))
        for k in range(-warmup, reps):
            if k == 0:
                if cuda:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                t0 = time.perf_counter()
            self.halo_exchange(v_ext)
        if cuda:
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        return (time.perf_counter() - t0) * 1e3 / reps

    def matvec(self, v_ext, y, want_dot=False):
        """y = A v for the owned rows; v_ext's owned part must be current.  The ghost
        exchange is started first and overlapped with the rows that need no ghost entry.
        want_dot: also return the device scalar sum v_owned[i]*y[i]."""
        if self.comm.world == 1:
            if want_dot:
                return self.be.matvec_dot(self.A, v_ext, self.plan.p_offset, y)
            self.be.matvec(self.A, v_ext, y)
            return None
        sends, recvs = self._halo_ops(v_ext)
        wait = self.comm.exchange_start(sends, recvs)
        return self.be.matvec_overlap(self.A, v_ext, self.plan.p_offset, y, self.plan.interior, wait, want_dot)


class PhaseTimer:
    """Where the time of one iteration goes (bench.py `phases`): marks between the phases of the device-scalar loops.
    Device tensors: a torch.cuda.Event per mark on the current stream -- the stream the library's kernels are
    enqueued on and the RCCL collectives synchronise with -- so a difference of two marks is time on the GPU's own
    timeline (a mark behind `wait()` completes when the ghost entries have arrived AND the interior rows are done: what
    is left of it after the interior rows is the part of the halo exchange the interior rows did not hide).  CPU
    tensors (the gloo dry runs): host clock."""

    PHASES = ("px_update", "spmv_interior", "halo_exposed", "spmv_boundary", "allreduce_1", "scalar_1", "r_update",
              "allreduce_2", "scalar_2")

    def __init__(self, cuda):
        self.cuda = bool(cuda)
        self.iters = []
        self.cur = None

    def begin(self):
        self.cur = []
        self.iters.append(self.cur)
        self.mark("begin")

    def mark(self, name):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.cur.append((name, e))
        else:
            import time
            self.cur.append((name, time.perf_counter()))

    def summary(self, skip=2):
        """mean milliseconds per phase over the recorded iterations (the first `skip` are warm-up)"""
        if self.cuda:
            torch.cuda.synchronize()
        acc, cnt = {}, 0
        for marks in self.iters[skip:] or self.iters:
            cnt += 1
            for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
                dt = a.elapsed_time(b) if self.cuda else (b - a) * 1e3
                acc[name] = acc.get(name, 0.0) + dt
        out = {k: v / max(cnt, 1) for k, v in acc.items()}
        out["iteration"] = sum(out.values())
        out["iterations_timed"] = cnt
        return out


def _matvec_state(A, kd, st, v_ext, y, timer=None, shake=None):
    """y = A v for the owned rows through a state-driven kernel (kd = be.kd_matvec_overlap or
    be.kd_minres_matvec): the ghost exchange is started first and overlapped with the interior rows;
    the local v_owned . y lands in be.scal[0]"""
    if A.comm.world == 1:
        kd(st, A.A, v_ext, A.plan.p_offset, y, (0, A.n_local), None)
        if timer is not None:
            timer.mark("spmv_interior")
        return
    if shake is not None:
        shake()  # the vector the sends read is late
    sends, recvs = A._halo_ops(v_ext)
    wait = A.comm.exchange_start(sends, recvs)
    if shake is not None:
        shake()  # the interior rows are late against the transfers
    if timer is not None or shake is not None:
        inner = wait

        def wait():  # called by the kernel driver between the interior and the boundary rows
            if timer is not None:
                timer.mark("spmv_interior")
            if shake is not None:
                shake()
            inner()
            if shake is not None:
                shake()  # the boundary rows are late against the next iteration's traffic
            if timer is not None:
                timer.mark("halo_exposed")
    kd(st, A.A, v_ext, A.plan.p_offset, y, A.plan.interior, wait)
    if timer is not None:
        timer.mark("spmv_boundary")


def _tuning(name, default):
    """A/B switches are read only when the process was started with PSP_TUNING=1"""
    return os.environ.get(name, default) if os.environ.get("PSP_TUNING") == "1" else default


SHAKE_INJECTED = 0  # spin kernels enqueued by _Shaker in this process (tests)


class _Shaker:
    """Delay injection at the cut points of the rank-per-process loops (the twin of psp::shake in psp_multi.hip): under
    PSP_TUNING=1 PSP_DIST_SHAKE="seed,max_us" every cut point enqueues, with probability 1/2, a spin kernel of up to
    max_us microseconds (psp_debug_spin) on the stream the library's kernels run on -- torch's current stream, the one
    RCCL's collectives and send / recv batches order themselves against.  Each rank draws its own sequence (seed +
    rank), so the ranks drift against each other as well.  The ordering edges this moves, per iteration:
      px / scale kernel -> [send reads p | recv writes the ghost zone]  RCCL waits for the current stream at post time
      recv -> boundary rows                                             wait() makes the current stream wait for RCCL
      send -> next px / scale kernel (overwrites p)                     the same wait(), before the boundary rows
      local sums -> all-reduce -> scalar step                           stream order (RCCL) / _dev_sync (gloo)
    Results must not depend on it: tests/test_gpu_shake.py::test_torch_ranks_under_delay_injection."""

    def __init__(self, be, rank):
        spec = _tuning("PSP_DIST_SHAKE", "")
        self.on = bool(spec) and hasattr(be, "L") and hasattr(be.L, "psp_debug_spin")
        self.count = 0
        if self.on:
            parts = [int(t) for t in spec.split(",") if t.strip()]
            self.rng = np.random.default_rng((parts[0] if parts else 0) + 7919 * rank)
            self.max_us = parts[1] if len(parts) > 1 else 200
            self.be = be

    def __call__(self):
        if self.on and self.rng.integers(0, 2):
            us = int(self.rng.integers(1, self.max_us + 1))
            self.be._capi.check(self.be.L.psp_debug_spin(us))
            self.count += 1
            global SHAKE_INJECTED
            SHAKE_INJECTED += 1


def dist_pcg_mode():
    """which loop dist_pcg runs (bench.py prints it)"""
    if _tuning("PSP_DIST_DEVSCALARS", "1") != "0":
        return "device-resident scalars, in-stream all-reduces, lazy x update"
    return "host scalars, lazy x update" if _tuning("PSP_DIST_LAZYX", "1") != "0" else "host scalars, eager"


def dist_pcg(A, b, x, tol, maxit, dinv=None, hist=None, timer=None):
    """info, iter, relres = dist_pcg(A: DistCSR, b, x, tol, maxit, dinv) on the owned slices; see
    _dist_pcg.  A dinv slice that holds one value everywhere (constant-diagonal operator) is
    announced to the vector kernels for the duration of the solve (psp_k_hint_constant).
    timer: a PhaseTimer that receives one set of marks per iteration (device-scalar loop only)."""
    if hasattr(A.be, "bind_current_stream"):
        A.be.bind_current_stream()
    hint = getattr(A.be, "hint_constant", None) if dinv is not None else None
    if hint is not None:
        hint(dinv)
    lazy = hasattr(A.be, "px_update") and _tuning("PSP_DIST_LAZYX", "1") != "0"
    devs = hasattr(A.be, "kd_px_update") and _tuning("PSP_DIST_DEVSCALARS", "1") != "0"
    try:
        if devs:
            return _dist_pcg_dev(A, b, x, tol, maxit, dinv, hist, timer)
        return (_dist_pcg_lazy if lazy else _dist_pcg)(A, b, x, tol, maxit, dinv, hist)
    finally:
        if hint is not None:
            A.be.unhint(dinv)


def _dist_pcg_lazy(A, b, x, tol, maxit, dinv=None, hist=None):
    """Same results as _dist_pcg with 8 bytes per row less traffic per iteration: the x update and the
    stagnation scan of iteration k ride in the p update of iteration k+1 (they share the read of p), and
    the scan's flag shares all-reduce #1 with p.q.  Exit order as in psp_solvers.hip's lazy loop: the
    stagnation of iteration k (-5, pcg.c:159-162) is known after all-reduce #1 of iteration k+1 and is
    tested before that iteration's rho == 0 / beta == 0 / p.q == 0 exits; convergence ends the loop at
    once and the pending update is applied afterwards; when the loop runs out the final scan decides
    between -5 and -1."""
    be, comm = A.be, A.comm
    n = A.n_local
    r, q = be.zeros(n), be.zeros(n)
    p_ext = A.new_ext()
    p = A.owned(p_ext)

    s = comm.allreduce_sum(be.dot(b, b).clone()).tolist()
    n2b = float(np.sqrt(s[0]))
    if n2b == 0.0:  # pcg.c:58-67
        x.zero_()
        return 0, 0, 0.0
    tolb = tol * n2b
    p.copy_(x)
    A.matvec(p_ext, r)
    s = comm.allreduce_sum(be.residual(b, r, dinv).clone()).tolist()
    normr = float(np.sqrt(s[0]))
    rho_next = s[1]
    if hist is not None:
        hist.append(normr)
    if normr <= tolb:  # pcg.c:77-84
        return 0, 0, normr / n2b
    info = -1
    rho = 1.0
    xpend, alpha_x, stag0, pend_maxit = False, 0.0, False, False
    it = 1
    while it <= maxit:
        rho1, rho = rho, rho_next
        beta = 0.0
        head_rho0 = rho == 0.0
        head_beta0 = False
        if it > 1 and not head_rho0:
            beta = rho / rho1
            head_beta0 = beta == 0.0
        nonstag = be.px_update(r, dinv, beta, it == 1, alpha_x, xpend, p, x)
        pq_dev = A.matvec(p_ext, q, want_dot=True)
        s = comm.allreduce_sum(torch.cat([pq_dev.reshape(1), nonstag.reshape(1)])).tolist()  # all-reduce #1
        if xpend:
            xpend = False
            if stag0 or s[1] == 0.0:  # iteration it-1 stagnated on every rank (pcg.c:159-162)
                info = -5
                it -= 1
                break
        if head_rho0:  # pcg.c:101-104
            info = -2
            break
        if head_beta0:  # pcg.c:109-112
            info = -6
            break
        pq = s[0]
        if pq == 0.0:  # pcg.c:118-120
            info = -6
            break
        alpha = rho / pq
        stag0 = alpha == 0.0
        alpha_x, xpend = alpha, True
        s = comm.allreduce_sum(be.r_update(alpha, q, dinv, r).clone()).tolist()  # all-reduce #2
        normr = float(np.sqrt(s[0]))
        rho_next = s[1]
        if hist is not None:
            hist.append(normr)
        if normr <= tolb:  # pcg.c:154-157
            info = 0
            break
        if it == maxit:
            pend_maxit = True
            break
        it += 1
    if xpend:  # the x update (and scan) of the last iteration
        s = comm.allreduce_sum(be.x_update(alpha_x, p, x).clone()).tolist()
        if pend_maxit:
            if stag0 or s[0] == 0.0:
                info = -5
            else:
                info, it = -1, maxit + 1  # pcg.c:165
    return info, it, normr / n2b


def _dist_pcg(A, b, x, tol, maxit, dinv=None, hist=None):
    """info, iter, relres = dist_pcg(A: DistCSR, b, x, tol, maxit, dinv) on the owned slices.

    Same control flow as Itsolvers_pcg_kernel (pcg.c:57-166) with K = None (dinv is None)
    or K = jacobi(steps=1) given by its local dinv slice; every rank returns the same
    triple.  x is updated in place."""
    be, comm = A.be, A.comm
    n = A.n_local
    r, q = be.zeros(n), be.zeros(n)
    p_ext = A.new_ext()
    p = A.owned(p_ext)

    s = comm.allreduce_sum(be.dot(b, b).clone()).tolist()
    n2b = float(np.sqrt(s[0]))
    if n2b == 0.0:  # pcg.c:58-67
        x.zero_()
        return 0, 0, 0.0
    info = -1
    tolb = tol * n2b
    # r = b - A x (pcg.c:72-75); also rho = r.z for the first iteration
    p.copy_(x)
    A.matvec(p_ext, r)
    s = comm.allreduce_sum(be.residual(b, r, dinv).clone()).tolist()
    normr = float(np.sqrt(s[0]))
    rho_next = s[1]
    if hist is not None:
        hist.append(normr)
    if normr <= tolb:  # pcg.c:77-84
        return 0, 0, normr / n2b
    rho = 1.0
    stag = 0
    it = 1
    while it <= maxit:
        rho1, rho = rho, rho_next
        if rho == 0.0:  # pcg.c:101-104
            info = -2
            break
        if it == 1:
            be.pupdate(r, dinv, 0.0, True, p)
        else:
            beta = rho / rho1
            if beta == 0.0:  # pcg.c:109-112
                info = -6
                break
            be.pupdate(r, dinv, beta, False, p)
        pq = comm.allreduce_sum(A.matvec(p_ext, q, want_dot=True).clone()).tolist()[0]  # all-reduce #1
        if pq == 0.0:  # pcg.c:118-120
            info = -6
            break
        alpha = rho / pq
        if alpha == 0.0:
            stag = 1
        s = comm.allreduce_sum(be.xr_update(alpha, p, q, dinv, x, r).clone()).tolist()  # all-reduce #2
        if stag == 0:
            stag = 1 if s[2] == 0.0 else 0  # every rank's local 1 + dmax == 1
        normr = float(np.sqrt(s[0]))
        rho_next = s[1]
        if hist is not None:
            hist.append(normr)
        if normr <= tolb:  # pcg.c:154-157
            info = 0
            break
        if stag == 1:  # pcg.c:159-162
            info = -5
            break
        it += 1
    return info, it, normr / n2b


PCG_BATCH = 16  # iterations enqueued between two reads of the device state


def _dist_pcg_dev(A, b, x, tol, maxit, dinv=None, hist=None, timer=None):
    """_dist_pcg_lazy with the scalars on the device: per iteration the host only enqueues -- px update,
    SpMV around the halo exchange, all-reduce #1 {p.q, nonstag} (in stream order), the scalar step that
    takes pcg.c:159-162 / :101-125's branches, r update, all-reduce #2 {r.r, r.z}, the scalar step of
    pcg.c:152-157 -- and reads the state back once per PCG_BATCH iterations.  Every kernel of an iteration
    that starts after the loop has ended is a no-op (the all-reduces still run; their operands are ignored).
    Same results as the host-scalar loops bit for bit (tests/test_distributed_cpu.py, test_gpu_distributed.py)."""
    be, comm = A.be, A.comm
    n = A.n_local
    r, q = be.zeros(n), be.zeros(n)
    p_ext = A.new_ext()
    p = A.owned(p_ext)

    s = comm.allreduce_sum(be.dot(b, b).clone()).tolist()
    n2b = float(np.sqrt(s[0]))
    if n2b == 0.0:  # pcg.c:58-67
        x.zero_()
        return 0, 0, 0.0
    tolb = tol * n2b
    p.copy_(x)
    A.matvec(p_ext, r)
    s = comm.allreduce_sum(be.residual(b, r, dinv).clone()).tolist()
    normr = float(np.sqrt(s[0]))
    if hist is not None:
        hist.append(normr)
    if normr <= tolb:  # pcg.c:77-84
        return 0, 0, normr / n2b
    if maxit < 1:
        return -1, 1, normr / n2b
    if s[1] == 0.0:  # pcg.c:101-104 in iteration 1
        return -2, 1, normr / n2b
    st = be.pcg_state(n2b, tolb, normr, s[1], maxit, hist is not None)
    shake = _Shaker(be, comm.rank)
    if not shake.on:
        shake = None
    try:
        enq = 0
        while True:
            batch = max(1, min(PCG_BATCH, maxit - enq))
            for _ in range(batch):
                if timer is not None:
                    timer.begin()
                if shake is not None:
                    shake()
                be.kd_px_update(st, r, dinv, p, x)                      # -> scal[1]
                if timer is not None:
                    timer.mark("px_update")
                _matvec_state(A, be.kd_matvec_overlap, st, p_ext, q, timer, shake)  # -> scal[0]
                if shake is not None:
                    shake()
                comm.allreduce_sum(be.scal[0:2])                        # all-reduce #1
                if timer is not None:
                    timer.mark("allreduce_1")
                if shake is not None:
                    shake()
                be.kd_pcg_scalar_xpq(st)
                if timer is not None:
                    timer.mark("scalar_1")
                be.kd_r_update(st, q, dinv, r)                          # -> scal[2:4]
                if timer is not None:
                    timer.mark("r_update")
                if shake is not None:
                    shake()
                comm.allreduce_sum(be.scal[2:4])                        # all-reduce #2
                if timer is not None:
                    timer.mark("allreduce_2")
                be.kd_pcg_scalar_r(st)
                if timer is not None:
                    timer.mark("scalar_2")
            enq += batch
            f = st.fetch()
            if f.status:
                break
        info, it, relres = f.info, f.iter, f.relres
        if f.xpend:  # the x update (and scan) of the last iteration
            flag = comm.allreduce_sum(be.x_update(f.alpha_x, p, x).clone()).tolist()
            if f.pend_maxit:
                stag = bool(f.stag0) or flag[0] == 0.0
                info, it = (-5, maxit) if stag else (-1, maxit + 1)  # pcg.c:159-165
                relres = f.normr / f.n2b
        if hist is not None:
            # -2 / -6 leave at the head of iteration `it` (or at p.q == 0) before its residual norm exists: that
            # slot was never written (NaN fill) and the host-scalar loops append nothing for it either
            cnt = min(it, maxit)
            if cnt >= 1:
                hist.extend(float(v) for v in st.hist(1, cnt) if not np.isnan(v))
        return info, it, relres
    finally:
        st.close()


def dist_minres(A, b, x, tol, maxit, dinv=None, hist=None):
    """info, iter, relres = dist_minres(A: DistCSR, b, x, tol, maxit, dinv) on the owned slices: the
    reference's preconditioned MINRES (pysparse/itsolvers/src/minres.c:43-200) on row blocks, K = None
    (dinv is None) or jacobi(steps=1) given by its local dinv slice.  Two all-reduces per iteration --
    alpha = v.Av (minres.c:129) and beta^2 = v_hat.y (:143) -- issued in stream order; the Lanczos / Givens
    recurrences run on the device (psp_minresstate_*), the host reads the state once per PCG_BATCH
    iterations.  x is updated in place; every rank returns the same triple."""
    be, comm = A.be, A.comm
    if hasattr(be, "bind_current_stream"):
        be.bind_current_stream()
    n = A.n_local
    v_hat, v_hat_old = be.zeros(n), be.zeros(n)
    wv, w_old, av = be.zeros(n), be.zeros(n), be.zeros(n)
    v_ext = A.new_ext()
    v = A.owned(v_ext)
    y = be.zeros(n) if dinv is not None else None
    hint = getattr(be, "hint_constant", None) if dinv is not None else None
    if hint is not None:
        hint(dinv)
    st = None
    try:
        # v_hat = b - A x, norm_r0 (minres.c:67-71); y = K v_hat, beta = sqrt(v_hat.y) (:73-82)
        v.copy_(x)
        A.matvec(v_ext, v_hat)
        s = comm.allreduce_sum(be.residual(b, v_hat, dinv).clone()).tolist()
        norm_r0 = float(np.sqrt(s[0]))
        beta = s[1]
        if dinv is not None:
            be.jacobi(v_hat, dinv, y)
        if beta < 0.0:  # minres.c:79-80
            return -3, 0, 0.0
        beta = float(np.sqrt(beta))
        if hist is not None:
            hist.append(norm_r0)
        conv0 = norm_r0 < tol * norm_r0
        if maxit < 1 or conv0:  # minres.c:114 before the first iteration
            return (0 if conv0 else -1), 0, float(np.float64(norm_r0) / np.float64(norm_r0))
        st = be.minres_state(norm_r0, beta, tol, maxit, hist is not None)
        shake = _Shaker(be, comm.rank)
        if not shake.on:
            shake = None
        enq = 0
        while True:
            batch = max(1, min(PCG_BATCH, maxit - enq))
            for _ in range(batch):
                if shake is not None:
                    shake()
                be.kd_minres_scale(st, y if dinv is not None else v_hat, v)   # v = y / beta
                _matvec_state(A, be.kd_minres_matvec, st, v_ext, av, None, shake)  # -> scal[0]
                if shake is not None:
                    shake()
                comm.allreduce_sum(be.scal[0:1])                              # all-reduce #1: alpha
                if shake is not None:
                    shake()
                be.kd_minres_scalar(st, 0)
                be.kd_minres_lanczos(st, av, v_hat, v_hat_old, dinv, y)       # -> scal[4]
                v_hat, v_hat_old = v_hat_old, v_hat
                if shake is not None:
                    shake()
                comm.allreduce_sum(be.scal[4:5])                              # all-reduce #2: beta^2
                be.kd_minres_scalar(st, 1)
                be.kd_minres_wx(st, v, wv, w_old, x)
                wv, w_old = w_old, wv
            enq += batch
            f = st.fetch()
            if f.status or f.stop:
                break
        if hist is not None:
            cnt = min(f.iter, maxit)
            if cnt >= 1:
                hist.extend(float(t) for t in st.hist(1, cnt) if not np.isnan(t))
        return f.info, f.iter, (f.relres if f.info in (0, -1) else 0.0)
    finally:
        if st is not None:
            st.close()
        if hint is not None:
            be.unhint(dinv)
