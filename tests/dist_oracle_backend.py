"""CPU stand-in for pysparse_amd.distributed.HipBackend used ONLY by the gloo tests: the
same method set on torch CPU tensors, arithmetic delegated to the oracle / numpy.  It lets
world_size-2 tests drive the product's partition + halo + reduction logic without a GPU."""
import numpy as np
import torch

from oracle import oracle as O
from pysparse_amd.distributed import HostStateOps


class OracleBackend(HostStateOps):
    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())

    def index_tensor(self, idx):
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32).copy())

    def dot(self, x, y):
        return torch.tensor([float(np.dot(x.numpy(), y.numpy()))], dtype=torch.float64)

    def residual(self, b, r, dinv):
        rn = r.numpy()
        rn[:] = b.numpy() - rn
        z = rn * dinv.numpy() if dinv is not None else rn
        return torch.tensor([float(np.dot(rn, rn)), float(np.dot(rn, z))], dtype=torch.float64)

    def pupdate(self, r, dinv, beta, first, p_owned):
        z = r.numpy() * dinv.numpy() if dinv is not None else r.numpy()
        pn = p_owned.numpy()
        pn[:] = z if first else z + beta * pn

    def matvec(self, A, p_ext, q):
        A.matvec(np.ascontiguousarray(p_ext.numpy()), q.numpy())

    def matvec_dot(self, A, p_ext, p_offset, q):
        self.matvec(A, p_ext, q)
        n = q.numel()
        return torch.tensor([float(np.dot(p_ext.numpy()[p_offset:p_offset + n], q.numpy()))], dtype=torch.float64)

    def matvec_overlap(self, A, p_ext, p_offset, q, interior, wait, want_dot):
        # rows of `interior` must not depend on ghost entries: multiply them BEFORE the
        # exchange completes and check that the result survives unchanged afterwards
        n = q.numel()
        before = q.numpy().copy()
        A.matvec(np.ascontiguousarray(p_ext.numpy()), before)
        wait()
        self.matvec(A, p_ext, q)
        lo, hi = interior
        assert np.array_equal(before[lo:hi], q.numpy()[lo:hi]), "interior rows touched ghost entries"
        if want_dot:
            return torch.tensor([float(np.dot(p_ext.numpy()[p_offset:p_offset + n], q.numpy()))], dtype=torch.float64)
        return None

    def xr_update(self, alpha, p, q, dinv, x, r):
        pn, qn, xn, rn = p.numpy(), q.numpy(), x.numpy(), r.numpy()
        dmax = 0.0
        with np.errstate(all="ignore"):
            nz = xn != 0.0
            if nz.any():
                dmax = float(np.nanmax(np.abs(alpha * pn[nz] / xn[nz]), initial=0.0))
            if ((~nz) & (pn != 0.0)).any():
                dmax = max(dmax, 1.0)
        if alpha != 0.0:
            xn += alpha * pn
            rn += (-alpha) * qn
        z = rn * dinv.numpy() if dinv is not None else rn
        return torch.tensor([float(np.dot(rn, rn)), float(np.dot(rn, z)), 1.0 if 1.0 + dmax != 1.0 else 0.0],
                            dtype=torch.float64)

    @staticmethod
    def _scan(alpha, pn, xn):
        dmax = 0.0
        with np.errstate(all="ignore"):
            nz = xn != 0.0
            if nz.any():
                dmax = float(np.nanmax(np.abs(alpha * pn[nz] / xn[nz]), initial=0.0))
            if ((~nz) & (pn != 0.0)).any():
                dmax = max(dmax, 1.0)
        return 1.0 if 1.0 + dmax != 1.0 else 0.0

    def px_update(self, r, dinv, beta, first, alpha_x, xpend, p_owned, x):
        pn, xn = p_owned.numpy(), x.numpy()
        flag = 0.0
        if xpend:
            flag = self._scan(alpha_x, pn, xn)
            if alpha_x != 0.0:
                xn += alpha_x * pn
        z = r.numpy() * dinv.numpy() if dinv is not None else r.numpy()
        pn[:] = z if first else z + beta * pn
        return torch.tensor([flag], dtype=torch.float64)

    def r_update(self, alpha, q, dinv, r):
        rn = r.numpy()
        if alpha != 0.0:
            rn += (-alpha) * q.numpy()
        z = rn * dinv.numpy() if dinv is not None else rn
        return torch.tensor([float(np.dot(rn, rn)), float(np.dot(rn, z))], dtype=torch.float64)

    def x_update(self, alpha, p_owned, x):
        pn, xn = p_owned.numpy(), x.numpy()
        flag = self._scan(alpha, pn, xn)
        if alpha != 0.0:
            xn += alpha * pn
        return torch.tensor([flag], dtype=torch.float64)

    # ---- MINRES pieces (minres.c:123-124, :131-143, :172-180), host scalars
    def jacobi(self, x, dinv, y):
        y.numpy()[:] = x.numpy() * dinv.numpy()

    def scale_div(self, y, beta, v):
        v.numpy()[:] = y.numpy() / beta

    def lanczos(self, av, c1, c2, v_hat, v_hat_old, dinv, y):
        vo = v_hat_old.numpy()
        vo[:] = av.numpy() - c1 * v_hat.numpy() - c2 * vo  # the caller swaps the names
        if dinv is not None:
            y.numpy()[:] = vo * dinv.numpy()
            return torch.tensor([float(np.dot(vo, y.numpy()))], dtype=torch.float64)
        return torch.tensor([float(np.dot(vo, vo))], dtype=torch.float64)

    def minres_wx(self, v, r1, r2, r3, c_eta, w, w_old, x):
        wo = w_old.numpy()
        wo[:] = (v.numpy() - r3 * wo - r2 * w.numpy()) / r1  # the caller swaps the names
        x.numpy()[:] += c_eta * wo

    def gather(self, idx, v, out):
        out.numpy()[:] = v.numpy()[idx.numpy()]

    def synchronize(self):
        pass


def local_poisson_from_oracle(nx, ny, nz, row_lo, row_hi, col_shift, ncols_local):
    A = O.poisson_csr(nx, ny, nz)
    a, b = A.ind[row_lo], A.ind[row_hi]
    return O.CSR((row_hi - row_lo, ncols_local), A.val[a:b], A.col[a:b] - col_shift, A.ind[row_lo:row_hi + 1] - a)


def bench_factory():
    """bench.py --test-backend tests.dist_oracle_backend:bench_factory (dry run of the launcher over gloo)"""
    return OracleBackend(), local_poisson_from_oracle
