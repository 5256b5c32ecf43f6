// psp_vec.hip -- the fp64 vector kernels of the Krylov loops (the reference's BLAS-1 calls
// and hand loops in pysparse/itsolvers/src/pcg.c and minres.c), fused so that every
// vector is streamed as few times as the data dependencies allow.
//
// All kernels: 256-thread workgroups, one contiguous span per workgroup with 16-byte-per-lane
// accesses (two 8-byte ones when a pointer is not 16-byte aligned or n is odd: same elements per
// thread, so the same bits), per-lane partial
// sums -> wave shuffle -> LDS -> one slot per workgroup; a one-block finishing kernel adds
// the slots in index order (bitwise reproducible, no atomics).  Per-element arithmetic
// follows the reference expression order; the library is built with -ffp-contract=off.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "psp_internal.h"

using namespace psp;

namespace {

constexpr int kBlock = 256;

template <int V>
struct alignas(8 * V) Pack {
  double v[V];
};

template <int V>
__device__ __forceinline__ Pack<V> ld(const double *p, long i) {
  return *reinterpret_cast<const Pack<V> *>(p + i);
}
typedef double d2nt __attribute__((ext_vector_type(2)));

// vector results are written once and next read after >= 1 GB of other traffic: store them
// non-temporally when PSP_VEC_NT_STORE is set (A/B: profiles/)
template <int V>
__device__ __forceinline__ void st(double *p, long i, const Pack<V> &x) {
#ifdef PSP_VEC_NT_STORE
  if constexpr (V == 2) {
    d2nt t;
    t.x = x.v[0];
    t.y = x.v[1];
    __builtin_nontemporal_store(t, reinterpret_cast<d2nt *>(p + i));
  } else {
    __builtin_nontemporal_store(x.v[0], p + i);
  }
#else
  *reinterpret_cast<Pack<V> *>(p + i) = x;
#endif
}

__device__ __forceinline__ double wave_sum(double v) { return psp::psp_wave_sum(v); }

// sums NV per-thread values over the block; thread 0 stores them to partials[j*kMaxParts + block]
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *partials) {
  __shared__ double sh[NV][4];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double s = wave_sum(v[j]);
    if ((threadIdx.x & 63) == 0) sh[j][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
      partials[(size_t)j * kMaxParts + blockIdx.x] = sh[j][0] + sh[j][1] + sh[j][2] + sh[j][3];
  }
}

// workgroup b streams the contiguous span [b*kVecSpan, (b+1)*kVecSpan) (and b + grid, ...).  Thread t owns
// elements 2t and 2t+1 of the span in BOTH forms -- V = 2: one 16-byte access per array; V = 1 (n odd or a
// pointer not 16-byte aligned): two 8-byte accesses -- so every per-thread partial sum, and with it every
// reduction result, is the same whichever form runs: results do not depend on where a buffer happens to
// be allocated (the owned slice of an extended vector starts at an odd offset as often as not).
#define PSP_VEC_LOOP(i, n)                                                             \
  for (long span_ = (long)blockIdx.x * kVecSpan; span_ < (n); span_ += (long)gridDim.x * kVecSpan) \
    _Pragma("unroll") for (int u_ = 0; u_ < kVecSpan / (kBlock * V); ++u_)              \
      for (long i = span_ + (V == 2 ? ((long)u_ * kBlock + threadIdx.x) * 2 : 2L * threadIdx.x + u_); i < (n); i = (n))
static_assert(kVecSpan == 2 * kBlock, "the V = 1 mapping above assumes two elements per thread and span");

// ---- dot: pcg.c:100,117  minres.c:78,129,143
template <int V>
__global__ __launch_bounds__(kBlock) void dot_kernel(long n, const double *__restrict__ x,
                                                     const double *__restrict__ y,
                                                     double *__restrict__ partials) {
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> a = ld<V>(x, i), b = ld<V>(y, i);
#pragma unroll
    for (int u = 0; u < V; ++u) acc[0] += a.v[u] * b.v[u];
  }
  block_reduce_store<1>(acc, partials);
}

// ---- r = b - r; partials {r.r, r.z}, z = dinv.*r or r: pcg.c:73-75 (+ :93-100 fused)
// PRE: 0 no preconditioner, 1 dinv array, 2 dinv is the constant dc everywhere (same product
// r_i * dinv_i, one 8-byte stream less)
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void residual_kernel(long n, const double *__restrict__ b,
                                                          double *__restrict__ r,
                                                          const double *__restrict__ dinv, double dc,
                                                          double *__restrict__ partials) {
  double acc[2] = {0.0, 0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> bb = ld<V>(b, i);
    Pack<V> rr = ld<V>(r, i);
    Pack<V> dd;
    if constexpr (PRE == 1) dd = ld<V>(dinv, i);
    if constexpr (PRE == 2) {
#pragma unroll
      for (int u = 0; u < V; ++u) dd.v[u] = dc;
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      const double t = bb.v[u] - rr.v[u];
      rr.v[u] = t;
      acc[0] += t * t;
      if constexpr (PRE != 0) {
        const double z = t * dd.v[u];
        acc[1] += t * z;
      }
    }
    st<V>(r, i, rr);
  }
  if constexpr (PRE == 0) acc[1] = acc[0];
  block_reduce_store<2>(acc, partials);
}

// ---- p = z + beta*p (pcg.c:113-114) or p = z (pcg.c:106); z = dinv.*r or r.
//      With a device state the "first iteration" decision is taken on the device (it == 1).
template <int V, int PRE, bool FIRST>
__global__ __launch_bounds__(kBlock) void pupdate_kernel(long n, const double *__restrict__ r,
                                                         const double *__restrict__ dinv, double dc,
                                                         double beta, double *__restrict__ p,
                                                         const PcgDev *__restrict__ dstate) {
  bool first = FIRST;
  if (dstate) {  // asynchronous loop: scalars live on the device
    if (dstate->status) return;
    beta = dstate->beta;
    first = dstate->it == 1;
  }
  PSP_VEC_LOOP(i, n) {
    Pack<V> z = ld<V>(r, i);
    if constexpr (PRE == 1) {
      const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] * dd.v[u];
    }
    if constexpr (PRE == 2) {
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] * dc;
    }
    if (!first) {
      const Pack<V> pp = ld<V>(p, i);
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] + beta * pp.v[u];
    }
    st<V>(p, i, z);
  }
}

// ---- lazy x-update (pcg_async_loop_lazy): ONE pass does the x update + stagnation scan of the
//      iteration that just finished (pcg.c:127-141, exactly x_update_kernel's arithmetic) and the
//      p update of the iteration that starts (pcg.c:105-115, pupdate_kernel's) -- they share the
//      read of p: 40 n bytes instead of 24 n + 24 n.  Scalars come from the device state.
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void px_update_kernel(long n, const double *__restrict__ r,
                                                           const double *__restrict__ dinv, double dc,
                                                           double *__restrict__ p, double *__restrict__ x,
                                                           double *__restrict__ partials,
                                                           const PcgDev *__restrict__ dstate, double beta,
                                                           double alpha, int first_, int xpend_) {
  bool first = first_ != 0, xp = xpend_ != 0;
  if (dstate) {  // asynchronous loop: scalars live on the device; otherwise they are the arguments
    if (dstate->status) return;
    beta = dstate->beta;
    alpha = dstate->alpha_x;
    first = dstate->it == 1;
    xp = dstate->xpend != 0;
  }
  const bool upd = alpha != 0.0;
  double dmax = 0.0;
  PSP_VEC_LOOP(i, n) {
    const Pack<V> pp = ld<V>(p, i);  // garbage in iteration 1: neither use below reads it then
    Pack<V> z = ld<V>(r, i);
    if (xp) {
      Pack<V> xx = ld<V>(x, i);
#pragma unroll
      for (int u = 0; u < V; ++u) {
        const double quot = fabs(alpha * pp.v[u] / xx.v[u]);
        const double ddum = (xx.v[u] != 0.0) ? quot : ((pp.v[u] != 0.0) ? 1.0 : 0.0);
        dmax = (ddum > dmax) ? ddum : dmax;
        if (upd) xx.v[u] = xx.v[u] + alpha * pp.v[u];
      }
      st<V>(x, i, xx);
    }
    if constexpr (PRE == 1) {
      const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] * dd.v[u];
    }
    if constexpr (PRE == 2) {
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] * dc;
    }
    if (!first) {
#pragma unroll
      for (int u = 0; u < V; ++u) z.v[u] = z.v[u] + beta * pp.v[u];
    }
    st<V>(p, i, z);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(dmax, off, 64);
    if (o > dmax) dmax = o;
  }
  double acc[1];
  acc[0] = ((threadIdx.x & 63) == 0 && (1.0 + dmax != 1.0)) ? 1.0 : 0.0;
  block_reduce_store<1>(acc, partials + 2 * (size_t)kMaxParts);
}

// ---- pcg.c:127-152 as TWO streaming kernels.  One fused pass over x, p, r, q, dinv (7 HBM
//      streams) measured 1.39 ms at n = 2^27; the two passes below (3 and 4 streams, the same
//      56 n bytes in total) take 0.54 + 0.74 ms (profiles/r1_vec_kernels.txt).
//      x_update: stagnation scan (:127-139) + x += alpha p (:141); partial slot 2 = nonstag
//      (1 for a workgroup whose local dmax has 1 + dmax != 1)
template <int V>
__global__ __launch_bounds__(kBlock) void x_update_kernel(long n, double alpha,
                                                          const double *__restrict__ p,
                                                          double *__restrict__ x,
                                                          double *__restrict__ partials,
                                                          const PcgDev *__restrict__ dstate) {
  if (dstate) {
    if (dstate->status) return;
    alpha = dstate->alpha;
  }
  double dmax = 0.0;
  const bool upd = alpha != 0.0;
  PSP_VEC_LOOP(i, n) {
    const Pack<V> pp = ld<V>(p, i);
    Pack<V> xx = ld<V>(x, i);
#pragma unroll
    for (int u = 0; u < V; ++u) {
      // pcg.c:128-137, branch-free: x != 0 -> ddum = |alpha*p/x| (a NaN never replaces dmax,
      // like `if (ddum > dmax)`); x == 0 and p != 0 -> the reference ASSIGNS dmax = 1.0, which
      // for the test 1 + dmax == 1 is equivalent to max(dmax, 1.0)
      const double quot = fabs(alpha * pp.v[u] / xx.v[u]);
      const double ddum = (xx.v[u] != 0.0) ? quot : ((pp.v[u] != 0.0) ? 1.0 : 0.0);
      dmax = (ddum > dmax) ? ddum : dmax;
      // daxpy (pcg.c:141) returns without touching y when the scalar is zero (netlib /
      // OpenBLAS quick return), which matters when p holds inf/NaN
      if (upd) xx.v[u] = xx.v[u] + alpha * pp.v[u];
    }
    st<V>(x, i, xx);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(dmax, off, 64);
    if (o > dmax) dmax = o;
  }
  double acc[1];
  acc[0] = ((threadIdx.x & 63) == 0 && (1.0 + dmax != 1.0)) ? 1.0 : 0.0;
  block_reduce_store<1>(acc, partials + 2 * (size_t)kMaxParts);
}

//      r_update: r -= alpha q (:142-143); partial slots 0, 1 = {r.r, r.z}, z = dinv.*r or r
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void r_update_kernel(long n, double alpha,
                                                          const double *__restrict__ q,
                                                          const double *__restrict__ dinv, double dc,
                                                          double *__restrict__ r,
                                                          double *__restrict__ partials,
                                                          const PcgDev *__restrict__ dstate) {
  if (dstate) {
    if (dstate->status) return;
    alpha = dstate->alpha;
  }
  double acc[2] = {0.0, 0.0};
  const double malpha = -alpha;
  const bool upd = alpha != 0.0;
  PSP_VEC_LOOP(i, n) {
    const Pack<V> qq = ld<V>(q, i);
    Pack<V> rr = ld<V>(r, i);
    Pack<V> dd;
    if constexpr (PRE == 1) dd = ld<V>(dinv, i);
    if constexpr (PRE == 2) {
#pragma unroll
      for (int u = 0; u < V; ++u) dd.v[u] = dc;
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      const double t = upd ? rr.v[u] + malpha * qq.v[u] : rr.v[u];
      rr.v[u] = t;
      acc[0] += t * t;
      if constexpr (PRE != 0) {
        const double z = t * dd.v[u];
        acc[1] += t * z;
      }
    }
    st<V>(r, i, rr);
  }
  if constexpr (PRE == 0) acc[1] = acc[0];
  block_reduce_store<2>(acc, partials);
}

// ---- Jacobi: y = x.*dinv (preconmodule.c:41-42)
template <int V>
__global__ __launch_bounds__(kBlock) void jacobi_first_kernel(long n, const double *__restrict__ x,
                                                              const double *__restrict__ dinv,
                                                              double *__restrict__ y) {
  PSP_VEC_LOOP(i, n) {
    Pack<V> a = ld<V>(x, i);
    const Pack<V> d = ld<V>(dinv, i);
#pragma unroll
    for (int u = 0; u < V; ++u) a.v[u] = a.v[u] * d.v[u];
    st<V>(y, i, a);
  }
}

// ---- Jacobi sweep: y = (x - y).*dinv + temp (preconmodule.c:50-51)
template <int V>
__global__ __launch_bounds__(kBlock) void jacobi_sweep_kernel(long n, const double *__restrict__ x,
                                                              const double *__restrict__ dinv,
                                                              const double *__restrict__ temp,
                                                              double *__restrict__ y) {
  PSP_VEC_LOOP(i, n) {
    const Pack<V> a = ld<V>(x, i), d = ld<V>(dinv, i), t = ld<V>(temp, i);
    Pack<V> yy = ld<V>(y, i);
#pragma unroll
    for (int u = 0; u < V; ++u) yy.v[u] = (a.v[u] - yy.v[u]) * d.v[u] + t.v[u];
    st<V>(y, i, yy);
  }
}

// ---- dinv = omega / diag with the singularity test 1 + d == 1 (preconmodule.c:395-400);
//      partial = number of singular entries
template <int V>
__global__ __launch_bounds__(kBlock) void dinv_kernel(long n, const double *__restrict__ diag,
                                                      double omega, double *__restrict__ dinv,
                                                      double *__restrict__ partials) {
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    Pack<V> d = ld<V>(diag, i);
#pragma unroll
    for (int u = 0; u < V; ++u) {
      if (1.0 + d.v[u] == 1.0) acc[0] += 1.0;
      d.v[u] = omega / d.v[u];
    }
    st<V>(dinv, i, d);
  }
  block_reduce_store<1>(acc, partials);
}

// ---- MINRES: v = y / beta (minres.c:123-124)
template <int V>
__global__ __launch_bounds__(kBlock) void scale_div_kernel(long n, const double *__restrict__ y,
                                                           double beta, double *__restrict__ v,
                                                           const MinresDev *__restrict__ ds) {
  if (ds) {  // asynchronous loop: scalars live on the device
    if (ds->skip) return;
    beta = ds->beta;
  }
  PSP_VEC_LOOP(i, n) {
    Pack<V> a = ld<V>(y, i);
#pragma unroll
    for (int u = 0; u < V; ++u) a.v[u] = a.v[u] / beta;
    st<V>(v, i, a);
  }
}

// ---- MINRES Lanczos update (minres.c:131-143):
//      t = v_hat; v_hat = av - c1*v_hat - c2*v_hat_old; v_hat_old = t;
//      y = dinv.*v_hat (PRE && y != nullptr) ; partial {v_hat . y}
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void lanczos_kernel(
    long n, const double *__restrict__ av, double c1, double c2, const double *__restrict__ v_hat,
    double *__restrict__ v_hat_old, const double *__restrict__ dinv, double dc, double *__restrict__ y,
    double *__restrict__ partials, const MinresDev *__restrict__ ds) {
  if (ds) {
    if (ds->skip) return;
    c1 = ds->c1;
    c2 = ds->c2;
  }
  // The new v_hat is written over v_hat_old (its old value is consumed here) and the caller swaps the
  // two names: "v_hat_old = old v_hat" (minres.c:125,135) then costs no store.  PRE as in residual_kernel.
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> a = ld<V>(av, i);
    const Pack<V> vh = ld<V>(v_hat, i);
    Pack<V> vo = ld<V>(v_hat_old, i);
    Pack<V> dd, yy;
    if constexpr (PRE == 1) dd = ld<V>(dinv, i);
    if constexpr (PRE == 2) {
#pragma unroll
      for (int u = 0; u < V; ++u) dd.v[u] = dc;
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      const double nv = a.v[u] - c1 * vh.v[u] - c2 * vo.v[u];
      vo.v[u] = nv;
      if constexpr (PRE != 0) {
        yy.v[u] = nv * dd.v[u];
        acc[0] += nv * yy.v[u];
      } else {
        acc[0] += nv * nv;
      }
    }
    st<V>(v_hat_old, i, vo);
    if constexpr (PRE != 0) st<V>(y, i, yy);
  }
  block_reduce_store<1>(acc, partials);
}

// ---- same recurrence without the preconditioner/dot (generic path: K applied afterwards)
template <int V>
__global__ __launch_bounds__(kBlock) void lanczos_plain_kernel(long n, const double *__restrict__ av,
                                                               double c1, double c2,
                                                               const double *__restrict__ v_hat,
                                                               double *__restrict__ v_hat_old) {
  PSP_VEC_LOOP(i, n) {  // new v_hat over v_hat_old; the caller swaps the names
    const Pack<V> a = ld<V>(av, i);
    const Pack<V> vh = ld<V>(v_hat, i);
    Pack<V> vo = ld<V>(v_hat_old, i);
#pragma unroll
    for (int u = 0; u < V; ++u) vo.v[u] = a.v[u] - c1 * vh.v[u] - c2 * vo.v[u];
    st<V>(v_hat_old, i, vo);
  }
}

// ---- MINRES update (minres.c:172-180): tmp = w; w = (v - r3*w_old - r2*tmp)/r1;
//      w_old = tmp; x += c_eta*w.  The new w is written over w_old and the caller swaps the names.
// VNEXT (round 5; unscaled products only -- csr_spmv_w3 / w2 operators): the NEXT iteration's v = y / beta (minres.c:123-124,
// scale_div_kernel's division; y and beta are final by now) is written over v in the same pass: one launch less per
// iteration where every launch sits on the ~5 us floor (FEM stand-in, n = 9.3e5: profiles/r5_fem_minres.txt).
template <int V, bool SCALED, bool VNEXT = false>
__global__ __launch_bounds__(kBlock) void minres_wx_kernel(long n, double *__restrict__ v,
                                                           double vdiv, double r1, double r2, double r3,
                                                           double c_eta, const double *__restrict__ w,
                                                           double *__restrict__ w_old,
                                                           double *__restrict__ x,
                                                           const MinresDev *__restrict__ ds,
                                                           const double *__restrict__ ynext = nullptr) {
  double beta_next = 1.0;
  if (ds) {  // the update of the running iteration is still due when only `stop` is set
    if (ds->status) return;
    vdiv = ds->beta_old;  // beta at the start of this iteration (the scalar step has moved on)
    r1 = ds->r1;
    r2 = ds->r2;
    r3 = ds->r3;
    c_eta = ds->c_eta;
    beta_next = ds->beta;
  }
  // SCALED: v holds the unnormalised Lanczos vector and v / vdiv is formed here (minres.c:123-124)
  PSP_VEC_LOOP(i, n) {
    Pack<V> vv = ld<V>(v, i);
    const Pack<V> ww = ld<V>(w, i);
    Pack<V> wo = ld<V>(w_old, i), xx = ld<V>(x, i);
#pragma unroll
    for (int u = 0; u < V; ++u) {
      if constexpr (SCALED) vv.v[u] = vv.v[u] / vdiv;
      const double nw = (vv.v[u] - r3 * wo.v[u] - r2 * ww.v[u]) / r1;
      wo.v[u] = nw;
      xx.v[u] += c_eta * nw;
    }
    st<V>(w_old, i, wo);
    st<V>(x, i, xx);
    if constexpr (VNEXT) {
      Pack<V> yn = ld<V>(ynext, i);
#pragma unroll
      for (int u = 0; u < V; ++u) yn.v[u] = yn.v[u] / beta_next;
      st<V>(v, i, yn);
    }
  }
}

// ---- z = a*x + b*y (separate roundings, as the reference's daxpy / hand loops); z may
//      alias x or y.  Building block of the unfused cgs / bicgstab / qmrs / gmres loops.
template <int V>
__global__ __launch_bounds__(kBlock) void lin2_kernel(long n, double a, const double *x, double b,
                                                      const double *y, double *z) {
  PSP_VEC_LOOP(i, n) {
    const Pack<V> xx = ld<V>(x, i), yy = ld<V>(y, i);
    Pack<V> zz;
#pragma unroll
    for (int u = 0; u < V; ++u) zz.v[u] = a * xx.v[u] + b * yy.v[u];
    st<V>(z, i, zz);
  }
}

// ---- fused passes of the BiCGSTAB loop (bicgstab.c; native matrix + None / jacobi(1)).  Element by element the
// same rounded operations, in the same order, as the unfused sequence of lin2 / jacobi kernels they replace.
//   p = r + beta*(p - omega*v)  (first iteration: p = r);  phat = K p            [PRE 0: phat is p itself]
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void bicg_p_kernel(long n, const double *__restrict__ r,
                                                        const double *__restrict__ v, double *__restrict__ p,
                                                        double *__restrict__ phat, const double *__restrict__ dinv,
                                                        double dc, double beta, double omega, int first, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    beta = ka.S->r[ka.i0];
    omega = ka.S->r[ka.i1];
    first = ka.S->iter == 1;
  }
  PSP_VEC_LOOP(i, n) {
    Pack<V> pp = ld<V>(r, i);
    if (!first) {
      const Pack<V> po = ld<V>(p, i), vv = ld<V>(v, i);
#pragma unroll
      for (int u = 0; u < V; ++u) {
        const double t = 1.0 * po.v[u] + (-omega) * vv.v[u];  // lin2(1, p, -omega, v)
        pp.v[u] = 1.0 * pp.v[u] + beta * t;                   // lin2(1, r, beta, t)
      }
    }
    st<V>(p, i, pp);
    if constexpr (PRE != 0) {
      Pack<V> ph;
      if constexpr (PRE == 1) {
        const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
        for (int u = 0; u < V; ++u) ph.v[u] = pp.v[u] * dd.v[u];
      } else {
#pragma unroll
        for (int u = 0; u < V; ++u) ph.v[u] = pp.v[u] * dc;
      }
      st<V>(phat, i, ph);
    }
  }
}

//   s = r - alpha*v;  shat = K s                                                 [PRE 0: shat is s itself]
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void bicg_s_kernel(long n, const double *__restrict__ r,
                                                        const double *__restrict__ v, double *__restrict__ sv,
                                                        double *__restrict__ shat, const double *__restrict__ dinv,
                                                        double dc, double alpha, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    alpha = ka.S->r[ka.i0];
  }
  PSP_VEC_LOOP(i, n) {
    const Pack<V> rr = ld<V>(r, i), vv = ld<V>(v, i);
    Pack<V> ss;
#pragma unroll
    for (int u = 0; u < V; ++u) ss.v[u] = 1.0 * rr.v[u] + (-alpha) * vv.v[u];
    st<V>(sv, i, ss);
    if constexpr (PRE != 0) {
      Pack<V> sh;
      if constexpr (PRE == 1) {
        const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
        for (int u = 0; u < V; ++u) sh.v[u] = ss.v[u] * dd.v[u];
      } else {
#pragma unroll
        for (int u = 0; u < V; ++u) sh.v[u] = ss.v[u] * dc;
      }
      st<V>(shat, i, sh);
    }
  }
}

//   x = (x + alpha*phat) + omega*shat;  r = s - omega*t;  partials {r.r, rhat.r}
template <int V>
__global__ __launch_bounds__(kBlock) void bicg_xr_kernel(long n, double *__restrict__ x,
                                                         const double *__restrict__ phat,
                                                         const double *__restrict__ shat,
                                                         const double *__restrict__ sv, const double *__restrict__ t,
                                                         double *__restrict__ r, const double *__restrict__ rhat,
                                                         double alpha, double omega, double *__restrict__ partials, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    alpha = ka.S->r[ka.i0];
    omega = ka.S->r[ka.i1];
  }
  double acc[2] = {0.0, 0.0};
  PSP_VEC_LOOP(i, n) {
    Pack<V> xx = ld<V>(x, i);
    const Pack<V> ph = ld<V>(phat, i), sh = ld<V>(shat, i), ss = ld<V>(sv, i), tt = ld<V>(t, i), rh = ld<V>(rhat, i);
    Pack<V> rn;
#pragma unroll
    for (int u = 0; u < V; ++u) {
      double xv = 1.0 * xx.v[u] + alpha * ph.v[u];
      xv = 1.0 * xv + omega * sh.v[u];
      xx.v[u] = xv;
      const double rv = 1.0 * ss.v[u] + (-omega) * tt.v[u];
      rn.v[u] = rv;
      acc[0] += rv * rv;
      acc[1] += rh.v[u] * rv;
    }
    st<V>(x, i, xx);
    st<V>(r, i, rn);
  }
  block_reduce_store<2>(acc, partials);
}

// ---- fused passes of the CGS loop (cgs.c; native matrix + None / jacobi(1)).  Each line is the copy + daxpy pair
// of the reference with daxpy's quick return for a zero coefficient (netlib: the vector is left untouched).
//   q = u - alpha*v;  tmp = u + q;  tmp2 = K tmp;  x = x + alpha*tmp2
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void cgs_q_kernel(long n, const double *__restrict__ u,
                                                       const double *__restrict__ v, double *__restrict__ x,
                                                       double *__restrict__ q, double *__restrict__ tmp2,
                                                       const double *__restrict__ dinv, double dc, double alpha, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    alpha = ka.S->r[ka.i0];
  }
  const bool upd = alpha != 0.0;
  PSP_VEC_LOOP(i, n) {
    const Pack<V> uu = ld<V>(u, i), vv = ld<V>(v, i);
    Pack<V> xx = ld<V>(x, i), qq, t2;
    Pack<V> dd;
    if constexpr (PRE == 1) dd = ld<V>(dinv, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      qq.v[k] = upd ? 1.0 * uu.v[k] + (-alpha) * vv.v[k] : uu.v[k];
      const double t = 1.0 * uu.v[k] + 1.0 * qq.v[k];
      t2.v[k] = PRE == 0 ? t : (PRE == 1 ? t * dd.v[k] : t * dc);
      if (upd) xx.v[k] = 1.0 * xx.v[k] + alpha * t2.v[k];
    }
    st<V>(q, i, qq);
    st<V>(tmp2, i, t2);
    st<V>(x, i, xx);
  }
}

//   r = r - alpha*t;  partials {r.r, r.r0}
template <int V>
__global__ __launch_bounds__(kBlock) void cgs_r_kernel(long n, double *__restrict__ r, const double *__restrict__ t,
                                                       const double *__restrict__ r0, double alpha,
                                                       double *__restrict__ partials, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    alpha = ka.S->r[ka.i0];
  }
  const bool upd = alpha != 0.0;
  double acc[2] = {0.0, 0.0};
  PSP_VEC_LOOP(i, n) {
    Pack<V> rr = ld<V>(r, i);
    const Pack<V> tt = ld<V>(t, i), r0v = ld<V>(r0, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (upd) rr.v[k] = 1.0 * rr.v[k] + (-alpha) * tt.v[k];
      acc[0] += rr.v[k] * rr.v[k];
      acc[1] += rr.v[k] * r0v.v[k];
    }
    st<V>(r, i, rr);
  }
  block_reduce_store<2>(acc, partials);
}

//   u = r + beta*q;  tmp = q + beta*p;  p = u + beta*tmp;  kp = K p                [PRE 0: kp is p itself]
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void cgs_p_kernel(long n, const double *__restrict__ r,
                                                       const double *__restrict__ q, double *__restrict__ p,
                                                       double *__restrict__ u, double *__restrict__ kp,
                                                       const double *__restrict__ dinv, double dc, double beta, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    beta = ka.S->r[ka.i0];
  }
  const bool upd = beta != 0.0;
  PSP_VEC_LOOP(i, n) {
    const Pack<V> rr = ld<V>(r, i), qq = ld<V>(q, i), po = ld<V>(p, i);
    Pack<V> uu, pn;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      uu.v[k] = upd ? 1.0 * rr.v[k] + beta * qq.v[k] : rr.v[k];
      const double t = upd ? 1.0 * qq.v[k] + beta * po.v[k] : qq.v[k];
      pn.v[k] = upd ? 1.0 * uu.v[k] + beta * t : uu.v[k];
    }
    st<V>(u, i, uu);
    st<V>(p, i, pn);
    if constexpr (PRE != 0) {
      Pack<V> kk;
      if constexpr (PRE == 1) {
        const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
        for (int k = 0; k < V; ++k) kk.v[k] = pn.v[k] * dd.v[k];
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) kk.v[k] = pn.v[k] * dc;
      }
      st<V>(kp, i, kk);
    }
  }
}

// ---- fused passes of the QMRS loop (qmrs.c; native matrix + None / jacobi(1)); per element the lin2 / scal / jacobi
// operations they replace.
//   wrk1 = K v1;  partial wrk1.v1                                                  (first iteration only; PRE != 0)
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void qmrs_kv_kernel(long n, const double *__restrict__ v1,
                                                         double *__restrict__ wrk1, const double *__restrict__ dinv,
                                                         double dc, double *__restrict__ partials) {
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> vv = ld<V>(v1, i);
    Pack<V> ww = vv;
    if constexpr (PRE == 1) {
      const Pack<V> dd = ld<V>(dinv, i);
#pragma unroll
      for (int k = 0; k < V; ++k) ww.v[k] = vv.v[k] * dd.v[k];
    }
    if constexpr (PRE == 2) {
#pragma unroll
      for (int k = 0; k < V; ++k) ww.v[k] = vv.v[k] * dc;
    }
    if constexpr (PRE != 0) st<V>(wrk1, i, ww);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[0] += ww.v[k] * vv.v[k];
  }
  block_reduce_store<1>(acc, partials);
}

//   p = v1 - cc*p;  g = wrk1 - cc*g
template <int V>
__global__ __launch_bounds__(kBlock) void qmrs_pg_kernel(long n, const double *__restrict__ v1,
                                                         const double *__restrict__ wrk1, double *__restrict__ p,
                                                         double *__restrict__ g, double cc, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    cc = ka.S->r[ka.i0];
  }
  PSP_VEC_LOOP(i, n) {
    const Pack<V> vv = ld<V>(v1, i), ww = ld<V>(wrk1, i);
    Pack<V> pp = ld<V>(p, i), gg = ld<V>(g, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      pp.v[k] = 1.0 * vv.v[k] + (-cc) * pp.v[k];
      gg.v[k] = 1.0 * ww.v[k] + (-cc) * gg.v[k];
    }
    st<V>(p, i, pp);
    st<V>(g, i, gg);
  }
}

//   v1 = t - beta*v1;  partial v1.v1
template <int V>
__global__ __launch_bounds__(kBlock) void qmrs_v_kernel(long n, const double *__restrict__ t, double *__restrict__ v1,
                                                        double beta, double *__restrict__ partials, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    beta = ka.S->r[ka.i0];
  }
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> tt = ld<V>(t, i);
    Pack<V> vv = ld<V>(v1, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      vv.v[k] = 1.0 * tt.v[k] + (-beta) * vv.v[k];
      acc[0] += vv.v[k] * vv.v[k];
    }
    st<V>(v1, i, vv);
  }
  block_reduce_store<1>(acc, partials);
}

//   d = eta*p + cc*d;  x = x + d;  v1 = rho1inv*v1;  and for the next iteration wrk1 = K v1, partial wrk1.v1
template <int V, int PRE>
__global__ __launch_bounds__(kBlock) void qmrs_dx_kernel(long n, const double *__restrict__ p, double *__restrict__ d,
                                                         double *__restrict__ x, double *__restrict__ v1,
                                                         double *__restrict__ wrk1, const double *__restrict__ dinv,
                                                         double dc, double eta, double cc, double rho1inv,
                                                         double *__restrict__ partials, KryArg ka) {
  if (ka.S) {
    if (ka.S->status) return;
    eta = ka.S->r[ka.i0];
    cc = ka.S->r[ka.i1];
    rho1inv = ka.S->r[ka.i2];
  }
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    const Pack<V> pp = ld<V>(p, i);
    Pack<V> dd = ld<V>(d, i), xx = ld<V>(x, i), vv = ld<V>(v1, i), ww;
    Pack<V> di;
    if constexpr (PRE == 1) di = ld<V>(dinv, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      dd.v[k] = eta * pp.v[k] + cc * dd.v[k];
      xx.v[k] = 1.0 * xx.v[k] + 1.0 * dd.v[k];
      vv.v[k] = rho1inv * vv.v[k];
      ww.v[k] = PRE == 0 ? vv.v[k] : (PRE == 1 ? vv.v[k] * di.v[k] : vv.v[k] * dc);
      acc[0] += ww.v[k] * vv.v[k];
    }
    st<V>(d, i, dd);
    st<V>(x, i, xx);
    st<V>(v1, i, vv);
    if constexpr (PRE != 0) st<V>(wrk1, i, ww);
  }
  block_reduce_store<1>(acc, partials);
}

// ---- y = y + a*x (daxpy, quick return for a == 0) and, in the same pass, the partial sums of y.z (z == nullptr: y.y):
// the modified Gram-Schmidt loop of gmres.c -- every dot needs the vector the previous axpy produced, so the pair
// "axpy k, dot k + 1" is one pass instead of two
template <int V, bool SELF>
__global__ __launch_bounds__(kBlock) void axpy_dot_kernel(long n, double a, const double *__restrict__ x,
                                                          double *__restrict__ y, const double *__restrict__ z,
                                                          double *__restrict__ partials,
                                                          const double *__restrict__ neg_a_dev) {
  // neg_a_dev (round 4): the coefficient is minus a value a previous reduction left on the device (the h of the
  // Gram-Schmidt step before) -- the chain dot -> axpy -> dot then needs no host round trip
  if (neg_a_dev) a = -*neg_a_dev;
  const bool upd = a != 0.0;
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    Pack<V> yy = ld<V>(y, i);
    if (upd) {
      const Pack<V> xx = ld<V>(x, i);
#pragma unroll
      for (int k = 0; k < V; ++k) yy.v[k] = 1.0 * yy.v[k] + a * xx.v[k];
      st<V>(y, i, yy);
    }
    if constexpr (SELF) {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[0] += yy.v[k] * yy.v[k];
    } else {
      const Pack<V> zz = ld<V>(z, i);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[0] += yy.v[k] * zz.v[k];
    }
  }
  block_reduce_store<1>(acc, partials);
}

// ---- the same step with the reduction that yields its coefficient folded in (round 5; gmres.c:110-116, modified
// Gram-Schmidt): every workgroup first adds the partial sums the step BEFORE left (prev[0 .. np_prev): reduce_block, the
// finishing block's own routine -- its result does not depend on how many waves run it, so these are the bits a finishing
// launch would have stored), takes a = -h, and workgroup 0 stores h where the host will read the column's coefficients.
// One launch per Gram-Schmidt step instead of two; the partial sums ping-pong between two arrays (a workgroup may still be
// adding the previous ones while another already stores its own).  np_prev <= 4096 (16 groups of 256).
template <int V, bool SELF>
__global__ __launch_bounds__(kBlock) void axpy_dot_chain_kernel(long n, const double *__restrict__ prev, int np_prev,
                                                                double *__restrict__ h_out, const double *__restrict__ x,
                                                                double *__restrict__ y, const double *__restrict__ z,
                                                                double *__restrict__ partials) {
  __shared__ double gsum[16];
  __shared__ double hsh;
  reduce_block(prev, np_prev, 1, 0, true, &hsh, gsum);  // (ends with a workgroup barrier)
  const double h = hsh;
  if (blockIdx.x == 0 && threadIdx.x == 0) *h_out = h;
  const double a = -h;
  const bool upd = a != 0.0;
  double acc[1] = {0.0};
  PSP_VEC_LOOP(i, n) {
    Pack<V> yy = ld<V>(y, i);
    if (upd) {
      const Pack<V> xx = ld<V>(x, i);
#pragma unroll
      for (int k = 0; k < V; ++k) yy.v[k] = 1.0 * yy.v[k] + a * xx.v[k];
      st<V>(y, i, yy);
    }
    if constexpr (SELF) {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[0] += yy.v[k] * yy.v[k];
    } else {
      const Pack<V> zz = ld<V>(z, i);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[0] += yy.v[k] * zz.v[k];
    }
  }
  block_reduce_store<1>(acc, partials);
}

// ---- x = a*x (dscal)
template <int V>
__global__ __launch_bounds__(kBlock) void scal_kernel(long n, double a, double *x) {
  PSP_VEC_LOOP(i, n) {
    Pack<V> xx = ld<V>(x, i);
#pragma unroll
    for (int u = 0; u < V; ++u) xx.v[u] = a * xx.v[u];
    st<V>(x, i, xx);
  }
}

// ---- halo packing
__global__ void gather_kernel(int count, const int *__restrict__ idx, const double *__restrict__ v,
                              double *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
    out[i] = v[idx[i]];
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <typename... P>
inline bool can_vec2(long n, P... ptrs) {
  return (n % 2 == 0) && (aligned16(ptrs) && ...);
}

}  // namespace

namespace {

__global__ void not_constant_kernel(long n, const double *__restrict__ v, int *__restrict__ flag) {
  const double c = v[0];
  int bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    bad |= v[i] != c;  // NaN != NaN: a vector holding NaN is never "constant"
  if (bad) atomicOr(flag, 1);
}

// device vectors known to hold one value everywhere (Jacobi dinv of a constant-diagonal operator):
// pointer -> (length, value).  Entries are added by dinv_register and removed by dinv_unregister;
// the owner guarantees the vector does not change in between.
struct ConstVec {
  long n;
  double c;
};
std::unordered_map<const double *, ConstVec> g_const;
std::mutex g_const_mu;

}  // namespace

namespace psp {

int dinv_register(const double *v, long n) {
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_DINV_CONST");
    return e && atoi(e) == 0;
  }();
  if (off || !v || n < 1) return PSP_OK;
  int *flag;
  PSP_HIP(hipMalloc((void **)&flag, sizeof(int)));
  PSP_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream()));
  hipLaunchKernelGGL(not_constant_kernel, dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, stream(),
                     n, v, flag);
  PSP_LAUNCH_CHECK();
  int bad = 1;
  double c = 0.0;
  PSP_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipMemcpyAsync(&c, v, sizeof(double), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(flag));
  std::lock_guard<std::mutex> lk(g_const_mu);
  if (bad)
    g_const.erase(v);
  else
    g_const[v] = {n, c};
  return PSP_OK;
}

void dinv_unregister(const double *v) {
  std::lock_guard<std::mutex> lk(g_const_mu);
  g_const.erase(v);
}

bool dinv_constant(const double *v, long n, double *c) {
  std::lock_guard<std::mutex> lk(g_const_mu);
  auto it = g_const.find(v);
  if (it == g_const.end() || it->second.n != n) return false;
  *c = it->second.c;
  return true;
}

// Launch helpers (device pointers, library stream).  `partials` = slot base in the workspace.

int k_dot(long n, const double *x, const double *y, double *partials, int *nparts) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x, y))
    hipLaunchKernelGGL(dot_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, x, y, partials);
  else
    hipLaunchKernelGGL(dot_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, x, y, partials);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_residual(long n, const double *b, double *r, const double *dinv, double *partials,
               int *nparts) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, b, r, dinv) : can_vec2(n, b, r);
#define L(V, PRE)                                                                              \
  hipLaunchKernelGGL((residual_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, b, r, \
                     dinv, dc, partials)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_pupdate(long n, const double *r, const double *dinv, double beta, bool first, double *p,
              const PcgDev *dstate) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, r, p, dinv) : can_vec2(n, r, p);
#define L(V, PRE, FIRST)                                                                  \
  hipLaunchKernelGGL((pupdate_kernel<V, PRE, FIRST>), dim3(grid), dim3(kBlock), 0, stream(), n, \
                     r, dinv, dc, beta, p, dstate)
  if (cst) {
    if (first) { if (v2) L(2, 2, true); else L(1, 2, true); }
    else { if (v2) L(2, 2, false); else L(1, 2, false); }
  } else if (dinv) {
    if (first) { if (v2) L(2, 1, true); else L(1, 1, true); }
    else { if (v2) L(2, 1, false); else L(1, 1, false); }
  } else {
    if (first) { if (v2) L(2, 0, true); else L(1, 0, true); }
    else { if (v2) L(2, 0, false); else L(1, 0, false); }
  }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_px_update(long n, const double *r, const double *dinv, double *p, double *x, double *partials,
                int *nparts, const PcgDev *dstate, double beta, double alpha, bool first, bool xpend) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, r, p, x, dinv) : can_vec2(n, r, p, x);
#define L(V, PRE)                                                                          \
  hipLaunchKernelGGL((px_update_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, r, \
                     dinv, dc, p, x, partials, dstate, beta, alpha, first ? 1 : 0, xpend ? 1 : 0)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

// the two halves of k_xr_update on their own (lazy loop: r update per iteration, x update at the end)
int k_x_update(long n, double alpha, const double *p, double *x, double *partials, int *nparts,
               const PcgDev *dstate) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, p, x))
    hipLaunchKernelGGL(x_update_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, alpha, p, x, partials, dstate);
  else
    hipLaunchKernelGGL(x_update_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, alpha, p, x, partials, dstate);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_r_update(long n, double alpha, const double *q, const double *dinv, double *r, double *partials,
               int *nparts, const PcgDev *dstate) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, q, r, dinv) : can_vec2(n, q, r);
#define L(V, PRE)                                                                         \
  hipLaunchKernelGGL((r_update_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, \
                     alpha, q, dinv, dc, r, partials, dstate)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_xr_update(long n, double alpha, const double *p, const double *q, const double *dinv,
                double *x, double *r, double *partials, int *nparts, const PcgDev *dstate) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, p, x))
    hipLaunchKernelGGL(x_update_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, alpha, p, x,
                       partials, dstate);
  else
    hipLaunchKernelGGL(x_update_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, alpha, p, x,
                       partials, dstate);
  PSP_LAUNCH_CHECK();
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, q, r, dinv) : can_vec2(n, q, r);
#define L(V, PRE)                                                                         \
  hipLaunchKernelGGL((r_update_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, \
                     alpha, q, dinv, dc, r, partials, dstate)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_jacobi_first(long n, const double *x, const double *dinv, double *y) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x, dinv, y))
    hipLaunchKernelGGL(jacobi_first_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, x, dinv, y);
  else
    hipLaunchKernelGGL(jacobi_first_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, x, dinv, y);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_jacobi_sweep(long n, const double *x, const double *dinv, const double *temp, double *y) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x, dinv, temp, y))
    hipLaunchKernelGGL(jacobi_sweep_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, x, dinv,
                       temp, y);
  else
    hipLaunchKernelGGL(jacobi_sweep_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, x, dinv,
                       temp, y);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_dinv(long n, const double *diag, double omega, double *dinv, double *partials, int *nparts) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, diag, dinv))
    hipLaunchKernelGGL(dinv_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, diag, omega, dinv,
                       partials);
  else
    hipLaunchKernelGGL(dinv_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, diag, omega, dinv,
                       partials);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_scale_div(long n, const double *y, double beta, double *v, const MinresDev *ds) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, y, v))
    hipLaunchKernelGGL(scale_div_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, y, beta, v, ds);
  else
    hipLaunchKernelGGL(scale_div_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, y, beta, v, ds);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_lanczos(long n, const double *av, double c1, double c2, const double *v_hat, double *v_hat_old,
              const double *dinv, double *y, double *partials, int *nparts, const MinresDev *ds) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, av, v_hat, v_hat_old, y) : can_vec2(n, av, v_hat, v_hat_old, dinv, y))
                       : can_vec2(n, av, v_hat, v_hat_old);
#define L(V, PRE)                                                                              \
  hipLaunchKernelGGL((lanczos_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, av, c1, \
                     c2, v_hat, v_hat_old, dinv, dc, y, partials, ds)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_lanczos_plain(long n, const double *av, double c1, double c2, const double *v_hat,
                    double *v_hat_old) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, av, v_hat, v_hat_old))
    hipLaunchKernelGGL(lanczos_plain_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, av, c1,
                       c2, v_hat, v_hat_old);
  else
    hipLaunchKernelGGL(lanczos_plain_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, av, c1,
                       c2, v_hat, v_hat_old);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_lin2(long n, double a, const double *x, double b, const double *y, double *z) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x, y, z))
    hipLaunchKernelGGL(lin2_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, a, x, b, y, z);
  else
    hipLaunchKernelGGL(lin2_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, a, x, b, y, z);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_bicg_p(long n, const double *r, const double *v, double *p, double *phat, const double *dinv, double beta,
             double omega, bool first, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, r, v, p, phat) : can_vec2(n, r, v, p, phat, dinv)) : can_vec2(n, r, v, p);
#define L(V, PRE)                                                                                       \
  hipLaunchKernelGGL((bicg_p_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, r, v, p, phat, \
                     dinv, dc, beta, omega, first ? 1 : 0, ka)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_bicg_s(long n, const double *r, const double *v, double *s, double *shat, const double *dinv, double alpha, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, r, v, s, shat) : can_vec2(n, r, v, s, shat, dinv)) : can_vec2(n, r, v, s);
#define L(V, PRE)                                                                                    \
  hipLaunchKernelGGL((bicg_s_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, r, v, s, shat, \
                     dinv, dc, alpha, ka)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_bicg_xr(long n, double *x, const double *phat, const double *shat, const double *s, const double *t, double *r,
              const double *rhat, double alpha, double omega, double *partials, int *nparts, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x, phat, shat, s, t, r, rhat))
    hipLaunchKernelGGL(bicg_xr_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, x, phat, shat, s, t, r, rhat,
                       alpha, omega, partials, ka);
  else
    hipLaunchKernelGGL(bicg_xr_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, x, phat, shat, s, t, r, rhat,
                       alpha, omega, partials, ka);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_cgs_q(long n, const double *u, const double *v, double *x, double *q, double *tmp2, const double *dinv,
            double alpha, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv && !cst ? can_vec2(n, u, v, x, q, tmp2, dinv) : can_vec2(n, u, v, x, q, tmp2);
#define L(V, PRE)                                                                                       \
  hipLaunchKernelGGL((cgs_q_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, u, v, x, q, tmp2, \
                     dinv, dc, alpha, ka)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_cgs_r(long n, double *r, const double *t, const double *r0, double alpha, double *partials, int *nparts, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, r, t, r0))
    hipLaunchKernelGGL(cgs_r_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, r, t, r0, alpha, partials, ka);
  else
    hipLaunchKernelGGL(cgs_r_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, r, t, r0, alpha, partials, ka);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_cgs_p(long n, const double *r, const double *q, double *p, double *u, double *kp, const double *dinv,
            double beta, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, r, q, p, u, kp) : can_vec2(n, r, q, p, u, kp, dinv)) : can_vec2(n, r, q, p, u);
#define L(V, PRE)                                                                                    \
  hipLaunchKernelGGL((cgs_p_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, r, q, p, u, kp, \
                     dinv, dc, beta, ka)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_qmrs_kv(long n, const double *v1, double *wrk1, const double *dinv, double *partials, int *nparts) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, v1, wrk1) : can_vec2(n, v1, wrk1, dinv)) : can_vec2(n, v1);
#define L(V, PRE)                                                                                      \
  hipLaunchKernelGGL((qmrs_kv_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, v1, wrk1, dinv, \
                     dc, partials)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_qmrs_pg(long n, const double *v1, const double *wrk1, double *p, double *g, double cc, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, v1, wrk1, p, g))
    hipLaunchKernelGGL(qmrs_pg_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, v1, wrk1, p, g, cc, ka);
  else
    hipLaunchKernelGGL(qmrs_pg_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, v1, wrk1, p, g, cc, ka);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_qmrs_v(long n, const double *t, double *v1, double beta, double *partials, int *nparts, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, t, v1))
    hipLaunchKernelGGL(qmrs_v_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, t, v1, beta, partials, ka);
  else
    hipLaunchKernelGGL(qmrs_v_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, t, v1, beta, partials, ka);
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_qmrs_dx(long n, const double *p, double *d, double *x, double *v1, double *wrk1, const double *dinv, double eta,
              double cc, double rho1inv, double *partials, int *nparts, KryArg ka) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  double dc = 0.0;
  const bool cst = dinv && dinv_constant(dinv, n, &dc);
  const bool v2 = dinv ? (cst ? can_vec2(n, p, d, x, v1, wrk1) : can_vec2(n, p, d, x, v1, wrk1, dinv))
                       : can_vec2(n, p, d, x, v1);
#define L(V, PRE)                                                                                          \
  hipLaunchKernelGGL((qmrs_dx_kernel<V, PRE>), dim3(grid), dim3(kBlock), 0, stream(), n, p, d, x, v1, wrk1, \
                     dinv, dc, eta, cc, rho1inv, partials, ka)
  if (cst) { if (v2) L(2, 2); else L(1, 2); }
  else if (dinv) { if (v2) L(2, 1); else L(1, 1); }
  else { if (v2) L(2, 0); else L(1, 0); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_axpy_dot(long n, double a, const double *x, double *y, const double *z, double *partials, int *nparts,
               const double *neg_a_dev) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  const bool v2 = z ? can_vec2(n, x, y, z) : can_vec2(n, x, y);
#define L(V, SELF)                                                                                      \
  hipLaunchKernelGGL((axpy_dot_kernel<V, SELF>), dim3(grid), dim3(kBlock), 0, stream(), n, a, x, y, z, partials, neg_a_dev)
  if (z) { if (v2) L(2, false); else L(1, false); }
  else { if (v2) L(2, true); else L(1, true); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_axpy_dot_chain(long n, const double *prev, int np_prev, double *h_out, const double *x, double *y, const double *z,
                     double *partials, int *nparts) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  if (np_prev < 1 || np_prev > 4096) return fail(PSP_EINVAL, "k_axpy_dot_chain: %d partial sums", np_prev);
  const int grid = vec_grid(*w, n);
  const bool v2 = z ? can_vec2(n, x, y, z) : can_vec2(n, x, y);
#define L(V, SELF)                                                                                                 \
  hipLaunchKernelGGL((axpy_dot_chain_kernel<V, SELF>), dim3(grid), dim3(kBlock), 0, stream(), n, prev, np_prev, h_out, x, y, \
                     z, partials)
  if (z) { if (v2) L(2, false); else L(1, false); }
  else { if (v2) L(2, true); else L(1, true); }
#undef L
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  return PSP_OK;
}

int k_scal(long n, double a, double *x) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, x))
    hipLaunchKernelGGL(scal_kernel<2>, dim3(grid), dim3(kBlock), 0, stream(), n, a, x);
  else
    hipLaunchKernelGGL(scal_kernel<1>, dim3(grid), dim3(kBlock), 0, stream(), n, a, x);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int k_minres_wx(long n, const double *v, double r1, double r2, double r3, double c_eta, const double *w_,
                double *w_old, double *x, bool scaled, double vdiv, const MinresDev *ds) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  const bool v2 = can_vec2(n, v, w_, w_old, x);
  double *vm = const_cast<double *>(v);  // (only the VNEXT form writes v)
#define L(V, S)                                                                                   \
  hipLaunchKernelGGL((minres_wx_kernel<V, S>), dim3(grid), dim3(kBlock), 0, stream(), n, vm, vdiv, r1, \
                     r2, r3, c_eta, w_, w_old, x, ds, (const double *)nullptr)
  if (scaled) { if (v2) L(2, true); else L(1, true); }
  else { if (v2) L(2, false); else L(1, false); }
#undef L
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// the w / x update of the running iteration AND v = ynext / beta of the next one (device-resident scalars, unscaled products)
int k_minres_wx_vnext(long n, double *v, const double *ynext, const double *w_, double *w_old, double *x,
                      const MinresDev *ds) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = vec_grid(*w, n);
  if (can_vec2(n, v, w_, w_old, x) && can_vec2(n, ynext, v))
    hipLaunchKernelGGL((minres_wx_kernel<2, false, true>), dim3(grid), dim3(kBlock), 0, stream(), n, v, 1.0, 0.0, 0.0, 0.0,
                       0.0, w_, w_old, x, ds, ynext);
  else
    hipLaunchKernelGGL((minres_wx_kernel<1, false, true>), dim3(grid), dim3(kBlock), 0, stream(), n, v, 1.0, 0.0, 0.0, 0.0,
                       0.0, w_, w_old, x, ds, ynext);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

}  // namespace psp

// ------------------------------------------------------------------ C ABI: phase kernels

extern "C" {

int psp_k_dot(int n, const double *x_dev, const double *y_dev, double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_dot(n, x_dev, y_dev, w->partials, &np));
  return finish_partials(w->partials, np, 1, out_dev);
}

int psp_k_hint_constant(const double *v_dev, int n) {
  PSP_API_GUARD;
  if (!v_dev || n < 1) return fail(PSP_EINVAL, "psp_k_hint_constant: bad argument");
  return dinv_register(v_dev, n);
}

int psp_k_unhint(const double *v_dev) {
  PSP_API_GUARD;
  dinv_unregister(v_dev);
  return PSP_OK;
}

int psp_k_residual(int n, const double *b_dev, double *r_dev, const double *dinv_dev,
                   double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_residual(n, b_dev, r_dev, dinv_dev, w->partials, &np));
  return finish_partials(w->partials, np, 2, out_dev);
}

int psp_k_pupdate(int n, const double *r_dev, const double *dinv_dev, double beta, int first,
                  double *p_dev) {
  PSP_API_GUARD;
  return k_pupdate(n, r_dev, dinv_dev, beta, first != 0, p_dev, nullptr);
}

int psp_k_csr_matvec_dot(psp_csr_t *A, const double *p_dev, int p_offset, double *q_dev,
                         double *out_dev) {
  PSP_API_GUARD_H(A);
  if (!A || !p_dev || !q_dev || !out_dev) return fail(PSP_EINVAL, "psp_k_csr_matvec_dot: NULL");
  if (p_offset < 0 || p_offset + A->nrows > A->ncols)
    return fail(PSP_EINVAL, "psp_k_csr_matvec_dot: owned rows do not fit the column space");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np = 0;
  if (A->nrows == 0) {
    PSP_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), stream()));
    return PSP_OK;
  }
  PSP_TRY(csr_spmv_launch(A, p_dev, q_dev, p_dev + p_offset, w->partials, &np));
  return finish_partials(w->partials, np, 1, out_dev);
}

int psp_k_csr_matvec_overlap(psp_csr_t *A, const double *x_dev, int x_offset, double *y_dev,
                             int row_a, int row_b, psp_wait_fn wait, void *ctx, double *dot_out_dev) {
  PSP_API_GUARD_H(A);
  if (!A || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_k_csr_matvec_overlap: NULL");
  if (x_offset < 0 || x_offset + A->nrows > A->ncols || row_a < 0 || row_b > A->nrows)
    return fail(PSP_EINVAL, "psp_k_csr_matvec_overlap: row range / offset out of bounds");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np = 0;
  if (A->nrows == 0) {
    if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
    if (dot_out_dev) PSP_HIP(hipMemsetAsync(dot_out_dev, 0, sizeof(double), stream()));
    return PSP_OK;
  }
  PSP_TRY(csr_spmv_overlap(A, x_dev, y_dev, dot_out_dev ? x_dev + x_offset : nullptr,
                           dot_out_dev ? w->partials : nullptr, &np, row_a, row_b, wait, ctx));
  if (dot_out_dev) return finish_partials(w->partials, np, 1, dot_out_dev);
  return PSP_OK;
}

int psp_k_px_update(int n, const double *r_dev, const double *dinv_dev, double beta, int first, double alpha_x,
                    int xpend, double *p_dev, double *x_dev, double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_px_update(n, r_dev, dinv_dev, p_dev, x_dev, w->partials, &np, nullptr, beta, alpha_x, first != 0,
                      xpend != 0));
  return finish_partials(w->partials + 2 * (size_t)kMaxParts, np, 1, out_dev);
}

int psp_k_r_update(int n, double alpha, const double *q_dev, const double *dinv_dev, double *r_dev,
                   double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_r_update(n, alpha, q_dev, dinv_dev, r_dev, w->partials, &np, nullptr));
  return finish_partials(w->partials, np, 2, out_dev);
}

int psp_k_x_update(int n, double alpha, const double *p_dev, double *x_dev, double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_x_update(n, alpha, p_dev, x_dev, w->partials, &np, nullptr));
  return finish_partials(w->partials + 2 * (size_t)kMaxParts, np, 1, out_dev);
}

int psp_k_xr_update(int n, double alpha, const double *p_dev, const double *q_dev,
                    const double *dinv_dev, double *x_dev, double *r_dev, double *out_dev) {
  PSP_API_GUARD;
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_xr_update(n, alpha, p_dev, q_dev, dinv_dev, x_dev, r_dev, w->partials, &np, nullptr));
  return finish_partials(w->partials, np, 3, out_dev);
}

int psp_k_jacobi(int n, const double *x_dev, const double *dinv_dev, double *y_dev) {
  PSP_API_GUARD;
  if (!x_dev || !dinv_dev || !y_dev || n < 0) return fail(PSP_EINVAL, "psp_k_jacobi: bad argument");
  if (n == 0) return PSP_OK;
  return k_jacobi_first(n, x_dev, dinv_dev, y_dev);
}

int psp_k_gather(int count, const int *idx_dev, const double *v_dev, double *send_dev) {
  PSP_API_GUARD;
  PSP_TRY(ensure_device());
  if (count <= 0) return PSP_OK;
  int grid = (count + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, stream(), count, idx_dev, v_dev,
                     send_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

}  // extern "C"
