// psp_solvers.hip -- operator protocol, Jacobi, and the device-resident PCG / MINRES loops.
//
// Control flow (scalars, exit tests, info codes) follows the reference line by line:
//   PCG     pysparse/itsolvers/src/pcg.c:22-171
//   MINRES  pysparse/itsolvers/src/minres.c:43-200
//   Jacobi  pysparse/precon/src/preconmodule.c:35-54, 352-412
// All n-vectors live in HBM for the whole solve.  Per iteration the host reads back the
// 1-3 reduced scalars it needs to take the reference's branches (alpha, beta, exit tests)
// in IEEE double exactly as the C code does; nothing else crosses PCIe.
//
// Two inner loops share the kernels of psp_vec.hip:
//   fused    A is a native csr/sss matrix and K is None or jacobi(steps=1):
//            z = dinv.*r is never stored; p.q is an epilogue of the SpMV; x/r update,
//            stagnation scan, ||r||^2 and the next rho are one pass.
//            HBM traffic per iteration: 12 nnz + 108 n bytes.
//   generic  any other operator pair (host callbacks = user-defined Python matvec/precon,
//            jacobi with steps > 1): same kernels with an explicit z vector; callback
//            operators are bridged with one D2H + one H2D copy per application.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <utility>
#include <vector>

#include "psp_internal.h"

using namespace psp;

namespace psp {
int k_dot(long n, const double *x, const double *y, double *partials, int *nparts);
int k_residual(long n, const double *b, double *r, const double *dinv, double *partials, int *nparts);
int k_pupdate(long n, const double *r, const double *dinv, double beta, bool first, double *p,
              const PcgDev *st = nullptr);
int k_px_update(long n, const double *r, const double *dinv, double *p, double *x, double *partials,
                int *nparts, const PcgDev *dstate, double beta = 0.0, double alpha = 0.0, bool first = false,
                bool xpend = false);
int k_x_update(long n, double alpha, const double *p, double *x, double *partials, int *nparts,
               const PcgDev *dstate);
int k_r_update(long n, double alpha, const double *q, const double *dinv, double *r, double *partials,
               int *nparts, const PcgDev *dstate);
int k_xr_update(long n, double alpha, const double *p, const double *q, const double *dinv,
                double *x, double *r, double *partials, int *nparts, const PcgDev *st = nullptr);
int k_jacobi_first(long n, const double *x, const double *dinv, double *y);
int k_jacobi_sweep(long n, const double *x, const double *dinv, const double *temp, double *y);
int k_dinv(long n, const double *diag, double omega, double *dinv, double *partials, int *nparts);
int k_scale_div(long n, const double *y, double beta, double *v, const MinresDev *ds = nullptr);
int k_lanczos(long n, const double *av, double c1, double c2, const double *v_hat, double *v_hat_old,
              const double *dinv, double *y, double *partials, int *nparts, const MinresDev *ds = nullptr);
int k_lanczos_plain(long n, const double *av, double c1, double c2, const double *v_hat, double *v_hat_old);
int k_minres_wx(long n, const double *v, double r1, double r2, double r3, double c_eta, const double *w,
                double *w_old, double *x, bool scaled = false, double vdiv = 1.0, const MinresDev *ds = nullptr);
int k_minres_wx_vnext(long n, double *v, const double *ynext, const double *w, double *w_old, double *x,
                      const MinresDev *ds);
int k_lin2(long n, double a, const double *x, double b, const double *y, double *z);
int k_scal(long n, double a, double *x);
int k_axpy_dot(long n, double a, const double *x, double *y, const double *z, double *partials, int *nparts,
               const double *neg_a_dev = nullptr);
int k_axpy_dot_chain(long n, const double *prev, int np_prev, double *h_out, const double *x, double *y, const double *z,
                     double *partials, int *nparts);
int k_qmrs_kv(long n, const double *v1, double *wrk1, const double *dinv, double *partials, int *nparts);
int k_qmrs_pg(long n, const double *v1, const double *wrk1, double *p, double *g, double cc, KryArg ka = KryArg());
int k_qmrs_v(long n, const double *t, double *v1, double beta, double *partials, int *nparts, KryArg ka = KryArg());
int k_qmrs_dx(long n, const double *p, double *d, double *x, double *v1, double *wrk1, const double *dinv, double eta,
              double cc, double rho1inv, double *partials, int *nparts, KryArg ka = KryArg());
int k_cgs_q(long n, const double *u, const double *v, double *x, double *q, double *tmp2, const double *dinv,
            double alpha, KryArg ka = KryArg());
int k_cgs_r(long n, double *r, const double *t, const double *r0, double alpha, double *partials, int *nparts,
            KryArg ka = KryArg());
int k_cgs_p(long n, const double *r, const double *q, double *p, double *u, double *kp, const double *dinv,
            double beta, KryArg ka = KryArg());
int k_bicg_p(long n, const double *r, const double *v, double *p, double *phat, const double *dinv, double beta,
             double omega, bool first, KryArg ka = KryArg());
int k_bicg_s(long n, const double *r, const double *v, double *s, double *shat, const double *dinv, double alpha,
             KryArg ka = KryArg());
int k_bicg_xr(long n, double *x, const double *phat, const double *shat, const double *s, const double *t, double *r,
              const double *rhat, double alpha, double omega, double *partials, int *nparts, KryArg ka = KryArg());
}  // namespace psp

extern "C" int psp_csr_diagonal_dev(const psp_csr_t *A, double *diag_dev);

namespace {

// Solver work vectors come from a small per-process pool: hipMalloc/hipFree of GB-sized
// vectors costs milliseconds each and would otherwise dominate short solves (measured:
// ~3.5 ms per iteration amortised over 40 iterations at n = 2^27).  psp_trim() empties it.
struct ScratchPool {
  struct Item {
    double *p;
    size_t cap;
    int dev;   // the device the vector lives on (threads may be on different devices)
    int role;  // 0 ordinary; 1 / 2: drawn for the output / input role of a product (psp_place.hip) -- handed out only
               // for that role and that exact length, so that a draw is paid once per (device, length) and process
  };
  std::vector<Item> free_;
  std::mutex mu;
  // up to 64 vectors and a third of the device memory stay cached (gmres(20) holds 41 work vectors: at n = 2^27 their
  // hipMalloc / hipFree per call cost more than the solve); psp_trim() and any failed allocation empty the pool
  static constexpr size_t kMaxCached = 64;
  size_t cached_bytes = 0;
  static size_t cap_bytes() {
    static const size_t cap = [] {
      size_t f = 0, t = 0;
      if (hipMemGetInfo(&f, &t) != hipSuccess) return (size_t)0;
      return t / 3;
    }();
    return cap;
  }
  // a cached vector of this role (role != 0: of exactly this length), or nullptr
  double *take(size_t n, int role) {
    std::lock_guard<std::mutex> lk(mu);
    int best = -1;
    const int dev = current_device();
    for (int i = 0; i < (int)free_.size(); ++i)
      if (free_[i].dev == dev && free_[i].role == role && (role ? free_[i].cap == n : free_[i].cap >= n) &&
          (best < 0 || free_[i].cap < free_[best].cap))
        best = i;
    if (best < 0 || (!role && free_[best].cap > 2 * n + 1024)) return nullptr;
    double *p = free_[best].p;
    cached_bytes -= free_[best].cap * sizeof(double);
    free_.erase(free_.begin() + best);
    return p;
  }
  int get(size_t n, double **out) {
    if ((*out = take(n, 0))) return PSP_OK;
    double *p = nullptr;
    hipError_t e = hipMalloc((void **)&p, sizeof(double) * (n ? n : 1));
    if (e != hipSuccess) {
      trim();  // cached vectors may be what exhausts the device: drop them and retry once
      e = hipMalloc((void **)&p, sizeof(double) * (n ? n : 1));
    }
    if (e != hipSuccess)
      return fail(PSP_ENOMEM, "solver work vector (%zu doubles): %s", n, hipGetErrorString(e));
    *out = p;
    return PSP_OK;
  }
  void put(double *p, size_t cap, int role = 0) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_.size() >= kMaxCached || cached_bytes + cap * sizeof(double) > cap_bytes()) {
      (void)hipFree(p);
      return;
    }
    free_.push_back({p, cap, current_device(), role});
    cached_bytes += cap * sizeof(double);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &it : free_) (void)hipFree(it.p);
    free_.clear();
    cached_bytes = 0;
  }
};
ScratchPool g_pool;
}  // namespace
namespace psp {
int scratch_get(size_t n, double **out) { return g_pool.get(n, out); }
void scratch_put(double *p, size_t n) {
  if (p) g_pool.put(p, n ? n : 1, 0);
}
}  // namespace psp
namespace {

struct DevVecs {
  struct Held {
    double *p;
    size_t cap;
    int role;
  };
  std::vector<Held> ptrs;
  ~DevVecs() {
    for (auto &h : ptrs)
      if (h.p) g_pool.put(h.p, h.cap, h.role);
  }
  int alloc(size_t n, double **out) {
    double *p = nullptr;
    PSP_TRY(g_pool.get(n, &p));
    ptrs.push_back({p, n ? n : 1, 0});
    *out = p;
    return PSP_OK;
  }
  // the operands of the solve's products y = A x (psp_place.hip): *y for the output role, *x0 / *x1 for the input role.
  // Where placement does not apply they are ordinary pool vectors.  The first solve of a (device, length) draws them;
  // later solves find them in the pool under their roles.
  int alloc_operands(const psp_csr *A, size_t n, double **y, double **x0, double **x1) {
    if (!placement_applies(A, n)) {
      PSP_TRY(alloc(n, y));
      PSP_TRY(alloc(n, x0));
      return alloc(n, x1);
    }
    double *py = g_pool.take(n, 1), *pa = g_pool.take(n, 2), *pb = g_pool.take(n, 2);
    if (!(py && pa && pb)) {  // an incomplete set (the pool let part of it go): draw afresh
      for (double *p : {py, pa, pb})
        if (p) (void)hipFree(p);
      double *xs[2] = {nullptr, nullptr};
      int rc = place_operands(A, n, n, 2, &py, xs, nullptr);
      if (rc == PSP_ENOMEM) {  // the draw holds several vectors at once: without the cached ones it may fit
        g_pool.trim();
        rc = place_operands(A, n, n, 2, &py, xs, nullptr);
      }
      PSP_TRY(rc);
      pa = xs[0];
      pb = xs[1];
    }
    ptrs.push_back({py, n, 1});
    ptrs.push_back({pa, n, 2});
    ptrs.push_back({pb, n, 2});
    *y = py;
    *x0 = pa;
    *x1 = pb;
    return PSP_OK;
  }
};

// reduce `nvals` slots of the workspace partials and bring them to the host
int reduce_fetch(Workspace *w, int nparts, int nvals, double *host) {
  return finish_partials_fetch(w->partials, nparts, nvals, w->scal_dev, host);
}

// the dinv array when K is a native single-step Jacobi, else nullptr
const double *fused_dinv(const psp_op *K) {
  if (K && K->kind == PSP_OP_JACOBI && K->jac->steps == 1) return K->jac->dinv;
  return nullptr;
}

}  // namespace

namespace psp {

int jacobi_apply_dev(psp_jacobi *K, const double *x, double *y) {
  PSP_TRY(k_jacobi_first(K->n, x, K->dinv, y));
  for (int step = 1; step < K->steps; ++step) {
    if (K->A.kind == 0) return fail(PSP_EINVAL, "jacobi: steps > 1 needs the matrix operator");
    PSP_HIP(hipMemcpyAsync(K->temp, y, sizeof(double) * (size_t)K->n, hipMemcpyDeviceToDevice,
                           stream()));
    PSP_TRY(op_apply(&K->A, K->temp, y));
    PSP_TRY(k_jacobi_sweep(K->n, x, K->dinv, K->temp, y));
  }
  return PSP_OK;
}

int op_apply(const psp_op *op, const double *x_dev, double *y_dev) {
  switch (op->kind) {
    case PSP_OP_CSR:
      return psp_csr_matvec_dev(op->csr, x_dev, y_dev);
    case PSP_OP_SSS:
      return psp_sss_matvec_dev(op->sss, x_dev, y_dev);
    case PSP_OP_JACOBI:
      return jacobi_apply_dev(op->jac, x_dev, y_dev);
    case PSP_OP_SSOR:
      return ssor_apply_dev(op->ssor, x_dev, y_dev);
    case PSP_OP_CALLBACK: {
      // SpMatrix_Matvec / SpMatrix_Precon (spmatrixmodule.c:169-248): the callee sees host
      // arrays, so bridge the device vectors through pinned staging buffers
      const size_t bytes = sizeof(double) * (size_t)op->n;
      PSP_HIP(hipMemcpyAsync(op->hx, x_dev, bytes, hipMemcpyDeviceToHost, stream()));
      PSP_HIP(hipStreamSynchronize(stream()));
      if (op->fn(op->ctx, op->n, op->hx, op->hy) != 0)
        return fail(PSP_ECALLBACK, "operator callback reported failure");
      PSP_HIP(hipMemcpyAsync(y_dev, op->hy, bytes, hipMemcpyHostToDevice, stream()));
      return PSP_OK;
    }
  }
  return fail(PSP_EINVAL, "op_apply: unknown operator kind %d", op->kind);
}

}  // namespace psp

// ====================================================================== PCG

// ---------------------------------------------------------------------- asynchronous PCG
//
// Same algorithm and the same kernels as pcg_device's fused path, but alpha, beta and the
// exit tests of pcg.c:100-162 are evaluated by one-thread kernels on the device, and a
// `status` word turns every later launch into a no-op once the loop has ended.  The host
// enqueues kBatch iterations, then reads the 96-byte state once: two host round trips per
// ITERATION become one per BATCH (matters when an iteration is shorter than ~1 ms).

__device__ __forceinline__ void pcg_finish(PcgDev *st, int code, int iter) {
  st->status = 1;
  st->info = code;
  st->iter = iter;
  st->relres = st->normr / st->n2b;  // pcg.c:166
}

// after q = A p and the p.q reduction: pcg.c:117-125
__device__ __forceinline__ void pcg_scalar_pq(PcgDev *st, const double *__restrict__ scal) {
  if (st->status) return;
  const double pq = scal[0];
  if (pq == 0.0) {
    pcg_finish(st, -6, st->it);
    return;
  }
  const double alpha = st->rho / pq;
  st->alpha = alpha;
  if (alpha == 0.0) st->stag = 1;
}

// after the x/r update and its reductions: pcg.c:127-162 for iteration `it`, then the head
// of iteration it+1 (pcg.c:99-112)
__device__ __forceinline__ void pcg_scalar_xr(PcgDev *st, const double *__restrict__ scal, double *__restrict__ hist) {
  if (st->status) return;
  const int it = st->it;
  if (st->stag == 0) st->stag = (scal[2] == 0.0) ? 1 : 0;
  const double normr = sqrt(scal[0]);
  st->normr = normr;
  if (hist) hist[it] = normr;
  if (normr <= st->tolb) {
    pcg_finish(st, 0, it);
  } else if (st->stag == 1) {
    pcg_finish(st, -5, it);
  } else if (it == st->maxit) {
    pcg_finish(st, -1, it + 1);  // pcg.c:165: the loop ran out
  } else {
    const double rho1 = st->rho, rho = scal[1];
    st->rho1 = rho1;
    st->rho = rho;
    st->it = it + 1;
    if (rho == 0.0) {
      pcg_finish(st, -2, it + 1);
    } else {
      const double beta = rho / rho1;
      if (beta == 0.0)
        pcg_finish(st, -6, it + 1);
      else
        st->beta = beta;
    }
  }
}

// The scalar recurrences ride in the block that finishes the reduction they depend on (one launch
// instead of finish_kernel + a one-thread kernel; the summation order is reduce_block's, so the values
// are the ones the separate kernels produced).
enum PcgScalarOp { kOpPq = 0, kOpXr = 1, kOpLazyX = 2, kOpLazyPq = 3, kOpLazyR = 4 };

template <int OP>
__global__ __launch_bounds__(kReduceBlock) void pcg_finish_scalar_kernel(const double *__restrict__ src, int count,
                                                                         int nvals, int stride, int raw,
                                                                         double *__restrict__ out, PcgDev *st,
                                                                         double *__restrict__ hist);
template <int OP>
static int pcg_reduce_then(const double *partials, int nparts, int nvals, double *out_dev, PcgDev *st,
                           double *hist_dev);

static int pcg_async_enabled() {
  static const int on = [] {
    const char *e = psp::tuning_env("PSP_PCG_ASYNC");
    return e ? atoi(e) : 1;
  }();
  return on;
}

// hipGraph replay of the batches is implemented but OFF by default: measured on MI355X it does
// not beat direct launches here (poisson2d(300): 33.7 vs 30.7 us/iteration; 512^3: 199 vs 210
// iterations/s) and capture + instantiate costs ~5 ms per solve.  PSP_PCG_GRAPH=1 enables it.
static int pcg_graph_enabled() {
  static const int on = [] {
    const char *e = psp::tuning_env("PSP_PCG_GRAPH");
    return e ? atoi(e) : 0;
  }();
  return on;
}

// one batch = kBatch iterations' worth of launches on the library stream; every kernel takes
// its scalars (and the iteration number) from the device state, so all batches are identical
// p / p2: the direction vector ping-pongs between two buffers when the p-update is folded into
// the SpMV (csr_spmv_w4_pf reads p_old while it writes p_new); *p is always the current one
static int pcg_enqueue_batch(psp_csr *Acsr, const double *dinv, int n, double *x, double *r, double **pp,
                             double **pp2, double *q, PcgDev *st, double *hist_dev, int batch) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  for (int i = 0; i < batch; ++i) {
    int fusedp = 0;
    PSP_TRY(csr_spmv_pfused_launch(Acsr, r, dinv, *pp, *pp2, q, 0.0, false, w->partials, &np, st, &fusedp));
    if (fusedp) {
      std::swap(*pp, *pp2);
    } else {
      PSP_TRY(k_pupdate(n, r, dinv, 0.0, false, *pp, st));
      PSP_TRY(csr_spmv_launch(Acsr, *pp, q, *pp, w->partials, &np, &st->status));
    }
    double *p = *pp;
    PSP_TRY(pcg_reduce_then<kOpPq>(w->partials, np, 1, w->scal_dev, st, nullptr));
    PSP_TRY(k_xr_update(n, 0.0, p, q, dinv, x, r, w->partials, &np, st));
    PSP_TRY(pcg_reduce_then<kOpXr>(w->partials, np, 3, w->scal_dev + 4, st, hist_dev));
  }
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// runs iterations 1..maxit; on entry r = b - A x, rho0 = r.z != 0, normr0 > tolb.
// The first batch is launched directly (it also performs the lazy table builds of the SpMV);
// when more batches are needed the batch is captured ONCE into a hipGraph and replayed.
static int pcg_async_loop(psp_csr *Acsr, const double *dinv, int n, double *x, double *r, double *p,
                          double *p2, double *q, double n2b, double tolb, double normr0, double rho0, int maxit,
                          int *info, int *iter, double *relres, double *hist) {
  constexpr int kBatch = 16;
  PcgDev *st = nullptr;
  PcgDev *hst = nullptr;
  double *hist_dev = nullptr;
  hipStream_t own = nullptr, prev = nullptr;
  bool swapped = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int enqueued = 0;  // iterations launched so far (never more than maxit: a no-op launch of a
                     // 512^3 grid still costs ~55 us)
  static_assert(sizeof(PcgDev) <= kStateBytes, "state slab");
  Workspace *wst;
  PSP_TRY(workspace(&wst));
  st = static_cast<PcgDev *>(wst->state_dev);  // the thread's state slab (psp_internal.h): nothing to allocate or free
  hst = static_cast<PcgDev *>(wst->state_host);
  hipError_t e = hipSuccess;
  if (hist) e = hipMalloc((void **)&hist_dev, sizeof(double) * ((size_t)maxit + 1));
  if (e != hipSuccess) {
    return fail(PSP_ENOMEM, "pcg: state allocation failed: %s", hipGetErrorString(e));
  }
  int rc = PSP_OK;
#define PCG_TRY(call)            \
  do {                           \
    rc = (call);                 \
    if (rc != PSP_OK) goto done; \
  } while (0)
#define PCG_HIP(call)                                                \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) {                                          \
      rc = fail(PSP_ENODEV, "%s: %s", #call, hipGetErrorString(e_)); \
      goto done;                                                     \
    }                                                                \
  } while (0)
  if (hist_dev) PCG_HIP(hipMemsetAsync(hist_dev, 0xff, sizeof(double) * ((size_t)maxit + 1), stream()));  // NaN
  memset(hst, 0, sizeof(PcgDev));
  hst->rho = rho0;
  hst->rho1 = 1.0;
  hst->normr = normr0;
  hst->tolb = tolb;
  hst->n2b = n2b;
  hst->it = 1;
  hst->maxit = maxit;
  PCG_HIP(hipMemcpyAsync(st, hst, sizeof(PcgDev), hipMemcpyHostToDevice, stream()));
  enqueued = std::min(kBatch, maxit);
  PCG_TRY(pcg_enqueue_batch(Acsr, dinv, n, x, r, &p, &p2, q, st, hist_dev, enqueued));
  PCG_HIP(hipMemcpyAsync(hst, st, sizeof(PcgDev), hipMemcpyDeviceToHost, stream()));
  PCG_HIP(hipStreamSynchronize(stream()));
  if (!hst->status) {
    // more batches: replay a captured graph (capture needs a non-null stream)
    bool use_graph = pcg_graph_enabled() != 0;
    if (use_graph) {
      if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess) {
        prev = swap_stream(own);
        swapped = true;
        if (hipStreamBeginCapture(own, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          const int brc = pcg_enqueue_batch(Acsr, dinv, n, x, r, &p, &p2, q, st, hist_dev, kBatch);
          const hipError_t ce = hipStreamEndCapture(own, &graph);
          if (brc != PSP_OK || ce != hipSuccess || graph == nullptr ||
              hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            exec = nullptr;
          }
        }
        if (exec == nullptr) {  // capture not possible here: fall back to direct launches
          (void)hipGetLastError();
          swap_stream(prev);
          swapped = false;
        }
      }
      use_graph = exec != nullptr;
    }
    while (!hst->status) {
      if (use_graph) {
        PCG_HIP(hipGraphLaunch(exec, stream()));
      } else {
        const int batch = std::max(1, std::min(kBatch, maxit - enqueued));
        PCG_TRY(pcg_enqueue_batch(Acsr, dinv, n, x, r, &p, &p2, q, st, hist_dev, batch));
        enqueued += batch;
      }
      PCG_HIP(hipMemcpyAsync(hst, st, sizeof(PcgDev), hipMemcpyDeviceToHost, stream()));
      PCG_HIP(hipStreamSynchronize(stream()));
    }
  }
  *info = hst->info;
  *iter = hst->iter;
  *relres = hst->relres;
  if (hist) {
    const int cnt = std::min(hst->iter, maxit);
    if (cnt >= 1)
      PCG_HIP(hipMemcpy(hist + 1, hist_dev + 1, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
  }
done:
#undef PCG_TRY
#undef PCG_HIP
  if (swapped) {
    (void)hipStreamSynchronize(own);
    swap_stream(prev);
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (own) (void)hipStreamDestroy(own);
  if (hist_dev) (void)hipFree(hist_dev);
  return rc;
}

// ---------------------------------------------------------------------- lazy x-update loop
//
// The x update of iteration k (x += alpha_k p_k, pcg.c:141) and its stagnation scan (:127-139) read
// p_k; so does the p update of iteration k+1 (:113-114).  Doing both in one pass (px_update_kernel)
// saves one read of p per iteration (146 n -> 138 n bytes with a constant Jacobi diagonal).  The
// price is bookkeeping: whether iteration k stagnated (flag -5, :159-162) is then only known inside
// iteration k+1, so
//   * the convergence test of iteration k (:154-157) still ends the loop at once (it comes first in
//     the reference too); the pending x update is applied by a final pass after the loop;
//   * otherwise iteration k+1 starts; right after its px pass the scalar kernel looks at the scan of
//     iteration k: stagnated -> finish with -5 / iter k / the residual of iteration k (x is x_k by
//     then, exactly what the reference returns); the exits at the head of iteration k+1 (rho == 0,
//     beta == 0, :101-112) are evaluated after that, in the reference's order;
//   * when the loop runs out (k == maxit) the final pass does the scan and picks -5 or -1 (:159-165).
// Same kernels' arithmetic, same reduction order: bitwise identical to the other loops (tested).

__device__ __forceinline__ void pcg_lazy_scalar_x(PcgDev *st, const double *__restrict__ scal) {
  if (st->status) return;
  if (st->xpend) {  // iteration it-1: pcg.c:159-162
    const int stag = st->stag0 || scal[0] == 0.0;
    st->xpend = 0;
    if (stag) {
      st->stag = 1;
      pcg_finish(st, -5, st->it - 1);
      return;
    }
  }
  if (st->head_rho0) {  // pcg.c:101-104 of iteration it
    pcg_finish(st, -2, st->it);
    return;
  }
  if (st->head_beta0) pcg_finish(st, -6, st->it);  // pcg.c:109-112
}

__device__ __forceinline__ void pcg_lazy_scalar_pq(PcgDev *st, const double *__restrict__ scal) {
  if (st->status) return;
  const double pq = scal[0];
  if (pq == 0.0) {  // pcg.c:118-120 (x holds the updates through it-1)
    pcg_finish(st, -6, st->it);
    return;
  }
  const double alpha = st->rho / pq;
  st->alpha = alpha;
  st->alpha_x = alpha;
  st->stag0 = alpha == 0.0 ? 1 : 0;
  st->xpend = 1;
}

__device__ __forceinline__ void pcg_lazy_scalar_r(PcgDev *st, const double *__restrict__ scal, double *__restrict__ hist) {
  if (st->status) return;
  const int it = st->it;
  const double normr = sqrt(scal[0]);
  st->normr = normr;
  if (hist) hist[it] = normr;
  if (normr <= st->tolb) {
    pcg_finish(st, 0, it);  // x update of iteration it still pending: final pass
  } else if (it == st->maxit) {
    st->pend_maxit = 1;     // -5 or -1: decided by the scan of the final pass
    st->status = 1;
  } else {
    const double rho1 = st->rho, rho = scal[1];
    st->rho1 = rho1;
    st->rho = rho;
    st->it = it + 1;
    st->head_rho0 = rho == 0.0 ? 1 : 0;
    st->head_beta0 = 0;
    if (rho != 0.0) {
      const double beta = rho / rho1;
      st->beta = beta;
      st->head_beta0 = beta == 0.0 ? 1 : 0;
    }
  }
}

template <int OP>
__global__ __launch_bounds__(kReduceBlock) void pcg_finish_scalar_kernel(const double *__restrict__ src, int count,
                                                                         int nvals, int stride, int raw,
                                                                         double *__restrict__ out, PcgDev *st,
                                                                         double *__restrict__ hist) {
  if (st->status) return;  // loop already over: nothing to reduce either
  __shared__ double sh[kOneBlockGroups];
  reduce_block(src, count, nvals, stride, raw != 0, out, sh);
  if (threadIdx.x == 0) {
    if constexpr (OP == kOpPq) pcg_scalar_pq(st, out);
    if constexpr (OP == kOpXr) pcg_scalar_xr(st, out, hist);
    if constexpr (OP == kOpLazyX) pcg_lazy_scalar_x(st, out);
    if constexpr (OP == kOpLazyPq) pcg_lazy_scalar_pq(st, out);
    if constexpr (OP == kOpLazyR) pcg_lazy_scalar_r(st, out, hist);
  }
}

// lazy loop: the scan of the px pass (`x`: nonstag, one value) and p.q of the product (`q`) reduced by ONE launch behind
// the product; thread 0 then takes pcg_lazy_scalar_x's branches and, if the loop goes on, pcg_lazy_scalar_pq's -- the
// order the separate launches run them in (and the order pcg_dist_scalar_xpq_kernel of the row-block drivers uses)
__global__ __launch_bounds__(kReduceBlock) void pcg_finish_xpq_kernel(const double *__restrict__ src_x, int count_x,
                                                                      int stride_x, int raw_x,
                                                                      const double *__restrict__ src_q, int count_q,
                                                                      int stride_q, int raw_q, double *__restrict__ out_x,
                                                                      double *__restrict__ out_q, PcgDev *st) {
  if (st->status) return;
  __shared__ double sh[kOneBlockGroups];
  reduce_block(src_x, count_x, 1, stride_x, raw_x != 0, out_x, sh);
  reduce_block(src_q, count_q, 1, stride_q, raw_q != 0, out_q, sh);
  if (threadIdx.x == 0) {
    pcg_lazy_scalar_x(st, out_x);
    pcg_lazy_scalar_pq(st, out_q);
  }
}

// the group fold (when there are more partials than one block takes) + the finishing block with scalar update OP
template <int OP>
static int pcg_reduce_then(const double *partials, int nparts, int nvals, double *out_dev, PcgDev *st,
                           double *hist_dev) {
  const double *src;
  int count, stride;
  bool raw;
  PSP_TRY(fold_stage(partials, nparts, nvals, &src, &count, &stride, &raw));
  hipLaunchKernelGGL((pcg_finish_scalar_kernel<OP>), dim3(1), dim3(kReduceBlock), 0, stream(), src, count, nvals,
                     stride, raw ? 1 : 0, out_dev, st, hist_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

static int pcg_reduce_xpq(Workspace *w, const double *parts_x, int np_x, const double *parts_q, int np_q, double *out_x,
                          double *out_q, PcgDev *st) {
  const double *const parts[2] = {parts_x, parts_q};
  const int np[2] = {np_x, np_q}, fslot[2] = {2, 0};
  const double *src[2];
  int count[2], stride[2];
  bool raw[2];
  (void)w;
  PSP_TRY(fold_stage2(parts, np, fslot, src, count, stride, raw));
  hipLaunchKernelGGL(pcg_finish_xpq_kernel, dim3(1), dim3(kReduceBlock), 0, stream(), src[0], count[0], stride[0],
                     raw[0] ? 1 : 0, src[1], count[1], stride[1], raw[1] ? 1 : 0, out_x, out_q, st);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

static int pcg_lazy_enabled() {  // read per solve (tools/reduce_ab.py alternates it inside one process)
  const char *e = psp::tuning_env("PSP_PCG_LAZYX");
  return e ? atoi(e) : 1;
}

// PSP_PCG_LAZYPF (tuning switch, read per solve: tools/lazypf_ab.py alternates it inside one process): the lazy loop with
// the p update AND the pending x update folded into the product on index-free operators (csr_spmv_pfx_launch): 4 launches
// and 130 n bytes per iteration instead of 5 and 138 n.  Same expressions on the same operands: the same bits.
// Measured in one process on the same buffers (tools/lazypf_ab.py, profiles/r5_pcg_lazypf_ab.txt), same bits everywhere:
// 2048^2 +5.2 %, 4096^2 +2.9 %, 512^3 +0.8 %, 256^3 +0.8 %, 1024^2 -0.4 % -- the saved 8 n bytes and the saved launch
// are partly paid back by reading r AND p_old at every neighbour position (twice the L1 traffic of the plain product):
// on from 2^21 unknowns.
static int pcg_lazypf_mode(int n) {
  const char *e = psp::tuning_env("PSP_PCG_LAZYPF");
  return e ? atoi(e) : (n >= (1 << 21) ? 1 : 0);
}

static int pcg_async_loop_lazy(psp_csr *Acsr, const double *dinv, int n, double *x, double *r, double *p, double *p2,
                               double *q, double n2b, double tolb, double normr0, double rho0, int maxit,
                               int *info, int *iter, double *relres, double *hist) {
  constexpr int kBatch = 16;
  double *const P[2] = {p, p2};  // folded form: iteration e reads p_{e-1} from P[(e-1) & 1] and writes p_e to P[e & 1]
  int usepf = (p2 && pcg_lazypf_mode(n)) ? 1 : 0;
  Workspace *w;
  PSP_TRY(workspace(&w));
  PcgDev *st = nullptr, *hst = nullptr;
  double *hist_dev = nullptr;
  static_assert(sizeof(PcgDev) <= kStateBytes, "state slab");
  Workspace *wst;
  PSP_TRY(workspace(&wst));
  st = static_cast<PcgDev *>(wst->state_dev);  // the thread's state slab (psp_internal.h): nothing to allocate or free
  hst = static_cast<PcgDev *>(wst->state_host);
  hipError_t e = hipSuccess;
  if (hist) e = hipMalloc((void **)&hist_dev, sizeof(double) * ((size_t)maxit + 1));
  if (e != hipSuccess) {
    return fail(PSP_ENOMEM, "pcg: state allocation failed: %s", hipGetErrorString(e));
  }
  int rc = PSP_OK;
  int enqueued = 0, np = 0;
  double *stag_parts = w->partials + 2 * (size_t)kMaxParts;  // slot 2: the scan's partials
  const bool merge_xpq = [] {  // read per solve: tools/reduce_ab.py alternates it inside one process
    const char *e = psp::tuning_env("PSP_PCG_MERGE_XPQ");
    return e ? atoi(e) != 0 : true;
  }();
#define PCG_TRY(call)            \
  do {                           \
    rc = (call);                 \
    if (rc != PSP_OK) goto done; \
  } while (0)
#define PCG_HIP(call)                                                \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) {                                          \
      rc = fail(PSP_ENODEV, "%s: %s", #call, hipGetErrorString(e_)); \
      goto done;                                                     \
    }                                                                \
  } while (0)
  if (hist_dev) PCG_HIP(hipMemsetAsync(hist_dev, 0xff, sizeof(double) * ((size_t)maxit + 1), stream()));
  memset(hst, 0, sizeof(PcgDev));
  hst->rho = rho0;
  hst->rho1 = 1.0;
  hst->normr = normr0;
  hst->tolb = tolb;
  hst->n2b = n2b;
  hst->it = 1;
  hst->maxit = maxit;
  PCG_HIP(hipMemcpyAsync(st, hst, sizeof(PcgDev), hipMemcpyHostToDevice, stream()));
  do {
    const int batch = std::max(1, std::min(kBatch, maxit - enqueued));
    for (int i = 0; i < batch; ++i) {
      // round 4: the scan of the px pass is reduced TOGETHER with p.q, behind the product (one launch for both up to
      // n = 2^25, the group fold + one beyond) -- 5 launches per iteration instead of 9 (7 beyond 2^25).  The product
      // then also runs in the one iteration that the scan ends (-5) or that starts with rho == 0 / beta == 0: its q is
      // not used.  Same scalar steps in the same order on the same reduced values: the same bits
      // (PSP_PCG_MERGE_XPQ=0 keeps the scan's reduction in front of the product: A/B and test_pcg_loop_variants_agree).
      int np_x = 0;
      if (usepf) {
        const int e = enqueued + i + 1;
        int av = 0;
        PCG_TRY(csr_spmv_pfx_launch(Acsr, r, dinv, P[(e - 1) & 1], P[e & 1], q, x, w->partials, &np, st, &av));
        if (av) {
          PCG_TRY(pcg_reduce_xpq(w, stag_parts, np, w->partials, np, w->scal_dev + 8, w->scal_dev, st));
          PCG_TRY(k_r_update(n, 0.0, q, dinv, r, w->partials, &np, st));
          PCG_TRY(pcg_reduce_then<kOpLazyR>(w->partials, np, 2, w->scal_dev + 4, st, hist_dev));
          continue;
        }
        usepf = 0;  // no index-free layout (decided by the operator: this is iteration 1, p = P[0] is untouched)
      }
      PCG_TRY(k_px_update(n, r, dinv, p, x, w->partials, &np_x, st));
      if (!merge_xpq) PCG_TRY(pcg_reduce_then<kOpLazyX>(stag_parts, np_x, 1, w->scal_dev + 8, st, nullptr));
      PCG_TRY(csr_spmv_launch(Acsr, p, q, p, w->partials, &np, &st->status));
      if (merge_xpq)
        PCG_TRY(pcg_reduce_xpq(w, stag_parts, np_x, w->partials, np, w->scal_dev + 8, w->scal_dev, st));
      else
        PCG_TRY(pcg_reduce_then<kOpLazyPq>(w->partials, np, 1, w->scal_dev, st, nullptr));
      PCG_TRY(k_r_update(n, 0.0, q, dinv, r, w->partials, &np, st));
      PCG_TRY(pcg_reduce_then<kOpLazyR>(w->partials, np, 2, w->scal_dev + 4, st, hist_dev));
    }
    PCG_HIP(hipGetLastError());
    enqueued += batch;
    PCG_HIP(hipMemcpyAsync(hst, st, sizeof(PcgDev), hipMemcpyDeviceToHost, stream()));
    PCG_HIP(hipStreamSynchronize(stream()));
  } while (!hst->status);
  if (usepf) p = P[hst->it & 1];  // the direction of the iteration the loop ended in
  if (hst->xpend) {  // the x update (and scan) of the last iteration
    double s[1];
    PCG_TRY(k_x_update(n, hst->alpha_x, p, x, w->partials, &np, nullptr));
    PCG_TRY(finish_partials(stag_parts, np, 1, w->scal_dev + 8));
    PCG_TRY(fetch_scalars(w->scal_dev + 8, 1, s));
    if (hst->pend_maxit) {
      const bool stag = hst->stag0 || s[0] == 0.0;
      hst->info = stag ? -5 : -1;               // pcg.c:159-165
      hst->iter = stag ? maxit : maxit + 1;
      hst->relres = hst->normr / hst->n2b;
    }
  }
  *info = hst->info;
  *iter = hst->iter;
  *relres = hst->relres;
  if (hist) {
    const int cnt = std::min(hst->iter, maxit);
    if (cnt >= 1)
      PCG_HIP(hipMemcpy(hist + 1, hist_dev + 1, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
  }
done:
#undef PCG_TRY
#undef PCG_HIP
  if (hist_dev) (void)hipFree(hist_dev);
  return rc;
}

// Acsr_forced != nullptr: the caller has moved the system into the numbering of a renumbered copy of A
// (pcg_device below): multiply with that handle, take z = dinv_forced .* r (or z = r), never touch A / K
// ||v||_2 where the sum of squares `sq` has left [1e-280, 1e280] (or is not a number): sqrt(sum v_i^2) cannot be
// trusted there, the reference's dnrm2 (scaled form: pcg.c:57,75; minres.c:71) still can.  The norm is then
// formed at a power-of-two scale (exact): 2^e * sqrt(sum (v_i * 2^-e)^2), e = exponent of max |v_i|.  Only the
// setup norms take this path; the reference's ddot (rho = r.z, p.q, alpha = v.Av) underflows / overflows at
// such scales exactly as the reductions here do, and the solvers then leave through the same exits as the CPU
// (rho == 0 -> -2; NaN iterates -> -5 / -1): tests/test_gpu_solvers.py::test_badly_scaled_right_hand_side.
static bool norm2_unsafe(double sq) { return !(sq >= 1e-280 && sq <= 1e280); }
static int abs_max(int n, const double *v, double *out);
static int robust_nrm2(Workspace *w, int n, const double *v, double sq, double *out) {
  if (!norm2_unsafe(sq)) {
    *out = sqrt(sq);
    return PSP_OK;
  }
  double vmax = 0.0;
  PSP_TRY(abs_max(n, v, &vmax));
  if (vmax == 0.0 || vmax != vmax || std::isinf(vmax)) {  // all zero / NaN / Inf: what sqrt(sq) says
    *out = vmax == 0.0 ? 0.0 : sqrt(sq);
    return PSP_OK;
  }
  int e = 0;
  (void)frexp(vmax, &e);
  DevVecs mem;
  double *t;
  PSP_TRY(mem.alloc(n, &t));
  PSP_HIP(hipMemcpyAsync(t, v, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_TRY(k_scal(n, ldexp(1.0, -e), t));
  int np;
  double s1;
  PSP_TRY(k_dot(n, t, t, w->partials, &np));
  PSP_TRY(reduce_fetch(w, np, 1, &s1));
  *out = ldexp(sqrt(s1), e);
  return PSP_OK;
}

static int pcg_device_core(const psp_op *A, const psp_op *K, psp_csr *Acsr_forced, const double *dinv_forced,
                           int n, double *x, const double *b, double tol, int maxit, int *info, int *iter,
                           double *relres, double *hist) {
  note_solve("pcg_no_iterations", 0, 0, 0);  // until one of the loops below is taken (b = 0, x0 already good enough)
  Workspace *w;
  PSP_TRY(workspace(&w));
  DevVecs mem;
  double *r, *p, *q, *z = nullptr;
  psp_csr *Acsr = Acsr_forced ? Acsr_forced : op_native_csr(A);
  const double *dinv = Acsr_forced ? dinv_forced : fused_dinv(K);
  const bool fused = Acsr != nullptr && (Acsr_forced || K == nullptr || dinv != nullptr);
  double *p2 = nullptr;  // second direction buffer of the p-update-in-SpMV path (csr_spmv_w4_pf)
  PSP_TRY(mem.alloc(n, &r));
  if (fused) {
    // q = A p is THE product of an iteration: q takes the output role, p (and its twin p2) the input role (psp_place.hip)
    PSP_TRY(mem.alloc_operands(Acsr, n, &q, &p, &p2));
  } else {
    PSP_TRY(mem.alloc(n, &p));
    PSP_TRY(mem.alloc(n, &q));
    if (K) PSP_TRY(mem.alloc(n, &z));
  }

  double s[4];
  int np;

  // n2b = ||b||  (pcg.c:57)
  PSP_TRY(k_dot(n, b, b, w->partials, &np));
  PSP_TRY(reduce_fetch(w, np, 1, s));
  double n2b;
  PSP_TRY(robust_nrm2(w, n, b, s[0], &n2b));
  if (n2b == 0.0) {  // pcg.c:58-67
    PSP_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    *info = 0;
    *relres = 0.0;
    *iter = 0;
    return PSP_OK;
  }

  *info = -1;  // pcg.c:70
  const double tolb = tol * n2b;

  // r = b - A x, normr (pcg.c:72-75); the fused form also yields rho = r.z for iteration 1
  if (Acsr_forced)
    PSP_TRY(csr_spmv_launch(Acsr, x, r, nullptr, nullptr, nullptr));
  else
    PSP_TRY(op_apply(A, x, r));
  PSP_TRY(k_residual(n, b, r, fused ? dinv : nullptr, w->partials, &np));
  PSP_TRY(reduce_fetch(w, np, 2, s));
  double normr;
  PSP_TRY(robust_nrm2(w, n, r, s[0], &normr));
  double rho_next = s[1];
  if (hist) hist[0] = normr;

  if (normr <= tolb) {  // pcg.c:77-84
    *info = 0;
    *relres = normr / n2b;
    *iter = 0;
    return PSP_OK;
  }

  const bool sk = single_kernel_loops_enabled();  // psp_set_single_kernel_loops(0): launch-per-phase loops only
  double dcst;
  const int dstream = (dinv && !dinv_constant(dinv, n, &dcst)) ? 1 : 0;
  if (sk && fused && maxit >= 1 && rho_next != 0.0 && mid_applicable(Acsr, n, dinv)) {
    // mid-size offset-structured system: the whole loop is one cooperative kernel, vectors in registers, p through LDS
    // (psp_mid.hip) -- the launch-per-phase loops' bits
    const int rc = pcg_mid_loop(Acsr, dinv, n, x, r, p, q, n2b, tolb, normr, rho_next, maxit, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("pcg_mid", 1, 0, dstream);
      return rc;
    }
    note_fallback();  // refused / gave up: x and r are untouched, the loops below take over
  }
  if (sk && fused && maxit >= 1 && rho_next != 0.0 && brick_applicable(Acsr, n)) {
    // 3-D grid operator whose slabs the loop above declines: the same loop with the points dealt out in bricks (psp_mid.hip)
    const int rc = pcg_brick_loop(Acsr, dinv, n, x, r, p, q, n2b, tolb, normr, rho_next, maxit, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("pcg_brick", 1, 0, dstream);
      return rc;
    }
    note_fallback();
  }
  if (sk && fused && maxit >= 1 && coop_applicable(Acsr, n)) {  // small system: the whole loop is one kernel (psp_coop.hip)
    const int rc = pcg_coop_loop(Acsr, dinv, n, x, r, p, q, n2b, tolb, normr, rho_next, maxit, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("pcg_coop", 1, 0, dstream);
      return rc;
    }
    note_fallback();  // refused / gave up: x and r are untouched, the loops below take over
  }
  if (fused && maxit >= 1 && pcg_async_enabled() && csr_spmv_has_skip(Acsr)) {
    if (rho_next == 0.0) {  // pcg.c:101-104 in iteration 1
      *info = -2;
      *iter = 1;
      *relres = normr / n2b;
      return PSP_OK;
    }
    // lazy x update: 8 n bytes less per iteration and, since a reduction is one launch (round 4), 5 launches against
    // the eager loop's 6.  In-process A/B on the same buffers (tools/lazy_ab.py, profiles/r4_pcg_lazy_threshold.txt):
    // +3 ... +10 % from 2^18 to 2^23.5 unknowns, +1 ... +4 % from 2^24.6 on -- and -0.5 / -2.9 % at exactly 2^24 (256^3,
    // 4096^2 = BASELINE's C2; vectors of exactly 128 MiB), -0.8 % at 280^3 (2^24.4), +3.6 % at 320^3 (2^24.97;
    // profiles/r4_pcg_lazy_threshold.txt, second table: after the reductions moved to the group fold), so the band
    // [2^24, 1.5 * 2^24) keeps the eager loop (round 1-3: lazy from 2^25 only, when either loop was 9-10 launches).
    // PSP_PCG_LAZYX=2 forces it at any size (tests), 0 disables it
    const int lazy_mode = pcg_lazy_enabled();
    // launches per iteration: a reduction is one launch up to 4096 partial sums (n <= 2^21), group fold + finish beyond
    const int rl = n > (1 << 21) ? 2 : 0;
    if ((lazy_mode == 2 || (lazy_mode == 1 && (n < (1 << 24) || n >= 3 * (1 << 23)))) && !pcg_graph_enabled()) {
      // bytes per row beside the product: px_update 40 (r, p, x read; p, x written) + r_update 24 (q, r read; r written);
      // folded form: the product forms p and x itself (x read and written, p written: 32 - 8 for the p it reads anyway)
      const bool pf = p2 && pcg_lazypf_mode(n);
      note_solve(pf ? "pcg_lazy_pf" : "pcg_lazy", (pf ? 4 : 5) + rl, (pf ? 56 : 64) + 16 * dstream, dstream);
      return pcg_async_loop_lazy(Acsr, dinv, n, x, r, p, p2, q, n2b, tolb, normr, rho_next, maxit, info, iter,
                                 relres, hist);
    }
    note_solve("pcg_eager", 6 + rl, 72 + 16 * dstream, dstream);
    return pcg_async_loop(Acsr, dinv, n, x, r, p, p2, q, n2b, tolb, normr, rho_next, maxit, info, iter,
                          relres, hist);
  }
  note_solve("pcg_host_scalars", -1, -1, dstream);

  double rho = 1.0, rho1, beta = 0.0, alpha, pq;
  int stag = 0;
  int it;
  for (it = 1; it <= maxit; ++it) {  // pcg.c:91
    const double *zsrc = r;           // vector the p-update reads z from
    const double *zdinv = dinv;       // fused: z = dinv.*r formed on the fly
    if (!fused) {
      zdinv = nullptr;
      if (K) {  // pcg.c:93-94
        PSP_TRY(op_apply(K, r, z));
        zsrc = z;
      }
      // rho = r.z (pcg.c:100); without K, z == r (pcg.c:96)
      PSP_TRY(k_dot(n, r, zsrc, w->partials, &np));
      PSP_TRY(reduce_fetch(w, np, 1, s));
      rho_next = s[0];
    }
    rho1 = rho;
    rho = rho_next;
    if (rho == 0.0) {  // pcg.c:101-104
      *info = -2;
      break;
    }
    if (it > 1) {
      beta = rho / rho1;
      if (beta == 0.0) {  // pcg.c:109-112
        *info = -6;
        break;
      }
    }
    int fusedp = 0;
    if (fused)  // p = z (+ beta p) and q = A p in one pass where the operator has the w4 layout
      PSP_TRY(csr_spmv_pfused_launch(Acsr, r, dinv, p, p2, q, beta, it == 1, w->partials, &np, nullptr, &fusedp));
    if (fusedp) {
      std::swap(p, p2);
    } else {
      PSP_TRY(k_pupdate(n, zsrc, zdinv, it == 1 ? 0.0 : beta, it == 1, p));  // pcg.c:106, :113-114
    }

    // q = A p, pq = p.q (pcg.c:116-117)
    if (fusedp) {
    } else if (Acsr) {
      PSP_TRY(csr_spmv_launch(Acsr, p, q, p, w->partials, &np));
    } else {
      PSP_TRY(op_apply(A, p, q));
      PSP_TRY(k_dot(n, p, q, w->partials, &np));
    }
    PSP_TRY(reduce_fetch(w, np, 1, s));
    pq = s[0];
    if (pq == 0.0) {  // pcg.c:118-120
      *info = -6;
      break;
    }
    alpha = rho / pq;
    if (alpha == 0.0) stag = 1;  // pcg.c:124-125

    // stagnation scan, x += alpha p, r -= alpha q, normr (pcg.c:127-152)
    PSP_TRY(k_xr_update(n, alpha, p, q, fused ? dinv : nullptr, x, r, w->partials, &np));
    PSP_TRY(reduce_fetch(w, np, 3, s));
    if (stag == 0) stag = (s[2] == 0.0);
    normr = sqrt(s[0]);
    rho_next = s[1];
    if (hist) hist[it] = normr;

    if (normr <= tolb) {  // pcg.c:154-157
      *info = 0;
      break;
    }
    if (stag == 1) {  // pcg.c:159-162
      *info = -5;
      break;
    }
  }
  *iter = it;  // pcg.c:165: maxit + 1 when the loop ran out
  *relres = normr / n2b;
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}


// The fused solver loops in the numbering of A's renumbered copy (psp_reorder.hip) when the product y = A x
// goes through one: b, x0 and dinv are permuted once, the loop multiplies with csr_spmv_w3 on the copy -- no
// permutation passes per iteration -- and x is permuted back once.  Same algorithm; the reductions add their
// terms in the new numbering (rounding-level differences, like any other summation order).
// PSP_SOLVE_PERMUTED=0 keeps the caller's numbering.
struct PermutedSystem {
  psp_csr *R = nullptr;
  const int *perm = nullptr, *inv = nullptr;
  double *xp = nullptr, *bp = nullptr, *dp = nullptr;
  bool registered = false;
  ~PermutedSystem() {
    if (registered) dinv_unregister(dp);
  }
  // *active = 1 when the system was moved
  int enter(const psp_op *A, const psp_op *K, int n, DevVecs &mem, const double *x, const double *b, int *active) {
    *active = 0;
    static const bool off = [] {
      const char *e = psp::tuning_env("PSP_SOLVE_PERMUTED");
      return e && atoi(e) == 0;
    }();
    psp_csr *Acsr = op_native_csr(A);
    const double *dinv = fused_dinv(K);
    if (off || !Acsr || !(K == nullptr || dinv != nullptr)) return PSP_OK;
    PSP_TRY(csr_reordered_view(Acsr, &R, &perm, &inv));
    if (!R) return PSP_OK;
    PSP_TRY(mem.alloc(n, &xp));
    PSP_TRY(mem.alloc(n, &bp));
    PSP_TRY(reorder_gather(n, perm, x, xp, nullptr));
    PSP_TRY(reorder_gather(n, perm, b, bp, nullptr));
    if (dinv) {
      PSP_TRY(mem.alloc(n, &dp));
      PSP_TRY(reorder_gather(n, perm, dinv, dp, nullptr));
      double c;
      if (dinv_constant(dinv, n, &c)) {  // a constant vector stays constant
        PSP_TRY(dinv_register(dp, n));
        registered = true;
      }
    }
    *active = 1;
    return PSP_OK;
  }
  int leave(int n, double *x) { return reorder_gather(n, inv, xp, x, nullptr); }  // x[j] = xp[inv[j]]
};

static int pcg_permuted(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol, int maxit,
                        int *info, int *iter, double *relres, double *hist) {
  DevVecs mem;
  PermutedSystem ps;
  int active = 0;
  PSP_TRY(ps.enter(A, K, n, mem, x, b, &active));
  if (!active)
    return pcg_device_core(A, K, nullptr, nullptr, n, x, b, tol, maxit, info, iter, relres, hist);
  const int rc = pcg_device_core(A, K, ps.R, ps.dp, n, ps.xp, ps.bp, tol, maxit, info, iter, relres, hist);
  if (rc != PSP_OK) return rc;
  PSP_TRY(ps.leave(n, x));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

static int pcg_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol, int maxit,
                      int *info, int *iter, double *relres, double *hist) {
  return pcg_permuted(A, K, n, x, b, tol, maxit, info, iter, relres, hist);
}

// ====================================================================== MINRES

// ---------------------------------------------------------------------- asynchronous MINRES
//
// Same kernels and the same arithmetic as the host-scalar loop in minres_device, but the Lanczos /
// Givens recurrences of minres.c:129-192 are evaluated on the device by the thread that finishes each
// of the two reductions, and the vector kernels read their coefficients from the MinresDev state.
// The host enqueues kBatch iterations, then reads the state once (two host round trips per iteration
// become one per batch: what bounds C1-sized solves).  Exit protocol:
//   * -3 (beta^2 < 0) and -6 (r1 == 0) return BEFORE the w / x update of the iteration (minres.c:144-146,
//     :160-162): status = 1, every later kernel is a no-op;
//   * the loop test (minres.c:114) belongs to the head of the NEXT iteration: the scalar step that ends
//     iteration k evaluates it and sets `stop`; the w / x update of iteration k still runs, everything
//     after it is skipped (the next scalar step turns stop into status).
enum MinresScalarOp { kMrAlpha = 0, kMrBeta = 1 };

__device__ __forceinline__ void minres_scalar_alpha(MinresDev *st, const double *__restrict__ scal) {
  if (st->status) return;
  if (st->stop) {  // the update of the last iteration has run by now
    st->status = 1;
    return;
  }
  const double alpha = scal[0];  // minres.c:129
  st->alpha = alpha;
  st->c1 = alpha / st->beta;  // minres.c:131
  st->c2 = st->beta / st->beta_old;
}

__device__ __forceinline__ void minres_scalar_beta(MinresDev *st, const double *__restrict__ scal,
                                                   double *__restrict__ hist) {
  if (st->status) return;
  const double alpha = st->alpha;
  const double beta_old = st->beta;
  st->beta_old = beta_old;
  double beta = scal[0];  // minres.c:143
  if (beta < 0.0) {       // minres.c:144-146
    st->status = 1;
    st->skip = 1;
    st->info = -3;
    return;
  }
  beta = sqrt(beta);
  st->beta = beta;
  // QR factorisation + Givens rotation (minres.c:151-164)
  const double c_oold = st->c_old;
  const double c_old = st->c;
  const double s_oold = st->s_old;
  const double s_old = st->s;
  st->c_old = c_old;
  st->s_old = s_old;
  const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
  const double r1 = sqrt(r1_hat * r1_hat + beta * beta);
  const double r2 = s_old * alpha + c_oold * c_old * beta_old;
  const double r3 = s_oold * beta_old;
  if (r1 == 0.0) {  // minres.c:160-162
    st->status = 1;
    st->skip = 1;
    st->info = -6;
    return;
  }
  const double c = r1_hat / r1;
  const double s_ = beta / r1;
  st->c = c;
  st->s = s_;
  st->r1 = r1;
  st->r2 = r2;
  st->r3 = r3;
  st->c_eta = c * st->eta;  // minres.c:180
  st->eta = -s_ * st->eta;
  const double norm_rmr = st->norm_rmr * fabs(s_);  // minres.c:192
  st->norm_rmr = norm_rmr;
  if (hist) hist[st->iter] = norm_rmr;
  // head of the next iteration (minres.c:114, strict <)
  const bool conv = norm_rmr < st->errtol * st->norm_r0;
  if (st->iter >= st->it_max || conv) {
    st->stop = 1;
    st->skip = 1;
    st->relres = norm_rmr / st->norm_r0;  // minres.c:195
    st->info = conv ? 0 : -1;
  } else {
    st->iter += 1;
  }
}

template <int OP>
__global__ __launch_bounds__(kReduceBlock) void minres_finish_scalar_kernel(const double *__restrict__ src, int count,
                                                                            int stride, int raw, double *__restrict__ out,
                                                                            MinresDev *st, double *__restrict__ hist) {
  if (st->status) return;
  __shared__ double sh[kOneBlockGroups];
  reduce_block(src, count, 1, stride, raw != 0, out, sh);
  if (threadIdx.x == 0) {
    if constexpr (OP == kMrAlpha) minres_scalar_alpha(st, out);
    if constexpr (OP == kMrBeta) minres_scalar_beta(st, out, hist);
  }
}

template <int OP>
static int minres_reduce_then(const double *partials, int nparts, double *out_dev, MinresDev *st, double *hist_dev) {
  const double *src;
  int count, stride;
  bool raw;
  PSP_TRY(fold_stage(partials, nparts, 1, &src, &count, &stride, &raw));
  hipLaunchKernelGGL((minres_finish_scalar_kernel<OP>), dim3(1), dim3(kReduceBlock), 0, stream(), src, count, stride,
                     raw ? 1 : 0, out_dev, st, hist_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

static int minres_async_enabled() {
  static const int on = [] {
    const char *e = psp::tuning_env("PSP_MINRES_ASYNC");
    return e ? atoi(e) : 1;
  }();
  return on;
}

// iterations 1.. on a native operator with K = None or jacobi(steps=1); on entry the setup of
// minres.c:62-94 is done (beta = sqrt(v_hat . y) >= 0, w = w_old = 0) and the first loop test passed
static int minres_async_loop(psp_csr *Acsr, const double *dinv, bool hasK, int n, double *x, double *v_hat,
                             double *v_hat_old, double *y, double *y2, double *wv, double *w_old, double *v,
                             double *av, double norm_r0, double beta0, double errtol, int it_max, int *info,
                             int *iter, double *relres, double *hist) {
  constexpr int kBatch = 16;
  Workspace *w;
  PSP_TRY(workspace(&w));
  MinresDev *st = nullptr, *hst = nullptr;
  double *hist_dev = nullptr;
  static_assert(sizeof(MinresDev) <= kStateBytes, "state slab");
  st = static_cast<MinresDev *>(w->state_dev);  // the thread's state slab (psp_internal.h)
  hst = static_cast<MinresDev *>(w->state_host);
  hipError_t e = hipSuccess;
  if (hist) e = hipMalloc((void **)&hist_dev, sizeof(double) * ((size_t)it_max + 1));
  if (e != hipSuccess) {
    return fail(PSP_ENOMEM, "minres: state allocation failed: %s", hipGetErrorString(e));
  }
  if (hist_dev) (void)hipMemsetAsync(hist_dev, 0xff, sizeof(double) * ((size_t)it_max + 1), stream());  // NaN, like the PCG loops
  int rc = PSP_OK;
  int enqueued = 0, np = 0;
#define MR_TRY(call)             \
  do {                           \
    rc = (call);                 \
    if (rc != PSP_OK) goto done; \
  } while (0)
#define MR_HIP(call)                                                 \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) {                                          \
      rc = fail(PSP_ENODEV, "%s: %s", #call, hipGetErrorString(e_)); \
      goto done;                                                     \
    }                                                                \
  } while (0)
  bool v_ready = false;
  const bool wx_vnext = [] {  // PSP_MINRES_WXV (tuning switch, read per solve): 0 keeps v = y / beta a pass of its own
    const char *e = psp::tuning_env("PSP_MINRES_WXV");
    return e ? atoi(e) != 0 : true;
  }();
  memset(hst, 0, sizeof(MinresDev));
  hst->beta = beta0;
  hst->beta_old = 1.0;
  hst->c = 1.0;
  hst->c_old = 1.0;
  hst->s = 0.0;
  hst->s_old = 0.0;
  hst->eta = beta0;
  hst->norm_rmr = norm_r0;
  hst->norm_r0 = norm_r0;
  hst->errtol = errtol;
  hst->iter = 1;
  hst->it_max = it_max;
  hst->info = -1;
  MR_HIP(hipMemcpyAsync(st, hst, sizeof(MinresDev), hipMemcpyHostToDevice, stream()));
  do {
    const int batch = std::max(1, std::min(kBatch, it_max - enqueued));
    for (int i = 0; i < batch; ++i) {
      // v = y / beta (minres.c:123-124), Av = A v, alpha = v . Av (:127-129)
      const double *vsrc = hasK ? y : v_hat;  // unnormalised Lanczos vector of this iteration
      int scaled = 0;
      if (!hasK || y2)
        MR_TRY(csr_spmv_scaled_launch(Acsr, vsrc, 1.0, av, w->partials, &np, &scaled, &st->skip, &st->beta));
      if (!scaled) {
        // v = y / beta: a pass of its own in the first iteration only -- afterwards the previous iteration's w / x update
        // has written it (k_minres_wx_vnext below)
        if (!v_ready) MR_TRY(k_scale_div(n, vsrc, 1.0, v, st));
        MR_TRY(csr_spmv_launch(Acsr, v, av, v, w->partials, &np, &st->skip));
      }
      MR_TRY(minres_reduce_then<kMrAlpha>(w->partials, np, w->scal_dev, st, nullptr));
      // v_hat = Av - c1 v_hat - c2 v_hat_old; y = K v_hat; beta^2 = v_hat . y (:131-143)
      double *ynew = (scaled && hasK) ? y2 : y;
      MR_TRY(k_lanczos(n, av, 0.0, 0.0, v_hat, v_hat_old, dinv, ynew, w->partials, &np, st));
      std::swap(v_hat, v_hat_old);
      if (scaled && hasK) std::swap(y, y2);
      MR_TRY(minres_reduce_then<kMrBeta>(w->partials, np, w->scal_dev + 4, st, hist_dev));
      // w, x update (:172-180); the new w lands in w_old's buffer
      if (!scaled && wx_vnext) {
        // the unnormalised vector of the NEXT iteration: y (with a preconditioner: the Lanczos pass just wrote it) or the
        // new v_hat (the names were swapped above)
        MR_TRY(k_minres_wx_vnext(n, v, hasK ? y : v_hat, wv, w_old, x, st));
        v_ready = true;
      } else {
        MR_TRY(k_minres_wx(n, scaled ? vsrc : v, 0.0, 0.0, 0.0, 0.0, wv, w_old, x, scaled != 0, 1.0, st));
      }
      std::swap(wv, w_old);
    }
    MR_HIP(hipGetLastError());
    enqueued += batch;
    MR_HIP(hipMemcpyAsync(hst, st, sizeof(MinresDev), hipMemcpyDeviceToHost, stream()));
    MR_HIP(hipStreamSynchronize(stream()));
  } while (!hst->status && !hst->stop);
  *info = hst->info;
  *iter = hst->iter;
  if (hst->info == 0 || hst->info == -1) *relres = hst->relres;  // untouched on -3 / -6, as in the reference
  if (hist) {
    // -3 / -6 end the running iteration before its history slot is written: that slot keeps the caller's fill
    const int cnt = std::min(hst->iter, it_max) - ((hst->info == -3 || hst->info == -6) ? 1 : 0);
    if (cnt >= 1)
      MR_HIP(hipMemcpy(hist + 1, hist_dev + 1, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
  }
done:
#undef MR_TRY
#undef MR_HIP
  if (hist_dev) (void)hipFree(hist_dev);
  return rc;
}

static int minres_device_core(const psp_op *A, const psp_op *K, psp_csr *Acsr_forced, const double *dinv_forced,
                              int n, double *x, const double *b, double errtol, int it_max, int *info, int *iter,
                              double *relres, double *hist) {
  note_solve("minres_no_iterations", 0, 0, 0);  // until one of the loops below is taken
  Workspace *w;
  PSP_TRY(workspace(&w));
  DevVecs mem;
  double *v_hat_old = nullptr, *v_hat = nullptr, *y = nullptr, *wv, *w_old, *v, *av = nullptr;
  const bool hasK = Acsr_forced ? dinv_forced != nullptr : K != nullptr;
  psp_csr *Acsr = Acsr_forced ? Acsr_forced : op_native_csr(A);
  const double *dinv = Acsr_forced ? dinv_forced : fused_dinv(K);
  const bool kfused = Acsr_forced || (K == nullptr) || dinv != nullptr;  // y = K v_hat can ride in the update
  // scaled mode (index-free SpMV layouts): v = y / beta is never stored; the unnormalised vector of
  // the iteration must then survive the Lanczos update, so y ping-pongs between two buffers
  double *y2 = nullptr;
  if (Acsr && kfused && placement_applies(Acsr, n)) {
    // Av = A (y / beta) is THE product of an iteration: av takes the output role; the vector it reads -- y and its twin
    // y2 with a preconditioner, v_hat and v_hat_old (they change places every iteration) without -- the input role
    double *xa, *xb;
    PSP_TRY(mem.alloc_operands(Acsr, n, &av, &xa, &xb));
    if (hasK) {
      y = xa;
      y2 = xb;
    } else {
      v_hat = xa;
      v_hat_old = xb;
    }
  }
  if (!v_hat_old) PSP_TRY(mem.alloc(n, &v_hat_old));
  if (!v_hat) PSP_TRY(mem.alloc(n, &v_hat));
  PSP_TRY(mem.alloc(n, &wv));
  PSP_TRY(mem.alloc(n, &w_old));
  PSP_TRY(mem.alloc(n, &v));
  if (!av) PSP_TRY(mem.alloc(n, &av));
  if (hasK && !y) PSP_TRY(mem.alloc(n, &y));
  if (hasK && kfused && Acsr && !y2) PSP_TRY(mem.alloc(n, &y2));
  const size_t bytes = sizeof(double) * (size_t)n;
  double s[4];
  int np;

  *iter = 0;
  PSP_HIP(hipMemsetAsync(v_hat_old, 0, bytes, stream()));  // minres.c:63-65
  // v_hat = b - A x; norm_r0 (minres.c:67-71); fused: also v_hat . (dinv.*v_hat)
  if (Acsr_forced)
    PSP_TRY(csr_spmv_launch(Acsr, x, v_hat, nullptr, nullptr, nullptr));
  else
    PSP_TRY(op_apply(A, x, v_hat));
  PSP_TRY(k_residual(n, b, v_hat, kfused ? dinv : nullptr, w->partials, &np));
  PSP_TRY(reduce_fetch(w, np, 2, s));
  double norm_r0_;
  PSP_TRY(robust_nrm2(w, n, v_hat, s[0], &norm_r0_));
  const double norm_r0 = norm_r0_;
  double beta = s[1];
  if (hasK) {  // y = K v_hat (minres.c:73-76)
    if (Acsr_forced)
      PSP_TRY(k_jacobi_first(n, v_hat, dinv, y));
    else
      PSP_TRY(op_apply(K, v_hat, y));
    if (!kfused) {
      PSP_TRY(k_dot(n, v_hat, y, w->partials, &np));  // minres.c:78
      PSP_TRY(reduce_fetch(w, np, 1, s));
      beta = s[0];
    }
  }
  if (beta < 0.0) {  // minres.c:79-80
    *info = -3;
    PSP_HIP(hipStreamSynchronize(stream()));
    return PSP_OK;
  }
  beta = sqrt(beta);
  double beta_old = 1.0;
  double c = 1.0, c_old = 1.0, s_ = 0.0, s_old = 0.0, c_oold, s_oold;
  PSP_HIP(hipMemsetAsync(wv, 0, bytes, stream()));  // minres.c:86-90
  PSP_HIP(hipMemsetAsync(w_old, 0, bytes, stream()));
  double eta = beta;
  double norm_rmr = norm_r0;
  if (hist) hist[0] = norm_rmr;

  const bool sk = single_kernel_loops_enabled();  // psp_set_single_kernel_loops(0): launch-per-phase loops only
  double dcst;
  const int dstream = (hasK && dinv && !dinv_constant(dinv, n, &dcst)) ? 1 : 0;
  if (sk && Acsr && kfused && it_max >= 1 && !(norm_rmr < errtol * norm_r0) && (!hasK || dinv) && mid_minres_applicable(Acsr, n)) {
    // mid-size offset-structured system: the whole loop is one cooperative kernel (psp_mid.hip), the launch-per-phase bits
    const int rc = minres_mid_loop(Acsr, hasK ? dinv : nullptr, n, x, v_hat, v_hat_old, y, wv, w_old, v, av, norm_r0,
                                   beta, errtol, it_max, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("minres_mid", 1, 0, dstream);
      return rc;
    }
    note_fallback();
  }
  if (sk && Acsr && kfused && it_max >= 1 && !(norm_rmr < errtol * norm_r0) && (!hasK || dinv) && brick_minres_applicable(Acsr, n)) {
    // 3-D grid operator: the same with the points dealt out in bricks
    const int rc = minres_brick_loop(Acsr, hasK ? dinv : nullptr, n, x, v_hat, v_hat_old, y, wv, w_old, v, av, norm_r0,
                                     beta, errtol, it_max, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("minres_brick", 1, 0, dstream);
      return rc;
    }
    note_fallback();
  }
  if (sk && Acsr && kfused && it_max >= 1 && !(norm_rmr < errtol * norm_r0) && coop_applicable(Acsr, n)) {
    // small system: the whole loop is one kernel (psp_coop.hip)
    const int rc = minres_coop_loop(Acsr, hasK ? dinv : nullptr, n, x, v_hat, v_hat_old, y, wv, w_old, v, av, norm_r0,
                                    beta, errtol, it_max, info, iter, relres, hist);
    if (rc != kCoopFallback) {
      note_solve("minres_coop", 1, 0, dstream);
      return rc;
    }
    note_fallback();  // refused / gave up: x, v_hat, y are untouched, the loops below take over
  }
  if (Acsr && kfused && minres_async_enabled() && it_max >= 1 && !(norm_rmr < errtol * norm_r0)) {
    // scaled product (reads y, writes Av) + lanczos (Av, v_hat, v_hat_old read; v_hat, y written: 40, + 8 when K streams
    // dinv) + w / x update (y, w, w_old, x read; w, x written: 48): 88 bytes per row beside the product
    note_solve("minres_async", 5 + (n > (1 << 21) ? 2 : 0), 88 + 8 * dstream, dstream);
    // the device loop writes hist[1 .. iter]; slots it never reaches keep the caller's fill
    return minres_async_loop(Acsr, dinv, hasK, n, x, v_hat, v_hat_old, y, y2, wv, w_old, v, av, norm_r0,
                             beta, errtol, it_max, info, iter, relres, hist);
  }
  note_solve("minres_host_scalars", -1, -1, dstream);

  while (true) {
    if (*iter >= it_max || norm_rmr < errtol * norm_r0) break;  // minres.c:114 (strict <)
    *iter += 1;

    // v = y / beta (minres.c:123-124); y = v_hat is implied: the update below keeps the
    // old v_hat in v_hat_old directly (minres.c:125,135)
    const double *vsrc = hasK ? y : v_hat;  // unnormalised Lanczos vector of this iteration
    const double vdiv = beta;
    int scaled = 0;
    if (Acsr && kfused && (!hasK || y2))
      PSP_TRY(csr_spmv_scaled_launch(Acsr, vsrc, vdiv, av, w->partials, &np, &scaled));
    if (!scaled) PSP_TRY(k_scale_div(n, vsrc, beta, v));
    // Av = A v, alpha = v.Av (minres.c:127-129)
    if (scaled) {
    } else if (Acsr) {
      PSP_TRY(csr_spmv_launch(Acsr, v, av, v, w->partials, &np));
    } else {
      PSP_TRY(op_apply(A, v, av));
      PSP_TRY(k_dot(n, v, av, w->partials, &np));
    }
    PSP_TRY(reduce_fetch(w, np, 1, s));
    const double alpha = s[0];
    const double dconst1 = alpha / beta, dconst2 = beta / beta_old;  // minres.c:131
    // v_hat = Av - c1 v_hat - c2 v_hat_old; v_hat_old = old v_hat; y = K v_hat; beta^2
    // (the kernels write the new v_hat over v_hat_old; swapping the names is "v_hat_old = old v_hat")
    if (kfused) {
      // scaled mode: the new y goes to the other buffer (vsrc = old y is still needed by the w update)
      double *ynew = (scaled && hasK) ? y2 : y;
      PSP_TRY(k_lanczos(n, av, dconst1, dconst2, v_hat, v_hat_old, dinv, ynew, w->partials, &np));
      std::swap(v_hat, v_hat_old);
      if (scaled && hasK) std::swap(y, y2);
    } else {
      PSP_TRY(k_lanczos_plain(n, av, dconst1, dconst2, v_hat, v_hat_old));
      std::swap(v_hat, v_hat_old);
      PSP_TRY(op_apply(K, v_hat, y));  // minres.c:137-140
      PSP_TRY(k_dot(n, v_hat, y, w->partials, &np));
    }
    PSP_TRY(reduce_fetch(w, np, 1, s));
    beta_old = beta;
    beta = s[0];  // minres.c:143
    if (beta < 0.0) {
      *info = -3;
      PSP_HIP(hipStreamSynchronize(stream()));
      return PSP_OK;
    }
    beta = sqrt(beta);

    // QR factorisation + Givens rotation (minres.c:151-164)
    c_oold = c_old;
    c_old = c;
    s_oold = s_old;
    s_old = s_;
    const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    const double r1 = sqrt(r1_hat * r1_hat + beta * beta);
    const double r2 = s_old * alpha + c_oold * c_old * beta_old;
    const double r3 = s_oold * beta_old;
    if (r1 == 0.0) {
      *info = -6;
      PSP_HIP(hipStreamSynchronize(stream()));
      return PSP_OK;
    }
    c = r1_hat / r1;
    s_ = beta / r1;

    // w, x update (minres.c:172-180)
    // new w lands in w_old's buffer; scaled: v = vsrc / vdiv on the fly (vsrc is v_hat_old resp. y2 now)
    PSP_TRY(k_minres_wx(n, scaled ? vsrc : v, r1, r2, r3, c * eta, wv, w_old, x, scaled != 0, vdiv));
    std::swap(wv, w_old);
    eta = -s_ * eta;
    norm_rmr *= fabs(s_);  // minres.c:192
    if (hist) hist[*iter] = norm_rmr;
  }

  *relres = norm_rmr / norm_r0;  // minres.c:195
  *info = (norm_rmr < errtol * norm_r0) ? 0 : -1;
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

static int minres_permuted(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double errtol,
                           int it_max, int *info, int *iter, double *relres, double *hist) {
  DevVecs mem;
  PermutedSystem ps;
  int active = 0;
  if (minres_async_enabled()) PSP_TRY(ps.enter(A, K, n, mem, x, b, &active));
  if (!active)
    return minres_device_core(A, K, nullptr, nullptr, n, x, b, errtol, it_max, info, iter, relres, hist);
  const int rc =
      minres_device_core(A, K, ps.R, ps.dp, n, ps.xp, ps.bp, errtol, it_max, info, iter, relres, hist);
  if (rc != PSP_OK) return rc;
  PSP_TRY(ps.leave(n, x));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

static int minres_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double errtol,
                         int it_max, int *info, int *iter, double *relres, double *hist) {
  return minres_permuted(A, K, n, x, b, errtol, it_max, info, iter, relres, hist);
}

// largest |v_i|: per-workgroup maxima on the device, the few thousand of them on the host (a path taken once
// per badly scaled solve)
__global__ __launch_bounds__(256) void abs_max_kernel(long n, const double *__restrict__ v, double *__restrict__ out) {
  double m = 0.0;
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double a = fabs(v[i]);
    bad |= a != a;
    m = a > m ? a : m;
  }
  if (bad) m = NAN;
  __shared__ double sh[256];
  sh[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    bool nan = false;
    for (int i = 0; i < 256; ++i) {
      nan |= sh[i] != sh[i];
      r = sh[i] > r ? sh[i] : r;
    }
    out[blockIdx.x] = nan ? NAN : r;
  }
}

static int abs_max(int n, const double *v, double *out) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  const int grid = std::min((n + 255) / 256, 4096);
  hipLaunchKernelGGL(abs_max_kernel, dim3(grid), dim3(256), 0, stream(), (long)n, v, w->partials);
  PSP_LAUNCH_CHECK();
  std::vector<double> h((size_t)grid);
  PSP_HIP(hipMemcpyAsync(h.data(), w->partials, sizeof(double) * (size_t)grid, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  double r = 0.0;
  for (double t : h) {
    if (t != t) {
      *out = NAN;
      return PSP_OK;
    }
    r = t > r ? t : r;
  }
  *out = r;
  return PSP_OK;
}

// ====================================================================== cgs / bicgstab / qmrs / gmres
//
// the four other kernels of pysparse/itsolvers on the same
// operator protocol and the same device vector ops.  Unfused on purpose: each BLAS-1 call /
// hand loop of the reference is one kernel with the reference's rounding order, the scalar
// recurrences run on the host in IEEE double like the C code.  (Parity: checked against vectors
// produced by the reference's own cgs.c / bicgstab.c / qmrs.c / gmres.c compiled unmodified --
// tests/golden/ref_krylov.json, tests/test_gpu_krylov_golden.py)

namespace {

struct Blas {
  Workspace *w;
  long n;
  int dot(const double *x, const double *y, double *out) {
    int np;
    PSP_TRY(k_dot(n, x, y, w->partials, &np));
    return reduce_fetch(w, np, 1, out);
  }
  int nrm2(const double *x, double *out) {
    PSP_TRY(dot(x, x, out));
    *out = sqrt(*out);
    return PSP_OK;
  }
  int copy(const double *x, double *y) {
    PSP_HIP(hipMemcpyAsync(y, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    return PSP_OK;
  }
  int axpy(double a, const double *x, double *y) {  // y = y + a*x; netlib quick return for a == 0
    if (a == 0.0) return PSP_OK;
    return k_lin2(n, 1.0, y, a, x, y);
  }
  int zero(double *x) {
    PSP_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, stream()));
    return PSP_OK;
  }
};

int apply_or_copy(const psp_op *K, long n, const double *x, double *y) {
  if (K) return op_apply(K, x, y);
  PSP_HIP(hipMemcpyAsync(y, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  return PSP_OK;
}

}  // namespace

// ---------------------------------------------------------------- cgs / bicgstab / qmrs with device-resident scalars
//
// Round 5 (VERDICT r4 "Next" #5).  The fused loops below (native matrix + None / jacobi(1)) used to read every reduced
// value back to the host -- through mapped memory and polling since round 4, but still one GPU-idle round trip per
// reduction: 2 (cgs), 4 (bicgstab), 4 (qmrs) per iteration, ~12 us each at 1024^2.  Now the recurrences of cgs.c /
// bicgstab.c / qmrs.c are evaluated by the thread that finishes each reduction (kry_finish_kernel<OP>: reduce_block's
// canonical order, so the reduced values are the ones the host used to get; then the same expressions in the same order,
// IEEE double, -ffp-contract=off on both sides), the vector kernels take their coefficients from the KryDev state
// (KryArg) and do nothing once its status is set, and the host enqueues kKryBatch iterations between two reads of the
// state.  Same bits as the host-scalar loops (tests/test_gpu_krylov_golden.py, test_krylov_more.py; PSP_KRY_DEVSCAL=0
// keeps the host-scalar loops for A/B).
namespace {

constexpr int kKryBatch = 8;

enum KryOp {
  kCgsD, kCgsR,
  kBicgD1, kBicgTs, kBicgTt, kBicgXr,
  kQmrsDelta0, kQmrsEps, kQmrsRho, kQmrsDelta,
};
// cgs registers
enum { CG_RHO = 0, CG_ALPHA, CG_BETA, CG_D, CG_RES, CG_RHONEW, CG_THR };
// bicgstab registers
enum { BI_RHO1 = 0, BI_RHO2, BI_ALPHA, BI_OMEGA, BI_BETA, BI_D1, BI_D2, BI_RR, BI_RHONEXT, BI_RES0, BI_TOL, BI_RES };
// qmrs registers
enum { QM_RHO0 = 0, QM_RHO1, QM_TAU, QM_C0, QM_C1, QM_EPS0, QM_XI1, QM_THETA0, QM_THETA, QM_ETA0, QM_DELTA, QM_CC, QM_BETA,
       QM_CC2, QM_RHO1INV, QM_ERR, QM_RESINIT, QM_TOL, QM_OUT };

__device__ __forceinline__ void kry_end(KryDev *S, int code) {
  S->status = 1;
  S->code = code;
}

// bicgstab.c: head of an iteration (the do-loop's first statements), after the previous one decided to go on
__device__ __forceinline__ void bicg_head(KryDev *S) {
  S->iter += 1;
  if (S->r[BI_RHO1] == 0.0) {
    kry_end(S, 3);  // "return" with info as initialised (-6)
    return;
  }
  S->r[BI_BETA] = (S->r[BI_RHO1] / S->r[BI_RHO2]) * (S->r[BI_ALPHA] / S->r[BI_OMEGA]);
}

// qmrs.c: head of an iteration (while test, ++iter, eps0 / delta tests, cc)
__device__ __forceinline__ void qmrs_head(KryDev *S) {
  if (!(S->r[QM_ERR] > S->r[QM_TOL] && S->iter < S->maxit)) {
    kry_end(S, 1);  // the loop test failed: normal end
    return;
  }
  S->iter += 1;
  if (S->r[QM_EPS0] == 0.0) {
    kry_end(S, 6);
    return;
  }
  if (S->r[QM_DELTA] == 0.0) {
    kry_end(S, 2);
    return;
  }
  S->r[QM_CC] = S->r[QM_XI1] * (S->r[QM_DELTA] / S->r[QM_EPS0]);
}

template <int OP>
__global__ __launch_bounds__(kReduceBlock) void kry_finish_kernel(const double *__restrict__ src, int count, int nvals,
                                                                  int stride, int raw, KryDev *S, int out_base) {
  if (S->status) return;
  __shared__ double sh[kOneBlockGroups];
  reduce_block(src, count, nvals, stride, raw != 0, S->r + out_base, sh);
  if (threadIdx.x != 0) return;
  double *r = S->r;
  if constexpr (OP == kCgsD) {  // cgs.c: alpha = rho / (v . r0)
    r[CG_ALPHA] = r[CG_RHO] / r[CG_D];
  }
  if constexpr (OP == kCgsR) {  // cgs.c: convergence test on r.r, beta, the for-loop's increment and test
    if (r[CG_RES] < r[CG_THR]) {
      kry_end(S, 1);
    } else {
      r[CG_BETA] = r[CG_RHONEW] / r[CG_RHO];
      r[CG_RHO] = r[CG_RHONEW];
      S->iter += 1;
      if (!(S->iter < S->maxit)) kry_end(S, 2);
    }
  }
  if constexpr (OP == kBicgD1) r[BI_ALPHA] = r[BI_RHO1] / r[BI_D1];  // alpha = rho / (rhat . v)
  if constexpr (OP == kBicgTs) {}                                    // t . s lands in r[BI_D1]
  if constexpr (OP == kBicgTt) r[BI_OMEGA] = r[BI_D1] / r[BI_D2];    // omega = (t . s) / (t . t)
  if constexpr (OP == kBicgXr) {  // res, the omega == 0 return, the loop test, then the next iteration's head
    const double res = sqrt(r[BI_RR]);
    r[BI_RES] = res;
    if (r[BI_OMEGA] == 0.0) {
      kry_end(S, 4);
    } else {
      r[BI_RHO2] = r[BI_RHO1];
      r[BI_RHO1] = r[BI_RHONEXT];
      if (!((res / r[BI_RES0] > r[BI_TOL]) && (S->iter < S->maxit))) kry_end(S, 1);
      else bicg_head(S);
    }
  }
  if constexpr (OP == kQmrsDelta0) {  // first iteration: delta = K v1 . v1, then the head
    r[QM_DELTA] = r[QM_OUT];
    qmrs_head(S);
  }
  if constexpr (OP == kQmrsEps) {  // eps0 = g . t; beta = eps0 / delta
    r[QM_EPS0] = r[QM_OUT];
    r[QM_BETA] = r[QM_EPS0] / r[QM_DELTA];
  }
  if constexpr (OP == kQmrsRho) {  // qmrs.c: rho1 = ||v1||, theta, c1, eta0, tau, the coefficients of the d / x update
    const double rho1 = sqrt(r[QM_OUT]);
    const double beta = r[QM_BETA], c0 = r[QM_C0];
    r[QM_RHO1] = rho1;
    r[QM_XI1] = rho1;
    if (c0 * fabs(beta) == 0.0) {
      kry_end(S, 6);
      return;
    }
    const double theta = rho1 / (c0 * fabs(beta));
    const double c1 = 1.0 / sqrt(theta * theta + 1.0);
    r[QM_THETA] = theta;
    r[QM_C1] = c1;
    if (beta * (c0 * c0) == 0.0) {
      kry_end(S, 6);
      return;
    }
    r[QM_ETA0] = -r[QM_ETA0] * r[QM_RHO0] * (c1 * c1) / (beta * (c0 * c0));
    r[QM_TAU] = r[QM_TAU] * theta * c1;
    if (rho1 == 0.0) {
      kry_end(S, 6);
      return;
    }
    const double d1 = r[QM_THETA0] * c1;
    r[QM_CC2] = d1 * d1;
    r[QM_RHO1INV] = 1.0 / rho1;
  }
  if constexpr (OP == kQmrsDelta) {  // delta of the next iteration; the tail of this one; the next one's head
    r[QM_DELTA] = r[QM_OUT];
    if (r[QM_XI1] == 0.0) {
      kry_end(S, 6);
      return;
    }
    r[QM_RHO0] = r[QM_RHO1];
    r[QM_ERR] = r[QM_TAU] / r[QM_RESINIT];
    r[QM_C0] = r[QM_C1];
    r[QM_THETA0] = r[QM_THETA];
    qmrs_head(S);
  }
}

template <int OP>
int kry_reduce_then(const double *partials, int nparts, int nvals, KryDev *S, int out_base, int fslot = 0) {
  const double *src;
  int count, stride;
  bool raw;
  PSP_TRY(fold_stage(partials, nparts, nvals, &src, &count, &stride, &raw, fslot));
  hipLaunchKernelGGL((kry_finish_kernel<OP>), dim3(1), dim3(kReduceBlock), 0, stream(), src, count, nvals, stride,
                     raw ? 1 : 0, S, out_base);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// two single-value reductions (their partial sums in different arrays, of different length) and the step that needs both,
// in ONE finishing launch -- bicgstab: t . s (the product's epilogue) and t . t (a pass of its own), then omega
template <int OP>
__global__ __launch_bounds__(kReduceBlock) void kry_finish2_kernel(const double *__restrict__ src0, int count0, int stride0,
                                                                   int raw0, int out0, const double *__restrict__ src1,
                                                                   int count1, int stride1, int raw1, int out1, KryDev *S) {
  if (S->status) return;
  __shared__ double sh[kOneBlockGroups];
  reduce_block(src0, count0, 1, stride0, raw0 != 0, S->r + out0, sh);
  reduce_block(src1, count1, 1, stride1, raw1 != 0, S->r + out1, sh);
  if (threadIdx.x != 0) return;
  double *r = S->r;
  if constexpr (OP == kBicgTt) r[BI_OMEGA] = r[BI_D1] / r[BI_D2];  // omega = (t . s) / (t . t)
}

template <int OP>
int kry_reduce2_then(const double *p0, int n0, int out0, const double *p1, int n1, int out1, KryDev *S) {
  const double *parts[2] = {p0, p1};
  const int np[2] = {n0, n1}, fslot[2] = {0, 1};
  const double *src[2];
  int count[2], stride[2];
  bool raw[2];
  PSP_TRY(fold_stage2(parts, np, fslot, src, count, stride, raw));
  hipLaunchKernelGGL((kry_finish2_kernel<OP>), dim3(1), dim3(kReduceBlock), 0, stream(), src[0], count[0], stride[0],
                     raw[0] ? 1 : 0, out0, src[1], count[1], stride[1], raw[1] ? 1 : 0, out1, S);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// the state on the device + its pinned host mirror
struct KryState {
  KryDev *dev = nullptr, *host = nullptr;  // in the thread's state slab (psp_internal.h)
  int init() {
    static_assert(sizeof(KryDev) <= kStateBytes, "state slab");
    Workspace *w;
    PSP_TRY(workspace(&w));
    dev = static_cast<KryDev *>(w->state_dev);
    host = static_cast<KryDev *>(w->state_host);
    memset(host, 0, sizeof(KryDev));
    return PSP_OK;
  }
  int upload() {
    PSP_HIP(hipMemcpyAsync(dev, host, sizeof(KryDev), hipMemcpyHostToDevice, stream()));
    return PSP_OK;
  }
  int fetch() {
    PSP_HIP(hipGetLastError());
    PSP_HIP(hipMemcpyAsync(host, dev, sizeof(KryDev), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    return PSP_OK;
  }
};

bool kry_devscal_enabled() {  // read per solve (A/B inside one process)
  const char *e = psp::tuning_env("PSP_KRY_DEVSCAL");
  return e ? atoi(e) != 0 : true;
}

}  // namespace

// pysparse/itsolvers/src/cgs.c:14-110
static int cgs_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol,
                      int maxit, int *info, int *iter, double *res) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  Blas B{w, n};
  DevVecs mem;
  double *r0, *r, *p, *q, *u, *v, *tmp, *tmp2;
  for (double **pp : {&r0, &r, &p, &q, &u, &v, &tmp, &tmp2}) PSP_TRY(mem.alloc(n, pp));
  const double tol_sq = tol * tol;
  double alpha, beta, rho, rho_new, bnrm_sq, d;
  *iter = 0;
  PSP_TRY(op_apply(A, x, tmp));
  PSP_TRY(B.copy(b, r0));
  PSP_TRY(B.axpy(-1.0, tmp, r0));
  PSP_TRY(B.copy(r0, r));
  PSP_TRY(B.copy(r0, u));
  PSP_TRY(B.copy(r0, p));
  PSP_TRY(B.dot(r0, r0, &rho));
  PSP_TRY(B.dot(b, b, &bnrm_sq));
  if (rho < bnrm_sq * tol_sq) {
    *res = sqrt(rho / bnrm_sq);
    *info = 0;
    return PSP_OK;
  }
  // Native matrix + None / jacobi(1): three fused passes (psp_vec.hip: cgs_q / cgs_r / cgs_p -- per element the copy +
  // daxpy pairs of the reference, quick returns included), v.r0 on the first product: ~17 instead of ~44 vector
  // streams per iteration.  PSP_CGS_FUSED=0 keeps the unfused sequence below (A/B).
  static const bool fuse_on = [] {
    const char *e = psp::tuning_env("PSP_CGS_FUSED");
    return e ? atoi(e) != 0 : true;
  }();
  psp_csr *Acsr = op_native_csr(A);
  const double *dinv = fused_dinv(K);
  if (fuse_on && Acsr && !Acsr->nparts && (K == nullptr || dinv != nullptr)) {
    double *kp = p;  // no preconditioner: K p is p
    if (K) {
      PSP_TRY(mem.alloc(n, &kp));
      PSP_TRY(op_apply(K, p, kp));
    }
    double sc[2];
    int np;
    if (kry_devscal_enabled() && maxit >= 1 && csr_spmv_has_skip(Acsr)) {
      KryState st;
      PSP_TRY(st.init());
      st.host->r[CG_RHO] = rho;
      st.host->r[CG_THR] = bnrm_sq * tol_sq;
      st.host->maxit = maxit;
      PSP_TRY(st.upload());
      KryDev *S = st.dev;
      do {
        for (int k = 0; k < kKryBatch; ++k) {
          PSP_TRY(csr_spmv_launch(Acsr, kp, v, r0, w->partials, &np, &S->status));
          PSP_TRY(kry_reduce_then<kCgsD>(w->partials, np, 1, S, CG_D));
          PSP_TRY(k_cgs_q(n, u, v, x, q, tmp2, dinv, 0.0, KryArg{S, CG_ALPHA}));
          PSP_TRY(csr_spmv_launch(Acsr, tmp2, tmp, nullptr, nullptr, nullptr, &S->status));
          PSP_TRY(k_cgs_r(n, r, tmp, r0, 0.0, w->partials, &np, KryArg{S, CG_ALPHA}));
          PSP_TRY(kry_reduce_then<kCgsR>(w->partials, np, 2, S, CG_RES));
          PSP_TRY(k_cgs_p(n, r, q, p, u, kp, dinv, 0.0, KryArg{S, CG_BETA}));
        }
        PSP_TRY(st.fetch());
      } while (!st.host->status);
      *iter = st.host->iter;
      *res = sqrt(st.host->r[CG_RES] / bnrm_sq);
      *info = st.host->code == 1 ? 0 : -1;
      return PSP_OK;
    }
    for (; *iter < maxit; (*iter)++) {
      PSP_TRY(csr_spmv_launch(Acsr, kp, v, r0, w->partials, &np, nullptr));
      PSP_TRY(reduce_fetch(w, np, 1, &d));
      alpha = rho / d;
      PSP_TRY(k_cgs_q(n, u, v, x, q, tmp2, dinv, alpha));
      PSP_TRY(csr_spmv_launch(Acsr, tmp2, tmp, nullptr, nullptr, nullptr, nullptr));
      PSP_TRY(k_cgs_r(n, r, tmp, r0, alpha, w->partials, &np));
      PSP_TRY(reduce_fetch(w, np, 2, sc));
      *res = sc[0];
      if (*res < bnrm_sq * tol_sq) {
        *res = sqrt(*res / bnrm_sq);
        *info = 0;
        return PSP_OK;
      }
      rho_new = sc[1];
      beta = rho_new / rho;
      rho = rho_new;
      PSP_TRY(k_cgs_p(n, r, q, p, u, kp, dinv, beta));
    }
    *res = sqrt(*res / bnrm_sq);
    *info = -1;
    return PSP_OK;
  }
  for (; *iter < maxit; (*iter)++) {
    if (K) {
      PSP_TRY(op_apply(K, p, tmp));
      PSP_TRY(op_apply(A, tmp, v));
    } else {
      PSP_TRY(op_apply(A, p, v));
    }
    PSP_TRY(B.dot(v, r0, &d));
    alpha = rho / d;
    const double ddummy = -alpha;
    PSP_TRY(B.copy(u, q));
    PSP_TRY(B.axpy(ddummy, v, q));
    PSP_TRY(B.copy(u, tmp));
    PSP_TRY(B.axpy(1.0, q, tmp));
    PSP_TRY(apply_or_copy(K, n, tmp, tmp2));
    PSP_TRY(B.axpy(alpha, tmp2, x));
    PSP_TRY(op_apply(A, tmp2, tmp));
    PSP_TRY(B.axpy(ddummy, tmp, r));
    PSP_TRY(B.dot(r, r, res));
    if (*res < bnrm_sq * tol_sq) {
      *res = sqrt(*res / bnrm_sq);
      *info = 0;
      return PSP_OK;
    }
    PSP_TRY(B.dot(r, r0, &rho_new));
    beta = rho_new / rho;
    rho = rho_new;
    PSP_TRY(B.copy(r, u));
    PSP_TRY(B.axpy(beta, q, u));
    PSP_TRY(B.copy(q, tmp));
    PSP_TRY(B.axpy(beta, p, tmp));
    PSP_TRY(B.copy(u, p));
    PSP_TRY(B.axpy(beta, tmp, p));
  }
  *res = sqrt(*res / bnrm_sq);
  *info = -1;
  return PSP_OK;
}

// pysparse/itsolvers/src/bicgstab.c:233-320
static int bicgstab_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b,
                           double tol, int maxit, int *info, int *iter, double *relres) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  Blas B{w, n};
  DevVecs mem;
  double *r, *rhat, *p, *phat, *v, *s, *shat, *t;
  for (double **pp : {&r, &rhat, &p, &phat, &v, &s, &shat, &t}) PSP_TRY(mem.alloc(n, pp));
  double alpha = 0.0, omega = 0.0, rho_im1, rho_im2 = 0.0, beta, res, res0, n2b, d1, d2;
  *info = -6;
  PSP_TRY(B.nrm2(b, &n2b));
  if (n2b == 0.0) {
    PSP_TRY(B.zero(x));
    *info = 0;
    *relres = 0.0;
    *iter = 0;
    return PSP_OK;
  }
  PSP_TRY(op_apply(A, x, r));
  PSP_TRY(k_lin2(n, 1.0, b, -1.0, r, r));  // r = b + -r
  PSP_TRY(B.nrm2(r, &res0));
  PSP_TRY(B.copy(r, rhat));
  *iter = 0;
  // Native matrix + None / jacobi(1): the vector passes of one iteration are three fused kernels (psp_vec.hip:
  // bicg_p / bicg_s / bicg_xr -- the same rounded operations per element), d1 = rhat.v and t.s ride on the two
  // products, the next rho on the last pass: ~21 instead of 32 vector streams per iteration.
  // PSP_BICGSTAB_FUSED=0 keeps the unfused sequence below (A/B; same iterates up to the order of the dot sums).
  static const bool fuse_on = [] {
    const char *e = psp::tuning_env("PSP_BICGSTAB_FUSED");
    return e ? atoi(e) != 0 : true;
  }();
  psp_csr *Acsr = op_native_csr(A);
  const double *dinv = fused_dinv(K);
  if (fuse_on && Acsr && !Acsr->nparts && (K == nullptr || dinv != nullptr)) {
    double *ph = K ? phat : p, *sh = K ? shat : s;  // no preconditioner: phat is p, shat is s
    double sc[2];
    int np;
    PSP_TRY(B.dot(rhat, r, &rho_im1));
    if (kry_devscal_enabled() && maxit >= 1 && csr_spmv_has_skip(Acsr)) {
      *iter = 1;
      if (rho_im1 == 0.0) return PSP_OK;  // the first iteration's head, on the host
      KryState st;
      PSP_TRY(st.init());
      st.host->r[BI_RHO1] = rho_im1;
      st.host->r[BI_RES0] = res0;
      st.host->r[BI_TOL] = tol;
      st.host->iter = 1;
      st.host->maxit = maxit;
      PSP_TRY(st.upload());
      KryDev *S = st.dev;
      int np2;
      do {
        for (int k = 0; k < kKryBatch; ++k) {
          PSP_TRY(k_bicg_p(n, r, v, p, ph, dinv, 0.0, 0.0, false, KryArg{S, BI_BETA, BI_OMEGA}));  // first: S->iter == 1
          PSP_TRY(csr_spmv_launch(Acsr, ph, v, rhat, w->partials, &np, &S->status));
          PSP_TRY(kry_reduce_then<kBicgD1>(w->partials, np, 1, S, BI_D1));
          PSP_TRY(k_bicg_s(n, r, v, s, sh, dinv, 0.0, KryArg{S, BI_ALPHA}));
          PSP_TRY(csr_spmv_launch(Acsr, sh, t, s, w->partials, &np, &S->status));
          // t . t into the second array of partial sums (runs once more after the loop has ended: its result is ignored);
          // ONE finishing launch for t . s, t . t and omega
          PSP_TRY(k_dot(n, t, t, w->partials + kMaxParts, &np2));
          PSP_TRY(kry_reduce2_then<kBicgTt>(w->partials, np, BI_D1, w->partials + kMaxParts, np2, BI_D2, S));
          PSP_TRY(k_bicg_xr(n, x, ph, sh, s, t, r, rhat, 0.0, 0.0, w->partials, &np, KryArg{S, BI_ALPHA, BI_OMEGA}));
          PSP_TRY(kry_reduce_then<kBicgXr>(w->partials, np, 2, S, BI_RR));
        }
        PSP_TRY(st.fetch());
      } while (!st.host->status);
      *iter = st.host->iter;
      if (st.host->code == 1) {
        *relres = st.host->r[BI_RES] / res0;
        *info = (*relres >= tol) ? -1 : 0;
      }
      return PSP_OK;  // codes 3 / 4: rho == 0 / omega == 0 -- info stays -6, relres untouched, like the returns below
    }
    do {
      (*iter)++;
      if (rho_im1 == 0.0) return PSP_OK;
      beta = *iter == 1 ? 0.0 : (rho_im1 / rho_im2) * (alpha / omega);
      PSP_TRY(k_bicg_p(n, r, v, p, ph, dinv, beta, omega, *iter == 1));
      PSP_TRY(csr_spmv_launch(Acsr, ph, v, rhat, w->partials, &np, nullptr));
      PSP_TRY(reduce_fetch(w, np, 1, &d1));
      alpha = rho_im1 / d1;
      PSP_TRY(k_bicg_s(n, r, v, s, sh, dinv, alpha));
      PSP_TRY(csr_spmv_launch(Acsr, sh, t, s, w->partials, &np, nullptr));
      PSP_TRY(reduce_fetch(w, np, 1, &d1));
      PSP_TRY(B.dot(t, t, &d2));
      omega = d1 / d2;
      PSP_TRY(k_bicg_xr(n, x, ph, sh, s, t, r, rhat, alpha, omega, w->partials, &np));
      PSP_TRY(reduce_fetch(w, np, 2, sc));
      res = sqrt(sc[0]);
      if (omega == 0.0) return PSP_OK;
      rho_im2 = rho_im1;
      rho_im1 = sc[1];  // rhat . r of the next iteration (bicgstab.c computes it at the top of the loop)
    } while ((res / res0 > tol) && (*iter < maxit));
    *relres = res / res0;
    *info = (*relres >= tol) ? -1 : 0;
    return PSP_OK;
  }
  do {
    (*iter)++;
    PSP_TRY(B.dot(rhat, r, &rho_im1));
    if (rho_im1 == 0.0) return PSP_OK;  // info stays -6 (the module ignores the kernel's -1)
    if (*iter == 1) {
      PSP_TRY(B.copy(r, p));
    } else {
      beta = (rho_im1 / rho_im2) * (alpha / omega);
      PSP_TRY(k_lin2(n, 1.0, p, -omega, v, t));  // t is free here: p - omega*v
      PSP_TRY(k_lin2(n, 1.0, r, beta, t, p));    // p = r + beta*(p - omega*v)
    }
    PSP_TRY(apply_or_copy(K, n, p, phat));
    PSP_TRY(op_apply(A, phat, v));
    PSP_TRY(B.dot(rhat, v, &d1));
    alpha = rho_im1 / d1;
    PSP_TRY(k_lin2(n, 1.0, r, -alpha, v, s));  // v_plus_cw(n, r, v, -alpha, s)
    PSP_TRY(apply_or_copy(K, n, s, shat));
    PSP_TRY(op_apply(A, shat, t));
    PSP_TRY(B.dot(t, s, &d1));
    PSP_TRY(B.dot(t, t, &d2));
    omega = d1 / d2;
    PSP_TRY(k_lin2(n, 1.0, x, alpha, phat, x));  // x = x + alpha*phat ...
    PSP_TRY(k_lin2(n, 1.0, x, omega, shat, x));  // ... + omega*shat
    PSP_TRY(k_lin2(n, 1.0, s, -omega, t, r));    // r = s - omega*t
    PSP_TRY(B.nrm2(r, &res));
    if (omega == 0.0) return PSP_OK;
    rho_im2 = rho_im1;
  } while ((res / res0 > tol) && (*iter < maxit));
  *relres = res / res0;
  *info = (*relres >= tol) ? -1 : 0;
  return PSP_OK;
}

// pysparse/itsolvers/src/qmrs.c:29-154
static int qmrs_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol,
                       int maxit, int *info, int *iter, double *err) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  Blas B{w, n};
  DevVecs mem;
  double *wrk1, *p, *d, *v1, *t, *g;
  for (double **pp : {&wrk1, &p, &d, &v1, &t, &g}) PSP_TRY(mem.alloc(n, pp));
  double beta, res_init, delta, theta, c0, c1, theta0, cc, xi1, rho1inv, tau, eta0, eps0, rho0, rho1, d1;
  PSP_TRY(B.copy(b, v1));
  PSP_TRY(B.nrm2(v1, &rho0));
  tau = rho0;
  PSP_TRY(k_scale_div(n, v1, rho0, v1));
  PSP_TRY(B.zero(p));
  PSP_TRY(B.zero(g));
  PSP_TRY(B.zero(d));
  PSP_TRY(B.zero(x));
  c0 = 1.0;
  eps0 = 1.0;
  xi1 = 1.0;
  theta0 = 0.0;
  eta0 = -1.0;
  res_init = rho0;
  *err = 1.0;
  *iter = 0;
#define QMRS_RET(code) do { *info = (code); PSP_HIP(hipStreamSynchronize(stream())); return PSP_OK; } while (0)
  // Native matrix + None / jacobi(1): four fused passes (psp_vec.hip: qmrs_kv / qmrs_pg / qmrs_v / qmrs_dx -- the same
  // rounded operations per element), g.t on the product, K v1 and its dot for the next iteration on the last pass:
  // 17 instead of 25 vector streams per iteration.  PSP_QMRS_FUSED=0 keeps the unfused sequence below (A/B).
  static const bool fuse_on = [] {
    const char *e = psp::tuning_env("PSP_QMRS_FUSED");
    return e ? atoi(e) != 0 : true;
  }();
  psp_csr *Acsr = op_native_csr(A);
  const double *dinv = fused_dinv(K);
  if (fuse_on && Acsr && !Acsr->nparts && (K == nullptr || dinv != nullptr)) {
    double *kv = K ? wrk1 : v1;  // no preconditioner: K v1 is v1
    int np;
    bool have_delta = false;
    if (kry_devscal_enabled() && csr_spmv_has_skip(Acsr)) {
      KryState st;
      PSP_TRY(st.init());
      double *R = st.host->r;
      R[QM_RHO0] = rho0;
      R[QM_TAU] = tau;
      R[QM_C0] = c0;
      R[QM_EPS0] = eps0;
      R[QM_XI1] = xi1;
      R[QM_THETA0] = theta0;
      R[QM_ETA0] = eta0;
      R[QM_ERR] = *err;
      R[QM_RESINIT] = res_init;
      R[QM_TOL] = tol;
      st.host->maxit = maxit;
      PSP_TRY(st.upload());
      KryDev *S = st.dev;
      // K v1 . v1 of the first iteration, then its head (the later ones get theirs from the d / x pass)
      PSP_TRY(k_qmrs_kv(n, v1, kv, dinv, w->partials, &np));
      PSP_TRY(kry_reduce_then<kQmrsDelta0>(w->partials, np, 1, S, QM_OUT));
      do {
        for (int k = 0; k < kKryBatch; ++k) {
          PSP_TRY(k_qmrs_pg(n, v1, kv, p, g, 0.0, KryArg{S, QM_CC}));
          PSP_TRY(csr_spmv_launch(Acsr, g, t, g, w->partials, &np, &S->status));
          PSP_TRY(kry_reduce_then<kQmrsEps>(w->partials, np, 1, S, QM_OUT));
          PSP_TRY(k_qmrs_v(n, t, v1, 0.0, w->partials, &np, KryArg{S, QM_BETA}));
          PSP_TRY(kry_reduce_then<kQmrsRho>(w->partials, np, 1, S, QM_OUT));
          PSP_TRY(k_qmrs_dx(n, p, d, x, v1, kv, dinv, 0.0, 0.0, 0.0, w->partials, &np, KryArg{S, QM_ETA0, QM_CC2, QM_RHO1INV}));
          PSP_TRY(kry_reduce_then<kQmrsDelta>(w->partials, np, 1, S, QM_OUT));
        }
        PSP_TRY(st.fetch());
      } while (!st.host->status);
      *iter = st.host->iter;
      *err = st.host->r[QM_ERR];
      if (st.host->code != 1) QMRS_RET(-st.host->code);  // 6 -> -6, 2 -> -2: left by a breakdown test
      if (K) {
        PSP_TRY(op_apply(K, x, wrk1));
        PSP_TRY(B.copy(wrk1, x));
      }
      *info = (*err < tol) ? 0 : -1;
      return PSP_OK;
    }
    while (*err > tol && *iter < maxit) {
      ++(*iter);
      if (eps0 == 0.0) QMRS_RET(-6);
      if (!have_delta) {
        PSP_TRY(k_qmrs_kv(n, v1, kv, dinv, w->partials, &np));
        PSP_TRY(reduce_fetch(w, np, 1, &delta));
      }
      if (delta == 0.0) QMRS_RET(-2);
      cc = xi1 * (delta / eps0);
      PSP_TRY(k_qmrs_pg(n, v1, kv, p, g, cc));
      PSP_TRY(csr_spmv_launch(Acsr, g, t, g, w->partials, &np, nullptr));
      PSP_TRY(reduce_fetch(w, np, 1, &eps0));
      beta = eps0 / delta;
      PSP_TRY(k_qmrs_v(n, t, v1, beta, w->partials, &np));
      PSP_TRY(reduce_fetch(w, np, 1, &rho1));
      rho1 = sqrt(rho1);
      xi1 = rho1;
      if (c0 * fabs(beta) == 0.0) QMRS_RET(-6);
      theta = rho1 / (c0 * fabs(beta));
      c1 = 1.0 / sqrt(theta * theta + 1.0);
      if (beta * (c0 * c0) == 0.0) QMRS_RET(-6);
      eta0 = -eta0 * rho0 * (c1 * c1) / (beta * (c0 * c0));
      tau = tau * theta * c1;
      if (rho1 == 0.0) QMRS_RET(-6);
      d1 = theta0 * c1;
      cc = d1 * d1;
      rho1inv = 1.0 / rho1;
      PSP_TRY(k_qmrs_dx(n, p, d, x, v1, kv, dinv, eta0, cc, rho1inv, w->partials, &np));
      PSP_TRY(reduce_fetch(w, np, 1, &delta));  // K v1 . v1 of the next iteration
      have_delta = true;
      if (xi1 == 0.0) QMRS_RET(-6);
      rho0 = rho1;
      *err = tau / res_init;
      c0 = c1;
      theta0 = theta;
    }
    if (K) {
      PSP_TRY(op_apply(K, x, wrk1));
      PSP_TRY(B.copy(wrk1, x));
    }
    *info = (*err < tol) ? 0 : -1;
    return PSP_OK;
  }
  while (*err > tol && *iter < maxit) {
    ++(*iter);
    if (eps0 == 0.0) QMRS_RET(-6);
    PSP_TRY(apply_or_copy(K, n, v1, wrk1));
    PSP_TRY(B.dot(wrk1, v1, &delta));
    if (delta == 0.0) QMRS_RET(-2);
    cc = xi1 * (delta / eps0);
    PSP_TRY(k_lin2(n, 1.0, v1, -cc, p, p));    // p = v1 - p*cc
    PSP_TRY(k_lin2(n, 1.0, wrk1, -cc, g, g));  // g = wrk1 - g*cc
    PSP_TRY(op_apply(A, g, t));
    PSP_TRY(B.dot(g, t, &eps0));
    beta = eps0 / delta;
    PSP_TRY(k_lin2(n, 1.0, t, -beta, v1, v1));  // v1 = t - v1*beta
    PSP_TRY(B.nrm2(v1, &rho1));
    xi1 = rho1;
    if (c0 * fabs(beta) == 0.0) QMRS_RET(-6);
    theta = rho1 / (c0 * fabs(beta));
    c1 = 1.0 / sqrt(theta * theta + 1.0);
    if (beta * (c0 * c0) == 0.0) QMRS_RET(-6);
    eta0 = -eta0 * rho0 * (c1 * c1) / (beta * (c0 * c0));
    tau = tau * theta * c1;
    if (rho1 == 0.0) QMRS_RET(-6);
    d1 = theta0 * c1;
    cc = d1 * d1;
    rho1inv = 1.0 / rho1;
    PSP_TRY(k_lin2(n, eta0, p, cc, d, d));  // d = p*eta0 + d*cc
    PSP_TRY(k_lin2(n, 1.0, x, 1.0, d, x));  // x += d
    PSP_TRY(k_scal(n, rho1inv, v1));        // v1 *= rho1inv
    if (xi1 == 0.0) QMRS_RET(-6);
    rho0 = rho1;
    *err = tau / res_init;
    c0 = c1;
    theta0 = theta;
  }
#undef QMRS_RET
  if (K) {
    PSP_TRY(op_apply(K, x, wrk1));
    PSP_TRY(B.copy(wrk1, x));
  }
  *info = (*err < tol) ? 0 : -1;
  return PSP_OK;
}

// pysparse/itsolvers/src/gmres.c:62-175 (rotations :40-61)
static void gen_rot(double dx, double dy, double *cs, double *sn) {
  if (dy == 0.0) {
    *cs = 1.0;
    *sn = 0.0;
  } else if (fabs(dy) > fabs(dx)) {
    const double temp = dx / dy;
    *sn = 1.0 / sqrt(1.0 + temp * temp);
    *cs = temp * *sn;
  } else {
    const double temp = dy / dx;
    *cs = 1.0 / sqrt(1.0 + temp * temp);
    *sn = temp * *cs;
  }
}
static void app_rot(double *dx, double *dy, double cs, double sn) {
  const double temp = cs * *dx + sn * *dy;
  *dy = -sn * *dx + cs * *dy;
  *dx = temp;
}

static int gmres_device(const psp_op *A, const psp_op *K, int n, double *x, const double *b,
                        double errtol, int it_max, int dim, int *info, int *it, double *relres) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  Blas B{w, n};
  if (dim < 1) return fail(PSP_EINVAL, "gmres: dim must be >= 1");
  DevVecs mem;
  std::vector<double *> V(dim + 1), W(dim);
  for (int i = 0; i <= dim; ++i) PSP_TRY(mem.alloc(n, &V[i]));
  for (int i = 0; i < dim; ++i) PSP_TRY(mem.alloc(n, &W[i]));
  const int m1 = dim + 1;
  std::vector<double> H((size_t)dim * m1), s(m1), cs(dim), sn(dim), hhost((size_t)dim + 2);
  double *hdev = nullptr;  // h[0..i] and ||w||^2 of the current column, on the device
  PSP_TRY(mem.alloc((size_t)dim + 2, &hdev));
  // PSP_GMRES_CHAIN (A/B; read per solve): 0 one read-back per Gram-Schmidt step, 1 the chain with a finishing launch per
  // step (round 4), 2 (default) the finishing reduction folded into the next step's kernel where a step is launch-bound
  const char *chain_env = psp::tuning_env("PSP_GMRES_CHAIN");
  const bool mgs_chain = !chain_env || atoi(chain_env) != 0;
  const bool mgs_fold = (!chain_env || atoi(chain_env) == 2) && vec_grid(*w, n) <= 2048;  // n <= 2^20
#define GH(i, j) (H[(size_t)(j) * m1 + (i)])
  int i, j, k, iter = 0;
  double beta, resid0 = 0.0, n2b, rel_resid = 0.0, d;
  *info = 0;
  PSP_TRY(B.nrm2(b, &n2b));
  if (n2b == 0.0) {
    PSP_TRY(B.zero(x));
    *relres = 0.0;
    *it = 0;
    return PSP_OK;
  }
  do {
    PSP_TRY(op_apply(A, x, V[0]));
    PSP_TRY(B.axpy(-1.0, b, V[0]));
    PSP_TRY(B.dot(V[0], V[0], &d));
    beta = sqrt(d);
    PSP_TRY(k_scal(n, -1.0 / beta, V[0]));
    if (iter == 0) resid0 = beta;
    for (i = 1; i < dim + 1; i++) s[i] = 0.0;
    s[0] = beta;
    i = -1;
    do {
      i++;
      iter++;
      PSP_TRY(apply_or_copy(K, n, V[i], W[i]));
      PSP_TRY(op_apply(A, W[i], V[i + 1]));
      // modified Gram-Schmidt (gmres.c:110-116): the axpy of step k and the dot of step k + 1 (at the end: the norm)
      // share one pass over V[i + 1]
      static const bool mgs_fused = [] {
        const char *e = psp::tuning_env("PSP_GMRES_FUSED");  // 0: one dot and one axpy kernel per step (A/B)
        return e ? atoi(e) != 0 : true;
      }();
      if (mgs_fused && mgs_chain) {
        // the whole Gram-Schmidt chain of the column enqueued at once (round 4): step k's axpy takes h[k] from where step
        // k - 1's reduction left it on the device; the host reads h[0..i] and the norm's square in ONE go at the end
        // (one scalar read-back per inner iteration instead of i + 2) -- the same operations on the same values
        int np;
        PSP_TRY(k_dot(n, V[i + 1], V[0], w->partials, &np));
        if (mgs_fold) {
          // round 5: step k adds the partial sums of step k - 1 itself (every workgroup, reduce_block: the finishing block's
          // bits) -- i + 3 launches for the column instead of 2 i + 4
          double *P[2] = {w->partials, w->partials + kMaxParts};
          for (k = 0; k <= i; k++)
            PSP_TRY(k_axpy_dot_chain(n, P[k & 1], np, hdev + k, V[k], V[i + 1], k < i ? V[k + 1] : nullptr, P[(k + 1) & 1], &np));
          PSP_TRY(finish_partials(P[(i + 1) & 1], np, 1, hdev + i + 1));
        } else {
          PSP_TRY(finish_partials(w->partials, np, 1, hdev));
          for (k = 0; k <= i; k++) {
            PSP_TRY(k_axpy_dot(n, 0.0, V[k], V[i + 1], k < i ? V[k + 1] : nullptr, w->partials, &np, hdev + k));
            PSP_TRY(finish_partials(w->partials, np, 1, hdev + k + 1));
          }
        }
        for (k = 0; k < i + 2; k += 16) PSP_TRY(fetch_scalars(hdev + k, std::min(16, i + 2 - k), hhost.data() + k));
        for (k = 0; k <= i; k++) GH(k, i) = hhost[k];
        d = hhost[i + 1];
      } else if (mgs_fused) {
        PSP_TRY(B.dot(V[i + 1], V[0], &d));
        for (k = 0; k <= i; k++) {
          GH(k, i) = d;
          int np;
          PSP_TRY(k_axpy_dot(n, -GH(k, i), V[k], V[i + 1], k < i ? V[k + 1] : nullptr, w->partials, &np));
          PSP_TRY(reduce_fetch(w, np, 1, &d));
        }
      } else {
        for (k = 0; k <= i; k++) {
          PSP_TRY(B.dot(V[i + 1], V[k], &d));
          GH(k, i) = d;
          PSP_TRY(B.axpy(-GH(k, i), V[k], V[i + 1]));
        }
        PSP_TRY(B.dot(V[i + 1], V[i + 1], &d));
      }
      GH(i + 1, i) = sqrt(d);
      PSP_TRY(k_scal(n, 1.0 / GH(i + 1, i), V[i + 1]));
      for (k = 0; k < i; k++) app_rot(&GH(k, i), &GH(k + 1, i), cs[k], sn[k]);
      gen_rot(GH(i, i), GH(i + 1, i), &cs[i], &sn[i]);
      app_rot(&GH(i, i), &GH(i + 1, i), cs[i], sn[i]);
      app_rot(&s[i], &s[i + 1], cs[i], sn[i]);
      rel_resid = fabs(s[i + 1]) / resid0;
      if (rel_resid <= errtol) break;
    } while (i + 1 < dim && iter + 1 <= it_max);
    for (j = i; j >= 0; j--) {
      s[j] /= GH(j, j);
      for (k = j - 1; k >= 0; k--) s[k] -= GH(k, j) * s[j];
    }
    for (j = 0; j <= i; j++) PSP_TRY(B.axpy(s[j], W[j], x));
  } while (rel_resid > errtol && iter + 1 <= it_max);
#undef GH
  PSP_TRY(op_apply(A, x, V[0]));
  PSP_TRY(B.axpy(-1.0, b, V[0]));
  PSP_TRY(B.dot(V[0], V[0], &d));
  *it = iter;
  *relres = sqrt(d) / resid0;
  return PSP_OK;
}


// ---------------------------------------------------------------- device-resident solver state for the
// row-partitioned driver (pysparse_amd/distributed.py).  Same state machine as pcg_async_loop_lazy /
// minres_async_loop, cut at the two reductions of an iteration so that an RCCL all-reduce (issued through
// torch.distributed on the same stream) can sit between "finish the local partial sums" and "take the
// reference's branches on the reduced values": the host enqueues whole batches of iterations and reads
// the state once per batch -- no host round trip per reduction.

struct psp_pcgstate {
  PcgDev *dev = nullptr;
  PcgDev *host = nullptr;  // pinned mirror
  double *hist_dev = nullptr;
  int hist_cap = 0;
};

struct psp_minresstate {
  MinresDev *dev = nullptr;
  MinresDev *host = nullptr;
  double *hist_dev = nullptr;
  int hist_cap = 0;
};

// {p.q, nonstag} reduced over all ranks: the stagnation verdict of the previous iteration and the exits at
// the head of this one (pcg.c:159-162, :101-112), then alpha (pcg.c:117-125)
__global__ void pcg_dist_scalar_xpq_kernel(PcgDev *st, const double *__restrict__ scal) {
  pcg_lazy_scalar_x(st, scal + 1);
  pcg_lazy_scalar_pq(st, scal);
}

// {r.r, r.z} reduced over all ranks: convergence test, next rho / beta (pcg.c:152-157, :99-112)
__global__ void pcg_dist_scalar_r_kernel(PcgDev *st, const double *__restrict__ scal, double *__restrict__ hist) {
  pcg_lazy_scalar_r(st, scal, hist);
}

__global__ void minres_dist_scalar_kernel(MinresDev *st, const double *__restrict__ scal, int op,
                                          double *__restrict__ hist) {
  if (op == kMrAlpha) minres_scalar_alpha(st, scal);
  else minres_scalar_beta(st, scal, hist);
}

extern "C" {

int psp_pcgstate_create(psp_pcgstate_t **out) {
  if (!out) return fail(PSP_EINVAL, "psp_pcgstate_create: NULL argument");
  PSP_TRY(ensure_device());
  psp_pcgstate *s = new psp_pcgstate();
  hipError_t e1 = hipMalloc((void **)&s->dev, sizeof(PcgDev));
  hipError_t e2 = hipHostMalloc((void **)&s->host, sizeof(PcgDev), hipHostMallocDefault);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    psp_pcgstate_destroy(s);
    return fail(PSP_ENOMEM, "psp_pcgstate_create: allocation failed");
  }
  *out = s;
  return PSP_OK;
}

int psp_pcgstate_destroy(psp_pcgstate_t *s) {
  if (!s) return PSP_OK;
  if (s->dev) (void)hipFree(s->dev);
  if (s->host) (void)hipHostFree(s->host);
  if (s->hist_dev) (void)hipFree(s->hist_dev);
  delete s;
  return PSP_OK;
}

int psp_pcgstate_init(psp_pcgstate_t *s, double n2b, double tolb, double normr0, double rho0, int maxit,
                      int want_hist) {
  PSP_API_GUARD;
  if (!s || maxit < 1) return fail(PSP_EINVAL, "psp_pcgstate_init: bad argument");
  if (want_hist && s->hist_cap < maxit + 1) {
    if (s->hist_dev) (void)hipFree(s->hist_dev);
    s->hist_dev = nullptr;
    s->hist_cap = 0;
    PSP_HIP(hipMalloc((void **)&s->hist_dev, sizeof(double) * ((size_t)maxit + 1)));
    s->hist_cap = maxit + 1;
  }
  if (want_hist) PSP_HIP(hipMemsetAsync(s->hist_dev, 0xff, sizeof(double) * ((size_t)maxit + 1), stream()));
  PSP_HIP(hipStreamSynchronize(stream()));  // the pinned mirror may still be the source of an earlier copy
  memset(s->host, 0, sizeof(PcgDev));
  s->host->rho = rho0;
  s->host->rho1 = 1.0;
  s->host->normr = normr0;
  s->host->tolb = tolb;
  s->host->n2b = n2b;
  s->host->it = 1;
  s->host->maxit = maxit;
  PSP_HIP(hipMemcpyAsync(s->dev, s->host, sizeof(PcgDev), hipMemcpyHostToDevice, stream()));
  return PSP_OK;
}

int psp_pcgstate_fetch(psp_pcgstate_t *s, psp_pcg_status_t *out) {
  PSP_API_GUARD;
  if (!s || !out) return fail(PSP_EINVAL, "psp_pcgstate_fetch: NULL argument");
  PSP_HIP(hipMemcpyAsync(s->host, s->dev, sizeof(PcgDev), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  const PcgDev *h = s->host;
  out->status = h->status;
  out->info = h->info;
  out->iter = h->iter;
  out->it = h->it;
  out->xpend = h->xpend;
  out->stag0 = h->stag0;
  out->pend_maxit = h->pend_maxit;
  out->relres = h->relres;
  out->normr = h->normr;
  out->n2b = h->n2b;
  out->alpha_x = h->alpha_x;
  return PSP_OK;
}

int psp_pcgstate_hist(psp_pcgstate_t *s, int first, int count, double *hist_host) {
  if (!s || !hist_host || !s->hist_dev || first < 0 || count < 0 || first + count > s->hist_cap)
    return fail(PSP_EINVAL, "psp_pcgstate_hist: bad argument");
  if (count) PSP_HIP(hipMemcpy(hist_host, s->hist_dev + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return PSP_OK;
}

int psp_kd_px_update(const psp_pcgstate_t *s, int n, const double *r_dev, const double *dinv_dev, double *p_dev,
                     double *x_dev, double *out_dev) {
  PSP_API_GUARD;
  if (!s || !r_dev || !p_dev || !x_dev || !out_dev) return fail(PSP_EINVAL, "psp_kd_px_update: NULL argument");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_px_update(n, r_dev, dinv_dev, p_dev, x_dev, w->partials, &np, s->dev));
  return finish_partials(w->partials + 2 * (size_t)kMaxParts, np, 1, out_dev);
}

int psp_kd_r_update(const psp_pcgstate_t *s, int n, const double *q_dev, const double *dinv_dev, double *r_dev,
                    double *out_dev) {
  PSP_API_GUARD;
  if (!s || !q_dev || !r_dev || !out_dev) return fail(PSP_EINVAL, "psp_kd_r_update: NULL argument");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_r_update(n, 0.0, q_dev, dinv_dev, r_dev, w->partials, &np, s->dev));
  return finish_partials(w->partials, np, 2, out_dev);
}

int psp_kd_csr_matvec_overlap(const psp_pcgstate_t *s, psp_csr_t *A, const double *x_dev, int x_offset,
                              double *y_dev, int row_a, int row_b, psp_wait_fn wait, void *ctx,
                              double *dot_out_dev) {
  PSP_API_GUARD_H(A);
  if (!s || !A || !x_dev || !y_dev || !dot_out_dev) return fail(PSP_EINVAL, "psp_kd_csr_matvec_overlap: NULL");
  if (x_offset < 0 || x_offset + A->nrows > A->ncols || row_a < 0 || row_b > A->nrows)
    return fail(PSP_EINVAL, "psp_kd_csr_matvec_overlap: row range / offset out of bounds");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np = 0;
  if (A->nrows == 0) {
    if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
    PSP_HIP(hipMemsetAsync(dot_out_dev, 0, sizeof(double), stream()));
    return PSP_OK;
  }
  PSP_TRY(csr_spmv_overlap(A, x_dev, y_dev, x_dev + x_offset, w->partials, &np, row_a, row_b, wait, ctx,
                           &s->dev->status));
  return finish_partials(w->partials, np, 1, dot_out_dev);
}

int psp_kd_pcg_scalar_xpq(psp_pcgstate_t *s, const double *scal_dev) {
  PSP_API_GUARD;
  if (!s || !scal_dev) return fail(PSP_EINVAL, "psp_kd_pcg_scalar_xpq: NULL argument");
  hipLaunchKernelGGL(pcg_dist_scalar_xpq_kernel, dim3(1), dim3(1), 0, stream(), s->dev, scal_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int psp_kd_pcg_scalar_r(psp_pcgstate_t *s, const double *scal_dev) {
  PSP_API_GUARD;
  if (!s || !scal_dev) return fail(PSP_EINVAL, "psp_kd_pcg_scalar_r: NULL argument");
  hipLaunchKernelGGL(pcg_dist_scalar_r_kernel, dim3(1), dim3(1), 0, stream(), s->dev, scal_dev, s->hist_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// ---- MINRES on row blocks (minres.c:96-193, two reductions per iteration)

int psp_minresstate_create(psp_minresstate_t **out) {
  if (!out) return fail(PSP_EINVAL, "psp_minresstate_create: NULL argument");
  PSP_TRY(ensure_device());
  psp_minresstate *s = new psp_minresstate();
  hipError_t e1 = hipMalloc((void **)&s->dev, sizeof(MinresDev));
  hipError_t e2 = hipHostMalloc((void **)&s->host, sizeof(MinresDev), hipHostMallocDefault);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    psp_minresstate_destroy(s);
    return fail(PSP_ENOMEM, "psp_minresstate_create: allocation failed");
  }
  *out = s;
  return PSP_OK;
}

int psp_minresstate_destroy(psp_minresstate_t *s) {
  if (!s) return PSP_OK;
  if (s->dev) (void)hipFree(s->dev);
  if (s->host) (void)hipHostFree(s->host);
  if (s->hist_dev) (void)hipFree(s->hist_dev);
  delete s;
  return PSP_OK;
}

int psp_minresstate_init(psp_minresstate_t *s, double norm_r0, double beta0, double errtol, int it_max,
                         int want_hist) {
  PSP_API_GUARD;
  if (!s || it_max < 1) return fail(PSP_EINVAL, "psp_minresstate_init: bad argument");
  if (want_hist && s->hist_cap < it_max + 1) {
    if (s->hist_dev) (void)hipFree(s->hist_dev);
    s->hist_dev = nullptr;
    s->hist_cap = 0;
    PSP_HIP(hipMalloc((void **)&s->hist_dev, sizeof(double) * ((size_t)it_max + 1)));
    s->hist_cap = it_max + 1;
  }
  if (want_hist) PSP_HIP(hipMemsetAsync(s->hist_dev, 0xff, sizeof(double) * ((size_t)it_max + 1), stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  MinresDev *h = s->host;
  memset(h, 0, sizeof(MinresDev));
  h->beta = beta0;
  h->beta_old = 1.0;
  h->c = 1.0;
  h->c_old = 1.0;
  h->eta = beta0;
  h->norm_rmr = norm_r0;
  h->norm_r0 = norm_r0;
  h->errtol = errtol;
  h->iter = 1;
  h->it_max = it_max;
  h->info = -1;
  PSP_HIP(hipMemcpyAsync(s->dev, h, sizeof(MinresDev), hipMemcpyHostToDevice, stream()));
  return PSP_OK;
}

int psp_minresstate_fetch(psp_minresstate_t *s, psp_minres_status_t *out) {
  PSP_API_GUARD;
  if (!s || !out) return fail(PSP_EINVAL, "psp_minresstate_fetch: NULL argument");
  PSP_HIP(hipMemcpyAsync(s->host, s->dev, sizeof(MinresDev), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  const MinresDev *h = s->host;
  out->status = h->status;
  out->stop = h->stop;
  out->info = h->info;
  out->iter = h->iter;
  out->relres = h->relres;
  out->norm_rmr = h->norm_rmr;
  return PSP_OK;
}

int psp_minresstate_hist(psp_minresstate_t *s, int first, int count, double *hist_host) {
  if (!s || !hist_host || !s->hist_dev || first < 0 || count < 0 || first + count > s->hist_cap)
    return fail(PSP_EINVAL, "psp_minresstate_hist: bad argument");
  if (count) PSP_HIP(hipMemcpy(hist_host, s->hist_dev + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return PSP_OK;
}

// the row-block form keeps v = y / beta as a vector of its own (its ghost entries are what the halo
// exchange moves); the single-GPU loop's scaled SpMV has no split-around-a-wait form
int psp_kd_minres_scale(const psp_minresstate_t *s, int n, const double *y_dev, double *v_dev) {
  PSP_API_GUARD;
  if (!s || !y_dev || !v_dev) return fail(PSP_EINVAL, "psp_kd_minres_scale: NULL argument");
  return k_scale_div(n, y_dev, 1.0, v_dev, s->dev);
}

int psp_kd_minres_matvec(const psp_minresstate_t *s, psp_csr_t *A, const double *v_dev, int v_offset,
                         double *av_dev, int row_a, int row_b, psp_wait_fn wait, void *ctx, double *dot_out_dev) {
  PSP_API_GUARD_H(A);
  if (!s || !A || !v_dev || !av_dev || !dot_out_dev) return fail(PSP_EINVAL, "psp_kd_minres_matvec: NULL");
  if (v_offset < 0 || v_offset + A->nrows > A->ncols || row_a < 0 || row_b > A->nrows)
    return fail(PSP_EINVAL, "psp_kd_minres_matvec: row range / offset out of bounds");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np = 0;
  if (A->nrows == 0) {  // a rank that owns no rows: nothing to multiply, alpha's local part is 0 (as in the PCG variant)
    if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
    PSP_HIP(hipMemsetAsync(dot_out_dev, 0, sizeof(double), stream()));
    return PSP_OK;
  }
  PSP_TRY(csr_spmv_overlap(A, v_dev, av_dev, v_dev + v_offset, w->partials, &np, row_a, row_b, wait, ctx,
                           &s->dev->skip));
  return finish_partials(w->partials, np, 1, dot_out_dev);
}

int psp_kd_minres_lanczos(const psp_minresstate_t *s, int n, const double *av_dev, const double *v_hat_dev,
                          double *v_hat_old_dev, const double *dinv_dev, double *y_dev, double *out_dev) {
  PSP_API_GUARD;
  if (!s || !av_dev || !v_hat_dev || !v_hat_old_dev || !out_dev)
    return fail(PSP_EINVAL, "psp_kd_minres_lanczos: NULL argument");
  if (dinv_dev && !y_dev) return fail(PSP_EINVAL, "psp_kd_minres_lanczos: y is needed with a preconditioner");
  Workspace *w;
  PSP_TRY(workspace(&w));
  int np;
  PSP_TRY(k_lanczos(n, av_dev, 0.0, 0.0, v_hat_dev, v_hat_old_dev, dinv_dev, y_dev, w->partials, &np, s->dev));
  return finish_partials(w->partials, np, 1, out_dev);
}

int psp_kd_minres_scalar(psp_minresstate_t *s, int which, const double *scal_dev) {
  PSP_API_GUARD;
  if (!s || !scal_dev || (which != 0 && which != 1)) return fail(PSP_EINVAL, "psp_kd_minres_scalar: bad argument");
  hipLaunchKernelGGL(minres_dist_scalar_kernel, dim3(1), dim3(1), 0, stream(), s->dev, scal_dev, which, s->hist_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int psp_kd_minres_wx(const psp_minresstate_t *s, int n, const double *v_dev, const double *w_dev,
                     double *w_old_dev, double *x_dev) {
  PSP_API_GUARD;
  if (!s || !v_dev || !w_dev || !w_old_dev || !x_dev) return fail(PSP_EINVAL, "psp_kd_minres_wx: NULL argument");
  return k_minres_wx(n, v_dev, 0.0, 0.0, 0.0, 0.0, w_dev, w_old_dev, x_dev, false, 1.0, s->dev);
}

}  // extern "C"

// ====================================================================== C ABI

// multi-device operands (psp_multi.hip): *multi = the row-partitioned matrix behind A (nullptr: an ordinary operand
// pair); K must then be absent or the jacobi of that very matrix (*jac).  Only psp_pcg / psp_minres accept them.
static int check_solver_args(const psp_op *A, const psp_op *K, int n, const void *x, const void *b,
                             int *info, int *iter, double *relres, psp_mcsr **multi = nullptr, bool *jac = nullptr) {
  if (!A || !x || !b || !info || !iter || !relres) return fail(PSP_EINVAL, "solver: NULL argument");
  if (n <= 0) return fail(PSP_EINVAL, "solver: n must be positive");
  if (A->n != n) return fail(PSP_EINVAL, "solver: operator order %d != n %d", A->n, n);
  if (K && K->n != n) return fail(PSP_EINVAL, "solver: preconditioner order %d != n %d", K->n, n);
  psp_mcsr *ma = (A->kind == PSP_OP_CSR && A->csr) ? A->csr->multi : nullptr;
  psp_mcsr *mk = (K && K->kind == PSP_OP_JACOBI && K->jac) ? K->jac->multi : nullptr;
  if (multi) *multi = ma;
  if (jac) *jac = mk != nullptr;
  if (!ma && !mk) return PSP_OK;
  if (!multi) return fail(PSP_EINVAL, "only pcg and minres (host vectors) run on a multi-device matrix");
  if (!ma || (K && mk != ma))
    return fail(PSP_EINVAL, "a multi-device matrix takes K = None or precon.jacobi of that same matrix");
  return PSP_OK;
}

// After a solve: did every application of the preconditioner produce a valid vector?  The SSOR brick sweeps report a
// sweep that gave up waiting through an error word that is read back asynchronously (psp_ssor.hip); the word is sticky
// across applications, so one waited-for look after the solver's last launch covers all of them.
static int precon_status(const psp_op *K) {
  if (!K || K->kind != PSP_OP_SSOR || !K->ssor) return PSP_OK;
  PSP_HIP(hipStreamSynchronize(stream()));
  return ssor_error_check(K->ssor);
}

extern "C" {

int psp_trim(void) {
  g_pool.trim();
  host_stage_trim();
  return PSP_OK;
}

int psp_op_from_csr(psp_csr_t *A, psp_op_t **out) {
  if (!A || !out) return fail(PSP_EINVAL, "psp_op_from_csr: NULL argument");
  if (A->nrows != A->ncols) return fail(PSP_EINVAL, "matrix is not square");
  psp_op *op = new psp_op();
  op->kind = PSP_OP_CSR;
  op->n = A->nrows;
  op->csr = A;
  *out = op;
  return PSP_OK;
}

int psp_op_from_sss(psp_sss_t *A, psp_op_t **out) {
  if (!A || !out) return fail(PSP_EINVAL, "psp_op_from_sss: NULL argument");
  psp_op *op = new psp_op();
  op->kind = PSP_OP_SSS;
  op->n = A->n;
  op->sss = A;
  *out = op;
  return PSP_OK;
}

int psp_op_from_jacobi(psp_jacobi_t *K, psp_op_t **out) {
  if (!K || !out) return fail(PSP_EINVAL, "psp_op_from_jacobi: NULL argument");
  psp_op *op = new psp_op();
  op->kind = PSP_OP_JACOBI;
  op->n = K->n;
  op->jac = K;
  *out = op;
  return PSP_OK;
}

int psp_op_from_ssor(psp_ssor_t *K, psp_op_t **out) {
  if (!K || !out) return fail(PSP_EINVAL, "psp_op_from_ssor: NULL argument");
  int n = 0;
  PSP_TRY(psp_ssor_info(K, &n, nullptr, nullptr));
  psp_op *op = new psp_op();
  op->kind = PSP_OP_SSOR;
  op->n = n;
  op->ssor = K;
  *out = op;
  return PSP_OK;
}

int psp_op_from_callback(int n, psp_host_apply_fn fn, void *ctx, psp_op_t **out) {
  if (!fn || !out || n <= 0) return fail(PSP_EINVAL, "psp_op_from_callback: bad argument");
  if (cpu_mode()) {  // host mode: the callback is applied to the solver's host vectors directly, no staging
    psp_op *op = new psp_op();
    op->kind = PSP_OP_CALLBACK;
    op->n = n;
    op->fn = fn;
    op->ctx = ctx;
    *out = op;
    return PSP_OK;
  }
  PSP_TRY(ensure_device());
  psp_op *op = new psp_op();
  op->kind = PSP_OP_CALLBACK;
  op->n = n;
  op->fn = fn;
  op->ctx = ctx;
  hipError_t e1 = hipHostMalloc((void **)&op->hx, sizeof(double) * (size_t)n, hipHostMallocDefault);
  hipError_t e2 = hipHostMalloc((void **)&op->hy, sizeof(double) * (size_t)n, hipHostMallocDefault);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    psp_op_destroy(op);
    return fail(PSP_ENOMEM, "psp_op_from_callback: pinned staging allocation failed");
  }
  *out = op;
  return PSP_OK;
}

int psp_op_destroy(psp_op_t *op) {
  if (!op) return PSP_OK;
  if (op->hx) (void)hipHostFree(op->hx);
  if (op->hy) (void)hipHostFree(op->hy);
  delete op;
  return PSP_OK;
}

// ---------------------------------------------------------------- jacobi

static int jacobi_from_diag_dev(int n, double *diag_dev_owned, double omega, int steps,
                                const psp_op *A, psp_jacobi_t **out) {
  // diag_dev_owned is turned into dinv in place
  Workspace *w;
  int rc = workspace(&w);
  int np = 0;
  double nsing = 0.0;
  if (rc == PSP_OK) rc = k_dinv(n, diag_dev_owned, omega, diag_dev_owned, w->partials, &np);
  if (rc == PSP_OK) rc = finish_partials(w->partials, np, 1, w->scal_dev);
  if (rc == PSP_OK) rc = fetch_scalars(w->scal_dev, 1, &nsing);
  if (rc == PSP_OK && nsing != 0.0)
    rc = fail(PSP_ESINGULAR, "diagonal element close to zero");  // preconmodule.c:395-397
  if (rc != PSP_OK) {
    (void)hipFree(diag_dev_owned);
    return rc;
  }
  (void)dinv_register(diag_dev_owned, n);  // constant diagonal: z = r .* dinv needs no dinv stream
  psp_jacobi *K = new psp_jacobi();
  K->n = n;
  K->omega = omega;
  K->steps = steps;
  K->dinv = diag_dev_owned;
  if (A) K->A = *A;
  if (steps > 1) {
    if (hipMalloc((void **)&K->temp, sizeof(double) * (size_t)n) != hipSuccess) {
      psp_jacobi_destroy(K);
      return fail(PSP_ENOMEM, "jacobi: temp allocation failed");
    }
  }
  *out = K;
  return PSP_OK;
}

int psp_jacobi_create_csr(psp_csr_t *A, double omega, int steps, psp_jacobi_t **out) {
  PSP_API_GUARD_H(A);
  if (!A || !out) return fail(PSP_EINVAL, "psp_jacobi_create_csr: NULL argument");
  if (A->nrows != A->ncols) return fail(PSP_EINVAL, "matrix is not square");
  if (steps < 1) return fail(PSP_EINVAL, "jacobi: steps must be >= 1");
  if (A->host) {
    std::vector<double> d((size_t)(A->nrows ? A->nrows : 1));
    PSP_TRY(cpu::csr_diagonal(A, d.data()));
    psp_op op;
    op.kind = PSP_OP_CSR;
    op.n = A->nrows;
    op.csr = A;
    return cpu::jacobi_create(A->nrows, d.data(), omega, steps, &op, out);
  }
  if (A->multi) {  // dinv lives with the row blocks; the handle says "jacobi of THIS matrix with THIS omega" -- every
                   // use of the handle re-establishes its own omega there (multi_jacobi_setup is a no-op when the
                   // factors in place are the handle's), so two handles with different omega never see each other's
    if (steps != 1) return fail(PSP_EINVAL, "jacobi of a multi-device matrix: steps must be 1");
    PSP_TRY(multi_jacobi_setup(A->multi, omega));
    psp_jacobi *K = new psp_jacobi();
    K->n = A->nrows;
    K->omega = omega;
    K->steps = 1;
    K->multi = A->multi;
    *out = K;
    return PSP_OK;
  }
  PSP_TRY(ensure_device());
  double *d;
  PSP_HIP(hipMalloc((void **)&d, sizeof(double) * (size_t)(A->nrows ? A->nrows : 1)));
  int rc = psp_csr_diagonal_dev(A, d);
  if (rc != PSP_OK) {
    (void)hipFree(d);
    return rc;
  }
  psp_op op;
  op.kind = PSP_OP_CSR;
  op.n = A->nrows;
  op.csr = A;
  return jacobi_from_diag_dev(A->nrows, d, omega, steps, &op, out);
}

int psp_jacobi_create_sss(psp_sss_t *A, double omega, int steps, psp_jacobi_t **out) {
  PSP_API_GUARD_H(A);
  if (!A || !out) return fail(PSP_EINVAL, "psp_jacobi_create_sss: NULL argument");
  if (steps < 1) return fail(PSP_EINVAL, "jacobi: steps must be >= 1");
  if (A->host) {
    psp_op op;
    op.kind = PSP_OP_SSS;
    op.n = A->n;
    op.sss = A;
    return cpu::jacobi_create(A->n, A->diag, omega, steps, &op, out);
  }
  PSP_TRY(ensure_device());
  double *d;
  PSP_HIP(hipMalloc((void **)&d, sizeof(double) * (size_t)(A->n ? A->n : 1)));
  hipError_t e = hipMemcpyAsync(d, A->diag, sizeof(double) * (size_t)A->n, hipMemcpyDeviceToDevice,
                                stream());
  if (e != hipSuccess) {
    (void)hipFree(d);
    return fail(PSP_ENODEV, "jacobi: %s", hipGetErrorString(e));
  }
  psp_op op;
  op.kind = PSP_OP_SSS;
  op.n = A->n;
  op.sss = A;
  return jacobi_from_diag_dev(A->n, d, omega, steps, &op, out);
}

int psp_jacobi_create_diag(int n, const double *diag_host, double omega, int steps,
                           const psp_op_t *A_or_null, psp_jacobi_t **out) {
  PSP_API_GUARD_OPS(A_or_null, nullptr);
  if (!diag_host || !out || n <= 0) return fail(PSP_EINVAL, "psp_jacobi_create_diag: bad argument");
  if (steps < 1) return fail(PSP_EINVAL, "jacobi: steps must be >= 1");
  if (steps > 1 && !A_or_null)
    return fail(PSP_EINVAL, "jacobi: steps > 1 needs the matrix operator");
  if (cpu_mode()) return cpu::jacobi_create(n, diag_host, omega, steps, A_or_null, out);
  PSP_TRY(ensure_device());
  double *d;
  PSP_HIP(hipMalloc((void **)&d, sizeof(double) * (size_t)n));
  hipError_t e = hipMemcpyAsync(d, diag_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice,
                                stream());
  if (e == hipSuccess) e = hipStreamSynchronize(stream());
  if (e != hipSuccess) {
    (void)hipFree(d);
    return fail(PSP_ENODEV, "jacobi: %s", hipGetErrorString(e));
  }
  return jacobi_from_diag_dev(n, d, omega, steps, A_or_null, out);
}

int psp_jacobi_destroy(psp_jacobi_t *K) {
  if (!K) return PSP_OK;
  if (K->host) return cpu::jacobi_destroy(K);
  if (K->multi) {
    delete K;
    return PSP_OK;
  }
  dinv_unregister(K->dinv);
  (void)hipFree(K->dinv);
  if (K->temp) (void)hipFree(K->temp);
  delete K;
  return PSP_OK;
}

int psp_jacobi_shape(const psp_jacobi_t *K, int *n) {
  if (!K || !n) return fail(PSP_EINVAL, "psp_jacobi_shape: NULL argument");
  *n = K->n;
  return PSP_OK;
}

int psp_jacobi_precon_dev(psp_jacobi_t *K, const double *x_dev, double *y_dev) {
  PSP_API_GUARD_JAC(K);
  if (!K || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_jacobi_precon_dev: NULL argument");
  if (K->multi) return fail(PSP_EINVAL, "psp_jacobi_precon_dev is not available on a multi-device matrix");
  return jacobi_apply_dev(K, x_dev, y_dev);
}

int psp_jacobi_precon(psp_jacobi_t *K, const double *x_host, double *y_host) {
  PSP_API_GUARD_JAC(K);
  if (!K || !x_host || !y_host) return fail(PSP_EINVAL, "psp_jacobi_precon: NULL argument");
  if (K->host) return cpu::jacobi_precon(K, x_host, y_host);
  if (K->multi) {
    PSP_TRY(multi_jacobi_setup(K->multi, K->omega));
    return multi_jacobi_apply_host(K->multi, x_host, y_host);
  }
  PSP_TRY(ensure_device());
  DevVecs mem;
  double *x, *y;
  PSP_TRY(mem.alloc(K->n, &x));
  PSP_TRY(mem.alloc(K->n, &y));
  const size_t bytes = sizeof(double) * (size_t)K->n;
  PSP_HIP(hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_TRY(jacobi_apply_dev(K, x, y));
  PSP_HIP(hipMemcpyAsync(y_host, y, bytes, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

// ---------------------------------------------------------------- solvers

int psp_pcg_dev(const psp_op_t *A, const psp_op_t *K, int n, double *x_dev, const double *b_dev,
                double tol, int maxit, int *info, int *iter, double *relres, double *hist_host) {
  PSP_API_GUARD_OPS(A, K);
  PSP_TRY(check_solver_args(A, K, n, x_dev, b_dev, info, iter, relres));
  PSP_TRY(ensure_device());
  PSP_TRY(pcg_device(A, K, n, x_dev, b_dev, tol, maxit, info, iter, relres, hist_host));
  return precon_status(K);
}

int psp_pcg(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
            double tol, int maxit, int *info, int *iter, double *relres, double *hist_host) {
  PSP_API_GUARD_OPS(A, K);
  psp_mcsr *multi = nullptr;
  bool multi_jac = false;
  PSP_TRY(check_solver_args(A, K, n, x_host, b_host, info, iter, relres, &multi, &multi_jac));
  if (multi) {
    if (multi_jac) PSP_TRY(multi_jacobi_setup(multi, K->jac->omega));  // this handle's omega, whatever was used last
    note_solve("pcg_multi", -1, -1, 0);
    return multi_pcg(multi, multi_jac, n, x_host, b_host, tol, maxit, info, iter, relres, hist_host);
  }
  if (cpu_mode()) note_solve("pcg_cpu_mode", 0, -1, 0);
  if (cpu_mode()) return cpu::pcg(A, K, n, x_host, b_host, tol, maxit, info, iter, relres, hist_host);
  PSP_TRY(ensure_device());
  DevVecs mem;
  double *x, *b;
  PSP_TRY(mem.alloc(n, &x));
  PSP_TRY(mem.alloc(n, &b));
  const size_t bytes = sizeof(double) * (size_t)n;
  PSP_HIP(hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_HIP(hipMemcpyAsync(b, b_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_TRY(pcg_device(A, K, n, x, b, tol, maxit, info, iter, relres, hist_host));
  PSP_HIP(hipMemcpyAsync(x_host, x, bytes, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return precon_status(K);
}

int psp_minres_dev(const psp_op_t *A, const psp_op_t *K, int n, double *x_dev,
                   const double *b_dev, double tol, int maxit, int *info, int *iter,
                   double *relres, double *hist_host) {
  PSP_API_GUARD_OPS(A, K);
  PSP_TRY(check_solver_args(A, K, n, x_dev, b_dev, info, iter, relres));
  PSP_TRY(ensure_device());
  PSP_TRY(minres_device(A, K, n, x_dev, b_dev, tol, maxit, info, iter, relres, hist_host));
  return precon_status(K);
}

int psp_minres(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
               double tol, int maxit, int *info, int *iter, double *relres, double *hist_host) {
  PSP_API_GUARD_OPS(A, K);
  psp_mcsr *multi = nullptr;
  bool multi_jac = false;
  PSP_TRY(check_solver_args(A, K, n, x_host, b_host, info, iter, relres, &multi, &multi_jac));
  if (multi) {
    if (multi_jac) PSP_TRY(multi_jacobi_setup(multi, K->jac->omega));
    note_solve("minres_multi", -1, -1, 0);
    return multi_minres(multi, multi_jac, n, x_host, b_host, tol, maxit, info, iter, relres, hist_host);
  }
  if (cpu_mode()) note_solve("minres_cpu_mode", 0, -1, 0);
  if (cpu_mode()) return cpu::minres(A, K, n, x_host, b_host, tol, maxit, info, iter, relres, hist_host);
  PSP_TRY(ensure_device());
  DevVecs mem;
  double *x, *b;
  PSP_TRY(mem.alloc(n, &x));
  PSP_TRY(mem.alloc(n, &b));
  const size_t bytes = sizeof(double) * (size_t)n;
  PSP_HIP(hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_HIP(hipMemcpyAsync(b, b_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_TRY(minres_device(A, K, n, x, b, tol, maxit, info, iter, relres, hist_host));
  PSP_HIP(hipMemcpyAsync(x_host, x, bytes, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return precon_status(K);
}

// ---------------------------------------------------------------- cgs / bicgstab / qmrs / gmres

#define PSP_HOST_SOLVER(NAME, CALL)                                                              \
  int NAME(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,    \
           double tol, int maxit, int *info, int *iter, double *relres) {                        \
    PSP_API_GUARD_OPS(A, K);                                                                     \
    PSP_TRY(check_solver_args(A, K, n, x_host, b_host, info, iter, relres));                     \
    PSP_TRY(ensure_device());                                                                    \
    DevVecs mem;                                                                                 \
    double *x, *b;                                                                               \
    PSP_TRY(mem.alloc(n, &x));                                                                   \
    PSP_TRY(mem.alloc(n, &b));                                                                   \
    const size_t bytes = sizeof(double) * (size_t)n;                                             \
    PSP_HIP(hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream()));                  \
    PSP_HIP(hipMemcpyAsync(b, b_host, bytes, hipMemcpyHostToDevice, stream()));                  \
    PSP_TRY(CALL);                                                                               \
    PSP_HIP(hipMemcpyAsync(x_host, x, bytes, hipMemcpyDeviceToHost, stream()));                  \
    PSP_HIP(hipStreamSynchronize(stream()));                                                     \
    return precon_status(K);                                                                     \
  }

PSP_HOST_SOLVER(psp_cgs, cgs_device(A, K, n, x, b, tol, maxit, info, iter, relres))
PSP_HOST_SOLVER(psp_bicgstab, bicgstab_device(A, K, n, x, b, tol, maxit, info, iter, relres))
PSP_HOST_SOLVER(psp_qmrs, qmrs_device(A, K, n, x, b, tol, maxit, info, iter, relres))
#undef PSP_HOST_SOLVER

int psp_gmres(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
              double tol, int maxit, int dim, int *info, int *iter, double *relres) {
  PSP_API_GUARD_OPS(A, K);
  PSP_TRY(check_solver_args(A, K, n, x_host, b_host, info, iter, relres));
  PSP_TRY(ensure_device());
  DevVecs mem;
  double *x, *b;
  PSP_TRY(mem.alloc(n, &x));
  PSP_TRY(mem.alloc(n, &b));
  const size_t bytes = sizeof(double) * (size_t)n;
  PSP_HIP(hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_HIP(hipMemcpyAsync(b, b_host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_TRY(gmres_device(A, K, n, x, b, tol, maxit, dim, info, iter, relres));
  PSP_HIP(hipMemcpyAsync(x_host, x, bytes, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return precon_status(K);
}

}  // extern "C"
