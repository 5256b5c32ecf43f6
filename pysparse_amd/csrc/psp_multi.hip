// psp_multi.hip -- row-partitioned operators and PCG / MINRES on a LIST OF DEVICES, one process.
//
// SURVEY.md section 8b asks the C ABI for "multi-GPU variants taking a device list", section 8e for the row-range
// partition with two packed all-reduces per PCG iteration.  pysparse_amd/distributed.py does that with one process
// per GPU (torch.distributed); this file does it behind the drop-in boundary: psp_csr_poisson_multi /
// psp_csr_create_multi return an ordinary psp_csr_t handle whose rows live on several devices, and psp_csr_matvec,
// psp_jacobi_create_csr, psp_pcg and psp_minres work on it -- so `krylov.pcg(A, b, x, tol, maxit, precon.jacobi(A))`
// of an unchanged user script runs on every GPU of the node when A was created with `devices=[...]`.
//
// One host thread drives all ranks (a rank = one entry of the device list; entries may repeat: ranks sharing a GPU,
// the rehearsal this one-GPU test pool allows).  Per rank: a compute stream, a copy stream, its row block as a
// psp_csr with columns renumbered into [ghost_lo | owned | ghost_hi], its vector slices.
// The loops are the device-resident state machines the torch driver uses (psp_kd_*, psp_solvers.hip; reference
// loops pcg.c:91-163, minres.c:96-193): per iteration the host only enqueues, for every rank,
//   px update -> [ghost planes: peer copies on the copy stream, overlapped with the interior rows] -> SpMV ->
//   all-reduce #1 {p.q, nonstag} -> scalar step -> r update -> all-reduce #2 {r.r, r.z} -> scalar step
// and reads rank 0's state once per 16 iterations.  Cross-rank ordering is by events only (no host waits):
//   evP[q]  q's vector is final (and its send buffers packed)      -> waited for by the copy streams of q's readers
//   evH[r]  r's ghost entries have arrived                         -> waited for by r's boundary rows, and by the
//           NEXT overwrite of the vectors r copied from (its writers' compute streams)
// Reductions: RCCL (ncclCommInitAll + ncclAllReduce in stream order, grouped over the ranks; librccl is dlopen'ed on
// first use so that single-GPU users never load it) when every rank has its own device; otherwise -- ranks sharing
// a device, or PSP_MULTI_REDUCE=local under PSP_TUNING=1 -- one small kernel on rank 0's stream that adds the
// ranks' operands in rank order through peer pointers and writes the sum back to all of them.
// Unmeasured on more than one GPU: the pool's boxes have one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "psp_internal.h"

using psp::fail;

extern "C" int psp_csr_diagonal_dev(const psp_csr_t *A, double *diag_dev);  // psp_csr.hip (internal)

namespace {

struct Link {              // one neighbour this rank RECEIVES from
  int q = -1;              // sending rank
  int recv_off = 0;        // where its entries land in my extended vector
  int count = 0;
  int send_off = -1;       // >= 0: they are q's owned entries [send_off, send_off + count) (slab planes) ...
  int *send_idx = nullptr;     // ... else q's owned-local indices (on q's device), packed by psp_k_gather
  double *send_buf = nullptr;  // into this buffer (on q's device)
};

struct RankOp {
  int dev = 0, rank = 0;
  hipStream_t s = nullptr, c = nullptr;
  hipEvent_t evP = nullptr, evH = nullptr, evA = nullptr;
  psp_csr *A = nullptr;
  int n = 0, ghost_lo = 0, ghost_hi = 0, n_ext = 0;
  int64_t row_lo = 0;
  int ia = 0, ib = 0;        // rows [ia, ib) reference no ghost entry
  std::vector<Link> links;   // what I receive
  std::vector<int> readers;  // ranks that copy from me
  double *scal = nullptr;    // 16 doubles: where the phase kernels leave their local sums
  double *dinv = nullptr;    // jacobi: omega / diag of my rows
};

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // the copy the process already has (torch ships one) before the system's
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (r.lib) break;
    }
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) {
      r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
      r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
      if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd) r.lib = nullptr;
    }
  }
  return r.lib ? &r : nullptr;
}

__global__ void multi_fold_kernel(double *const *__restrict__ ptrs, int nranks, int off, int cnt) {
  const int j = threadIdx.x;
  if (j >= cnt) return;
  double s = ptrs[0][off + j];
  for (int r = 1; r < nranks; ++r) s += ptrs[r][off + j];  // rank order: the same sum on every run
  for (int r = 0; r < nranks; ++r) ptrs[r][off + j] = s;
}

// diagonal of a row block whose columns live in extended-vector coordinates: A[i, i] sits in column p_offset + i
__global__ void block_diag_kernel(int nrows, int p_offset, const int *__restrict__ ind, const int *__restrict__ col,
                                  const double *__restrict__ val, double *__restrict__ diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  double d = 0.0;
  for (int k = ind[i]; k < ind[i + 1]; ++k)
    if (col[k] == p_offset + i) d = val[k];
  diag[i] = d;
}

}  // namespace

struct psp_mcsr {
  int nranks = 0, distinct = 0;
  int n_global = 0;
  int64_t nnz = 0;
  std::vector<RankOp> r;
  bool use_rccl = false;
  std::vector<ncclComm_t> comms;
  double **scal_ptrs = nullptr;  // on rank 0's device: the ranks' scal arrays (local fold)
  hipEvent_t evB = nullptr;
  bool has_dinv = false;
  double jac_omega = 0.0;
  char kind[32] = "";
};

namespace {

#define M_HIP(call) PSP_HIP(call)

int use(const psp_mcsr *M, int r) { return psp::use_device(M->r[r].dev, M->r[r].s, r + 1); }

struct DeviceRestore {  // the caller's device / stream / workspace come back when a multi entry point returns
  psp::ThreadCtxSave saved;
  DeviceRestore() : saved(psp::save_thread_ctx()) {}
  ~DeviceRestore() { psp::restore_thread_ctx(saved); }
};

int wait_halo(void *ctx) {  // psp_wait_fn of the overlapped SpMV: the boundary rows wait for the ghost copies in stream order
  RankOp *R = (RankOp *)ctx;
  if (psp::shake(R->s, psp::kShakeBoundary, R->rank) != PSP_OK) return 1;
  return hipStreamWaitEvent(R->s, R->evH, 0) == hipSuccess ? 0 : 1;
}

// delay injection on every rank's compute stream (a kernel is launched with its stream's device current)
int shake_ranks(psp_mcsr *M, int point) {
  if (!psp::shake_armed()) return PSP_OK;
  for (int r = 0; r < M->nranks; ++r) {
    M_HIP(hipSetDevice(M->r[r].dev));
    PSP_TRY(psp::shake(M->r[r].s, point, r));
  }
  return PSP_OK;
}

// before rank q overwrites a vector its readers copy from (or repacks its send buffers): their last copies must be done
int guard_overwrite(psp_mcsr *M, int q) {
  RankOp &Q = M->r[q];
  for (int t : Q.readers) M_HIP(hipStreamWaitEvent(Q.s, M->r[t].evH, 0));
  PSP_TRY(psp::shake(Q.s, psp::kShakeOverwrite, q));
  return PSP_OK;
}

// ghost exchange of the extended vectors vext[r]: packs / marks ready on the compute streams, copies on the copy streams;
// afterwards evH[r] says "r's ghosts are in"
int exchange(psp_mcsr *M, double *const *vext) {
  if (M->nranks == 1) return PSP_OK;
  for (int q = 0; q < M->nranks; ++q) {
    RankOp &Q = M->r[q];
    PSP_TRY(use(M, q));
    for (int t : Q.readers)
      for (Link &L : M->r[t].links)
        if (L.q == q && L.send_idx) PSP_TRY(psp_k_gather(L.count, L.send_idx, vext[q] + Q.ghost_lo, L.send_buf));
    PSP_TRY(psp::shake(Q.s, psp::kShakePack, q));
    M_HIP(hipEventRecord(Q.evP, Q.s));
  }
  for (int t = 0; t < M->nranks; ++t) {
    RankOp &T = M->r[t];
    M_HIP(hipSetDevice(T.dev));
    // the receiver's own stream first: what it enqueued on its ghost zone before this exchange (the clearing of a fresh
    // vector, the previous product's reads) must be over before a halo lands there -- the copy stream is ordered against
    // the SENDER by evP below, and nothing else ordered it against the receiver (round 4: a 3-rank product came back
    // with a ghost zone zeroed after the halo had arrived)
    // (ordering edges of the whole driver: DESIGN.md section 5; psp::shake moves the streams against each other at
    // every one of them, tests/test_gpu_shake.py; kShakeRevertGhostWait takes this wait out again to show that the
    // stress finds the race it was added for)
    PSP_TRY(psp::shake(T.c, psp::kShakeCopyPre, t));
    if (!T.links.empty() && !psp::shake_revert(psp::kShakeRevertGhostWait)) M_HIP(hipStreamWaitEvent(T.c, T.evP, 0));
    for (Link &L : T.links) {
      RankOp &Q = M->r[L.q];
      M_HIP(hipStreamWaitEvent(T.c, Q.evP, 0));
      const double *src = L.send_idx ? L.send_buf : vext[L.q] + Q.ghost_lo + L.send_off;
      double *dst = vext[t] + L.recv_off;
      const size_t bytes = sizeof(double) * (size_t)L.count;
      if (Q.dev == T.dev)
        M_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, T.c));
      else
        M_HIP(hipMemcpyPeerAsync(dst, T.dev, src, Q.dev, bytes, T.c));
    }
    PSP_TRY(psp::shake(T.c, psp::kShakeCopyPost, t));
    M_HIP(hipEventRecord(T.evH, T.c));
    PSP_TRY(psp::shake(T.s, psp::kShakePosted, t));
  }
  return PSP_OK;
}

// sum of scal[off .. off+cnt) over the ranks, left on every rank, in stream order
int allreduce(psp_mcsr *M, int off, int cnt) {
  if (M->nranks == 1) return PSP_OK;
  PSP_TRY(shake_ranks(M, psp::kShakeReducePre));
  if (M->use_rccl) {
    Rccl *R = rccl();
    ncclResult_t e = R->GroupStart();
    for (int r = 0; r < M->nranks && e == ncclSuccess; ++r)
      e = R->AllReduce(M->r[r].scal + off, M->r[r].scal + off, (size_t)cnt, ncclDouble, ncclSum, M->comms[r], M->r[r].s);
    const ncclResult_t e2 = R->GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess)
      return fail(PSP_ENODEV, "RCCL all-reduce failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?");
    return shake_ranks(M, psp::kShakeReducePost);
  }
  for (int r = 1; r < M->nranks; ++r) {
    M_HIP(hipSetDevice(M->r[r].dev));
    M_HIP(hipEventRecord(M->r[r].evA, M->r[r].s));
  }
  PSP_TRY(use(M, 0));
  for (int r = 1; r < M->nranks; ++r) M_HIP(hipStreamWaitEvent(M->r[0].s, M->r[r].evA, 0));
  hipLaunchKernelGGL(multi_fold_kernel, dim3(1), dim3(64), 0, M->r[0].s, M->scal_ptrs, M->nranks, off, cnt);
  PSP_LAUNCH_CHECK();
  PSP_TRY(psp::shake(M->r[0].s, psp::kShakeReduceMid, 0));  // (rank 0's device is current: use(M, 0) above)
  M_HIP(hipEventRecord(M->evB, M->r[0].s));
  for (int r = 1; r < M->nranks; ++r) {
    M_HIP(hipSetDevice(M->r[r].dev));
    M_HIP(hipStreamWaitEvent(M->r[r].s, M->evB, 0));
  }
  return shake_ranks(M, psp::kShakeReducePost);
}

int fetch0(psp_mcsr *M, int off, int cnt, double *host) {  // rank 0's copy of reduced scalars (synchronises its stream)
  PSP_TRY(use(M, 0));
  M_HIP(hipMemcpyAsync(host, M->r[0].scal + off, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, M->r[0].s));
  M_HIP(hipStreamSynchronize(M->r[0].s));
  return PSP_OK;
}

int sync_all(psp_mcsr *M) {
  for (int r = 0; r < M->nranks; ++r) {
    M_HIP(hipSetDevice(M->r[r].dev));
    M_HIP(hipStreamSynchronize(M->r[r].s));
    M_HIP(hipStreamSynchronize(M->r[r].c));
  }
  return PSP_OK;
}

// per-rank device vectors of one solve; freed with it
struct Vecs {
  std::vector<double *> all;
  ~Vecs() {
    for (double *p : all) (void)hipFree(p);
  }
  int get(psp_mcsr *M, int r, size_t n, double **out, bool zero = true) {
    M_HIP(hipSetDevice(M->r[r].dev));
    double *p = nullptr;
    if (hipMalloc((void **)&p, sizeof(double) * (n ? n : 1)) != hipSuccess) {
      (void)hipGetLastError();
      (void)psp_trim();
      M_HIP(hipMalloc((void **)&p, sizeof(double) * (n ? n : 1)));
    }
    all.push_back(p);
    if (zero) {
      PSP_TRY(psp::shake(M->r[r].s, psp::kShakeClear, r));
      M_HIP(hipMemsetAsync(p, 0, sizeof(double) * (n ? n : 1), M->r[r].s));
    }
    *out = p;
    return PSP_OK;
  }
};

int scatter_host(psp_mcsr *M, const double *host, ptrdiff_t inc, std::vector<double *> &dst, int dst_off_is_ghost) {
  std::vector<double> pack;
  for (int r = 0; r < M->nranks; ++r) {
    RankOp &R = M->r[r];
    M_HIP(hipSetDevice(R.dev));
    const double *src = host + R.row_lo * inc;
    if (inc != 1) {
      pack.resize((size_t)R.n);
      for (int i = 0; i < R.n; ++i) pack[i] = src[(ptrdiff_t)i * inc];
      src = pack.data();
    }
    PSP_TRY(psp::shake(R.s, psp::kShakeScatter, r));
    M_HIP(hipMemcpyAsync(dst[r] + (dst_off_is_ghost ? R.ghost_lo : 0), src, sizeof(double) * (size_t)R.n,
                         hipMemcpyHostToDevice, R.s));
    if (inc != 1) M_HIP(hipStreamSynchronize(R.s));  // pack is reused
  }
  return PSP_OK;
}

int gather_host(psp_mcsr *M, const std::vector<double *> &src, double *host, ptrdiff_t inc) {
  std::vector<double> pack;
  for (int r = 0; r < M->nranks; ++r) {
    RankOp &R = M->r[r];
    M_HIP(hipSetDevice(R.dev));
    if (inc == 1) {
      M_HIP(hipMemcpyAsync(host + R.row_lo, src[r], sizeof(double) * (size_t)R.n, hipMemcpyDeviceToHost, R.s));
    } else {
      pack.resize((size_t)R.n);
      M_HIP(hipMemcpyAsync(pack.data(), src[r], sizeof(double) * (size_t)R.n, hipMemcpyDeviceToHost, R.s));
      M_HIP(hipStreamSynchronize(R.s));
      for (int i = 0; i < R.n; ++i) host[(R.row_lo + i) * inc] = pack[i];
    }
  }
  return sync_all(M);
}

// y[r] = A_r * vext[r] (owned rows) after a ghost exchange, not overlapped (set-up products)
int matvec_plain(psp_mcsr *M, double *const *vext, double *const *y) {
  PSP_TRY(exchange(M, vext));
  for (int r = 0; r < M->nranks; ++r) {
    RankOp &R = M->r[r];
    PSP_TRY(use(M, r));
    if (M->nranks > 1) M_HIP(hipStreamWaitEvent(R.s, R.evH, 0));
    if (R.n) PSP_TRY(psp_csr_matvec_dev(R.A, vext[r], y[r]));
  }
  return PSP_OK;
}

int finish_setup(psp_mcsr *M, const int *devices, int ndev) {
  std::vector<int> uniq(devices, devices + ndev);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  M->distinct = (int)uniq.size();
  for (int a : uniq)
    for (int b : uniq)
      if (a != b) {
        M_HIP(hipSetDevice(a));
        int can = 0;
        M_HIP(hipDeviceCanAccessPeer(&can, a, b));
        if (!can) return fail(PSP_ENODEV, "device %d cannot access device %d: no peer path for the halo copies", a, b);
        const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(PSP_ENODEV, "hipDeviceEnablePeerAccess(%d -> %d): %s", a, b, hipGetErrorString(e));
        (void)hipGetLastError();
      }
  for (int r = 0; r < M->nranks; ++r) {
    RankOp &R = M->r[r];
    M_HIP(hipSetDevice(R.dev));
    M_HIP(hipMalloc((void **)&R.scal, sizeof(double) * 16));
    M_HIP(hipMemset(R.scal, 0, sizeof(double) * 16));
    for (Link &L : R.links) M->r[L.q].readers.push_back(r);
  }
  // how the reductions travel
  const char *e = psp::tuning_env("PSP_MULTI_REDUCE");
  const bool force_local = e && !strcmp(e, "local");
  const bool force_rccl = e && !strcmp(e, "rccl");
  const bool all_distinct = M->distinct == M->nranks;
  if (!force_local && all_distinct && (M->nranks > 1 || force_rccl)) {
    Rccl *R = rccl();
    if (R) {
      M->comms.resize(M->nranks);
      const ncclResult_t rc = R->CommInitAll(M->comms.data(), M->nranks, devices);
      if (rc == ncclSuccess) {
        M->use_rccl = true;
      } else {
        M->comms.clear();
        if (force_rccl) return fail(PSP_ENODEV, "ncclCommInitAll failed: %s", R->GetErrorString ? R->GetErrorString(rc) : "?");
      }
    } else if (force_rccl) {
      return fail(PSP_ENODEV, "librccl.so could not be loaded");
    }
  } else if (force_rccl) {
    return fail(PSP_EINVAL, "RCCL needs one device per rank (the device list repeats a device)");
  }
  if (!M->use_rccl && M->nranks > 1) {
    std::vector<double *> ptrs(M->nranks);
    for (int r = 0; r < M->nranks; ++r) ptrs[r] = M->r[r].scal;
    M_HIP(hipSetDevice(M->r[0].dev));
    M_HIP(hipMalloc((void **)&M->scal_ptrs, sizeof(double *) * M->nranks));
    M_HIP(hipMemcpy(M->scal_ptrs, ptrs.data(), sizeof(double *) * M->nranks, hipMemcpyHostToDevice));
    M_HIP(hipEventCreateWithFlags(&M->evB, hipEventDisableTiming));
  }
  return PSP_OK;
}

int new_ranks(psp_mcsr *M, const int *devices, int ndev) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
    return fail(PSP_ENODEV, "no HIP device available; libpysparse_hip has no CPU fallback");
  M->nranks = ndev;
  M->r.resize(ndev);
  for (int r = 0; r < ndev; ++r) {
    if (devices[r] < 0 || devices[r] >= cnt) return fail(PSP_EINVAL, "device %d out of range (0..%d)", devices[r], cnt - 1);
    RankOp &R = M->r[r];
    R.dev = devices[r];
    R.rank = r;
    M_HIP(hipSetDevice(R.dev));
    M_HIP(hipStreamCreateWithFlags(&R.s, hipStreamNonBlocking));
    M_HIP(hipStreamCreateWithFlags(&R.c, hipStreamNonBlocking));
    M_HIP(hipEventCreateWithFlags(&R.evP, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&R.evH, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&R.evA, hipEventDisableTiming));
  }
  return PSP_OK;
}

void row_range(int64_t n, int world, int rank, int64_t *lo, int64_t *hi) {  // distributed.py row_range
  const int64_t base = n / world, rem = n % world;
  *lo = rank * base + std::min<int64_t>(rank, rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

// ---- the partition of a general CSR matrix, pure host code (no device is touched): what rank r of ndev owns, which ghost
// entries it needs from whom, its columns in [ghost_lo | owned | ghost_hi] numbering, and its widest ghost-free row range.
// Exported for the CPU tests as psp_multi_plan (the twin of general_halo_plan in pysparse_amd/distributed.py).
struct HostLink {
  int q, recv_off, count;
  int send_off;          // >= 0: q's owned entries [send_off, send_off + count); -1: the index list below
  std::vector<int> idx;  // q's owned-local indices, ascending
};
struct HostPlan {
  int64_t lo = 0, hi = 0;
  int n = 0, ghost_lo = 0, ghost_hi = 0, ia = 0, ib = 0;
  std::vector<int> ghosts;  // sorted global ids
  std::vector<int> lind, lcol;
  std::vector<HostLink> links;
};

int plan_block(int nrows, int ncols, const int *ind, const int *col, int ndev, int r, HostPlan *P) {
  std::vector<int64_t> lo(ndev), hi(ndev);
  for (int q = 0; q < ndev; ++q) row_range(nrows, ndev, q, &lo[q], &hi[q]);
  auto owner = [&](int g) {
    int a = 0, b = ndev - 1;
    while (a < b) {
      const int m = (a + b) / 2;
      if (g >= hi[m]) a = m + 1; else b = m;
    }
    return a;
  };
  P->lo = lo[r];
  P->hi = hi[r];
  P->n = (int)(hi[r] - lo[r]);
  const int a = ind[lo[r]], b = ind[hi[r]];
  std::vector<int> &g = P->ghosts;
  g.clear();
  for (int k = a; k < b; ++k) {
    if (col[k] < 0 || col[k] >= ncols) return fail(PSP_EINVAL, "column index %d out of range", col[k]);
    if (col[k] < lo[r] || col[k] >= hi[r]) g.push_back(col[k]);
  }
  std::sort(g.begin(), g.end());
  g.erase(std::unique(g.begin(), g.end()), g.end());
  P->ghost_lo = (int)(std::lower_bound(g.begin(), g.end(), (int)lo[r]) - g.begin());
  P->ghost_hi = (int)g.size() - P->ghost_lo;
  P->lind.assign((size_t)P->n + 1, 0);
  P->lcol.assign((size_t)(b - a), 0);
  // widest run of rows that reference no ghost entry (overlaps the exchange with the product)
  int best_a = 0, best_b = 0, run_a = 0;
  for (int i = 0; i < P->n; ++i) {
    P->lind[i] = ind[lo[r] + i] - a;
    bool touches = false;
    for (int k = ind[lo[r] + i]; k < ind[lo[r] + i + 1]; ++k) {
      const int c = col[k];
      int lc;
      if (c >= lo[r] && c < hi[r]) {
        lc = P->ghost_lo + (int)(c - lo[r]);
      } else {
        const int pos = (int)(std::lower_bound(g.begin(), g.end(), c) - g.begin());
        lc = pos < P->ghost_lo ? pos : P->n + pos;
        touches = true;
      }
      P->lcol[k - a] = lc;
    }
    if (touches) {
      if (i - run_a > best_b - best_a) best_a = run_a, best_b = i;
      run_a = i + 1;
    }
  }
  if (P->n - run_a > best_b - best_a) best_a = run_a, best_b = P->n;
  P->lind[P->n] = b - a;
  P->ia = best_a;
  P->ib = best_b;
  // one link per owner of my ghosts (they are contiguous in the sorted list)
  P->links.clear();
  for (size_t i = 0; i < g.size();) {
    const int q = owner(g[i]);
    size_t j = i;
    while (j < g.size() && g[j] < hi[q]) ++j;
    HostLink L;
    L.q = q;
    L.recv_off = (int)i < P->ghost_lo ? (int)i : P->n + (int)i;
    L.count = (int)(j - i);
    L.idx.resize((size_t)L.count);
    for (int t = 0; t < L.count; ++t) L.idx[t] = g[i + t] - (int)lo[q];
    L.send_off = (L.idx.back() - L.idx.front() + 1 == L.count) ? L.idx.front() : -1;
    P->links.push_back(L);
    i = j;
  }
  return PSP_OK;
}

psp_csr *wrap(psp_mcsr *M, int nrows, int ncols) {
  psp_csr *A = new psp_csr();
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz64 = M->nnz;
  A->nnz = M->nnz > 2147483647LL ? -1 : (int)M->nnz;
  A->no_reorder = true;
  A->multi = M;
  return A;
}

}  // namespace

namespace psp {

int k_dinv(long n, const double *diag, double omega, double *dinv, double *partials, int *nparts);  // psp_vec.hip

int multi_destroy(psp_mcsr *M) {
  if (!M) return PSP_OK;
  DeviceRestore keep;
  (void)sync_all(M);
  if (M->use_rccl && rccl())
    for (ncclComm_t c : M->comms) (void)rccl()->CommDestroy(c);
  for (RankOp &R : M->r) {
    (void)hipSetDevice(R.dev);
    for (Link &L : R.links) {
      if (L.send_idx) (void)hipFree(L.send_idx);
      if (L.send_buf) (void)hipFree(L.send_buf);
    }
    if (R.dinv) {
      dinv_unregister(R.dinv);
      (void)hipFree(R.dinv);
    }
    if (R.scal) (void)hipFree(R.scal);
    if (R.A) psp_csr_destroy(R.A);
    if (R.evP) (void)hipEventDestroy(R.evP);
    if (R.evH) (void)hipEventDestroy(R.evH);
    if (R.evA) (void)hipEventDestroy(R.evA);
    if (R.s) (void)hipStreamDestroy(R.s);
    if (R.c) (void)hipStreamDestroy(R.c);
  }
  if (M->scal_ptrs) (void)hipFree(M->scal_ptrs);
  if (M->evB) (void)hipEventDestroy(M->evB);
  delete M;
  return PSP_OK;
}

int multi_describe(const psp_mcsr *M, char *buf, int cap) {
  snprintf(buf, cap, "multi[%d ranks on %d device(s), %s, reductions: %s]", M->nranks, M->distinct, M->kind,
           M->nranks == 1 ? "none" : (M->use_rccl ? "rccl" : "local fold"));
  return PSP_OK;
}

int multi_matvec_host(psp_mcsr *M, const double *x_host, ptrdiff_t incx, double *y_host, ptrdiff_t incy) {
  DeviceRestore keep;
  Vecs mem;
  std::vector<double *> vext(M->nranks), y(M->nranks);
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n_ext, &vext[r]));
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &y[r], false));
  }
  PSP_TRY(scatter_host(M, x_host, incx, vext, 1));
  PSP_TRY(matvec_plain(M, vext.data(), y.data()));
  return gather_host(M, y, y_host, incy);
}

static int block_diag(psp_mcsr *M, int r, double *d) {
  RankOp &R = M->r[r];
  PSP_TRY(use(M, r));
  if (R.n == 0) return PSP_OK;
  if (R.A->w4_only) return psp_csr_diagonal_dev(R.A, d);  // the slot of A[r, r] is known to the index-free layout
  hipLaunchKernelGGL(block_diag_kernel, dim3((R.n + 255) / 256), dim3(256), 0, R.s, R.n, R.ghost_lo, R.A->ind,
                     R.A->col, R.A->val, d);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int multi_diagonal_host(psp_mcsr *M, double *diag_host) {
  DeviceRestore keep;
  Vecs mem;
  std::vector<double *> d(M->nranks);
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &d[r]));
    PSP_TRY(block_diag(M, r, d[r]));
  }
  return gather_host(M, d, diag_host, 1);
}

int multi_jacobi_setup(psp_mcsr *M, double omega) {
  DeviceRestore keep;
  if (M->has_dinv && M->jac_omega == omega) return PSP_OK;
  double nsing_total = 0.0;
  for (int r = 0; r < M->nranks; ++r) {
    RankOp &R = M->r[r];
    PSP_TRY(use(M, r));
    if (R.dinv) {
      dinv_unregister(R.dinv);
      (void)hipFree(R.dinv);
      R.dinv = nullptr;
    }
    M_HIP(hipMalloc((void **)&R.dinv, sizeof(double) * (size_t)(R.n ? R.n : 1)));
    PSP_TRY(block_diag(M, r, R.dinv));
    if (R.n == 0) continue;
    Workspace *w;
    PSP_TRY(workspace(&w));
    int np = 0;
    double nsing = 0.0;
    PSP_TRY(k_dinv(R.n, R.dinv, omega, R.dinv, w->partials, &np));
    PSP_TRY(finish_partials(w->partials, np, 1, w->scal_dev));
    PSP_TRY(fetch_scalars(w->scal_dev, 1, &nsing));
    nsing_total += nsing;
    (void)dinv_register(R.dinv, R.n);  // constant diagonal: the vector kernels skip the dinv stream
  }
  if (nsing_total != 0.0) return fail(PSP_ESINGULAR, "diagonal element close to zero");  // preconmodule.c:395-397
  M->has_dinv = true;
  M->jac_omega = omega;
  return PSP_OK;
}

int multi_jacobi_apply_host(psp_mcsr *M, const double *x_host, double *y_host) {
  DeviceRestore keep;
  if (!M->has_dinv) return fail(PSP_EINVAL, "jacobi of a multi-device matrix was not set up");
  Vecs mem;
  std::vector<double *> x(M->nranks), y(M->nranks);
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &x[r], false));
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &y[r], false));
  }
  PSP_TRY(scatter_host(M, x_host, 1, x, 0));
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(use(M, r));
    if (M->r[r].n) PSP_TRY(psp_k_jacobi(M->r[r].n, x[r], M->r[r].dinv, y[r]));
  }
  return gather_host(M, y, y_host, 1);
}

static constexpr int kBatch = 16;  // iterations enqueued between two reads of the state (PCG_BATCH of the torch driver)

// pcg.c:22-171 on row blocks -- the C++ twin of _dist_pcg_dev (pysparse_amd/distributed.py)
int multi_pcg(psp_mcsr *M, bool jacobi, int n, double *x_host, const double *b_host, double tol, int maxit,
              int *info, int *iter, double *relres, double *hist) {
  if (n != M->n_global) return fail(PSP_EINVAL, "incompatible operand shapes");
  if (jacobi && !M->has_dinv) return fail(PSP_EINVAL, "jacobi of a multi-device matrix was not set up");
  DeviceRestore keep;
  const int nr = M->nranks;
  Vecs mem;
  std::vector<double *> rv(nr), q(nr), pext(nr), x(nr), b(nr), dinv(nr, nullptr);
  std::vector<psp_pcgstate_t *> st(nr, nullptr);
  struct StGuard {
    psp_mcsr *M;
    std::vector<psp_pcgstate_t *> &st;
    ~StGuard() {
      for (size_t r = 0; r < st.size(); ++r)
        if (st[r]) {
          (void)use(M, (int)r);
          psp_pcgstate_destroy(st[r]);
        }
    }
  } guard{M, st};
  for (int r = 0; r < nr; ++r) {
    const RankOp &R = M->r[r];
    PSP_TRY(mem.get(M, r, (size_t)R.n, &rv[r]));
    PSP_TRY(mem.get(M, r, (size_t)R.n, &q[r]));
    PSP_TRY(mem.get(M, r, (size_t)R.n_ext, &pext[r]));
    PSP_TRY(mem.get(M, r, (size_t)R.n, &x[r], false));
    PSP_TRY(mem.get(M, r, (size_t)R.n, &b[r], false));
    if (jacobi) dinv[r] = R.dinv;
  }
  PSP_TRY(scatter_host(M, x_host, 1, x, 0));
  PSP_TRY(scatter_host(M, b_host, 1, b, 0));
  double s[4];
  for (int r = 0; r < nr; ++r) {
    PSP_TRY(use(M, r));
    PSP_TRY(psp_k_dot(M->r[r].n, b[r], b[r], M->r[r].scal));
  }
  PSP_TRY(allreduce(M, 0, 1));
  PSP_TRY(fetch0(M, 0, 1, s));
  const double n2b = std::sqrt(s[0]);
  if (n2b == 0.0) {  // pcg.c:58-67
    std::memset(x_host, 0, sizeof(double) * (size_t)n);
    *info = 0;
    *iter = 0;
    *relres = 0.0;
    return sync_all(M);
  }
  const double tolb = tol * n2b;
  for (int r = 0; r < nr; ++r) {
    const RankOp &R = M->r[r];
    M_HIP(hipSetDevice(R.dev));
    M_HIP(hipMemcpyAsync(pext[r] + R.ghost_lo, x[r], sizeof(double) * (size_t)R.n, hipMemcpyDeviceToDevice, R.s));
  }
  PSP_TRY(matvec_plain(M, pext.data(), rv.data()));
  for (int r = 0; r < nr; ++r) {
    PSP_TRY(use(M, r));
    PSP_TRY(psp_k_residual(M->r[r].n, b[r], rv[r], dinv[r], M->r[r].scal));  // r := b - A x; {r.r, r.z}
  }
  PSP_TRY(allreduce(M, 0, 2));
  PSP_TRY(fetch0(M, 0, 2, s));
  const double normr = std::sqrt(s[0]);
  if (hist) hist[0] = normr;
  auto leave = [&](int fl, int it, double rr) {
    *info = fl;
    *iter = it;
    *relres = rr;
    return gather_host(M, x, x_host, 1);
  };
  if (normr <= tolb) return leave(0, 0, normr / n2b);  // pcg.c:77-84
  if (maxit < 1) return leave(-1, 1, normr / n2b);
  if (s[1] == 0.0) return leave(-2, 1, normr / n2b);   // pcg.c:101-104 in iteration 1
  for (int r = 0; r < nr; ++r) {
    PSP_TRY(use(M, r));
    PSP_TRY(psp_pcgstate_create(&st[r]));
    PSP_TRY(psp_pcgstate_init(st[r], n2b, tolb, normr, s[1], maxit, hist != nullptr));
  }
  psp_pcg_status_t f;
  int enq = 0;
  for (;;) {
    const int batch = std::max(1, std::min(kBatch, maxit - enq));
    for (int k = 0; k < batch; ++k) {
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        PSP_TRY(guard_overwrite(M, r));
        PSP_TRY(psp_kd_px_update(st[r], R.n, rv[r], dinv[r], pext[r] + R.ghost_lo, x[r], R.scal + 1));
      }
      PSP_TRY(exchange(M, pext.data()));
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        if (nr == 1)
          PSP_TRY(psp_kd_csr_matvec_overlap(st[r], R.A, pext[r], R.ghost_lo, q[r], 0, R.n, nullptr, nullptr, R.scal));
        else
          PSP_TRY(psp_kd_csr_matvec_overlap(st[r], R.A, pext[r], R.ghost_lo, q[r], R.ia, R.ib, wait_halo, &R, R.scal));
      }
      PSP_TRY(allreduce(M, 0, 2));  // #1 {p.q, nonstag}
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        PSP_TRY(psp_kd_pcg_scalar_xpq(st[r], R.scal));
        PSP_TRY(psp_kd_r_update(st[r], R.n, q[r], dinv[r], rv[r], R.scal + 2));
      }
      PSP_TRY(allreduce(M, 2, 2));  // #2 {r.r, r.z}
      for (int r = 0; r < nr; ++r) {
        PSP_TRY(use(M, r));
        PSP_TRY(psp_kd_pcg_scalar_r(st[r], M->r[r].scal + 2));
      }
    }
    enq += batch;
    PSP_TRY(use(M, 0));
    PSP_TRY(psp_pcgstate_fetch(st[0], &f));
    if (f.status) break;
  }
  int fl = f.info, it = f.iter;
  double rr = f.relres;
  if (f.xpend) {  // the x update (and stagnation scan) of the last iteration
    for (int r = 0; r < nr; ++r) {
      RankOp &R = M->r[r];
      PSP_TRY(use(M, r));
      PSP_TRY(psp_k_x_update(R.n, f.alpha_x, pext[r] + R.ghost_lo, x[r], R.scal + 1));
    }
    PSP_TRY(allreduce(M, 1, 1));
    PSP_TRY(fetch0(M, 1, 1, s));
    if (f.pend_maxit) {
      const bool stag = f.stag0 || s[0] == 0.0;
      fl = stag ? -5 : -1;  // pcg.c:159-165
      it = stag ? maxit : maxit + 1;
      rr = f.normr / f.n2b;
    }
  }
  if (hist) {
    const int cnt = std::min(it, maxit);
    PSP_TRY(use(M, 0));
    if (cnt >= 1) PSP_TRY(psp_pcgstate_hist(st[0], 1, cnt, hist + 1));
  }
  return leave(fl, it, rr);
}

// minres.c:43-200 on row blocks -- the C++ twin of dist_minres (pysparse_amd/distributed.py)
int multi_minres(psp_mcsr *M, bool jacobi, int n, double *x_host, const double *b_host, double tol, int maxit,
                 int *info, int *iter, double *relres, double *hist) {
  if (n != M->n_global) return fail(PSP_EINVAL, "incompatible operand shapes");
  if (jacobi && !M->has_dinv) return fail(PSP_EINVAL, "jacobi of a multi-device matrix was not set up");
  DeviceRestore keep;
  const int nr = M->nranks;
  Vecs mem;
  std::vector<double *> vhat(nr), vhat_old(nr), wv(nr), w_old(nr), av(nr), vext(nr), y(nr, nullptr), x(nr), b(nr),
      dinv(nr, nullptr);
  std::vector<psp_minresstate_t *> st(nr, nullptr);
  struct StGuard {
    psp_mcsr *M;
    std::vector<psp_minresstate_t *> &st;
    ~StGuard() {
      for (size_t r = 0; r < st.size(); ++r)
        if (st[r]) {
          (void)use(M, (int)r);
          psp_minresstate_destroy(st[r]);
        }
    }
  } guard{M, st};
  for (int r = 0; r < nr; ++r) {
    const RankOp &R = M->r[r];
    for (std::vector<double *> *v : {&vhat, &vhat_old, &wv, &w_old, &av}) PSP_TRY(mem.get(M, r, (size_t)R.n, &(*v)[r]));
    PSP_TRY(mem.get(M, r, (size_t)R.n_ext, &vext[r]));
    if (jacobi) {
      PSP_TRY(mem.get(M, r, (size_t)R.n, &y[r]));
      dinv[r] = R.dinv;
    }
    PSP_TRY(mem.get(M, r, (size_t)R.n, &x[r], false));
    PSP_TRY(mem.get(M, r, (size_t)R.n, &b[r], false));
  }
  PSP_TRY(scatter_host(M, x_host, 1, x, 0));
  PSP_TRY(scatter_host(M, b_host, 1, b, 0));
  // v_hat = b - A x, norm_r0 (minres.c:67-71); y = K v_hat, beta = sqrt(v_hat.y) (:73-82)
  for (int r = 0; r < nr; ++r) {
    const RankOp &R = M->r[r];
    M_HIP(hipSetDevice(R.dev));
    M_HIP(hipMemcpyAsync(vext[r] + R.ghost_lo, x[r], sizeof(double) * (size_t)R.n, hipMemcpyDeviceToDevice, R.s));
  }
  PSP_TRY(matvec_plain(M, vext.data(), vhat.data()));
  for (int r = 0; r < nr; ++r) {
    PSP_TRY(use(M, r));
    PSP_TRY(psp_k_residual(M->r[r].n, b[r], vhat[r], dinv[r], M->r[r].scal));
    if (jacobi && M->r[r].n) PSP_TRY(psp_k_jacobi(M->r[r].n, vhat[r], dinv[r], y[r]));
  }
  PSP_TRY(allreduce(M, 0, 2));
  double s[2];
  PSP_TRY(fetch0(M, 0, 2, s));
  const double norm_r0 = std::sqrt(s[0]);
  auto leave = [&](int fl, int it, bool set_rr, double rr) {
    *info = fl;
    *iter = it;
    if (set_rr) *relres = rr;
    return gather_host(M, x, x_host, 1);
  };
  if (s[1] < 0.0) return leave(-3, 0, false, 0.0);  // minres.c:79-80
  const double beta = std::sqrt(s[1]);
  if (hist) hist[0] = norm_r0;
  const bool conv0 = norm_r0 < tol * norm_r0;
  if (maxit < 1 || conv0) return leave(conv0 ? 0 : -1, 0, true, norm_r0 / norm_r0);  // minres.c:114 before iteration 1
  for (int r = 0; r < nr; ++r) {
    PSP_TRY(use(M, r));
    PSP_TRY(psp_minresstate_create(&st[r]));
    PSP_TRY(psp_minresstate_init(st[r], norm_r0, beta, tol, maxit, hist != nullptr));
  }
  psp_minres_status_t f;
  int enq = 0;
  for (;;) {
    const int batch = std::max(1, std::min(kBatch, maxit - enq));
    for (int k = 0; k < batch; ++k) {
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        PSP_TRY(guard_overwrite(M, r));
        PSP_TRY(psp_kd_minres_scale(st[r], R.n, jacobi ? y[r] : vhat[r], vext[r] + R.ghost_lo));  // v = y / beta
      }
      PSP_TRY(exchange(M, vext.data()));
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        if (nr == 1)
          PSP_TRY(psp_kd_minres_matvec(st[r], R.A, vext[r], R.ghost_lo, av[r], 0, R.n, nullptr, nullptr, R.scal));
        else
          PSP_TRY(psp_kd_minres_matvec(st[r], R.A, vext[r], R.ghost_lo, av[r], R.ia, R.ib, wait_halo, &R, R.scal));
      }
      PSP_TRY(allreduce(M, 0, 1));  // #1 alpha = v.Av
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        PSP_TRY(psp_kd_minres_scalar(st[r], 0, R.scal));
        PSP_TRY(psp_kd_minres_lanczos(st[r], R.n, av[r], vhat[r], vhat_old[r], dinv[r], y[r], R.scal + 4));
      }
      std::swap(vhat, vhat_old);
      PSP_TRY(allreduce(M, 4, 1));  // #2 beta^2 = v_hat.y
      for (int r = 0; r < nr; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        PSP_TRY(psp_kd_minres_scalar(st[r], 1, R.scal + 4));
        PSP_TRY(psp_kd_minres_wx(st[r], R.n, vext[r] + R.ghost_lo, wv[r], w_old[r], x[r]));
      }
      std::swap(wv, w_old);
    }
    enq += batch;
    PSP_TRY(use(M, 0));
    PSP_TRY(psp_minresstate_fetch(st[0], &f));
    if (f.status || f.stop) break;
  }
  if (hist) {
    const int cnt = std::min(f.iter, maxit) - ((f.info == -3 || f.info == -6) ? 1 : 0);
    PSP_TRY(use(M, 0));
    if (cnt >= 1) PSP_TRY(psp_minresstate_hist(st[0], 1, cnt, hist + 1));
  }
  return leave(f.info, f.iter, f.info == 0 || f.info == -1, f.relres);
}

}  // namespace psp

extern "C" {

int psp_csr_poisson_multi(int nx, int ny, int nz, const int *devices, int ndev, psp_csr_t **out) {
  PSP_API_GUARD;
  if (!out || !devices || ndev < 1 || ndev > 64) return fail(PSP_EINVAL, "psp_csr_poisson_multi: bad device list");
  if (nx < 1 || ny < 1 || nz < 0) return fail(PSP_EINVAL, "psp_csr_poisson_multi: bad grid");
  PSP_TRY(psp::ensure_device());
  const bool three_d = nz > 0;
  const int planes = three_d ? nz : ny;
  const int64_t plane_rows = three_d ? (int64_t)nx * ny : nx;
  const int64_t n = plane_rows * planes;
  if (planes < ndev) return fail(PSP_EINVAL, "fewer grid planes (%d) than ranks (%d)", planes, ndev);
  if (n > 2147483647LL) return fail(PSP_EINVAL, "grid exceeds 2^31 - 1 rows");
  DeviceRestore keep;
  psp_mcsr *M = new psp_mcsr();
  snprintf(M->kind, sizeof M->kind, "poisson %s slabs", three_d ? "z" : "y");
  M->n_global = (int)n;
  int rc = new_ranks(M, devices, ndev);
  for (int r = 0; r < ndev && rc == PSP_OK; ++r) {
    RankOp &R = M->r[r];
    int64_t plo, phi;
    row_range(planes, ndev, r, &plo, &phi);  // whole planes per rank (slab_range of the torch driver)
    const int64_t lo = plo * plane_rows, hi = phi * plane_rows;
    R.row_lo = lo;
    R.n = (int)(hi - lo);
    R.ghost_lo = (int)std::min<int64_t>(plane_rows, lo);
    R.ghost_hi = (int)std::min<int64_t>(plane_rows, n - hi);
    R.n_ext = R.ghost_lo + R.n + R.ghost_hi;
    R.ia = R.ghost_lo ? (int)plane_rows : 0;
    R.ib = R.ghost_hi ? R.n - (int)plane_rows : R.n;
    if (R.ib < R.ia) R.ia = R.ib = 0;  // a one-plane slab with two neighbours has no ghost-free row
    if (R.ghost_lo) {  // the lower neighbour's last plane
      Link L;
      L.q = r - 1;
      L.recv_off = 0;
      L.count = R.ghost_lo;
      L.send_off = -2;  // patched below: needs the neighbour's n
      R.links.push_back(L);
    }
    if (R.ghost_hi) {  // the upper neighbour's first plane
      Link L;
      L.q = r + 1;
      L.recv_off = R.ghost_lo + R.n;
      L.count = R.ghost_hi;
      L.send_off = 0;
      R.links.push_back(L);
    }
    rc = use(M, r);
    if (rc == PSP_OK)
      rc = psp_csr_poisson_big_slab(nx, ny, nz, lo, hi, lo - R.ghost_lo, R.n_ext, &R.A);
    if (rc == PSP_OK) M->nnz += psp_csr_nnz64(R.A);
  }
  for (int r = 0; r < ndev && rc == PSP_OK; ++r)
    for (Link &L : M->r[r].links)
      if (L.send_off == -2) L.send_off = M->r[L.q].n - L.count;
  if (rc == PSP_OK) rc = finish_setup(M, devices, ndev);
  if (rc == PSP_OK) rc = sync_all(M);
  if (rc != PSP_OK) {
    psp::multi_destroy(M);
    return rc;
  }
  *out = wrap(M, (int)n, (int)n);
  return PSP_OK;
}

int psp_csr_create_multi(int nrows, int ncols, int nnz, const int *ind, const int *col, const double *val,
                         const int *devices, int ndev, psp_csr_t **out) {
  PSP_API_GUARD;
  if (!out || !devices || ndev < 1 || ndev > 64) return fail(PSP_EINVAL, "psp_csr_create_multi: bad device list");
  if (!ind || (nnz > 0 && (!col || !val)) || nrows < 0 || nnz < 0 || ind[0] != 0 || ind[nrows] != nnz)
    return fail(PSP_EINVAL, "psp_csr_create_multi: bad CSR arrays");
  if (nrows != ncols) return fail(PSP_EINVAL, "a multi-device matrix must be square (rows and vector slices share the partition)");
  PSP_TRY(psp::ensure_device());
  DeviceRestore keep;
  psp_mcsr *M = new psp_mcsr();
  snprintf(M->kind, sizeof M->kind, "csr row blocks");
  M->n_global = nrows;
  M->nnz = nnz;
  int rc = new_ranks(M, devices, ndev);
  for (int r = 0; r < ndev && rc == PSP_OK; ++r) {
    RankOp &R = M->r[r];
    HostPlan P;
    rc = plan_block(nrows, ncols, ind, col, ndev, r, &P);
    if (rc != PSP_OK) break;
    R.row_lo = P.lo;
    R.n = P.n;
    R.ghost_lo = P.ghost_lo;
    R.ghost_hi = P.ghost_hi;
    R.n_ext = P.ghost_lo + P.n + P.ghost_hi;
    R.ia = P.ia;
    R.ib = P.ib;
    rc = use(M, r);
    if (rc == PSP_OK)
      rc = psp_csr_create(R.n, R.n_ext, (int)P.lcol.size(), P.lind.data(), P.lcol.data(), val + ind[P.lo], &R.A);
    if (rc == PSP_OK) R.A->no_reorder = true;  // the renumbered copy would need x in another numbering per rank
    for (const HostLink &H : P.links) {
      if (rc != PSP_OK) break;
      Link L;
      L.q = H.q;
      L.recv_off = H.recv_off;
      L.count = H.count;
      L.send_off = H.send_off;
      if (H.send_off < 0) {  // scattered: the sender packs them with psp_k_gather into a buffer on ITS device
        const int qdev = M->r[H.q].dev;
        if (hipSetDevice(qdev) != hipSuccess || hipMalloc((void **)&L.send_idx, sizeof(int) * (size_t)H.count) != hipSuccess ||
            hipMalloc((void **)&L.send_buf, sizeof(double) * (size_t)H.count) != hipSuccess ||
            hipMemcpy(L.send_idx, H.idx.data(), sizeof(int) * (size_t)H.count, hipMemcpyHostToDevice) != hipSuccess)
          rc = fail(PSP_ENOMEM, "psp_csr_create_multi: send list allocation failed");
      }
      R.links.push_back(L);
    }
  }
  if (rc == PSP_OK) rc = finish_setup(M, devices, ndev);
  if (rc == PSP_OK) rc = sync_all(M);
  if (rc != PSP_OK) {
    psp::multi_destroy(M);
    return rc;
  }
  *out = wrap(M, nrows, ncols);
  return PSP_OK;
}

namespace {
// a start / stop event per rank for the timing entry points: destroyed on every way out
struct RankEvents {
  std::vector<hipEvent_t> e0, e1;
  bool started = false;
  ~RankEvents() {
    for (hipEvent_t e : e0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : e1) if (e) (void)hipEventDestroy(e);
  }
  int create(psp_mcsr *M) {
    e0.assign(M->nranks, nullptr);
    e1.assign(M->nranks, nullptr);
    for (int r = 0; r < M->nranks; ++r) {
      M_HIP(hipSetDevice(M->r[r].dev));
      M_HIP(hipEventCreate(&e0[r]));
      M_HIP(hipEventCreate(&e1[r]));
    }
    return PSP_OK;
  }
  int start(psp_mcsr *M) {  // everything before is over; the clocks of all ranks start together
    PSP_TRY(sync_all(M));
    for (int r = 0; r < M->nranks; ++r) {
      M_HIP(hipSetDevice(M->r[r].dev));
      M_HIP(hipEventRecord(e0[r], M->r[r].s));
    }
    started = true;
    return PSP_OK;
  }
  int stop(psp_mcsr *M, double *worst_ms) {  // the slowest rank's stream time
    if (!started) return fail(PSP_EINVAL, "timing: no repetition was started");
    *worst_ms = 0.0;
    for (int r = 0; r < M->nranks; ++r) {
      float ms = 0.f;
      M_HIP(hipSetDevice(M->r[r].dev));
      M_HIP(hipEventRecord(e1[r], M->r[r].s));
      M_HIP(hipEventSynchronize(e1[r]));
      M_HIP(hipEventElapsedTime(&ms, e0[r], e1[r]));
      *worst_ms = std::max(*worst_ms, (double)ms);
    }
    return PSP_OK;
  }
};
}  // namespace

int psp_csr_multi_spmv_time(psp_csr_t *A, int warmup, int reps, double *ms_per_product) {
  PSP_API_GUARD_H(A);
  if (!A || !A->multi || reps < 1 || warmup < 0 || !ms_per_product)
    return fail(PSP_EINVAL, "psp_csr_multi_spmv_time: needs a multi-device matrix and reps >= 1");
  psp_mcsr *M = A->multi;
  DeviceRestore keep;
  Vecs mem;
  std::vector<double *> vext(M->nranks), y(M->nranks);
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n_ext, &vext[r]));
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &y[r]));
    PSP_TRY(use(M, r));
    if (M->r[r].n) PSP_TRY(psp_k_jacobi(M->r[r].n, y[r], y[r], y[r]));  // touch
  }
  RankEvents ev;
  PSP_TRY(ev.create(M));
  for (int k = -warmup; k < reps; ++k) {
    if (k == 0) PSP_TRY(ev.start(M));
    // one product the way a solver iteration does it: exchange on the copy streams, interior rows meanwhile
    for (int r = 0; r < M->nranks; ++r) {
      PSP_TRY(use(M, r));
      PSP_TRY(guard_overwrite(M, r));
    }
    PSP_TRY(exchange(M, vext.data()));
    for (int r = 0; r < M->nranks; ++r) {
      RankOp &R = M->r[r];
      PSP_TRY(use(M, r));
      if (!R.n) continue;
      if (M->nranks == 1)
        PSP_TRY(psp_k_csr_matvec_overlap(R.A, vext[r], R.ghost_lo, y[r], 0, R.n, nullptr, nullptr, nullptr));
      else
        PSP_TRY(psp_k_csr_matvec_overlap(R.A, vext[r], R.ghost_lo, y[r], R.ia, R.ib, wait_halo, &R, nullptr));
    }
  }
  double worst = 0.0;
  PSP_TRY(ev.stop(M, &worst));
  PSP_TRY(sync_all(M));
  *ms_per_product = worst / reps;  // the slowest rank's stream time: what an iteration waits for
  return PSP_OK;
}

// The pieces of one iteration on their own (bench.py: `phases` of the single-process line): the slowest rank's stream
// time per repetition of
//   what 0  the ghost exchange alone (packing, peer copies on the copy streams, the compute streams wait for evH)
//   what 1  the local product alone (all owned rows, ghost entries as they are -- zeros: no exchange, no wait)
//   what 2  one packed reduction of two doubles (RCCL all-reduce or the fold kernel), in stream order
// psp_csr_multi_spmv_time is the product as a solver does it (0 overlapped with 1); overlap = (t0 + t1 - t_spmv) / t0.
int psp_csr_multi_phase_time(psp_csr_t *A, int what, int warmup, int reps, double *ms_per_rep) {
  PSP_API_GUARD_H(A);
  if (!A || !A->multi || reps < 1 || warmup < 0 || !ms_per_rep || what < 0 || what > 2)
    return fail(PSP_EINVAL, "psp_csr_multi_phase_time: needs a multi-device matrix, what in 0..2 and reps >= 1");
  psp_mcsr *M = A->multi;
  DeviceRestore keep;
  Vecs mem;
  std::vector<double *> vext(M->nranks), y(M->nranks);
  for (int r = 0; r < M->nranks; ++r) {
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n_ext, &vext[r]));  // cleared (Vecs::get)
    PSP_TRY(mem.get(M, r, (size_t)M->r[r].n, &y[r]));
  }
  RankEvents ev;
  PSP_TRY(ev.create(M));
  for (int k = -warmup; k < reps; ++k) {
    if (k == 0) PSP_TRY(ev.start(M));
    if (what == 0) {
      for (int r = 0; r < M->nranks; ++r) {
        PSP_TRY(use(M, r));
        PSP_TRY(guard_overwrite(M, r));
      }
      PSP_TRY(exchange(M, vext.data()));
      for (int r = 0; r < M->nranks && M->nranks > 1; ++r) {
        M_HIP(hipSetDevice(M->r[r].dev));
        M_HIP(hipStreamWaitEvent(M->r[r].s, M->r[r].evH, 0));
      }
    } else if (what == 1) {
      for (int r = 0; r < M->nranks; ++r) {
        RankOp &R = M->r[r];
        PSP_TRY(use(M, r));
        if (R.n) PSP_TRY(psp_k_csr_matvec_overlap(R.A, vext[r], R.ghost_lo, y[r], 0, R.n, nullptr, nullptr, nullptr));
      }
    } else {
      PSP_TRY(allreduce(M, 0, 2));
    }
  }
  double worst = 0.0;
  PSP_TRY(ev.stop(M, &worst));
  PSP_TRY(sync_all(M));
  *ms_per_rep = worst / reps;
  return PSP_OK;
}

int psp_multi_plan(int nrows, int ncols, const int *ind, const int *col, int ndev, int rank, int64_t *row_range_out,
                   int *counts_out, int *ghost_ids, int ghost_cap, int *links, int links_cap, int *col_local) {
  if (!ind || nrows < 0 || ndev < 1 || rank < 0 || rank >= ndev || !row_range_out || !counts_out)
    return fail(PSP_EINVAL, "psp_multi_plan: bad argument");
  if (ind[nrows] > 0 && !col) return fail(PSP_EINVAL, "psp_multi_plan: NULL column array");
  HostPlan P;
  PSP_TRY(plan_block(nrows, ncols, ind, col, ndev, rank, &P));
  row_range_out[0] = P.lo;
  row_range_out[1] = P.hi;
  counts_out[0] = P.ghost_lo;
  counts_out[1] = P.ghost_hi;
  counts_out[2] = P.ia;
  counts_out[3] = P.ib;
  counts_out[4] = (int)P.links.size();
  if (ghost_ids) {
    if ((int)P.ghosts.size() > ghost_cap) return fail(PSP_EINVAL, "psp_multi_plan: ghost_cap too small");
    std::copy(P.ghosts.begin(), P.ghosts.end(), ghost_ids);
  }
  if (links) {
    if ((int)P.links.size() > links_cap) return fail(PSP_EINVAL, "psp_multi_plan: links_cap too small");
    for (size_t i = 0; i < P.links.size(); ++i) {
      links[4 * i] = P.links[i].q;
      links[4 * i + 1] = P.links[i].recv_off;
      links[4 * i + 2] = P.links[i].count;
      links[4 * i + 3] = P.links[i].send_off;
    }
  }
  if (col_local) std::copy(P.lcol.begin(), P.lcol.end(), col_local);
  return PSP_OK;
}

int psp_csr_multi_info(const psp_csr_t *A, int *nranks, int *distinct_devices, int *uses_rccl) {
  if (!A) return fail(PSP_EINVAL, "psp_csr_multi_info: NULL argument");
  const psp_mcsr *M = A->multi;
  if (nranks) *nranks = M ? M->nranks : 0;
  if (distinct_devices) *distinct_devices = M ? M->distinct : 0;
  if (uses_rccl) *uses_rccl = M ? (int)M->use_rccl : 0;
  return PSP_OK;
}

}  // extern "C"
