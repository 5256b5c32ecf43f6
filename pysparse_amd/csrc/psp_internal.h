// psp_internal.h -- shared declarations inside libpysparse_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <mutex>
#include <string>

#include "pysparse_hip.h"

namespace psp {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// The A/B switches of the kernels and loops (PSP_SPMV_*, PSP_PCG_*, ... -- INTEGRATION.md section 7) are read
// through this function only: it returns getenv(name) when the process was started with PSP_TUNING=1 and
// nullptr otherwise, so a stray PSP_* variable in a user's environment cannot change the code path of the
// drop-in.  Tests and tools that select a variant set both.
const char *tuning_env(const char *name);
// psp_last_solve_info / psp_set_single_kernel_loops (psp_runtime.hip): the pcg / minres entry points note which loop they
// ran -- per calling thread -- with its launches per iteration and the vector bytes per row it moves beside its product
// (-1: not modelled); note_fallback counts single-kernel loops that were refused or gave up after they were chosen
void note_solve(const char *loop, int launches_per_iter, int vec_bytes_per_row, int dinv_streamed);
void note_fallback();
bool single_kernel_loops_enabled();
// PSP_TUNING=1 PSP_SETUP_TRACE=1: where the set-up of a handle goes -- setup_mark(label) waits for the calling thread's
// stream and prints the milliseconds since the previous mark to stderr (label == nullptr: restart the clock silently).
// One relaxed load otherwise.  tools/setup_breakdown.py reads the lines.
void setup_mark(const char *label);
// Threading model (round 4; SURVEY 8b: "one HIP stream per handle; handles not thread-safe").
//   * What an entry point enqueues on -- device, stream, reduction workspace, host staging -- belongs to the CALLING
//     THREAD (a thread-local context).  The first thread that touches the library keeps the null stream (or whatever it
//     sets with psp_set_stream, e.g. torch's current stream); every other thread gets a non-blocking stream of its own on
//     first use.  psp_set_device / psp_set_stream act on the calling thread; a new thread starts on the device the
//     process selected last.  So two host threads with two handles overlap on the GPU instead of taking turns.
//   * A HANDLE (csr / sss / jacobi / ssor / callback operator) carries scratch of its own (lazily built tables, the
//     renumbered copy's vectors, SSOR's sweep vectors, pinned staging): every entry point locks the handles it is given
//     (HandleLock: one recursive mutex per handle, taken in address order), so two threads that share a handle take turns
//     on it -- the reference serialises the same calls by holding the GIL -- and never corrupt it.  Solvers lock the
//     operator, the preconditioner and the matrix behind the preconditioner; host-callback operators take the GIL
//     while those are held, so callers must not hold the GIL while they wait (ctypes and the extension modules
//     release it first).
//   * Entry points without a handle (the psp_k_* vector kernels, memory, events) lock nothing.
//   * Device-side order between threads (round 5).  The locks only serialise the HOST side of two calls on one handle;
//     the asynchronous *_dev entry points return with their kernels still in flight, and the next caller may enqueue on
//     another stream -- on the same scratch.  So every handle also carries a "last use" event: once a second stream has
//     been seen anywhere in the process (a second thread, or psp_set_stream to another stream; the one-stream process
//     pays nothing), an entry point records the event on its stream before it releases the handle, and the next entry
//     point on that handle makes its stream wait for it when the streams differ.  At the moment the second stream
//     appears the devices used so far are synchronised once, which covers everything enqueued before events were kept.
struct HandleEntry {
  std::recursive_mutex mu;
  hipEvent_t ev = nullptr;      // recorded behind the last call's work ...
  hipStream_t last = nullptr;   // ... on this stream
  int ev_dev = -1;
  bool has_last = false;
};
HandleEntry &handle_entry(const void *handle);
inline std::recursive_mutex &handle_mutex(const void *handle) { return handle_entry(handle).mu; }
void handles_enter(HandleEntry *const *e, int n);  // the calling thread's stream waits for the handles' last users
void handles_leave(HandleEntry *const *e, int n);  // ... and is recorded as their last user
class HandleLock {
 public:
  HandleLock() {}
  HandleLock(std::initializer_list<const void *> hs) {
    for (const void *h : hs) add(h);
    lock();
  }
  ~HandleLock() {
    if (n_ > 0 && locked_) handles_leave(e_, n_);
    for (int i = (locked_ ? n_ : 0) - 1; i >= 0; --i) e_[i]->mu.unlock();
  }
  HandleLock(const HandleLock &) = delete;
  HandleLock &operator=(const HandleLock &) = delete;
  void add(const void *h) {
    if (!h || n_ >= kMax) return;
    for (int i = 0; i < n_; ++i)
      if (h_[i] == h) return;
    h_[n_++] = h;
  }
  void lock() {  // address order: two threads that want the same set never wait for each other crosswise
    for (int i = 1; i < n_; ++i)
      for (int j = i; j > 0 && (uintptr_t)h_[j - 1] > (uintptr_t)h_[j]; --j) {
        const void *t = h_[j];
        h_[j] = h_[j - 1];
        h_[j - 1] = t;
      }
    for (int i = 0; i < n_; ++i) {
      e_[i] = &handle_entry(h_[i]);
      e_[i]->mu.lock();
    }
    locked_ = true;
    if (n_ > 0) handles_enter(e_, n_);
  }

 private:
  static constexpr int kMax = 8;
  const void *h_[kMax];
  HandleEntry *e_[kMax];
  int n_ = 0;
  bool locked_ = false;
};
#define PSP_API_GUARD psp::HandleLock psp_api_guard_
#define PSP_API_GUARD_H(...) psp::HandleLock psp_api_guard_({__VA_ARGS__})
hipStream_t stream();
hipStream_t swap_stream(hipStream_t s);  // returns the previous stream (graph capture needs a non-null one)
// Delay injection (round 5; psp_runtime.hip "shake").  The multi-device drivers order their streams by events only; an
// ordering edge that is missing shows as a wrong vector only when the timing happens to open the window (round 4 found
// one such race once in ~15 suite runs).  Under PSP_TUNING=1 a test arms the facility (psp_debug_shake or
// PSP_SHAKE="seed,min_us,max_us,points,ranks,revert"); every cut point of psp_multi.hip then enqueues a one-wave spin
// kernel of a pseudo-random duration on the stream it names, which moves that stream's later work against all others.
// Not armed (always, without PSP_TUNING=1): shake() is one relaxed load and a return.
enum ShakePoint {
  kShakePack = 0,        // sender's compute stream, before "vector final" (evP) is recorded
  kShakeCopyPre = 1,     // receiver's copy stream, before it waits for anybody
  kShakeCopyPost = 2,    // receiver's copy stream, after the ghost copies, before evH is recorded
  kShakePosted = 3,      // compute stream, after the exchange was posted (in front of the interior rows)
  kShakeBoundary = 4,    // compute stream, in front of the wait for the ghost copies (boundary rows)
  kShakeReducePre = 5,   // a rank's stream in front of an all-reduce
  kShakeReduceMid = 6,   // rank 0's stream between the fold kernel and evB
  kShakeReducePost = 7,  // a rank's stream behind an all-reduce
  kShakeClear = 8,       // compute stream, in front of the clearing of a fresh vector
  kShakeOverwrite = 9,   // compute stream, in front of a kernel that overwrites what readers copy from
  kShakeScatter = 10,    // compute stream, in front of a host -> device slice copy
  kShakePoints = 11
};
// reverting switches (bits of `revert`): test that the facility FINDS a known race when its fix is taken out
constexpr int kShakeRevertGhostWait = 1;  // round 4: the copy stream's wait for the receiver's own evP (psp_multi.hip exchange)
int shake(hipStream_t s, int point, int rank);  // the stream's device must be current
bool shake_armed();
bool shake_revert(int bit);
int ensure_device();  // PSP_OK, or PSP_ENODEV (with message) when no GPU is usable
// PSP_DEVICE=cpu (read once): the opt-in host mode of psp_cpu.hip -- never a fallback, see that file
bool cpu_mode();
// multi-device driver (psp_multi.hip): make `device` current (hipSetDevice) for THIS THREAD and enqueue on `s` from now
// on; the reduction workspace is per (device, thread, ws_slot), so the phase kernels can be driven for one rank after
// another.  ws_slot selects the reduction workspace (0 = the thread's ordinary one): ranks of a multi-device matrix that
// share a device enqueue on different streams and so must not share partial-sum buffers
int use_device(int device, hipStream_t s, int ws_slot = 0);
struct ThreadCtxSave {
  int device, dev_state, ws_slot;
  hipStream_t stream;
  bool stream_given;
};
ThreadCtxSave save_thread_ctx();
void restore_thread_ctx(const ThreadCtxSave &c);
int current_device();
int current_ws_slot();
int current_thread_slot();  // 0 for the first thread that used the library, small integers (reused) for the others

#define PSP_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return psp::fail(e_ == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "%s: %s (%s:%d)", \
                       #call, hipGetErrorString(e_), __FILE__, __LINE__);                \
  } while (0)

#define PSP_TRY(call)        \
  do {                       \
    int rc_ = (call);        \
    if (rc_ != PSP_OK)       \
      return rc_;            \
  } while (0)

#define PSP_LAUNCH_CHECK() PSP_HIP(hipGetLastError())

// Reductions: every reducing kernel is launched with at most kMaxParts workgroups; workgroup b leaves its partial
// sums in partials[slot*kMaxParts + b].  They are added in ONE canonical order, whichever kernel does the adding:
//     R(v[0..m))        = wave_sum over the 64 lanes of ( v[l] + v[l+64] + v[l+128] + ... ), lane l, left to right
//     reduce(parts, np) = np <= 256 ?  R(parts)  :  R( [ R(parts[256 g .. 256 g + 256)) for every group g ] )
// (wave_sum = the fixed shuffle-down tree).  Fixed grid + fixed order => bitwise reproducible results, no
// floating-point atomics.  Up to kOneBlockGroups groups ONE workgroup of 1024 threads does both levels (one wave per
// group, then wave 0 over the group sums): a reduction is one small launch up to n = 2^25 (round 4; two before);
// beyond that group_fold_kernel (one wave per group, many workgroups) + a finishing block.
// Round 4 also tried the reduction inside the kernel that produces the partial sums (the workgroup that draws the
// last ticket of its group adds the group, the last group adds the group sums and runs the solver's scalar step: three
// launches per PCG iteration instead of nine).  Same bits, and 20 % SLOWER at 512^3 (profiles/
// r4_reduce_tail_per_workgroup_ab.txt): the wave that waits for its ticket keeps its workgroup's slot ~3 us longer, a
// quarter of the lifetime of a workgroup of these bandwidth-bound kernels, and 2.6e5 returning device-scope atomics
// per kernel are not free either.  Backed out.
constexpr int kMaxParts = 1 << 21;  // one span per workgroup up to n = 2^30 (64 MB of partial-sum slots); round 1: 2^18,
                                    // i.e. looping grids -- 10-20 % slower vector kernels -- from 2^27 + 1 elements on
constexpr int kSlots = 4;
constexpr int kFold = 1024;  // outputs of the first-level folds of psp_csr.hip (grids beyond kMaxParts, parts of a big matrix)
constexpr int kTailGroup = 256;                      // partials per group
constexpr int kTailGroups = kMaxParts / kTailGroup;  // 8192
constexpr int kOneBlockGroups = 256;                 // <= this many groups (65536 partials): one workgroup CAN do it all
constexpr int kFoldAboveGroups = 16;                 // ... and does, up to this many (4096 partials; psp_runtime.hip one_block_groups():
                                                     // beyond, one workgroup's ~45 GB/s costs more than the second launch --
                                                     // profiles/r4_fold_threshold_ab.txt: 4096^2 PCG 368 -> 338 us per iteration)
// streaming vector kernels: one workgroup per contiguous span of kVecSpan elements
// (non-persistent grids measured faster than grid-stride loops on MI355X, profiles/)
constexpr int kVecSpan = 512;  // one 16-byte access per lane per array: measured best (profiles/)

struct Workspace {
  double *partials = nullptr;   // kSlots * kMaxParts doubles (device)
  double *folded = nullptr;     // kSlots * kTailGroups doubles (device): the group sums
  double *scal_dev = nullptr;   // 16 doubles (device)
  double *scal_host = nullptr;  // 16 doubles + a sequence word (pinned, mapped host memory)
  double *scal_host_dev = nullptr;      // the device's address of it (fetch_scalars)
  unsigned long long scal_seq = 0;
  // Round 5: what a solve needs besides its vectors -- the device-resident state of its loop and the pinned mirror the
  // host reads it into (PcgDev / MinresDev / KryDev; the control block of the single-kernel loops) and the partial sums of
  // the single-kernel loops -- lives here for the life of the thread instead of a hipMalloc / hipHostMalloc / hipFree per
  // solve (a solve of ONE iteration took 430 us at 2048^2, 110-150 us in the single-kernel range: tools/solve_overhead.py).
  // One solve at a time per thread uses them (the loops that do never run a caller's callback).
  void *state_dev = nullptr;    // kStateBytes (device)
  void *state_host = nullptr;   // kStateBytes (pinned)
  double *ctl_part = nullptr;   // kCtlPartDoubles (device)
  int num_cu = 0;
  int device = -1;
};
constexpr size_t kStateBytes = 8192;
constexpr size_t kCtlPartDoubles = 4 * 4096;
int workspace(Workspace **out);
// the solvers' pool of device work vectors (psp_solvers.hip) for the other translation units: at least n doubles,
// contents undefined; give back with the n asked for
int scratch_get(size_t n, double **out);
void scratch_put(double *p, size_t n);

// grid for streaming n-element vector kernels: one workgroup per span (capped; the kernels
// loop over spans beyond the cap)
inline int vec_grid(const Workspace &, long n) {
  long want = (n + kVecSpan - 1) / kVecSpan;
  if (want < 1) want = 1;
  return (int)(want < kMaxParts ? want : kMaxParts);
}

// finish: out_dev[j] = reduce(partials + j*kMaxParts, nparts), j < nvals
int finish_partials(const double *partials, int nparts, int nvals, double *out_dev);
// its two stages separately (the solvers append their scalar recurrences to the finishing block): fold_stage launches
// the group fold when there are more than kOneBlockGroups groups and returns what the finishing block (1024 threads:
// reduce_block below) reads
int fold_stage(const double *partials, int nparts, int nvals, const double **src, int *count, int *stride, bool *raw,
               int fslot = 0);
// two single-value reductions (different partial arrays / counts): one group-fold launch when either needs the stage
int fold_stage2(const double *const partials[2], const int nparts[2], const int fslot[2], const double *src[2],
                int count[2], int stride[2], bool raw[2]);
struct FoldJobs {  // kernel argument of group_fold_kernel
  const double *in[2] = {nullptr, nullptr};
  int nparts[2] = {0, 0};
  int nvals[2] = {0, 0};
  double *out[2] = {nullptr, nullptr};
};

#ifdef __HIPCC__
// The fixed tree of a wave's 64 values: v[l] + v[l + 32], then + 16, 8, 4, 2, 1 -- lane 0 holds the sum (the other lanes
// hold nothing a caller may use).  Round 5: the operands come through v_permlane32_swap / v_permlane16_swap (gfx950) and
// DPP row shifts instead of ds_bpermute (__shfl_down): the same additions on the same values -- the same bits
// (tools/wave_sum_check.hip) -- without six dependent trips through the LDS crossbar (the single-kernel loops of
// psp_mid.hip add a dozen waves' worth per iteration on their critical path).
template <int CTRL>
__device__ __forceinline__ double psp_dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double psp_lane_plus_32(double v) {  // lanes 0 .. 31 receive lanes 32 .. 63
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double psp_lane_plus_16(double v) {  // lanes 0 .. 15 receive lanes 16 .. 31 (32 .. 47: 48 .. 63)
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double psp_wave_sum(double v) {
  v = v + psp_lane_plus_32(v);
  v = v + psp_lane_plus_16(v);
  v = v + psp_dpp_mov<0x108>(v);  // row_shl:8 -- lane l receives lane l + 8 of its row of 16
  v = v + psp_dpp_mov<0x104>(v);
  v = v + psp_dpp_mov<0x102>(v);
  v = v + psp_dpp_mov<0x101>(v);
  return v;
}
__device__ __forceinline__ double psp_wave_max(double v) {  // lane 0: the largest of the wave's values (v >= 0 or any: plain >)
  double o = psp_lane_plus_32(v);
  v = o > v ? o : v;
  o = psp_lane_plus_16(v);
  v = o > v ? o : v;
  o = psp_dpp_mov<0x108>(v);
  v = o > v ? o : v;
  o = psp_dpp_mov<0x104>(v);
  v = o > v ? o : v;
  o = psp_dpp_mov<0x102>(v);
  v = o > v ? o : v;
  o = psp_dpp_mov<0x101>(v);
  v = o > v ? o : v;
  return v;
}
// R(v[0..count)) by one wave: lane l adds v[l], v[l+64], ... in order, then the shuffle tree; lane 0 holds the result.
// COH: the values were written by other workgroups of the SAME launch -> device-coherent (sc1) loads
template <bool COH>
__device__ __forceinline__ double psp_wave_reduce(const double *__restrict__ v, int count) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int base = 0; base < count; base += 512) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // eight loads in flight, added in order
      const int i = base + lane + 64 * k;
      if constexpr (COH)
        t[k] = i < count ? __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      else
        t[k] = i < count ? v[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s += t[k];
  }
  return psp_wave_sum(s);
}
// the finishing block (1024 threads): out[j] = reduce(...) of value j.  raw: `partials` are per-workgroup partial sums,
// nparts <= 256 * kOneBlockGroups of them -- wave w adds groups w, w + 16, ... (R over each group's <= 256 entries),
// then wave 0 adds the group sums (R); one group (nparts <= 256) is added by wave 0 directly.  !raw: `partials` are the
// group sums group_fold_kernel left: wave 0 adds them (R).  `sh`: kOneBlockGroups doubles of LDS.  The block is
// synchronised on return and thread 0 has written out[].
constexpr int kReduceBlock = 1024;
__device__ __forceinline__ void reduce_block(const double *__restrict__ partials, int nparts, int nvals, int stride,
                                             bool raw, double *__restrict__ out, double *__restrict__ sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int ngroups = raw ? (nparts + kTailGroup - 1) / kTailGroup : 1;
  for (int j = 0; j < nvals; ++j) {
    const double *p = partials + (size_t)j * stride;
    if (ngroups <= 1) {
      if (wid == 0) {
        const double s = psp_wave_reduce<false>(p, nparts);
        if (lane == 0) out[j] = s;
      }
    } else {
      for (int g = wid; g < ngroups; g += nw) {
        const double s = psp_wave_reduce<false>(p + (size_t)g * kTailGroup, min(kTailGroup, nparts - g * kTailGroup));
        if (lane == 0) sh[g] = s;
      }
      __syncthreads();
      if (wid == 0) {
        const double s = psp_wave_reduce<false>(sh, ngroups);
        if (lane == 0) out[j] = s;
      }
    }
    __syncthreads();
  }
}
#endif
// copy k scalars device -> host (synchronises the stream)
int fetch_scalars(const double *src_dev, int k, double *dst_host);
// finish_partials (below) and fetch_scalars in one launch
int finish_partials_fetch(const double *partials, int nparts, int nvals, double *out_dev, double *dst_host);

}  // namespace psp

struct psp_csr {
  int nrows = 0, ncols = 0, nnz = 0;
  int *ind = nullptr;     // nrows + 1
  int *col = nullptr;     // padded to a multiple of 4 entries (+4)
  double *val = nullptr;  // same padding
  size_t padded = 0;
  int rows_per_chunk = 0;  // SpMV work decomposition (see psp_csr.hip)
  int nchunks = 0;
  int variant = -1;  // kernel variant override, -1 = default
  int max_row_nnz = 0;
  int sched_strip_rows = -1;  // psp_csr_set_schedule: -1 automatic, 0 off, > 0 forced strip width
  const struct psp_sss *sym_owner = nullptr;  // set on the full mirror of an sss_mat (sss_spmv_w4)
  // psp_csr_poisson_big: the operator exists ONLY in the offset-major w4 layout (ind / col / val are
  // null; nnz may exceed 32 bits: nnz64 holds it, nnz is -1 then)
  bool w4_only = false;
  int64_t nnz64 = 0;
  int w4_diag_slot = -1;  // w4_only: which offset slot holds A[r, r]
  bool no_reorder = false;  // internal copies (renumbered / transposed) are never renumbered again
  // More than 2^31 - 8192 nonzeros (psp_csr_create64, psp_csr_random_banded): the rows are cut into parts of
  // < 2^30 nonzeros, each an ordinary handle with 32-bit offsets over the same column space; part p holds rows
  // [part_row0[p], part_row0[p+1]).  The kernels never see a 64-bit offset; nnz is -1, nnz64 the count.
  int nparts = 0;
  psp_csr **parts = nullptr;
  int *part_row0 = nullptr;
  // psp_csr_poisson_multi / psp_csr_create_multi (psp_multi.hip): the rows live on several devices as row blocks;
  // this handle then only carries the shape -- matvec, the diagonal, jacobi, pcg and minres go through `multi`
  struct psp_mcsr *multi = nullptr;
  bool host = false;  // PSP_DEVICE=cpu: ind / col / val are host arrays (psp_cpu.hip)
};

struct psp_sss {
  int n = 0, nnz_lower = 0;
  int *ind = nullptr;
  int *col = nullptr;
  double *val = nullptr;
  double *diag = nullptr;
  psp_csr *full = nullptr;  // expanded full-CSR device mirror used by matvec
  // sss_spmv_w4: offset-major values of the strict lower triangle + row masks (psp_csr.hip);
  // state -1 not examined, 0 not eligible, 1 built
  int w4_state = -1;
  int w4_nol = 0;
  int w4_offs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool w4_soa = false;  // w4_val: one array per offset (PSP_SSS_SOA) instead of 128-row blocks
  double *w4_val = nullptr;
  unsigned short *w4_mask = nullptr;
  bool host = false;  // PSP_DEVICE=cpu: ind / col / val / diag are host arrays (psp_cpu.hip)
};

enum psp_op_kind { PSP_OP_CSR = 1, PSP_OP_SSS = 2, PSP_OP_JACOBI = 3, PSP_OP_CALLBACK = 4, PSP_OP_SSOR = 5 };

struct psp_op {
  int kind = 0;
  int n = 0;
  psp_csr *csr = nullptr;
  psp_sss *sss = nullptr;
  psp_jacobi *jac = nullptr;
  psp_ssor *ssor = nullptr;
  psp_host_apply_fn fn = nullptr;
  void *ctx = nullptr;
  // pinned staging for callback operators
  double *hx = nullptr, *hy = nullptr;
};

struct psp_jacobi {
  int n = 0;
  double omega = 1.0;
  int steps = 1;
  double *dinv = nullptr;
  double *temp = nullptr;  // steps > 1
  psp_op A;                // operator for the extra sweeps (kind == 0 when absent)
  struct psp_mcsr *multi = nullptr;  // jacobi of a multi-device matrix: dinv lives with the row blocks (psp_multi.hip)
  bool host = false;                 // PSP_DEVICE=cpu: dinv / temp are host arrays (psp_cpu.hip)
};

namespace psp {
// the handles a call touches through operator `op`: the operator object itself (a callback operator owns pinned staging),
// the matrix / preconditioner behind it and, for a jacobi, the matrix its extra sweeps multiply with
inline void op_lock_add(HandleLock &L, const psp_op *op) {
  if (!op || op->kind == 0) return;
  L.add(op);
  L.add(op->csr);
  L.add(op->sss);
  L.add(op->ssor);
  if (op->jac) {
    L.add(op->jac);
    L.add(op->jac->A.csr);
    L.add(op->jac->A.sss);
  }
}
inline void jacobi_lock_add(HandleLock &L, const psp_jacobi *K) {
  if (!K) return;
  L.add(K);
  L.add(K->A.csr);
  L.add(K->A.sss);
}
}  // namespace psp
#define PSP_API_GUARD_OPS(A, K)      \
  psp::HandleLock psp_api_guard_;    \
  psp::op_lock_add(psp_api_guard_, A); \
  psp::op_lock_add(psp_api_guard_, K); \
  psp_api_guard_.lock()
#define PSP_API_GUARD_JAC(K)            \
  psp::HandleLock psp_api_guard_;       \
  psp::jacobi_lock_add(psp_api_guard_, K); \
  psp_api_guard_.lock()

namespace psp {
// Device-resident scalar state of the asynchronous PCG loop (psp_solvers.hip): the kernels
// of one iteration read alpha / beta / status from here instead of from host arguments, so
// the host can enqueue many iterations without synchronising.
struct PcgDev {
  double rho, rho1, alpha, beta, normr, tolb, n2b, relres;
  int status;  // 0 running, 1 finished (every later kernel is a no-op)
  int info, iter, stag;
  int it;      // iteration the enqueued kernels are working on (1-based), advanced on the device
  int maxit;
  // lazy x-update loop (pcg_async_loop_lazy): x += alpha_x * p of the last finished iteration is
  // applied by the NEXT iteration's p-update pass (or by the final pass after the loop)
  int xpend;       // an x update (and its stagnation scan) is pending
  int stag0;       // alpha == 0 in the pending iteration (pcg.c:124-125)
  int head_rho0;   // deferred exits of the iteration about to start: rho == 0 (pcg.c:101-104) ...
  int head_beta0;  // ... and beta == 0 (pcg.c:109-112); they come AFTER the pending stagnation test
  int pend_maxit;  // the loop ran out at it == maxit: -5 or -1 is decided by the final scan
  double alpha_x;
};
// Device-resident scalar state of the asynchronous MINRES loop (psp_solvers.hip; minres.c:96-193): the
// Lanczos / Givens recurrences are evaluated by the thread that finishes each reduction; the vector
// kernels read their coefficients from here.
struct MinresDev {
  double beta, beta_old, alpha, c, c_old, s, s_old, eta, norm_rmr, norm_r0, errtol, relres;
  double c1, c2;             // Lanczos coefficients of the running iteration: alpha/beta, beta/beta_old
  double r1, r2, r3, c_eta;  // its w / x update (minres.c:172-180)
  int status;  // 1: ended by -3 / -6 (minres.c:144-146, :160-162) or after the last w/x update: all kernels no-ops
  int stop;    // the loop test at the head of the NEXT iteration failed (minres.c:114): only the w/x update
               // of the running iteration is still to be done
  int skip;    // status | stop -- what the SpMV / scale / Lanczos kernels look at
  int info, iter, it_max;
};
// Device-resident scalars of the cgs / bicgstab / qmrs loops (round 5; psp_solvers.hip): the recurrences of
// cgs.c / bicgstab.c / qmrs.c are evaluated by the thread that finishes each reduction, the fused vector kernels read their
// coefficients from here, the host enqueues a batch of iterations and reads the state once -- no host round trip per
// reduction (they cost 12 us each at 1024^2: profiles/r4_extra_solvers_1024sq.txt).
struct KryDev {
  double r[32];
  int status;  // 1: the loop is over -- every later kernel is a no-op
  int code;    // which way it ended (per solver)
  int iter, maxit;
};
// scalar operands of a fused vector kernel: S == nullptr: the by-value arguments; else S->r[i0 / i1 / i2] (and the
// kernel does nothing once S->status is set)
struct KryArg {
  const KryDev *S = nullptr;
  int i0 = 0, i1 = 0, i2 = 0;
};
// y = op(x) on device vectors; y must not alias x
int op_apply(const psp_op *op, const double *x_dev, double *y_dev);
// the csr that a native operator multiplies with (csr, or sss->full); nullptr otherwise
inline psp_csr *op_native_csr(const psp_op *op) {
  if (!op) return nullptr;
  if (op->kind == PSP_OP_CSR) return op->csr;
  if (op->kind == PSP_OP_SSS) return op->sss->full;
  return nullptr;
}
int csr_spmv_launch(const psp_csr *A, const double *x, double *y, const double *dotv,
                    double *partials, int *nparts, const int *skip = nullptr);
// PCG: p_new = z + beta p_old and q = A p_new (+ p_new.q partials) in one pass over a w4 operator;
// *available = 0 when A has no such layout (nothing was launched)
int csr_spmv_pfused_launch(const psp_csr *A, const double *r, const double *dinv, const double *p_old,
                           double *p_new, double *q, double beta, bool first, double *partials, int *nparts,
                           const PcgDev *dstate, int *available);
// the lazy PCG loop's product on a w4 operator: the pending x update + stagnation scan of the previous iteration, p_new,
// q = A p_new and the p_new.q partials in one pass (psp_csr.hip csr_spmv_w4_pf<.., XU>); *available = 0 otherwise
int csr_spmv_pfx_launch(const psp_csr *A, const double *r, const double *dinv, const double *p_old, double *p_new,
                        double *q, double *x, double *partials, int *nparts, const PcgDev *dstate, int *available);
// MINRES: y = A (x ./ xdiv) + partials of (x ./ xdiv) . y on the index-free layouts; *available = 0 otherwise
// xdiv_dev != nullptr: the divisor is read from the device (asynchronous MINRES loop)
int csr_spmv_scaled_launch(const psp_csr *A, const double *x, double xdiv, double *y, double *partials,
                           int *nparts, int *available, const int *skip = nullptr,
                           const double *xdiv_dev = nullptr);
int csr_reordered_view(const psp_csr *A, psp_csr **R, const int **perm, const int **inv);  // psp_csr.hip
int reorder_gather(int n, const int *perm_dev, const double *x, double *xp, const int *skip);  // xp[i] = x[perm[i]]
// true when the SpMV kernel selected for A honours the `skip` flag (csr_spmv_w2)
bool csr_spmv_has_skip(const psp_csr *A);
int csr_spmv_overlap(const psp_csr *A, const double *x, double *y, const double *dotv,
                     double *partials, int *nparts, int row_a, int row_b, int (*wait)(void *),
                     void *ctx, const int *skip = nullptr);
// constant-vector registry (psp_vec.hip): the PCG vector kernels skip the dinv stream when the
// preconditioner's dinv holds one value everywhere
int dinv_register(const double *v, long n);
void dinv_unregister(const double *v);
bool dinv_constant(const double *v, long n, double *c);
int jacobi_apply_dev(psp_jacobi *K, const double *x, double *y);
// psp_csr.hip -> psp_mid.hip: an operator's index-free (csr_spmv_w4) layout
struct W4View {
  int no;       // offsets (<= 8)
  int offs[12];  // col - row, ascending (the first `no`)
  int grid3[3];  // nx, ny, nz when the operator is a 7-offset one of a 3-D grid without couplings across line ends, else 0
  int constv;    // 1: every stored entry at offset o has the value cval[o] (constant-coefficient stencil)
  double cval[12];
  const double *valT;          // blocks of 128 rows, offset-major inside a block
  const unsigned short *mask;  // bit o of mask[r]: row r stores an entry at offset o
  int stripe, grid;            // XCD stripe and grid of the launch-per-phase product (order of its dot partials)
};
int csr_w4_view(const psp_csr *A, W4View *out, int *available);
// psp_mid.hip: the whole PCG loop as one cooperative kernel for mid-size offset-structured systems (vectors in
// registers, the direction vector exchanged through LDS); kCoopFallback as for the small-system loops
bool mid_applicable(const psp_csr *A, int n, const double *dinv);
int pcg_mid_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                 double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres, double *hist);
// psp_mid.hip: the same for the 7-offset operators of 3-D grids, the points dealt out in bricks (iterates agree with the
// launch-per-phase loops' to rounding)
bool brick_applicable(const psp_csr *A, int n);
int pcg_brick_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                   double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres, double *hist);
bool brick_minres_applicable(const psp_csr *A, int n);
int minres_brick_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old, double *y,
                      double *w, double *w_old, double *v, double *av, double norm_r0, double beta0, double errtol, int it_max,
                      int *info, int *iter, double *relres, double *hist);
bool mid_minres_applicable(const psp_csr *A, int n);
int minres_mid_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old, double *y,
                    double *w, double *w_old, double *v, double *av, double norm_r0, double beta0, double errtol, int it_max,
                    int *info, int *iter, double *relres, double *hist);
// psp_coop.hip: the whole loop as one kernel for small systems (grid barriers instead of dependent launches).
// The two loops return kCoopFallback (not an error; nothing was changed) when the cooperative launch is refused or a
// grid barrier gives up: the caller then runs its launch-per-phase loop from the same vectors.
constexpr int kCoopFallback = 1;
bool coop_applicable(const psp_csr *A, int n);
int pcg_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                  double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres,
                  double *hist);
int minres_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old,
                     double *y, double *w, double *w_old, double *v, double *av, double norm_r0, double beta0,
                     double errtol, int it_max, int *info, int *iter, double *relres, double *hist);
// psp_cpu.hip: the host loops behind the entry points when PSP_DEVICE=cpu
namespace cpu {
int csr_create(int nrows, int ncols, int nnz, const int *ind, const int *col, const double *val, psp_csr **out);
int csr_destroy(psp_csr *A);
int csr_poisson(int nx, int ny, int nz, psp_csr **out);
int csr_download(const psp_csr *A, int *ind, int *col, double *val);
int csr_diagonal(const psp_csr *A, double *diag);
int csr_matvec(const psp_csr *A, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy, bool transp);
int sss_create(int n, int nnz, const int *ind, const int *col, const double *val, const double *diag, psp_sss **out);
int sss_destroy(psp_sss *S);
int sss_poisson(int nx, int ny, int nz, psp_sss **out);
int sss_download(const psp_sss *S, int *ind, int *col, double *val, double *diag);
int sss_getitem(const psp_sss *S, int i, int j, double *value);
int sss_matvec(const psp_sss *S, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy);
int jacobi_create(int n, const double *diag, double omega, int steps, const psp_op *A, psp_jacobi **out);
int jacobi_destroy(psp_jacobi *K);
int jacobi_precon(const psp_jacobi *K, const double *x, double *y);
int pcg(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol, int maxit, int *info,
        int *iter, double *relres, double *hist);
int minres(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol, int maxit, int *info,
           int *iter, double *relres, double *hist);
}  // namespace cpu
// psp_place.hip: placement of the vectors the library owns (solver work vectors, host-pointer staging)
bool placement_enabled();
bool placement_applies(const psp_csr *A, size_t n);
int place_operands(const psp_csr *A, size_t nx, size_t ny, int want_x, double **y_out, double **x_out, double *report);
// psp_csr.hip: device staging of the host-pointer products (kept between calls, released by psp_trim)
int host_stage(const psp_csr *A, size_t nx, size_t ny, double **x, double **y);
void host_stage_trim();
// psp_multi.hip: row-partitioned operators on a list of devices, one process
int multi_destroy(psp_mcsr *M);
int multi_matvec_host(psp_mcsr *M, const double *x_host, ptrdiff_t incx, double *y_host, ptrdiff_t incy);
int multi_diagonal_host(psp_mcsr *M, double *diag_host);
int multi_jacobi_setup(psp_mcsr *M, double omega);  // dinv slices = omega / diag on every rank (PSP_ESINGULAR)
int multi_jacobi_apply_host(psp_mcsr *M, const double *x_host, double *y_host);
int multi_pcg(psp_mcsr *M, bool jacobi, int n, double *x_host, const double *b_host, double tol, int maxit,
              int *info, int *iter, double *relres, double *hist_host);
int multi_minres(psp_mcsr *M, bool jacobi, int n, double *x_host, const double *b_host, double tol, int maxit,
                 int *info, int *iter, double *relres, double *hist_host);
int multi_describe(const psp_mcsr *M, char *buf, int cap);
int ssor_apply_dev(psp_ssor *K, const double *b, double *x);  // psp_ssor.hip
// waits for the error word of the last application's brick sweeps (sticky across applications until reported)
int ssor_error_check(psp_ssor *K);
int ssor_apply_host(psp_ssor *K, const double *b, double *x);  // psp_ssor.hip: PSP_DEVICE=cpu, host arrays
}  // namespace psp
