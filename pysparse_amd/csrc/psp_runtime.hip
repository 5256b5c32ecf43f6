// psp_runtime.hip -- device selection, stream, memory, events, reduction workspace.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "psp_internal.h"

namespace psp {

static thread_local std::string g_err;
static std::mutex g_mu;
// ---- the calling thread's context (psp_internal.h, "Threading model")
static std::atomic<int> g_default_device{0};  // what a new thread starts on: the device the process selected last
static std::vector<int> g_free_slots;         // thread slots given back by threads that ended (guarded by g_mu)
static int g_next_slot = 0;
struct ThreadCtx {
  int device = -1;     // -1: not initialised yet (takes g_default_device on first use)
  int dev_state = 0;   // 0 unknown, 1 ok, -1 none
  hipStream_t stream = nullptr;
  bool stream_given = false;  // psp_set_stream / use_device chose the stream: no automatic one
  int thread_slot = -1;
  int ws_slot = 0;
  std::map<int, hipStream_t> own;  // automatic non-blocking streams of a secondary thread, by device
  int slot() {
    if (thread_slot < 0) {
      std::lock_guard<std::mutex> lk(g_mu);
      if (!g_free_slots.empty()) {
        thread_slot = g_free_slots.back();
        g_free_slots.pop_back();
        if (thread_slot == 0) thread_slot = g_next_slot++;  // slot 0 (null stream) belongs to the first thread for good
      } else {
        thread_slot = g_next_slot++;
      }
    }
    return thread_slot;
  }
  int dev() {
    if (device < 0) device = g_default_device.load();
    return device;
  }
  ~ThreadCtx() {
    // the slot (and the reduction workspaces cached under it) goes to the next thread that asks: nothing of this thread
    // may still be in flight on them (best effort: the runtime may already be shutting down)
    for (auto &kv : own) {
      if (hipStreamSynchronize(kv.second) != hipSuccess) (void)hipGetLastError();
      (void)hipStreamDestroy(kv.second);
    }
    if (thread_slot > 0) {
      std::lock_guard<std::mutex> lk(g_mu);
      g_free_slots.push_back(thread_slot);  // its workspaces stay cached for the next thread that gets the slot
    }
  }
};
static thread_local ThreadCtx tl;
// one reduction workspace per (device, thread slot, rank slot): the multi-device driver switches devices and gives every
// rank its own slot -- ranks that share a device run on different streams and must not share partial-sum buffers
static std::map<long, Workspace> g_wss;

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

const char *last_error() { return g_err.c_str(); }

const char *tuning_env(const char *name) {
  static const bool on = [] {
    const char *e = getenv("PSP_TUNING");
    return e && e[0] == '1' && e[1] == 0;
  }();
  return on ? getenv(name) : nullptr;
}

// ---- which loop the calling thread's last solve ran (psp_last_solve_info), single-kernel loops on / off
struct SolveNote {
  char loop[48];
  int launches, vec_bytes, dinv_streamed;
};
static thread_local SolveNote g_note = {"", 0, 0, 0};
static std::atomic<long long> g_sk_fallbacks{0};
static std::atomic<int> g_sk_enabled{1};

void note_solve(const char *loop, int launches_per_iter, int vec_bytes_per_row, int dinv_streamed) {
  snprintf(g_note.loop, sizeof g_note.loop, "%s", loop);
  g_note.launches = launches_per_iter;
  g_note.vec_bytes = vec_bytes_per_row;
  g_note.dinv_streamed = dinv_streamed;
}
void note_fallback() { g_sk_fallbacks.fetch_add(1, std::memory_order_relaxed); }
bool single_kernel_loops_enabled() { return g_sk_enabled.load(std::memory_order_relaxed) != 0; }

void setup_mark(const char *label) {
  static const bool on = [] {
    const char *e = tuning_env("PSP_SETUP_TRACE");
    return e && atoi(e) != 0;
  }();
  if (!on) return;
  static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  if (tl.dev_state == 1) {
    if (hipStreamSynchronize(stream()) != hipSuccess) (void)hipGetLastError();
  }
  const auto now = std::chrono::steady_clock::now();
  if (label) fprintf(stderr, "[psp setup] %-44s +%9.3f ms\n", label, std::chrono::duration<double, std::milli>(now - last).count());
  last = now;
}

HandleEntry &handle_entry(const void *handle) {
  static std::mutex mu;
  static std::unordered_map<const void *, std::unique_ptr<HandleEntry>> tab;  // entries live as long as the process
  std::lock_guard<std::mutex> lk(mu);
  auto &e = tab[handle];
  if (!e) e.reset(new HandleEntry());
  return *e;
}

// ---- device-side order between the streams that use one handle (psp_internal.h "Device-side order between threads")
static std::atomic<int> g_track{0};                        // a second stream has been seen: keep last-use events
static std::atomic<unsigned long long> g_devices_used{0};  // devices some thread of the library has made current
static std::mutex g_first_mu;
static std::atomic<bool> g_first_set{false};  // stored (release) after g_first_stream: readers need no mutex
static hipStream_t g_first_stream = nullptr;  // the one stream everything ran on so far

void handles_enter(HandleEntry *const *e, int n) {
  if (cpu_mode()) return;
  if (tl.dev_state == 0) (void)ensure_device();  // a thread's first call: its stream must exist before it can wait
  if (tl.dev_state != 1) return;                 // no device: nothing was, or can be, enqueued
  const hipStream_t cur = stream();
  if (!g_track.load(std::memory_order_acquire) &&
      !(g_first_set.load(std::memory_order_acquire) && cur == g_first_stream)) {  // the one-stream process: no mutex
    std::lock_guard<std::mutex> lk(g_first_mu);
    if (!g_first_set.load(std::memory_order_relaxed)) {
      g_first_stream = cur;
      g_first_set.store(true, std::memory_order_release);
    } else if (cur != g_first_stream && !g_track.load(std::memory_order_relaxed)) {
      // the second stream of the process.  Tracking starts FIRST, so that every handles_leave from here on records its
      // event -- also the one of a call that is in flight on the first stream right now; the one-off synchronisation
      // below then only has to cover what was enqueued before events were kept
      g_track.store(1, std::memory_order_release);
      const unsigned long long used = g_devices_used.load();
      for (int d = 0; d < 64; ++d)
        if (used >> d & 1ull) {
          if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
          (void)hipGetLastError();
        }
      (void)hipSetDevice(tl.dev());
    }
  }
  if (!g_track.load(std::memory_order_acquire)) return;
  for (int i = 0; i < n; ++i)
    if (e[i]->has_last && e[i]->last != cur) {
      if (hipStreamWaitEvent(cur, e[i]->ev, 0) != hipSuccess) (void)hipGetLastError();
    }
}

void handles_leave(HandleEntry *const *e, int n) {
  if (!g_track.load(std::memory_order_acquire) || cpu_mode() || tl.dev_state != 1) return;
  const hipStream_t cur = tl.stream;  // what the call enqueued on (stream() ran in handles_enter)
  const int d = tl.dev();
  for (int i = 0; i < n; ++i) {
    HandleEntry &h = *e[i];
    if (h.ev && h.ev_dev != d) {  // the handle moved to a stream of another device: events belong to a device
      (void)hipEventDestroy(h.ev);
      h.ev = nullptr;
    }
    if (!h.ev) {
      if (hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        h.ev = nullptr;
        h.has_last = false;
        continue;
      }
      h.ev_dev = d;
    }
    if (hipEventRecord(h.ev, cur) == hipSuccess) {
      h.last = cur;
      h.has_last = true;
    } else {
      (void)hipGetLastError();
      h.has_last = false;
    }
  }
}

hipStream_t stream() {
  if (!tl.stream_given && tl.stream == nullptr && tl.slot() > 0 && !cpu_mode()) {
    // a secondary thread: its own non-blocking stream on the device it is on (created once per device)
    const int d = tl.dev();
    auto it = tl.own.find(d);
    if (it == tl.own.end()) {
      hipStream_t s = nullptr;
      if (hipSetDevice(d) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess)
        it = tl.own.emplace(d, s).first;
      else
        (void)hipGetLastError();
    }
    if (it != tl.own.end()) tl.stream = it->second;
  }
  return tl.stream;
}
hipStream_t swap_stream(hipStream_t s) {
  hipStream_t old = stream();
  tl.stream = s;
  tl.stream_given = true;
  return old;
}

int ensure_device() {
  if (cpu_mode())
    return fail(PSP_ENODEV, "PSP_DEVICE=cpu: this entry point has no host loop (it runs on the GPU only); the host mode "
                            "covers csr_mat / sss_mat products, jacobi, pcg and minres");
  (void)tl.slot();
  if (tl.dev_state == 1) return PSP_OK;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0) {
    tl.dev_state = -1;
    return fail(PSP_ENODEV,
                "no HIP device available (%s); libpysparse_hip has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (tl.dev() >= cnt)
    return fail(PSP_ENODEV, "device %d requested but only %d visible", tl.dev(), cnt);
  PSP_HIP(hipSetDevice(tl.dev()));  // the HIP runtime's current device is per host thread too
  tl.dev_state = 1;
  g_devices_used.fetch_or(1ull << (tl.dev() & 63));
  return PSP_OK;
}

int workspace(Workspace **out) {
  PSP_TRY(ensure_device());
  const int d = tl.dev();
  std::lock_guard<std::mutex> lk(g_mu);
  Workspace &ws = g_wss[((long)d << 40) | ((long)tl.thread_slot << 16) | (long)tl.ws_slot];
  if (ws.device != d) {
    hipDeviceProp_t prop;
    PSP_HIP(hipGetDeviceProperties(&prop, d));
    ws.num_cu = prop.multiProcessorCount;
    // all or nothing: a failed allocation releases what the earlier ones got (ws.device stays unset, so the next call
    // starts over -- without this it would allocate everything again on top of the survivors; round-5 advisor finding)
    auto undo = [&ws]() {
      if (ws.partials) (void)hipFree(ws.partials);
      if (ws.folded) (void)hipFree(ws.folded);
      if (ws.scal_dev) (void)hipFree(ws.scal_dev);
      if (ws.scal_host) (void)hipHostFree(ws.scal_host);
      if (ws.state_dev) (void)hipFree(ws.state_dev);
      if (ws.state_host) (void)hipHostFree(ws.state_host);
      if (ws.ctl_part) (void)hipFree(ws.ctl_part);
      ws.partials = ws.folded = ws.scal_dev = ws.scal_host = ws.scal_host_dev = ws.ctl_part = nullptr;
      ws.state_dev = ws.state_host = nullptr;
      (void)hipGetLastError();
    };
#define WS_HIP(call)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      undo();                                                                                             \
      return fail(e_ == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "%s: %s", #call, hipGetErrorString(e_)); \
    }                                                                                                     \
  } while (0)
    WS_HIP(hipMalloc((void **)&ws.partials, sizeof(double) * kSlots * kMaxParts));
    WS_HIP(hipMalloc((void **)&ws.folded, sizeof(double) * kSlots * kTailGroups));
    WS_HIP(hipMalloc((void **)&ws.scal_dev, sizeof(double) * 16));
    // 16 doubles + a sequence word (fetch_scalars): pinned, mapped, coherent -- the device stores into it directly
    bool mapped = true;
    if (hipHostMalloc((void **)&ws.scal_host, sizeof(double) * 24, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();  // no mapped coherent host memory here: plain pinned memory, scalars come back by copy
      mapped = false;
      ws.scal_host = nullptr;
      WS_HIP(hipHostMalloc((void **)&ws.scal_host, sizeof(double) * 24, hipHostMallocDefault));
    }
    memset(ws.scal_host, 0, sizeof(double) * 24);
    ws.scal_host_dev = nullptr;
    if (!mapped || hipHostGetDevicePointer((void **)&ws.scal_host_dev, ws.scal_host, 0) != hipSuccess) {
      (void)hipGetLastError();
      ws.scal_host_dev = nullptr;  // fetch_scalars copies then
    }
    WS_HIP(hipMalloc(&ws.state_dev, kStateBytes));
    WS_HIP(hipHostMalloc(&ws.state_host, kStateBytes, hipHostMallocDefault));
    WS_HIP(hipMalloc((void **)&ws.ctl_part, sizeof(double) * kCtlPartDoubles));
#undef WS_HIP
    ws.device = d;
  }
  *out = &ws;
  return PSP_OK;
}

int use_device(int device, hipStream_t s, int ws_slot) {
  PSP_HIP(hipSetDevice(device));  // unconditionally: the multi-device driver also switches with plain hipSetDevice
  (void)tl.slot();
  tl.device = device;
  tl.dev_state = 1;
  tl.stream = s;
  tl.stream_given = true;
  tl.ws_slot = ws_slot;
  g_devices_used.fetch_or(1ull << (device & 63));
  return PSP_OK;
}

// what a multi-device entry point puts back when it returns (psp_multi.hip DeviceRestore): the thread's context as it was,
// including WHETHER its stream had been chosen by the caller -- going through use_device would pin a secondary thread's
// automatic per-device stream as if the user had supplied it, and a later psp_set_device would keep enqueueing on a
// stream of the old device
ThreadCtxSave save_thread_ctx() { return {tl.device, tl.dev_state, tl.ws_slot, tl.stream, tl.stream_given}; }
void restore_thread_ctx(const ThreadCtxSave &c) {
  if (c.device >= 0) {
    if (hipSetDevice(c.device) != hipSuccess) (void)hipGetLastError();
  }
  tl.device = c.device;
  tl.dev_state = c.dev_state;
  tl.ws_slot = c.ws_slot;
  tl.stream = c.stream;
  tl.stream_given = c.stream_given;
}

int current_device() { return tl.dev(); }
int current_ws_slot() { return tl.ws_slot; }
int current_thread_slot() { return tl.slot(); }

// ---- delay injection (psp_internal.h "shake")
// one wave that does nothing until `ticks` of the 100 MHz wall clock have passed; the poll count is bounded too, so the
// kernel ends even where the clock does not advance
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 22) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}

namespace {
struct ShakeState {
  std::atomic<int> armed{0};
  std::mutex mu;
  unsigned long long rng = 0;
  int min_us = 0, max_us = 0;
  unsigned points = 0, ranks = 0;
  int revert = 0;
  long long injected = 0;
  bool env_read = false;
};
ShakeState g_shake;

void shake_configure(long long seed, int min_us, int max_us, unsigned points, unsigned ranks, int revert) {
  std::lock_guard<std::mutex> lk(g_shake.mu);
  g_shake.rng = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)seed * 0xD1B54A32D192ED03ull + 1ull);
  g_shake.min_us = std::max(0, min_us);
  g_shake.max_us = std::min(100000, std::max(g_shake.min_us, max_us));  // <= 100 ms per injection
  g_shake.points = points;
  g_shake.ranks = ranks;
  g_shake.revert = revert;
  g_shake.injected = 0;
  g_shake.armed.store(seed >= 0 && (points || revert) ? 1 : 0, std::memory_order_release);
}

void shake_read_env() {  // PSP_SHAKE="seed,min_us,max_us,points,ranks,revert" (tuning switch), once
  static std::once_flag once;
  std::call_once(once, [] {
    const char *e = tuning_env("PSP_SHAKE");
    if (!e) return;
    long long seed = 0;
    int mn = 0, mx = 200, rv = 0;
    unsigned pts = 0xffffffffu, rks = 0xffffffffu;
    (void)sscanf(e, "%lld,%d,%d,%i,%i,%d", &seed, &mn, &mx, (int *)&pts, (int *)&rks, &rv);
    shake_configure(seed, mn, mx, pts, rks, rv);
  });
}
}  // namespace

int shake(hipStream_t s, int point, int rank) {
  shake_read_env();
  if (!g_shake.armed.load(std::memory_order_acquire)) return PSP_OK;
  long long us;
  {
    std::lock_guard<std::mutex> lk(g_shake.mu);
    if (!(g_shake.points >> point & 1u) || !(g_shake.ranks >> (rank & 31) & 1u)) return PSP_OK;
    unsigned long long x = g_shake.rng;  // xorshift64*: the same seed draws the same delays
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    g_shake.rng = x;
    const unsigned long long r = x * 0x2545F4914F6CDD1Dull;
    if (g_shake.min_us == g_shake.max_us) {
      us = g_shake.max_us;
    } else {
      if (r >> 63) return PSP_OK;  // half of the visits inject nothing: a delayed stream next to an undelayed one
      us = g_shake.min_us + (long long)((r >> 20) % (unsigned long long)(g_shake.max_us - g_shake.min_us + 1));
    }
    if (us <= 0) return PSP_OK;
    g_shake.injected += 1;
  }
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, us * 100);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

bool shake_armed() {
  shake_read_env();
  return g_shake.armed.load(std::memory_order_acquire) != 0;
}

bool shake_revert(int bit) {
  shake_read_env();
  return g_shake.armed.load(std::memory_order_acquire) && (g_shake.revert & bit) != 0;
}

// psp_stream_probe: R read streams (the first with ordinary loads, the others non-temporal, like the value streams
// of csr_spmv_w4) and optionally one non-temporal write stream; one 16-byte element per thread and stream, full grid
template <int R, bool W>
__global__ __launch_bounds__(256) void stream_probe_kernel(const double2 *__restrict__ a, double2 *__restrict__ b,
                                                           long n2) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  const d2 *p = reinterpret_cast<const d2 *>(a);
  d2 s = d2{1.5, 2.5};
  if (R > 0) s = p[i];
#pragma unroll
  for (int k = 1; k < R; ++k) {
    const d2 v = __builtin_nontemporal_load(p + i + (long)k * n2);
    s.x += v.x;
    s.y += v.y;
  }
  if (W) __builtin_nontemporal_store(s, reinterpret_cast<d2 *>(b) + i);
  else if (s.x + s.y == 12345.678) b[0] = double2{s.x, s.y};  // never true: keeps the loads alive
}

// first level of reduce() (psp_internal.h) for up to two sets of partial sums in one launch (blockIdx.y = job * nvals
// of job 0 ...): out[j*kTailGroups + g] = R(in[j*kMaxParts + 256 g .. + 256)), one wave per group, four groups per
// workgroup
__global__ __launch_bounds__(256) void group_fold_kernel(FoldJobs jobs) {
  int y = blockIdx.y, k = 0;
  if (y >= jobs.nvals[0]) {
    y -= jobs.nvals[0];
    k = 1;
  }
  const int nparts = jobs.nparts[k];
  const int ngroups = (nparts + kTailGroup - 1) / kTailGroup;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= ngroups) return;
  const int gsize = min(kTailGroup, nparts - g * kTailGroup);
  const double s = psp_wave_reduce<false>(jobs.in[k] + (size_t)y * kMaxParts + (size_t)g * kTailGroup, gsize);
  if ((threadIdx.x & 63) == 0) jobs.out[k][(size_t)y * kTailGroups + g] = s;
}

// one block of 1024 threads (reduce_block, psp_internal.h)
// host_dst != nullptr: the sums also go to mapped host memory, a sequence number behind them (finish_partials_fetch)
__global__ __launch_bounds__(kReduceBlock) void finish_kernel(const double *__restrict__ partials, int nparts, int nvals,
                                                              int stride, int raw, double *out, double *host_dst,
                                                              unsigned long long *seq_word, unsigned long long seq) {
  __shared__ double sh[kOneBlockGroups];
  reduce_block(partials, nparts, nvals, stride, raw != 0, out, sh);
  if (host_dst) {
    __syncthreads();  // out[] was written by whichever lanes finished the sums
    if (threadIdx.x < 64) {  // ONE wave publishes: a system-scope fence writes the L2 back, sixteen of them do it sixteen times
      if ((int)threadIdx.x < nvals)
        __builtin_nontemporal_store(__hip_atomic_load(out + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    host_dst + threadIdx.x);
      __threadfence_system();  // the wave's stores are complete (and visible to the host) before ...
      if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Up to how many groups of 256 partial sums the finishing block reads the raw partial sums itself.  One workgroup pulls
// its operands at ~45 GB/s (profiles/r4_c2_kernel_durations.txt: 22 us for 32 768 x 3 partial sums at 4096^2), a group
// fold over the whole chip + the finishing block cost two launches (~10 us): PSP_FOLD_ONE_BLOCK_GROUPS (tuning switch,
// read per reduction; tools/fold_threshold_ab.py) moves the boundary; both routes add in the same order (R, psp_internal.h)
// A reduction of ONE value reads a third of what the PCG reductions read (two or three values): its boundary lies at
// three times as many groups (PSP_FOLD_ONE_BLOCK_GROUPS1; MINRES' alpha over the ~10 000 workgroup sums of csr_spmv_w3 at
// n = 9.3e5 is one launch instead of two: profiles/r5_fem_minres.txt).
static int one_block_groups(int nvals) {
  const char *e = psp::tuning_env(nvals == 1 ? "PSP_FOLD_ONE_BLOCK_GROUPS1" : "PSP_FOLD_ONE_BLOCK_GROUPS");
  if (e) {
    const int v = atoi(e);
    if (v >= 1 && v <= kOneBlockGroups) return v;
  }
  return nvals == 1 ? 3 * kFoldAboveGroups : kFoldAboveGroups;
}

// first stage of a reduction over more partials than one block takes: the group sums (into slot `fslot` of the
// workspace's group-sum array); tells the caller what the finishing block has to read (raw: still the per-workgroup
// partial sums).  fold_stage2: two sets of partial sums, ONE launch when both need the stage.
static void fold_plan(Workspace *w, const double *partials, int nparts, int nvals, int fslot, const double **src, int *count,
                      int *stride, bool *raw, bool *need) {
  const int ngroups = (nparts + kTailGroup - 1) / kTailGroup;
  *need = ngroups > one_block_groups(nvals);
  if (*need) {
    *src = w->folded + (size_t)fslot * kTailGroups;
    *count = ngroups;
    *stride = kTailGroups;
    *raw = false;
  } else {
    *src = partials;
    *count = nparts;
    *stride = kMaxParts;
    *raw = true;
  }
}

int fold_stage(const double *partials, int nparts, int nvals, const double **src, int *count, int *stride, bool *raw,
               int fslot) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  bool need;
  fold_plan(w, partials, nparts, nvals, fslot, src, count, stride, raw, &need);
  if (need) {
    FoldJobs jobs;
    jobs.in[0] = partials;
    jobs.nparts[0] = nparts;
    jobs.nvals[0] = nvals;
    jobs.out[0] = w->folded + (size_t)fslot * kTailGroups;
    hipLaunchKernelGGL(group_fold_kernel, dim3((*count + 3) / 4, nvals), dim3(256), 0, stream(), jobs);
    PSP_LAUNCH_CHECK();
  }
  return PSP_OK;
}

int fold_stage2(const double *const partials[2], const int nparts[2], const int fslot[2], const double *src[2],
                int count[2], int stride[2], bool raw[2]) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  bool need[2];
  for (int k = 0; k < 2; ++k)
    fold_plan(w, partials[k], nparts[k], 2, fslot[k], &src[k], &count[k], &stride[k], &raw[k], &need[k]);
  if (need[0] || need[1]) {
    FoldJobs jobs;
    int nj = 0, maxg = 0;
    for (int k = 0; k < 2; ++k)
      if (need[k]) {
        jobs.in[nj] = partials[k];
        jobs.nparts[nj] = nparts[k];
        jobs.nvals[nj] = 1;
        jobs.out[nj] = w->folded + (size_t)fslot[k] * kTailGroups;
        maxg = std::max(maxg, count[k]);
        ++nj;
      }
    if (nj == 1) jobs.nvals[1] = 0;
    hipLaunchKernelGGL(group_fold_kernel, dim3((maxg + 3) / 4, nj), dim3(256), 0, stream(), jobs);
    PSP_LAUNCH_CHECK();
  }
  return PSP_OK;
}

int finish_partials(const double *partials, int nparts, int nvals, double *out_dev) {
  const double *src;
  int count, stride;
  bool raw;
  PSP_TRY(fold_stage(partials, nparts, nvals, &src, &count, &stride, &raw, 0));
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kReduceBlock), 0, stream(), src, count, nvals, stride, raw ? 1 : 0,
                     out_dev, (double *)nullptr, (unsigned long long *)nullptr, 0ull);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// k <= 16 scalars device -> host.  The host-scalar loops (generic PCG / MINRES, cgs, bicgstab, qmrs, gmres) do this several
// times per iteration; as a copy + stream synchronisation it left the GPU idle ~18 us each time (trace of the four solvers
// at 4096^2, profiles/r4_host_scalar_readback.txt).  Instead a one-wave kernel stores the values into mapped host memory,
// then a sequence number behind a system-scope fence; the host polls the sequence word (a store over the link is visible
// within a microsecond or two).  Bounded: after 2 s of polling, or where the mapping is not available, or with
// PSP_FETCH_POLL=0 (tuning switch), the old path runs -- and reports whatever error the stream holds.
__global__ void publish_scalars_kernel(const double *__restrict__ src, int k, double *dst, unsigned long long *seq_word,
                                       unsigned long long seq) {
  if ((int)threadIdx.x < k) __builtin_nontemporal_store(src[threadIdx.x], dst + threadIdx.x);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static bool fetch_poll_enabled() {
  const char *e = tuning_env("PSP_FETCH_POLL");
  return !e || atoi(e) != 0;
}

// wait (at most 2 s) until the kernel that carries `seq` has stored its values into the workspace's mapped host words
static bool poll_scalars(Workspace *w, unsigned long long seq, int k, double *dst_host) {
  volatile unsigned long long *word = reinterpret_cast<volatile unsigned long long *>(w->scal_host + 16);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (*word != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xfff) == 0) {  // a value is normally there within microseconds; beyond ~4000 polls give the core away
      std::this_thread::yield();
      if ((spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
  }
  if (*word != seq) {
    // stuck, faulted or starved stream: the synchronising copy reports it, and this workspace does not poll again (a
    // second publish kernel + another 2 s of spinning per read-back would be the price on a GPU shared with other jobs)
    w->scal_host_dev = nullptr;
    return false;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  memcpy(dst_host, w->scal_host, sizeof(double) * k);
  return true;
}

// finish_partials + fetch_scalars in one: the finishing block itself stores the sums to the host (one launch and one
// dependent kernel start less per read-back of the host-scalar loops)
int finish_partials_fetch(const double *partials, int nparts, int nvals, double *out_dev, double *dst_host) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  if (!(nvals <= 16 && w->scal_host_dev && fetch_poll_enabled())) {
    PSP_TRY(finish_partials(partials, nparts, nvals, out_dev));
    return fetch_scalars(out_dev, nvals, dst_host);
  }
  const double *src;
  int count, stride;
  bool raw;
  PSP_TRY(fold_stage(partials, nparts, nvals, &src, &count, &stride, &raw, 0));
  const unsigned long long seq = ++w->scal_seq;
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kReduceBlock), 0, stream(), src, count, nvals, stride, raw ? 1 : 0,
                     out_dev, w->scal_host_dev, reinterpret_cast<unsigned long long *>(w->scal_host_dev + 16), seq);
  PSP_LAUNCH_CHECK();
  if (poll_scalars(w, seq, nvals, dst_host)) return PSP_OK;
  return fetch_scalars(out_dev, nvals, dst_host);  // timed out (polling is off for this workspace now): copy + synchronise
}

int fetch_scalars(const double *src_dev, int k, double *dst_host) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  if (k <= 16 && w->scal_host_dev && fetch_poll_enabled()) {
    const unsigned long long seq = ++w->scal_seq;
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, stream(), src_dev, k, w->scal_host_dev,
                       reinterpret_cast<unsigned long long *>(w->scal_host_dev + 16), seq);
    if (hipGetLastError() == hipSuccess && poll_scalars(w, seq, k, dst_host)) return PSP_OK;
    // fall through: the stream is stuck or faulted -- let the synchronising path say so
  }
  PSP_HIP(hipMemcpyAsync(w->scal_host, src_dev, sizeof(double) * k, hipMemcpyDeviceToHost,
                         stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  memcpy(dst_host, w->scal_host, sizeof(double) * k);
  return PSP_OK;
}

}  // namespace psp

using namespace psp;

extern "C" {

const char *psp_last_error(void) { return psp::last_error(); }
const char *psp_version(void) {
  return psp::cpu_mode() ? "pysparse_hip 0.1 (gfx950; PSP_DEVICE=cpu: host loops, no GPU used)" : "pysparse_hip 0.1 (gfx950)";
}

int psp_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

int psp_peer_access(int device, int peer, int *can_access) {
  if (!can_access) return fail(PSP_EINVAL, "psp_peer_access: NULL argument");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
    return fail(PSP_ENODEV, "no HIP device available; libpysparse_hip has no CPU fallback");
  if (device < 0 || device >= cnt || peer < 0 || peer >= cnt)
    return fail(PSP_EINVAL, "device %d / peer %d out of range (0..%d)", device, peer, cnt - 1);
  if (device == peer) {
    *can_access = 1;
    return PSP_OK;
  }
  PSP_HIP(hipDeviceCanAccessPeer(can_access, device, peer));
  return PSP_OK;
}

int psp_set_device(int device) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
    return fail(PSP_ENODEV, "no HIP device available; libpysparse_hip has no CPU fallback");
  if (device < 0 || device >= cnt)
    return fail(PSP_EINVAL, "device %d out of range (0..%d)", device, cnt - 1);
  PSP_HIP(hipSetDevice(device));
  (void)tl.slot();
  if (tl.device != device && !tl.stream_given) tl.stream = nullptr;  // a secondary thread's own stream is per device
  tl.device = device;
  tl.dev_state = 1;
  g_devices_used.fetch_or(1ull << (device & 63));
  g_default_device.store(device);  // threads that start later begin here
  return PSP_OK;
}

int psp_set_stream(void *hip_stream) {
  (void)tl.slot();
  tl.stream = (hipStream_t)hip_stream;
  tl.stream_given = true;
  return PSP_OK;
}

int psp_debug_hold_handles(const void *h1, const void *h2, int milliseconds) {
  PSP_API_GUARD_H(h1, h2);
  if (milliseconds > 0) std::this_thread::sleep_for(std::chrono::milliseconds(milliseconds));
  return PSP_OK;
}

int psp_debug_shake(long long seed, int min_us, int max_us, unsigned point_mask, unsigned rank_mask, int revert_mask) {
  if (!psp::tuning_env("PSP_TUNING"))  // (PSP_TUNING itself is set whenever the tuning switches are honoured)
    return fail(PSP_EINVAL, "psp_debug_shake: delay injection needs a process started with PSP_TUNING=1");
  psp::shake_read_env();  // an environment setting is consumed first, so that it cannot overwrite this call later
  psp::shake_configure(seed, min_us, max_us, point_mask, rank_mask, revert_mask);
  return PSP_OK;
}

int psp_debug_shake_count(long long *injected) {
  if (!injected) return fail(PSP_EINVAL, "psp_debug_shake_count: NULL argument");
  std::lock_guard<std::mutex> lk(psp::g_shake.mu);
  *injected = psp::g_shake.injected;
  return PSP_OK;
}

int psp_debug_spin(int microseconds) {
  if (!psp::tuning_env("PSP_TUNING"))
    return fail(PSP_EINVAL, "psp_debug_spin: needs a process started with PSP_TUNING=1");
  PSP_TRY(ensure_device());
  if (microseconds <= 0) return PSP_OK;
  hipLaunchKernelGGL(psp::spin_kernel, dim3(1), dim3(64), 0, stream(), (long long)std::min(microseconds, 100000) * 100);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int psp_thread_info(int *thread_slot, int *device, void **hip_stream) {
  if (thread_slot) *thread_slot = current_thread_slot();
  if (device) *device = current_device();
  if (hip_stream) *hip_stream = (void *)psp::stream();
  return PSP_OK;
}

int psp_synchronize(void) {
  if (psp::cpu_mode()) return PSP_OK;
  PSP_TRY(ensure_device());
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int psp_device_info(char *name, int name_len, int *compute_units, int64_t *hbm_bytes) {
  PSP_TRY(ensure_device());
  hipDeviceProp_t prop;
  PSP_HIP(hipGetDeviceProperties(&prop, tl.dev()));
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return PSP_OK;
}

int psp_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
  PSP_TRY(ensure_device());
  size_t f = 0, t = 0;
  PSP_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return PSP_OK;
}

int psp_malloc(void **dev, size_t bytes) {
  if (!dev) return fail(PSP_EINVAL, "psp_malloc: NULL out pointer");
  PSP_TRY(ensure_device());
  if (hipMalloc(dev, bytes ? bytes : 8) != hipSuccess) {
    (void)hipGetLastError();
    (void)psp_trim();  // the solvers' cached work vectors may be what fills the device
    PSP_HIP(hipMalloc(dev, bytes ? bytes : 8));
  }
  return PSP_OK;
}

int psp_free(void *dev) {
  if (!dev) return PSP_OK;
  PSP_HIP(hipFree(dev));
  return PSP_OK;
}

int psp_memcpy_h2d(void *dev, const void *host, size_t bytes) {
  PSP_TRY(ensure_device());
  PSP_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int psp_memcpy_d2h(void *host, const void *dev, size_t bytes) {
  PSP_TRY(ensure_device());
  PSP_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int psp_memset(void *dev, int byte, size_t bytes) {
  PSP_TRY(ensure_device());
  PSP_HIP(hipMemsetAsync(dev, byte, bytes, stream()));
  return PSP_OK;
}

int psp_event_create(void **event) {
  PSP_TRY(ensure_device());
  hipEvent_t e;
  PSP_HIP(hipEventCreate(&e));
  *event = (void *)e;
  return PSP_OK;
}

int psp_event_destroy(void *event) {
  if (event) PSP_HIP(hipEventDestroy((hipEvent_t)event));
  return PSP_OK;
}

int psp_event_record(void *event) {
  PSP_HIP(hipEventRecord((hipEvent_t)event, stream()));
  return PSP_OK;
}

int psp_event_elapsed_ms(void *start, void *stop, float *ms) {
  PSP_HIP(hipEventSynchronize((hipEvent_t)stop));
  PSP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return PSP_OK;
}

int psp_last_solve_info(char *name, int name_cap, int *info) {
  if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", psp::g_note.loop);
  if (info) {
    info[0] = psp::g_note.launches;
    info[1] = psp::g_note.vec_bytes;
    info[2] = psp::g_note.dinv_streamed;
    const long long f = psp::g_sk_fallbacks.load(std::memory_order_relaxed);
    info[3] = f > 0x7fffffffLL ? 0x7fffffff : (int)f;
  }
  return PSP_OK;
}

int psp_set_single_kernel_loops(int on) {
  psp::g_sk_enabled.store(on ? 1 : 0, std::memory_order_relaxed);
  return PSP_OK;
}

const char *psp_build_id(void) {
#ifdef PSP_BUILD_ID
  return PSP_BUILD_ID;
#else
  return "unstamped";
#endif
}

int psp_stream_probe(int reads, int writes, size_t bytes_per_stream, int reps, float *avg_ms, float *min_ms) {
  PSP_API_GUARD;
  if (reads < 0 || reads > 8 || writes < 0 || writes > 1 || reads + writes == 0 || reps < 1 || !avg_ms ||
      bytes_per_stream < 4096 || (bytes_per_stream & 4095))
    return psp::fail(PSP_EINVAL, "psp_stream_probe: 0..8 reads, 0..1 writes, stream size a multiple of 4096 bytes");
  PSP_TRY(psp::ensure_device());
  double2 *a = nullptr, *b = nullptr;
  const long n2 = (long)(bytes_per_stream / 16);
  hipError_t e = hipMalloc((void **)&a, bytes_per_stream * (size_t)(reads ? reads : 1));
  if (e == hipSuccess) e = hipMalloc((void **)&b, bytes_per_stream);
  if (e != hipSuccess) {
    if (a) (void)hipFree(a);
    return psp::fail(PSP_ENOMEM, "psp_stream_probe: %s", hipGetErrorString(e));
  }
  (void)hipMemsetAsync(a, 0, bytes_per_stream * (size_t)(reads ? reads : 1), psp::stream());
  (void)hipMemsetAsync(b, 0, bytes_per_stream, psp::stream());
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((n2 + 255) / 256);
  float sum = 0.f, best = 3.4e38f;
  int rc = PSP_OK;
  for (int r = -2; r < reps && rc == PSP_OK; ++r) {  // two untimed launches first
    (void)hipEventRecord(e0, psp::stream());
    switch (reads) {
#define PSP_PROBE_CASE(R)                                                                                          \
  case R:                                                                                                          \
    if (writes) hipLaunchKernelGGL((psp::stream_probe_kernel<R, true>), dim3(grid), dim3(256), 0, psp::stream(), a, b, n2); \
    else hipLaunchKernelGGL((psp::stream_probe_kernel<R, false>), dim3(grid), dim3(256), 0, psp::stream(), a, b, n2);       \
    break;
      PSP_PROBE_CASE(0) PSP_PROBE_CASE(1) PSP_PROBE_CASE(2) PSP_PROBE_CASE(3) PSP_PROBE_CASE(4) PSP_PROBE_CASE(5)
      PSP_PROBE_CASE(6) PSP_PROBE_CASE(7) PSP_PROBE_CASE(8)
#undef PSP_PROBE_CASE
    }
    (void)hipEventRecord(e1, psp::stream());
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
      rc = psp::fail(PSP_ENODEV, "psp_stream_probe: %s", hipGetErrorString(hipGetLastError()));
    if (r >= 0) {
      sum += ms;
      if (ms < best) best = ms;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (rc != PSP_OK) return rc;
  *avg_ms = sum / (float)reps;
  if (min_ms) *min_ms = best;
  return PSP_OK;
}

}  // extern "C"
