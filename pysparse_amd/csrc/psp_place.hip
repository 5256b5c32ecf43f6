// psp_place.hip -- where the vectors the LIBRARY owns lie in device memory (round 5).
//
// Round 4 (profiles/r4_modes.txt) established that what a bandwidth-bound product takes depends on WHICH PAGES of HBM its
// vector operands occupy: the same launch of csr_spmv_w4 at 512^3 sits on one of two levels ~8 % apart (1.50-1.56 ms /
// 1.63-1.70 ms) according to the allocation its y -- or its x -- came from; consecutive allocations share a level, a
// buffer keeps its level for as long as it lives, and nothing user space can see (offset, memory type, translation
// reach, load latency) tells the levels apart beforehand.  The caller's own vectors (psp_csr_matvec_dev: bench.py's
// `value`) are the caller's business.  But inside a solve (itsolversmodule.c:32-118: the solver owns its work array) and in
// the host-pointer products the library allocates the operands of every product itself -- and can simply LOOK: draw a few
// candidate buffers spread over the address space, time the handle's own product on each, keep the best for the role it
// was best in (y: the product's output, x: its input), release the rest.  Once per (device, vector length) in a process:
// the winners stay in the solvers' scratch pool (psp_solvers.hip) / the thread's staging pair (psp_csr.hip).
// Same kernels on other addresses: no bit of any result changes.  Opt-in: psp_set_placement(1) (see placement_enabled
// for what it measured to be worth).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <vector>

#include "psp_internal.h"

namespace psp {

namespace {
std::atomic<int> g_place{-1};  // -1: not decided yet (PSP_PLACE under PSP_TUNING=1, else off)
std::atomic<long long> g_draws{0};
std::atomic<long long> g_draw_us{0};
}  // namespace

bool placement_enabled() {
  int v = g_place.load(std::memory_order_relaxed);
  if (v < 0) {
    // OFF unless asked for (psp_set_placement(1); PSP_PLACE=1 under PSP_TUNING=1).  Measured over 11 fresh processes at
    // 512^3 (profiles/r5_placement_ab_*.jsonl): the product on the drawn pair 0 ... 4.6 % faster than on the process'
    // first allocations (mean 1.8 %: in most processes NONE of 8 candidates spread over 22 GB lies on the fast level),
    // Jacobi-PCG +0.5 ... +1.7 % in the mean, MINRES 0 %; a draw costs 57-79 ms, and twice it took 3-4 s (the
    // allocator).  Not enough to pay a draw for in everybody's first solve.
    const char *e = tuning_env("PSP_PLACE");
    v = (e && atoi(e) == 1) ? 1 : 0;
    g_place.store(v);
  }
  return v != 0;
}

// below this many elements a vector is a few MB: the product is not HBM-bound and no level was ever seen
constexpr size_t kPlaceMinElems = (size_t)1 << 23;  // 64 MiB per vector

bool placement_applies(const psp_csr *A, size_t n) {
  return placement_enabled() && A && !A->multi && !A->host && n >= kPlaceMinElems && !cpu_mode();
}

namespace {

struct Cand {
  double *p = nullptr;
  float ty = 0.f, tx = 0.f;
};

// milliseconds per product y = A x: one untimed launch, then `reps` timed ones between two events on the thread's stream
int time_product(const psp_csr *A, const double *x, double *y, int reps, hipEvent_t e0, hipEvent_t e1, float *ms) {
  PSP_TRY(csr_spmv_launch(A, x, y, nullptr, nullptr, nullptr));
  PSP_HIP(hipEventRecord(e0, stream()));
  for (int r = 0; r < reps; ++r) PSP_TRY(csr_spmv_launch(A, x, y, nullptr, nullptr, nullptr));
  PSP_HIP(hipEventRecord(e1, stream()));
  PSP_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  PSP_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)reps;
  return PSP_OK;
}

}  // namespace

// Draws the operands of y = A x: one buffer of ny doubles for the output role and want_x (<= 2) buffers of nx doubles for
// the input role.  All come back hipMalloc'ed and zeroed; the caller owns them.  report (may be null): [0] candidates
// drawn, [1] best / [2] worst output-role time, [3] best / [4] worst input-role time (ms), [5] milliseconds the draw took.
// Falls back to plain allocations (report[0] = 0) when memory is short -- never fails for lack of candidates.
int place_operands(const psp_csr *A, size_t nx, size_t ny, int want_x, double **y_out, double **x_out, double *report) {
  if (want_x < 1 || want_x > 2) return fail(PSP_EINVAL, "place_operands: 1 or 2 input buffers");
  PSP_TRY(ensure_device());
  const size_t n = std::max(nx, ny);  // one candidate serves either role (square operators: nx == ny)
  const size_t bytes = sizeof(double) * (n ? n : 1);
  if (report) std::fill(report, report + 6, 0.0);
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
    (void)hipGetLastError();
    fr = 0;
  }
  // candidates spread over the address space: pads of twice the vector in between (released at the end) -- neighbouring
  // allocations share a level (runs of 3-5 in profiles/r4_modes.txt), the pads make eight draws reach as far as twenty
  int m = 7;
  size_t pad = 2 * bytes;
  if (fr < 2 * ((size_t)(m + 1) * bytes + (size_t)m * pad)) pad = 0;
  if (fr < 2 * (size_t)(m + 1) * bytes) m = 3;
  const bool draw = placement_applies(A, std::min(nx, ny)) && fr >= 2 * (size_t)(m + 1) * bytes;
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<Cand> c;
  std::vector<void *> pads;
  auto release = [&](bool all) {
    for (void *p : pads) (void)hipFree(p);
    pads.clear();
    if (all)
      for (Cand &k : c)
        if (k.p) (void)hipFree(k.p);
  };
  const int total = draw ? m + 1 : want_x + 1;
  for (int i = 0; i < total; ++i) {
    Cand k;
    // no draw (the default): plain allocations of each role's own size -- y (candidate 0) ny doubles, x nx doubles -- so a
    // strongly rectangular host-pointer product does not pay for two vectors of max(nx, ny) (round-5 advisor finding)
    const size_t want = draw ? bytes : sizeof(double) * std::max<size_t>(i == 0 ? ny : nx, 1);
    hipError_t e = hipMalloc((void **)&k.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release(true);
      return fail(PSP_ENOMEM, "place_operands: %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    c.push_back(k);
    if (hipMemsetAsync(k.p, 0, want, stream()) != hipSuccess) {
      release(true);
      return fail(PSP_ENODEV, "place_operands: %s", hipGetErrorString(hipGetLastError()));
    }
    if (draw && pad && i + 1 < total) {
      void *p = nullptr;
      if (hipMalloc(&p, pad) == hipSuccess) pads.push_back(p);
      else (void)hipGetLastError();  // no pad here: the draw goes on without it
    }
  }
  if (!draw) {
    if (hipStreamSynchronize(stream()) != hipSuccess) {
      release(true);
      return fail(PSP_ENODEV, "place_operands: %s", hipGetErrorString(hipGetLastError()));
    }
    *y_out = c[0].p;
    for (int j = 0; j < want_x; ++j) x_out[j] = c[1 + j].p;
    return PSP_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PSP_OK;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(PSP_ENODEV, "place_operands: no events");
  // output role: every candidate as y, the input fixed (the last candidate; it is timed as y against the first)
  for (int i = 0; i < total && rc == PSP_OK; ++i)
    rc = time_product(A, c[i == total - 1 ? 0 : total - 1].p, c[i].p, 2, e0, e1, &c[i].ty);
  int by = 0;
  for (int i = 1; i < total; ++i)
    if (c[i].ty < c[by].ty) by = i;
  // input role: every other candidate as x with the chosen y
  for (int i = 0; i < total && rc == PSP_OK; ++i)
    if (i != by) rc = time_product(A, c[i].p, c[by].p, 2, e0, e1, &c[i].tx);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (rc != PSP_OK) {
    release(true);
    return rc;
  }
  std::vector<int> order;
  for (int i = 0; i < total; ++i)
    if (i != by) order.push_back(i);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return c[a].tx < c[b].tx; });
  *y_out = c[by].p;
  for (int j = 0; j < want_x; ++j) x_out[j] = c[order[j]].p;
  if (report) {
    report[0] = total;
    report[1] = c[by].ty;
    report[2] = c[0].ty;
    for (int i = 0; i < total; ++i) report[2] = std::max(report[2], (double)c[i].ty);
    report[3] = c[order[0]].tx;
    report[4] = c[order.back()].tx;
  }
  // the timed products multiplied zero vectors, but a matrix that holds Inf / NaN turns those into NaN: back to zero
  (void)hipMemsetAsync(*y_out, 0, bytes, stream());
  for (int j = 0; j < want_x; ++j) (void)hipMemsetAsync(x_out[j], 0, bytes, stream());
  c[by].p = nullptr;
  for (int j = 0; j < want_x; ++j) c[order[j]].p = nullptr;
  (void)hipStreamSynchronize(stream());
  release(true);
  const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
  g_draws.fetch_add(1);
  g_draw_us.fetch_add(us);
  if (report) report[5] = us / 1e3;
  return PSP_OK;
}

}  // namespace psp

extern "C" {

int psp_set_placement(int on) {
  psp::g_place.store(on ? 1 : 0);
  return PSP_OK;
}

int psp_placement_info(int *enabled, long long *draws, double *draw_ms_total) {
  if (enabled) *enabled = psp::placement_enabled() ? 1 : 0;
  if (draws) *draws = psp::g_draws.load();
  if (draw_ms_total) *draw_ms_total = psp::g_draw_us.load() / 1e3;
  return PSP_OK;
}

int psp_place_operands(const psp_csr_t *A, double **y_dev, double **x_dev, double *report6) {
  PSP_API_GUARD_H(A);
  if (!A || !y_dev || !x_dev) return psp::fail(PSP_EINVAL, "psp_place_operands: NULL argument");
  if (A->multi || A->host) return psp::fail(PSP_EINVAL, "psp_place_operands: needs a single-device matrix");
  return psp::place_operands(A, (size_t)A->ncols, (size_t)A->nrows, 1, y_dev, x_dev, report6);
}

}  // extern "C"
