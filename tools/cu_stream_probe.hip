// cu_stream_probe.hip -- how fast can ONE workgroup (one CU) stream from HBM / L2?  (tuning aid, not part of the product)
// Question behind it (SSOR on 2-D grids): a sweep whose levels are walked by a single workgroup has to
// pull all of the triangle's static data (values, positions, diagonal, b) through one CU's L1.
//   hipcc -O3 --offload-arch=gfx950 tools/cu_stream_probe.hip -o tools/cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

// active workgroups: those with blockIdx.x % 8 == xcd (same XCD, same L2), `g` of them; each streams its own slice
template <int U>
__global__ __launch_bounds__(1024) void stream_k(const d2 *__restrict__ p, long n2_per_wg, int xcd, double *out) {
  if ((int)(blockIdx.x & 7) != xcd) return;
  const long w = blockIdx.x >> 3;
  const d2 *q = p + w * n2_per_wg;
  double s = 0.0;
  for (long i = threadIdx.x; i + (U - 1) * 1024L < n2_per_wg; i += U * 1024L) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(q + i + u * 1024L);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
  }
  if (s == 12345.678) out[0] = s;
}

// the same with a write stream beside it (x and y of a sweep): reads 3/4, writes 1/4 of the bytes
template <int U>
__global__ __launch_bounds__(1024) void stream_rw_k(const d2 *__restrict__ p, d2 *__restrict__ o, long n2_per_wg, int xcd) {
  if ((int)(blockIdx.x & 7) != xcd) return;
  const long w = blockIdx.x >> 3;
  const d2 *q = p + w * n2_per_wg;
  d2 *r = o + w * (n2_per_wg / 3);
  for (long i = threadIdx.x, j = threadIdx.x; i + (U * 3 - 1) * 1024L < n2_per_wg; i += 3 * U * 1024L, j += U * 1024L) {
    d2 v[3 * U];
#pragma unroll
    for (int u = 0; u < 3 * U; ++u) v[u] = __builtin_nontemporal_load(q + i + u * 1024L);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d2 t = v[3 * u];
      t.x += v[3 * u + 1].x + v[3 * u + 2].x;
      t.y += v[3 * u + 1].y + v[3 * u + 2].y;
      r[j + u * 1024L] = t;
    }
  }
}

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));          \
      return 1;                                                       \
    }                                                                 \
  } while (0)

template <int U>
static int run(const d2 *p, d2 *o, double *out, long bytes_per_wg, int g, int rw, int warm_l2) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const long n2 = bytes_per_wg / 16;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    if (warm_l2)  // second pass over a slice that fits the XCD's 4 MiB L2
      hipLaunchKernelGGL(stream_k<U>, dim3(8 * g), dim3(1024), 0, 0, p, n2, 0, out);
    CK(hipEventRecord(a, 0));
    if (rw)
      hipLaunchKernelGGL(stream_rw_k<U>, dim3(8 * g), dim3(1024), 0, 0, p, o, n2, 0);
    else
      hipLaunchKernelGGL(stream_k<U>, dim3(8 * g), dim3(1024), 0, 0, p, n2, 0, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double moved = (double)bytes_per_wg * g * (rw ? 4.0 / 3.0 : 1.0);
  printf("{\"workgroups_on_one_xcd\": %d, \"unroll\": %d, \"mode\": \"%s\", \"bytes_per_wg\": %ld, \"ms\": %.4f, \"GBps_total\": %.1f, "
         "\"GBps_per_wg\": %.1f}\n",
         g, U, rw ? "3 reads + 1 write" : (warm_l2 ? "read, L2-resident" : "read, HBM"), bytes_per_wg, best,
         moved / best * 1e-6, moved / best * 1e-6 / g);
  return 0;
}

int main() {
  const long total = 1L << 30;
  d2 *p = nullptr, *o = nullptr;
  double *out = nullptr;
  CK(hipMalloc((void **)&p, total));
  CK(hipMalloc((void **)&o, total / 2));
  CK(hipMalloc((void **)&out, 8));
  CK(hipMemset(p, 0, total));
  CK(hipMemset(o, 0, total / 2));
  for (int g : {1, 2, 4, 8}) {
    if (run<4>(p, o, out, 96L << 20, g, 0, 0)) return 1;
    if (run<8>(p, o, out, 96L << 20, g, 0, 0)) return 1;
    if (run<16>(p, o, out, 96L << 20, g, 0, 0)) return 1;
    if (run<4>(p, o, out, 96L << 20, g, 1, 0)) return 1;
  }
  // L2-resident: 3 MiB slice read twice, second pass timed (what a helper workgroup running ahead would give)
  if (run<8>(p, o, out, 3L << 20, 1, 0, 1)) return 1;
  if (run<16>(p, o, out, 3L << 20, 1, 0, 1)) return 1;
  return 0;
}
