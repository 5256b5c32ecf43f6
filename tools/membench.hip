// membench.hip -- streaming-bandwidth probes on one MI355X (tuning aid, not part of the product).
// hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o build/membench && build/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// read-only: every thread sums its elements; U independent 16-B loads in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_k(const d2 *__restrict__ p, long n2, double *out) {
  double acc = 0.0;
  const long T = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * T < n2; i += U * T) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * T) : p[i + u * T];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  for (; i < n2; i += T) acc += p[i].x + p[i].y;
  if (acc == 12345.678) out[0] = acc;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const d2 *__restrict__ p, d2 *__restrict__ q, long n2) {
  const long T = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * T < n2; i += U * T) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * T];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], q + i + u * T); else q[i + u * T] = v[u]; }
  }
  for (; i < n2; i += T) q[i] = p[i];
}

// block-contiguous read: block b owns [b*span, (b+1)*span)
template <int U>
__global__ __launch_bounds__(256) void read_blk(const d2 *__restrict__ p, long n2, long span, double *out) {
  double acc = 0.0;
  for (long b = blockIdx.x; b * span < n2; b += gridDim.x) {
    long lo = b * span, hi = lo + span < n2 ? lo + span : n2;
    long i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
      d2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (; i < hi; i += 256) acc += p[i].x + p[i].y;
  }
  if (acc == 12345.678) out[0] = acc;
}

template <typename F>
double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, (double)ms / reps);
  }
  return best;
}

int main(int argc, char **argv) {
  double *out; CK(hipMalloc(&out, 64));
  size_t sizes_mb[] = {256, 1024, 4096, 12288};
  for (size_t mb : sizes_mb) {
    size_t bytes = mb << 20; long n2 = bytes / 16;
    d2 *p, *q; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 0, bytes));
    int reps = mb >= 4096 ? 3 : 10;
    for (int gpc : {4, 8, 16, 32}) {
      int grid = 256 * gpc;
      double t1 = timeit([&] { read_k<1, false><<<grid, 256>>>(p, n2, out); }, reps);
      double t4 = timeit([&] { read_k<4, false><<<grid, 256>>>(p, n2, out); }, reps);
      double t8 = timeit([&] { read_k<8, false><<<grid, 256>>>(p, n2, out); }, reps);
      double t4n = timeit([&] { read_k<4, true><<<grid, 256>>>(p, n2, out); }, reps);
      double tb = timeit([&] { read_blk<4><<<grid, 256>>>(p, n2, 2048, out); }, reps);
      double c1 = timeit([&] { copy_k<1, false><<<grid, 256>>>(p, q, n2); }, reps);
      double c4 = timeit([&] { copy_k<4, false><<<grid, 256>>>(p, q, n2); }, reps);
      double c4n = timeit([&] { copy_k<4, true><<<grid, 256>>>(p, q, n2); }, reps);
      printf("size %5zu MiB grid %5d | read U1 %6.0f U4 %6.0f U8 %6.0f U4nt %6.0f blk2048 %6.0f | copy U1 %6.0f U4 %6.0f U4nt %6.0f GB/s\n",
             mb, grid, bytes / t1 / 1e6, bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / t4n / 1e6, bytes / tb / 1e6,
             2.0 * bytes / c1 / 1e6, 2.0 * bytes / c4 / 1e6, 2.0 * bytes / c4n / 1e6);
      fflush(stdout);
    }
    // non-persistent: one 16-B element per thread
    {
      long blocks = (n2 + 255) / 256;
      double t = timeit([&] { read_k<1, false><<<(int)std::min<long>(blocks, 2147483647L), 256>>>(p, n2, out); }, reps);
      double c = timeit([&] { copy_k<1, false><<<(int)std::min<long>(blocks, 2147483647L), 256>>>(p, q, n2); }, reps);
      double m = timeit([&] { CK(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, 0)); }, reps);
      printf("size %5zu MiB full grid      | read %6.0f copy %6.0f hipMemcpyD2D %6.0f GB/s\n", mb, bytes / t / 1e6, 2.0 * bytes / c / 1e6, 2.0 * bytes / m / 1e6);
    }
    CK(hipFree(p)); CK(hipFree(q));
  }
  return 0;
}
