// Stand-alone check (GPU box): psp_wave_sum's permlane / DPP form against the shuffle-down tree it replaces, lane 0's
// bits on random data (values of mixed magnitude, so that a different association would show), and the wave maximum.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ipysparse_amd/csrc -Iinclude tools/wave_sum_check.hip -o /tmp/wsc && /tmp/wsc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "psp_internal.h"

__device__ __forceinline__ double tree_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double tree_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    if (o > v) v = o;
  }
  return v;
}

__global__ void k(const double *x, double *out, int nw) {
  const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (w >= nw) return;
  const double v = x[(size_t)w * 64 + (threadIdx.x & 63)];
  const double a = tree_sum(v), b = psp::psp_wave_sum(v), c = tree_max(v), d = psp::psp_wave_max(v);
  if ((threadIdx.x & 63) == 0) {
    out[4 * w] = a;
    out[4 * w + 1] = b;
    out[4 * w + 2] = c;
    out[4 * w + 3] = d;
  }
}

int main() {
  const int nw = 1 << 16;
  std::vector<double> h((size_t)nw * 64), o((size_t)nw * 4);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = u(g) * std::ldexp(1.0, (int)(g() % 40) - 20);
  double *dx, *dout;
  if (hipMalloc((void **)&dx, h.size() * 8) != hipSuccess || hipMalloc((void **)&dout, o.size() * 8) != hipSuccess) return 2;
  hipMemcpy(dx, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nw / 4), dim3(256), 0, 0, dx, dout, nw);
  if (hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  long bad = 0;
  for (int w = 0; w < nw; ++w)
    if (memcmp(&o[4 * w], &o[4 * w + 1], 8) || memcmp(&o[4 * w + 2], &o[4 * w + 3], 8)) ++bad;
  printf("waves %d, lane-0 results that differ from the shuffle tree: %ld\n", nw, bad);
  return bad ? 1 : 0;
}
