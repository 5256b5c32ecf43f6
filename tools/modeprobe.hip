// modeprobe.hip -- helpers of tools/mode_probe2.py (tuning aid, not part of the product):
// streaming probes and a census of where waves run, callable from Python in the SAME process as the SpMV.
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/modeprobe.hip -o tools/libmodeprobe.so
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void mp_read_k(const d2 *__restrict__ p, long n2, double *out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) {
    d2 v = p[i];
    if (v.x + v.y == 12345.678) out[0] = v.x;
  }
}
__global__ __launch_bounds__(256) void mp_fill_k(d2 *__restrict__ q, long n2, double c) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) q[i] = d2{c, c};
}
__global__ __launch_bounds__(256) void mp_fill_nt_k(d2 *__restrict__ q, long n2, double c) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) __builtin_nontemporal_store(d2{c, c}, q + i);
}
__global__ __launch_bounds__(256) void mp_copy_k(const d2 *__restrict__ p, d2 *__restrict__ q, long n2) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) q[i] = p[i];
}
// 7 read streams + 1 write stream, like csr_spmv_w4 at 512^3 without the x re-reads
__global__ __launch_bounds__(256) void mp_r7w1_k(const d2 *__restrict__ p, d2 *__restrict__ q, long n2) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) {
    d2 s = p[i];
#pragma unroll
    for (int k = 1; k < 7; ++k) {
      d2 v = __builtin_nontemporal_load(p + i + k * n2);
      s.x += v.x;
      s.y += v.y;
    }
    __builtin_nontemporal_store(s, q + i);
  }
}
__global__ void mp_census_k(unsigned *hw, unsigned *xcc) {
  if (threadIdx.x == 0) {
    unsigned v, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    xcc[blockIdx.x] = v;
    hw[blockIdx.x] = h;
  }
}

// one lane walks `steps` dependent loads through [p, p + n) doubles, `stride` doubles apart (wrapping): every address
// depends on the value loaded before it, so the time per step is one full load latency -- L2 / HBM plus, when consecutive
// addresses lie in different translation fragments, whatever the translation costs.  out[0] = device clock ticks (100 MHz).
__global__ void mp_chase_k(const double *__restrict__ p, long n, long stride, int warm, int steps, long long *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long idx = 0;
  double acc = 0.0;
  for (int i = 0; i < warm; ++i) {  // untimed: the code path, and (warm = one wrap) the caches / translations of a first pass
    const double v = __builtin_nontemporal_load(p + idx);
    idx = (idx + stride + (v == 12345.678 ? 1 : 0)) % n;
    acc += v;
  }
  const long long t0 = wall_clock64();
  for (int i = 0; i < steps; ++i) {
    const double v = __builtin_nontemporal_load(p + idx);
    idx = (idx + stride + (v == 12345.678 ? 1 : 0)) % n;
    acc += v;
  }
  const long long t1 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = (long long)acc + idx;
}

extern "C" {
// average nanoseconds per dependent load (wall_clock64 ticks at 100 MHz)
double mp_chase_ns(void *buf, long n_doubles, long stride_doubles, int warm, int steps) {
  long long *out = nullptr, host[2] = {0, 0};
  if (hipMalloc(&out, 16) != hipSuccess) return -1.0;
  mp_chase_k<<<1, 64>>>((const double *)buf, n_doubles, stride_doubles, warm, steps, out);
  hipMemcpy(host, out, 16, hipMemcpyDeviceToHost);
  hipFree(out);
  return (double)host[0] * 10.0 / steps;
}
void *mp_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  hipMemset(p, 0, bytes);
  return p;
}
void mp_free(void *p) { hipFree(p); }
// kind: 0 read, 1 fill, 2 copy, 3 fill (non-temporal), 4 seven reads + one write; n2 = 16-byte elements per stream
double mp_stream_ms(int kind, void *a, void *b, long n2, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned grid = (unsigned)((n2 + 255) / 256);
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0);
    if (kind == 0) mp_read_k<<<grid, 256>>>((const d2 *)a, n2, (double *)b);
    else if (kind == 1) mp_fill_k<<<grid, 256>>>((d2 *)b, n2, 1.5);
    else if (kind == 2) mp_copy_k<<<grid, 256>>>((const d2 *)a, (d2 *)b, n2);
    else if (kind == 3) mp_fill_nt_k<<<grid, 256>>>((d2 *)b, n2, 1.5);
    else mp_r7w1_k<<<grid, 256>>>((const d2 *)a, (d2 *)b, n2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best;
}
int mp_census(int nblocks, unsigned *hw_host, unsigned *xcc_host) {
  unsigned *hw, *xc;
  if (hipMalloc(&hw, nblocks * 4) != hipSuccess || hipMalloc(&xc, nblocks * 4) != hipSuccess) return -1;
  mp_census_k<<<nblocks, 256>>>(hw, xc);
  hipMemcpy(hw_host, hw, nblocks * 4, hipMemcpyDeviceToHost);
  hipMemcpy(xcc_host, xc, nblocks * 4, hipMemcpyDeviceToHost);
  hipFree(hw);
  hipFree(xc);
  return 0;
}
}
