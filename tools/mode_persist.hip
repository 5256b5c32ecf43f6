// tools/mode_persist.hip -- the decisive experiment on the timing modes (VERDICT r3 "Next" #3).
//
// Fresh processes of the same csr_spmv_w4 launch (7-point Poisson 512^3) differ by up to 8 %; round 3 narrowed the
// difference to "fewer read requests in flight at an unchanged latency" and left two leads:
//   (A) how the hardware dispatcher's workgroup order falls against the XCD-stripe remap in a given process;
//   (B) the physical pages / channel balance a process's allocations get.
// This program holds the index-free operator of the library's layout (values offset-major in blocks of 128 rows, 16-bit
// row masks) and times, in ONE process, with HIP events:
//   plain     the library's kernel shape: one workgroup per 4 row blocks, 262144 workgroups, XCD stripe remap of blockIdx
//   persist   a PERSISTENT grid (256 CUs x 6 workgroups): every wave draws row blocks from a per-XCD ticket counter (the XCD
//             is read from HW_REG_XCC_ID), so each XCD walks ITS stripes in ascending order whatever the dispatcher did --
//             the dispatcher places 1536 workgroups once and is out of the picture
//   pads      lead (B): the operator re-allocated with leading pads of different sizes (all copies stay alive, so every
//             copy lies somewhere else physically), `plain` timed on each copy
// Run it as N fresh processes (tools/mode_persist.sh): if `persist` spreads as much over the processes as `plain` does,
// the dispatcher's order is not the cause; if the copies inside one process spread as much as processes do, placement is.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/mode_persist tools/mode_persist.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);       \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

constexpr int kRows = 128;  // rows per block (kDiaRows of the library)
constexpr int NO = 7;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // x pairs: 8-byte aligned (as in the library)
struct Offs {
  long o[NO];
};

__global__ void build_kernel(int nx, int ny, int nz, long n, double *valT, unsigned short *mask) {
  const long nxy = (long)nx * ny;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
    const int i = (int)(r % nx), j = (int)((r / nx) % ny);
    const long l = r / nxy;
    double *v = valT + (size_t)(r / kRows) * NO * kRows + (size_t)(r % kRows);
    unsigned m = 0;
    const bool have[NO] = {l > 0, j > 0, i > 0, true, i < nx - 1, j < ny - 1, l < nz - 1};
    for (int b = 0; b < NO; ++b) {
      v[(size_t)b * kRows] = have[b] ? (b == 3 ? 6.0 : -1.0) : 0.0;
      if (have[b]) m |= 1u << b;
    }
    mask[r] = (unsigned short)m;
  }
}

__global__ void fill_kernel(long n, double *x) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = 1.0 + (double)((i * 2654435761u) & 1023) * (1.0 / 1024.0);
}

// one row block by one wave: the arithmetic of csr_spmv_w4 (stored products in offset order, separately rounded)
__device__ __forceinline__ void block_rows(long blk, long nrows, const Offs &offs, const double *__restrict__ valT,
                                           const unsigned short *__restrict__ mask, const double *__restrict__ x,
                                           double *__restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long r = blk * kRows + 2 * lane;
  if (r >= nrows) return;
  const unsigned mm = *reinterpret_cast<const unsigned *>(mask + r);
  const unsigned m0 = mm & 0xffffu, m1 = mm >> 16;
  const double *vp = valT + (size_t)blk * NO * kRows + 2 * lane;
  d2 v[NO], xv[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) v[o] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vp + o * kRows));
  const long cmax = nrows - 2;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const long c = r + offs.o[o];
    const long cc = c < 0 ? 0 : (c > cmax ? cmax : c);
    const d2u t = *reinterpret_cast<const d2u *>(x + cc);
    xv[o].x = t.x;
    xv[o].y = t.y;
  }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const double t0 = a0 + v[o].x * xv[o].x, t1 = a1 + v[o].y * xv[o].y;
    a0 = ((m0 >> o) & 1u) ? t0 : a0;
    a1 = ((m1 >> o) & 1u) ? t1 : a1;
  }
  d2u out;
  out.x = a0;
  out.y = a1;
  __builtin_nontemporal_store(out, reinterpret_cast<d2u *>(y + r));
}

__global__ __launch_bounds__(256) void plain_kernel(long nblk, long nrows, int stripe, Offs offs,
                                                    const double *__restrict__ valT,
                                                    const unsigned short *__restrict__ mask,
                                                    const double *__restrict__ x, double *__restrict__ y) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const long blk = (long)vb * 4 + wid;
  if (blk < nblk) block_rows(blk, nrows, offs, valT, mask, x, y);
}

// persistent grid: every WAVE draws units of one row block from its XCD's ticket counter; ticket k of XCD c is row block
// ((k / sb) * 8 + c) * sb + k % sb  (sb = 4 * stripe row blocks: the stripes the remap above gives XCD c, in ascending
// order).  The next ticket is drawn before the current block is processed, so its latency hides behind the block.
__global__ __launch_bounds__(256) void persist_kernel(long nblk, long nrows, int sb, Offs offs,
                                                      const double *__restrict__ valT,
                                                      const unsigned short *__restrict__ mask,
                                                      const double *__restrict__ x, double *__restrict__ y,
                                                      unsigned *counters) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  unsigned *ctr = counters + xcc * 32;  // one 128-byte line per XCD
  const long per_xcd = (nblk + 8L * sb - 1) / (8L * sb) * sb;
  const int lane = threadIdx.x & 63;
  unsigned k = 0;
  if (lane == 0) k = atomicAdd(ctr, 1u);
  k = __builtin_amdgcn_readfirstlane(k);
  while ((long)k < per_xcd) {
    unsigned kn = 0;
    if (lane == 0) kn = atomicAdd(ctr, 1u);
    const long blk = ((long)(k / sb) * 8 + xcc) * sb + k % sb;
    if (blk < nblk) block_rows(blk, nrows, offs, valT, mask, x, y);
    k = __builtin_amdgcn_readfirstlane(kn);
  }
}

// the plain kernel without the row-mask load for row blocks whose 128 rows all store every offset (`full[blk]` != 0): what
// would a mask-free interior buy?  (2 of the kernel's 74 bytes per row)
__global__ __launch_bounds__(256) void plain_nomask_kernel(long nblk, long nrows, int stripe, Offs offs,
                                                           const double *__restrict__ valT,
                                                           const unsigned short *__restrict__ mask,
                                                           const unsigned char *__restrict__ full,
                                                           const double *__restrict__ x, double *__restrict__ y) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const long blk = (long)vb * 4 + wid;
  if (blk >= nblk) return;
  if (!__builtin_amdgcn_readfirstlane((int)full[blk])) {
    block_rows(blk, nrows, offs, valT, mask, x, y);
    return;
  }
  const int lane = threadIdx.x & 63;
  const long r = blk * kRows + 2 * lane;
  const double *vp = valT + (size_t)blk * NO * kRows + 2 * lane;
  d2 v[NO], xv[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) v[o] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vp + o * kRows));
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const d2u t = *reinterpret_cast<const d2u *>(x + r + offs.o[o]);  // a full block lies in the interior: no clamping
    xv[o].x = t.x;
    xv[o].y = t.y;
  }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    a0 = a0 + v[o].x * xv[o].x;
    a1 = a1 + v[o].y * xv[o].y;
  }
  d2u out;
  out.x = a0;
  out.y = a1;
  __builtin_nontemporal_store(out, reinterpret_cast<d2u *>(y + r));
}

__global__ void full_flags_kernel(long nblk, long nrows, const unsigned short *__restrict__ mask, unsigned char *full) {
  for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (long)gridDim.x * blockDim.x) {
    bool f = (b + 1) * kRows <= nrows;
    for (int i = 0; f && i < kRows; ++i) f = mask[b * kRows + i] == 0x7f;
    full[b] = f ? 1 : 0;
  }
}

// translation probe: one 8-byte read per `stride` bytes of a buffer, every thread a different page -- the time is
// dominated by address translation (few bytes move), so it tells buffers backed by large page fragments from others
__global__ void touch_kernel(const double *__restrict__ v, long n, long stride_d, long phase, double *sink) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * stride_d + phase;
  if (i < n && v[i] == 12345.678) *sink = 1.0;
}
static double touch_us(const double *v, long n, double *sink) {
  const long stride_d = 4096 / 8;  // one read per 4 KiB
  const long cnt = n / stride_d;
  double best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    auto run = [&]() { touch_kernel<<<(unsigned)((cnt + 255) / 256), 256>>>(v, n, stride_d, (rep * 37) % stride_d, sink); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    run();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    if (rep) best = std::min(best, (double)ms * 1e3);
  }
  return best;
}

struct Op {
  void *pad = nullptr;
  double *valT = nullptr;
  unsigned short *mask = nullptr;
};

static Op make_op(int nx, int ny, int nz, size_t pad_bytes) {
  Op A;
  const long n = (long)nx * ny * nz;
  const long nblk = (n + kRows - 1) / kRows;
  if (pad_bytes) CK(hipMalloc(&A.pad, pad_bytes));
  CK(hipMalloc((void **)&A.valT, sizeof(double) * (size_t)nblk * NO * kRows));
  CK(hipMalloc((void **)&A.mask, sizeof(unsigned short) * ((size_t)nblk * kRows + 8)));
  CK(hipMemset(A.mask, 0, sizeof(unsigned short) * ((size_t)nblk * kRows + 8)));
  build_kernel<<<4096, 256>>>(nx, ny, nz, n, A.valT, A.mask);
  CK(hipDeviceSynchronize());
  return A;
}

template <typename F>
static double time_ms(F launch, int warm, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; ++i) launch();
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ts[ts.size() / 2];
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512;
  const int npads = argc > 2 ? atoi(argv[2]) : 4;
  const int nx = N, ny = N, nz = N;
  const long n = (long)nx * ny * nz, nblk = (n + kRows - 1) / kRows;
  const int stripe = 128;
  Offs offs;
  const long o7[NO] = {-(long)nx * ny, -nx, -1, 0, 1, nx, (long)nx * ny};
  for (int i = 0; i < NO; ++i) offs.o[i] = o7[i];
  double *x, *y, *yref;
  CK(hipMalloc((void **)&x, sizeof(double) * n));
  CK(hipMalloc((void **)&y, sizeof(double) * n));
  CK(hipMalloc((void **)&yref, sizeof(double) * n));
  fill_kernel<<<4096, 256>>>(n, x);
  unsigned *counters;
  CK(hipMalloc((void **)&counters, 8 * 32 * sizeof(unsigned)));
  Op A = make_op(nx, ny, nz, 0);
  int grid = (int)((nblk + 3) / 4);
  grid = (grid + 8 * stripe - 1) / (8 * stripe) * (8 * stripe);
  auto plain = [&](const Op &B, double *out) {
    return [&, out]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, B.valT, B.mask, x, out); };
  };
  const double t_plain = time_ms(plain(A, yref), 10, 40);
  printf("{\"n\": %ld, \"plain_ms\": %.4f", n, t_plain);
  // the persistent grid, 6 and 8 workgroups per CU; the ticket counters are cleared by a memset in front of each launch
  // (inside the timed region: ~2 us)
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  for (int per_cu : {6, 8}) {
    auto persist = [&]() {
      CK(hipMemsetAsync(counters, 0, 8 * 32 * sizeof(unsigned), 0));
      persist_kernel<<<cus * per_cu, 256>>>(nblk, n, 4 * stripe, offs, A.valT, A.mask, x, y, counters);
    };
    const double t = time_ms(persist, 10, 40);
    // same bits as the plain kernel (it is the same arithmetic per row)
    std::vector<double> a(1 << 16), b(1 << 16);
    CK(hipMemcpy(a.data(), y + n / 3, sizeof(double) * a.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), yref + n / 3, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
    printf(", \"persist%d_ms\": %.4f, \"persist%d_same\": %s", per_cu, t, per_cu, a == b ? "true" : "false");
  }
  // lead (B): copies of the operator behind pads of different sizes, all alive at once
  printf(", \"pad_copies_ms\": [");
  std::vector<Op> copies;
  const size_t pads[] = {4096, (size_t)3 << 20, ((size_t)1 << 30) + 12288, ((size_t)5 << 30) + (1 << 16), (size_t)777 << 20};
  for (int i = 0; i < npads && i < 5; ++i) {
    copies.push_back(make_op(nx, ny, nz, pads[i]));
    const double t = time_ms(plain(copies.back(), y), 5, 30);
    printf("%s%.4f", i ? ", " : "", t);
  }
  printf("], \"pad_copies_valT\": [");
  for (size_t i = 0; i < copies.size(); ++i) printf("%s\"%p\"", i ? ", " : "", (void *)copies[i].valT);
  printf("], \"valT\": \"%p\", \"mask\": \"%p\", \"x\": \"%p\", \"y\": \"%p\"", (void *)A.valT, (void *)A.mask, (void *)x,
         (void *)yref);
  printf(", \"plain_again_ms\": %.4f", time_ms(plain(A, yref), 5, 40));
  // lead (B), second form: ONE allocation, the operator rebuilt at different byte offsets inside it (the physical
  // addresses of everything it streams shift by exactly that much); then the same for x and for y
  if (argc > 3) {
    for (Op &c : copies) {
      CK(hipFree(c.valT));
      CK(hipFree(c.mask));
      if (c.pad) CK(hipFree(c.pad));
    }
    const size_t slack = (size_t)96 << 20;
    const size_t vbytes = sizeof(double) * (size_t)nblk * NO * kRows;
    char *big, *xbig, *ybig;
    CK(hipMalloc((void **)&big, vbytes + slack));
    CK(hipMalloc((void **)&xbig, sizeof(double) * n + slack));
    CK(hipMalloc((void **)&ybig, sizeof(double) * n + slack));
    const size_t offs_b[] = {0, 4096, 65536, (size_t)1 << 20, (size_t)2 << 20, ((size_t)2 << 20) + 4096, (size_t)3 << 20,
                             (size_t)4 << 20, (size_t)6 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)24 << 20,
                             (size_t)32 << 20, (size_t)48 << 20, (size_t)64 << 20, (size_t)80 << 20};
    printf(", \"sweep_base\": [\"%p\", \"%p\", \"%p\"]", (void *)big, (void *)xbig, (void *)ybig);
    for (int which = 0; which < 3; ++which) {
      printf(", \"sweep_%s_ms\": [", which == 0 ? "valT" : which == 1 ? "x" : "y");
      for (size_t k = 0; k < sizeof(offs_b) / sizeof(offs_b[0]); ++k) {
        Op B = A;
        double *xs = x, *ys = y;
        if (which == 0) {
          B.valT = (double *)(big + offs_b[k]);
          build_kernel<<<4096, 256>>>(nx, ny, nz, n, B.valT, B.mask);
        } else if (which == 1) {
          xs = (double *)(xbig + offs_b[k]);
          fill_kernel<<<4096, 256>>>(n, xs);
        } else {
          ys = (double *)(ybig + offs_b[k]);
        }
        CK(hipDeviceSynchronize());
        auto run = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, B.valT, B.mask, xs, ys); };
        printf("%s%.4f", k ? ", " : "", time_ms(run, 5, 30));
      }
      printf("]");
    }
  }
  // lead (B), third form: the operator stays where it is; eight different allocations of y, then of x (pads of odd sizes
  // in between, everything stays alive)
  if (argc > 4) {
    std::vector<void *> keep;
    double touch[2][8];
    printf(", \"touch_us_x0_y0_yref_valT\": [%.1f, %.1f, %.1f, %.1f]", touch_us(x, n, (double *)counters),
           touch_us(y, n, (double *)counters), touch_us(yref, n, (double *)counters),
           touch_us(A.valT, n, (double *)counters));
    for (int which = 0; which < 2; ++which) {
      printf(", \"%s_lottery_ms\": [", which == 0 ? "y" : "x");
      for (int k = 0; k < 8; ++k) {
        void *pad;
        double *v;
        CK(hipMalloc(&pad, ((size_t)(37 + 101 * k) << 20) + 4096 * (size_t)k));
        CK(hipMalloc((void **)&v, sizeof(double) * n));
        keep.push_back(pad);
        keep.push_back(v);
        if (which == 1) fill_kernel<<<4096, 256>>>(n, v);
        CK(hipDeviceSynchronize());
        double *xs = which == 1 ? v : x, *ys = which == 0 ? v : y;
        auto run = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, A.valT, A.mask, xs, ys); };
        printf("%s%.4f", k ? ", " : "", time_ms(run, 5, 30));
        touch[which][k] = touch_us(v, n, (double *)counters);
      }
      printf("]");
      printf(", \"%s_lottery_touch_us\": [", which == 0 ? "y" : "x");
      for (int k = 0; k < 8; ++k) printf("%s%.1f", k ? ", " : "", touch[which][k]);
      printf("]");
    }
  }
  // what if a stream does not go through the caches at all: y, then the operator's values, then both in fine-grained /
  // uncached device memory (hipExtMallocWithFlags), against the SAME lottery of ordinary y allocations as reference
  if (argc > 5) {
    double *yu = nullptr, *vu = nullptr;
    const size_t vbytes = sizeof(double) * (size_t)nblk * NO * kRows;
    const unsigned flag = (unsigned)atoi(argv[5]);  // 3 = hipDeviceMallocUncached, 1 = hipDeviceMallocFinegrained
    if (hipExtMallocWithFlags((void **)&yu, sizeof(double) * n, flag) == hipSuccess &&
        hipExtMallocWithFlags((void **)&vu, vbytes, flag) == hipSuccess) {
      Op U = A;
      U.valT = vu;
      build_kernel<<<4096, 256>>>(nx, ny, nz, n, U.valT, U.mask);
      CK(hipDeviceSynchronize());
      auto r1 = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, A.valT, A.mask, x, yu); };
      auto r2 = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, U.valT, U.mask, x, y); };
      auto r3 = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, U.valT, U.mask, x, yu); };
      printf(", \"flag\": %u, \"y_special_ms\": %.4f", flag, time_ms(r1, 5, 30));
      printf(", \"valT_special_ms\": %.4f", time_ms(r2, 5, 30));
      printf(", \"both_special_ms\": %.4f", time_ms(r3, 5, 30));
    } else {
      (void)hipGetLastError();
      printf(", \"special\": \"hipExtMallocWithFlags(%u) failed\"", flag);
    }
  }
  // confirmation: fine-grained and ordinary allocations of y ALTERNATED (so that "later allocations are luckier" cannot
  // pass for an effect of the memory type), the operator once in ordinary and once in fine-grained memory
  if (argc > 6) {
    std::vector<void *> keep;
    const size_t vbytes = sizeof(double) * (size_t)nblk * NO * kRows;
    Op U = A;
    CK(hipExtMallocWithFlags((void **)&U.valT, vbytes, hipDeviceMallocFinegrained));
    build_kernel<<<4096, 256>>>(nx, ny, nz, n, U.valT, U.mask);
    CK(hipDeviceSynchronize());
    for (int opk = 0; opk < 2; ++opk) {
      const Op &B = opk ? U : A;
      printf(", \"alt_%s\": [", opk ? "valT_fine" : "valT_ordinary");
      for (int k = 0; k < 8; ++k) {
        void *pad;
        double *v;
        CK(hipMalloc(&pad, ((size_t)(53 + 67 * k) << 20) + 8192 * (size_t)k));
        if (k & 1) CK(hipExtMallocWithFlags((void **)&v, sizeof(double) * n, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void **)&v, sizeof(double) * n));
        keep.push_back(pad);
        keep.push_back(v);
        auto run = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, B.valT, B.mask, x, v); };
        printf("%s[\"%s\", %.4f]", k ? ", " : "", (k & 1) ? "fine" : "ord", time_ms(run, 5, 30));
      }
      printf("]");
    }
  }
  // is the level a RELATION between the placements of x and y?  Both inside ONE allocation, x at its start, y at distances
  // of 1 GiB + k * 96 MiB behind it
  if (argc > 7) {
    char *blk;
    const size_t gib = (size_t)1 << 30;
    CK(hipMalloc((void **)&blk, 5 * gib));
    double *xs = (double *)blk;
    fill_kernel<<<4096, 256>>>(n, xs);
    CK(hipDeviceSynchronize());
    printf(", \"xy_one_allocation_ms\": [");
    for (int k = 0; k < 30; ++k) {
      double *ys = (double *)(blk + gib + (size_t)k * (96 << 20));
      auto run = [&]() { plain_kernel<<<grid, 256>>>(nblk, n, stripe, offs, A.valT, A.mask, xs, ys); };
      printf("%s%.4f", k ? ", " : "", time_ms(run, 5, 30));
    }
    printf("]");
    // the same kernel without the 2 bytes of row mask per row for row blocks whose rows all store every offset (a per-block
    // byte says so), alternated with the plain kernel on the SAME buffers
    unsigned char *full;
    CK(hipMalloc((void **)&full, nblk));
    full_flags_kernel<<<1024, 256>>>(nblk, n, A.mask, full);
    CK(hipDeviceSynchronize());
    double tp = 1e9, tn = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      tp = std::min(tp, time_ms(plain(A, y), 3, 30));  // (both forms write the SAME y: placement decides 8 %, section 2)
      auto run = [&]() { plain_nomask_kernel<<<grid, 256>>>(nblk, n, stripe, offs, A.valT, A.mask, full, x, y); };
      tn = std::min(tn, time_ms(run, 3, 30));
    }
    std::vector<double> a(1 << 16), b(1 << 16);
    CK(hipMemcpy(a.data(), y + n / 3, sizeof(double) * a.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), yref + n / 3, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
    printf(", \"nomask_ab\": {\"plain_ms\": %.4f, \"mask_free_interior_ms\": %.4f, \"same_bits\": %s}", tp, tn,
           a == b ? "true" : "false");
  }
  printf("}\n");
  return 0;
}
