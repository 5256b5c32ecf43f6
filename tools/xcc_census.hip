// xcc_census.hip -- which XCD does workgroup b land on?  (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void census(int *xcc, int *cu) {
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    xcc[blockIdx.x] = v & 0xf;
    unsigned h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    cu[blockIdx.x] = h;
  }
}
int main() {
  for (int nb : {64, 4096, 262144}) {
    int *dx, *dc; hipMalloc(&dx, nb * 4); hipMalloc(&dc, nb * 4);
    census<<<nb, 256>>>(dx, dc); hipDeviceSynchronize();
    std::vector<int> x(nb), c(nb); hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost);
    int agree = 0; for (int b = 0; b < nb; ++b) agree += ((x[b] - x[0] + 8) % 8) == (b % 8);
    printf("grid %d: first 24 xcc:", nb); for (int b = 0; b < 24; ++b) printf(" %d", x[b]);
    printf(" | (xcc[b]-xcc[0])%%8 == b%%8 for %d / %d blocks\n", agree, nb);
    hipFree(dx); hipFree(dc);
  }
  return 0;
}
