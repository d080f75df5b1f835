// Which XCD does workgroup i of a launch land on?  (ci_wide.h places the workgroups of one chain
// on one XCD by their ids, and verifies it at run time with this register.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 64;
  int* d; hipMalloc(&d, n * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    int h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%d%s", h[i] & 15, i % 16 == 15 ? "\n" : " ");
    printf("\n");
  }
  return 0;
}
