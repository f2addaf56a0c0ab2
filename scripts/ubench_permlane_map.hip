// which lanes' values v_permlane16_swap / v_permlane32_swap deliver: inputs a = lane, b = 100 + lane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  u2 t = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[4 * threadIdx.x] = r[0]; o[4 * threadIdx.x + 1] = r[1]; o[4 * threadIdx.x + 2] = t[0]; o[4 * threadIdx.x + 3] = t[1];
}
int main() {
  unsigned h[256], *d;
  (void)hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) printf("lane %2d: p16 (%3u, %3u)  p32 (%3u, %3u)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
