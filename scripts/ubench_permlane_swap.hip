// v_permlane16_swap / v_permlane32_swap (gfx950) as the cross-row sum of csrc/ndq_mlp.h quad_sum: both results of a swap
// of a value with itself added up = value + partner row, in every lane; compared bit for bit with the ds_bpermute form.
//   hipcc --offload-arch=gfx950 scripts/ubench_permlane_swap.hip -o /tmp/ubench_permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  a = __builtin_bit_cast(unsigned, v); b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float shfl_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__global__ void k(const float* in, float* a, float* b) {
  const float v = in[threadIdx.x];
  a[threadIdx.x] = swap_sum(v);
  b[threadIdx.x] = shfl_sum(v);
}
int main() {
  float h[64], ha[64], hb[64], *d, *da, *db;
  unsigned s = 12345u;
  for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8) / 7777.f - 1000.f; }
  hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&db, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, da, db);
  hipMemcpy(ha, da, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (memcmp(&ha[i], &hb[i], 4)) { if (bad < 8) printf("lane %d: swap %a shfl %a\n", i, ha[i], hb[i]); ++bad; }
  printf(bad ? "MISMATCH in %d lanes\n" : "permlane swap sum == shfl sum in all lanes (%d differences)\n", bad);
  return bad != 0;
}
