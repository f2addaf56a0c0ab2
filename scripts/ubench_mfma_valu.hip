// Micro-benchmark (MI355X): does MFMA work overlap with f32 VALU work on one SIMD?
//   build: hipcc --offload-arch=gfx950 -O3 [-DUSE_BF16] scripts/ubench_mfma_valu.hip -o ubench
// Roles are wave-uniform (scalar branch), each role runs its own tight loop.
//   MODE 0: every wave MFMA only | 1: every wave VALU only | 2: waves < W/2 MFMA, waves >= W/2 VALU (co-resident pairs)
//   MODE 3: one wave alternates 1 MFMA + 8 FMA (fine-grained interleave inside a wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) {
#ifdef USE_BF16
  bf16x8 av, bv;
  for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  const bool do_mfma = MODE == 0 || (MODE == 2 && wave < nw / 2);
  const bool do_valu = MODE == 1 || (MODE == 2 && wave >= nw / 2);
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = mm(a, b, acc[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[(i * 8 + j) & 15] = fmaf(v[(i * 8 + j) & 15], b, a);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // 8 VALU
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = mm(a, b, acc[i]);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], b, a);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(int threads, int iters) {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return ms * 1e3f;
}

int main() {
  const int iters = 20000;   // per iteration and wave: 4 MFMA and/or 32 independent-chain FMAs
  printf("1 wave/SIMD  (256 thr): mfma %.1f us | valu %.1f us | 1 MFMA + 8 FMA interleaved in one wave %.1f us\n",
         run<0>(256, iters), run<1>(256, iters), run<3>(256, iters));
  printf("2 waves/SIMD (512 thr): mfma %.1f us | valu %.1f us | half the waves MFMA-only, half VALU-only %.1f us\n",
         run<0>(512, iters), run<1>(512, iters), run<2>(512, iters));
  return 0;
}
