// micro-benchmark: do f32 MFMA (16x16x4) and f32 VALU overlap on one SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: MFMA only, 1: VALU only, 2: waves alternate roles (even wave MFMA, odd wave VALU), 3: interleaved in one wave
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], b, a);   // 32 dependent-chain-free-ish VALU FMAs
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
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
  const int iters = 20000;
  // per iteration: 4 MFMA (128 cycles of matrix pipe) and/or 32 VALU FMAs (wave64 -> 64+ cycles)
  printf("1 wave/SIMD  (256 thr): mfma %.1f us  valu %.1f us  interleaved-in-wave %.1f us\n", run<0>(256, iters), run<1>(256, iters), run<3>(256, iters));
  printf("2 waves/SIMD (512 thr): mfma %.1f us  valu %.1f us  split-roles %.1f us  both-interleaved %.1f us\n", run<0>(512, iters), run<1>(512, iters), run<2>(512, iters), run<3>(512, iters));
  return 0;
}
