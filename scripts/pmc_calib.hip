// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this package's kernels
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of a 16 B/lane streaming read; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels over a
// buffer of known size, each launched a few times:
//   calib_read4   4 B/lane coalesced reads  (the coordinate / stream reads of the closure and pointwise kernels)
//   calib_read16  16 B/lane coalesced reads (the guide's calibrated pattern, as a cross-check)
//   calib_write4  4 B/lane coalesced writes (stream / partial-sum writes)
// usage: pmc_calib [MiB]   -- run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (scripts/gpu_pmc.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void calib_read4(const float* __restrict__ x, float* __restrict__ out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 12345.678f) out[0] = s;      // never true: keeps the loads alive without a write stream
}

__global__ void calib_read16(const float4* __restrict__ x, float* __restrict__ out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}

__global__ void calib_write4(float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = (float)(i & 1023);
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 512;     // default 512 MiB: past the 256 MiB Infinity Cache
  const size_t n = mib * 1024 * 1024 / 4;
  float *x, *y, *o;
  if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&y, n * 4) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
  hipMemset(x, 0, n * 4);
  hipMemset(y, 0, n * 4);
  for (int it = 0; it < 4; ++it) {
    calib_read4<<<2048, 256>>>(x, o, n);
    calib_read16<<<2048, 256>>>(reinterpret_cast<const float4*>(x), o, n / 4);
    calib_write4<<<2048, 256>>>(y, n);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("pmc_calib: %zu MiB per kernel launch (bytes = %zu)\n", mib, n * 4);
  return 0;
}
