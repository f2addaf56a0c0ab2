// Device-side collocation-point sampler (include/ndq.h: ndq_sample).  One Philox4x32-10 block per point, keyed by
// (seed, draw counter, stream id): counter-based, so a batch is reproducible from three integers, shards on different
// ranks never overlap and no generator state lives in HBM.  HBM-bound by construction: writes d floats per point.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ndq.h"

namespace ndq {

struct U4 { unsigned x, y, z, w; };

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), the generator torch uses on
// GPUs; restated in oracle/philox_ref.py and pinned there to the Random123 known-answer vectors.
__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    U4 n;
    n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0;
    n.y = (unsigned)p1;
    n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1;
    n.w = (unsigned)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ inline float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }          // [0, 1)
__device__ inline float u01_open(unsigned x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f; }  // (0, 1]

// torch.linspace(lo, hi, n)[i]: stepped from the nearer end with one fused multiply-add (ATen RangeFactories)
__device__ inline float linspace_at(float lo, float hi, int n, int i) {
  if (n <= 1) return lo;
  const float step = (hi - lo) / (float)(n - 1);
  return (i < n / 2) ? fmaf(step, (float)i, lo) : fmaf(-step, (float)(n - 1 - i), hi);
}

struct SampleArgs {
  ndq_sampler_desc s;
  unsigned k0, k1, c1, c2, c3;
  float* coords;
  int ldc, total;
};

// point i of the batch described by a (one thread per point; also called from the epoch tail kernel's extra
// workgroups, which draw the NEXT batch while the optimiser step is applied: csrc/ndq_api.hip)
__device__ __forceinline__ void sample_point_store(const SampleArgs& a, int i) {
  if (i >= a.total) return;
  const U4 r = philox4x32_10(U4{(unsigned)i, a.c1, a.c2, a.c3}, a.k0, a.k1);
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
  const ndq_sampler_desc& s = a.s;
  if (s.kind == NDQ_SAMPLE_UNIFORM) {                       // generators.py:150-152 (Generator1D 'uniform')
    for (int c = 0; c < s.d; ++c) a.coords[(size_t)c * a.ldc + i] = s.lo[c] + (s.hi[c] - s.lo[c]) * u01(w[c]);
  } else if (s.kind == NDQ_SAMPLE_GRID) {                   // generators.py:253-266: ij-meshgrid + N(0, std^2) jitter
    // Box-Muller: (w0, w1) -> two normals, (w2, w3) -> two more
    const float r0 = sqrtf(-2.0f * __logf(u01_open(w[0]))), t0 = 6.283185307179586f * u01(w[1]);
    const float r1 = sqrtf(-2.0f * __logf(u01_open(w[2]))), t1 = 6.283185307179586f * u01(w[3]);
    const float z[3] = {r0 * __cosf(t0), r0 * __sinf(t0), r1 * __cosf(t1)};
    int rem = i;
    int idx[3] = {0, 0, 0};
    for (int c = s.d - 1; c >= 0; --c) { idx[c] = rem % s.n[c]; rem /= s.n[c]; }
    for (int c = 0; c < s.d; ++c) {
      float v = linspace_at(s.lo[c], s.hi[c], s.n[c], idx[c]);
      if (s.noise_std[c] != 0.0f) v = fmaf(s.noise_std[c], z[c], v);
      a.coords[(size_t)c * a.ldc + i] = v;
    }
  } else {                                                  // generators.py:622-646 (GeneratorSpherical)
    const float p = u01_open(w[0]), q = u01_open(w[1]), t = u01_open(w[2]);
    const float inv = 1.0f / (p + q + t);
    float x = sqrtf(p * inv) + 1e-6f, y = sqrtf(q * inv) + 1e-6f, z = fminf(sqrtf(t * inv) + 1e-6f, 1.0f);
    if (w[0] & 1u) x = -x;                                  // the low 8 bits of each word are not used by u01
    if (w[1] & 1u) y = -y;
    if (w[2] & 1u) z = -z;
    const float u = u01(w[3]);
    const float lo = s.lo[0], hi = s.hi[0];
    const float rad = s.radial ? lo + (hi - lo) * u : sqrtf((hi * hi - lo * lo) * u + lo * lo);
    a.coords[i] = rad;
    a.coords[(size_t)a.ldc + i] = acosf(z);
    a.coords[(size_t)2 * a.ldc + i] = 3.14159265358979f - atan2f(y, x);
  }
}

__global__ void __launch_bounds__(256) sample_kernel(SampleArgs a) { sample_point_store(a, blockIdx.x * 256 + threadIdx.x); }

// validated launch arguments of one draw; returns 0 or NDQ_EINVAL
inline int fill_sample_args(SampleArgs& a, const ndq_sampler_desc* s, unsigned long long seed, unsigned long long draw,
                            unsigned stream_id, float* coords, int ldc) {
  if (!s || !coords || s->d < 1 || s->d > 3) return NDQ_EINVAL;
  long long total = 0;
  if (s->kind == NDQ_SAMPLE_GRID) {
    total = 1;
    for (int c = 0; c < s->d; ++c) {
      if (s->n[c] < 1) return NDQ_EINVAL;
      total *= s->n[c];
    }
  } else if (s->kind == NDQ_SAMPLE_UNIFORM || s->kind == NDQ_SAMPLE_SPHERICAL) {
    total = s->n[0];
    if (s->kind == NDQ_SAMPLE_SPHERICAL && s->d != 3) return NDQ_EINVAL;
  } else {
    return NDQ_EINVAL;
  }
  if (total < 1 || total > 0x7fffffffLL || ldc < total) return NDQ_EINVAL;
  a.s = *s;
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32);
  a.c1 = (unsigned)draw; a.c2 = (unsigned)(draw >> 32); a.c3 = stream_id;
  a.coords = coords; a.ldc = ldc; a.total = (int)total;
  return 0;
}

inline int launch_sample(const ndq_sampler_desc* s, unsigned long long seed, unsigned long long draw, unsigned stream_id,
                         float* coords, int ldc, hipStream_t stream) {
  SampleArgs a;
  const int rc = fill_sample_args(a, s, seed, draw, stream_id, coords, ldc);
  if (rc) return rc;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

}  // namespace ndq
