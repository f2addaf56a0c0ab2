// End of a training epoch on the device -- second-stage sums of the gradient / loss partials, loss history, best-network
// snapshot, Adam (solvers.py:331-341, 407-441 without a host round trip) -- as device functions shared by
//   * the sums / tail kernels of libndq.so (csrc/ndq_api.hip: reduce_tail_kernel ...), one launch behind a closure launch;
//   * the PULL PROLOGUE of the closure kernels (csrc/ndq_mlp.h), round 3: inside fit() the closure launch of epoch e
//     first finishes epoch e - 1 itself -- every workgroup adds up the (few) partial rows of the previous launch, applies
//     Adam to ALL parameters in registers and stages its weight image from the result; workgroup 0 alone writes the new
//     parameters / moments / history.  One launch per epoch instead of two, no parameter round trip through HBM between
//     them.  Used for small grids only (every workgroup reads every partial row: <= 32 rows).
// Both routes perform the SAME floating-point operations in the SAME order (the summation orders of the tail kernel are
// spelled out below and re-enacted by the prologue), so a training run does not depend on which one served an epoch.
#pragma once
#include <hip/hip_runtime.h>

namespace ndq {

constexpr int kPullMaxRows = 32;        // partial rows (closure workgroups) up to which the pull prologue is used
constexpr int kPullMaxNets = 4;

// Adam, torch.optim.Adam single-tensor formula (amsgrad = False, maximize = False); bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t)
struct AdamConsts { float lr, b1, b2, eps, wd, bc1, bc2s; };
__device__ __forceinline__ void adam_value(const AdamConsts& c, float pi, float g, float m0, float v0, float& p, float& m,
                                           float& v) {
  float gi = g;
  if (c.wd != 0.f) gi = fmaf(c.wd, pi, gi);
  m = fmaf(c.b1, m0, (1.f - c.b1) * gi);
  v = fmaf(c.b2, v0, (1.f - c.b2) * gi * gi);
  p = pi - (c.lr / c.bc1) * (m / (sqrtf(v) / c.bc2s + c.eps));
}

// ---- the tail kernel's summation orders, re-enacted ------------------------------------------------------------------
// loss: 1024 threads, thread t adds part[t], part[t + 1024] ...; shuffle tree inside each of the 16 waves; the 16 wave
// sums are added in order.  Here: any number of whole waves plays the 16 "virtual" waves one after the other.
// scratch: 16 floats of LDS.  Every thread returns the total.  (n <= 1024: one term per virtual thread.)
__device__ __forceinline__ float tail_loss_total(const float* __restrict__ part, int n, float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  for (int w = wave; w < 16; w += waves) {
    float x = 0.f;
    for (int r = 64 * w + lane; r < n; r += 1024) x += part[r];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if (lane == 0) scratch[w] = x;
  }
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += scratch[w];
  __syncthreads();
  return s;
}

// gradient column i: 16 row groups rg = 0 .. 15, each four chains over the rows rg, rg + 16, rg + 32, rg + 48 (+ 64 k),
// combined as (s0 + s1) + (s2 + s3); the 16 group sums are added in order.
__device__ __forceinline__ float tail_column_total(const float* __restrict__ part, int nparts, int len, int i) {
  float tot = 0.f;
#pragma unroll 4
  for (int rg = 0; rg < 16; ++rg) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 48 < nparts; r += 64) {
      s0 += part[(size_t)r * len + i];
      s1 += part[(size_t)(r + 16) * len + i];
      s2 += part[(size_t)(r + 32) * len + i];
      s3 += part[(size_t)(r + 48) * len + i];
    }
    for (; r < nparts; r += 16) s0 += part[(size_t)r * len + i];
    tot += (s0 + s1) + (s2 + s3);
  }
  return tot;
}

// ---- pull prologue ----------------------------------------------------------------------------------------------------
struct PullNet {
  const float* part;                   // [nparts][len] gradient partial rows of the epoch being finished
  const float* p_in; const float* m_in; const float* v_in;     // parameters / Adam moments that epoch started from
  float* p_out; float* m_out; float* v_out;                    // ... and where workgroup 0 puts the updated ones
  float* grad;                         // [len] the reduced gradient (what p.grad views show), written by workgroup 0
  float* best_flat;                    // snapshot of p_in when the tracked loss improved (nullptr: no snapshot)
  int len;
  AdamConsts adam;
};
struct PullArgs {
  int enabled;                         // 0: the launch starts from the parameters it is given (first epoch of a call)
  int n_nets, nparts;
  const float* lpart; int nlparts; float lscale;               // training loss of the epoch being finished
  float* loss_hist; int hist_index; float* loss_slot;
  const float* vpart; int nvparts; float vscale;               // validation loss of the epoch before it (nullptr: none)
  float* valid_hist; int valid_index; int best_on_valid;
  float* best_loss; int parity;        // 2-slot ping-pong like the tail kernel: read [parity], write [parity ^ 1]
  PullNet net[kPullMaxNets];
};

// tail_column_total for nparts <= 32 on values that are already in registers (a[r] = row r of the column, 0 where
// r >= nparts): the same additions in the same order -- row group rg holds the chain (0 + a[rg]) + a[rg + 16], the other
// three chains of the group are empty, the 16 groups are added in order.
__device__ __forceinline__ float tail_column_total_regs(const float (&a)[kPullMaxRows], int nparts) {
  float tot = 0.f;
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    float s0 = 0.f;
    const float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (rg < nparts) s0 += a[rg];
    if (rg + 16 < nparts) s0 += a[rg + 16];
    tot += (s0 + s1) + (s2 + s3);
  }
  return tot;
}

// Finish the previous epoch for one network: every thread of the workgroup takes columns tid, tid + nt, ...; the updated
// parameters go to pnew (LDS, [len]) for the weight staging that follows.  `writer`: this workgroup also writes the
// global state.  `better`: the tracked loss improved (snapshot the pre-update parameters).  The loop is latency-bound
// (every value comes from the previous launch, i.e. from HBM / the other XCDs' L2): the loads of CH columns -- up to 32
// partial rows, parameter, two moments each -- are all issued before the first addition (a column at a time measured
// 9 us for one row and 28 us for 16 rows: ~0.45 us per dependent load).
template <int CH = 4>
__device__ __forceinline__ void pull_update_net(const PullNet& n, int nparts, float* pnew, bool writer, bool better, int tid,
                                                int nt) {
  for (int base = tid; base < n.len; base += CH * nt) {
    float a[CH][kPullMaxRows], pi[CH], m0[CH], v0[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      // unconditional loads from clamped addresses (a predicated load is a branch with its own wait), masked afterwards
      const int i = base + c * nt < n.len ? base + c * nt : n.len - 1;
#pragma unroll
      for (int r = 0; r < kPullMaxRows; ++r) {
        const float x = n.part[(size_t)(r < nparts ? r : nparts - 1) * n.len + i];
        a[c][r] = r < nparts ? x : 0.f;
      }
      pi[c] = n.p_in[i];
      m0[c] = n.m_in[i];
      v0[c] = n.v_in[i];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int i = base + c * nt;
      if (i >= n.len) continue;
      const float g = tail_column_total_regs(a[c], nparts);
      float p, m, v;
      adam_value(n.adam, pi[c], g, m0[c], v0[c], p, m, v);
      pnew[i] = p;
      if (writer) {
        n.p_out[i] = p; n.m_out[i] = m; n.v_out[i] = v;
        if (n.grad) n.grad[i] = g;
        if (better && n.best_flat) n.best_flat[i] = pi[c];
      }
    }
  }
}

// The scalars of the epoch being finished: losses -> history, best-loss ping-pong.  Returns `better` to every thread.
// scratch: 16 floats of LDS.
__device__ __forceinline__ bool pull_scalars(const PullArgs& a, float* scratch, bool writer) {
  const bool has_valid = a.vpart != nullptr;
  // both loss totals in one pass (tail_loss_total's order for each; scratch: 2 x 16 floats)
  float loss, vloss = 0.f;
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    for (int w = wave; w < 16; w += waves) {
      float x = 0.f, y = 0.f;
      for (int r = 64 * w + lane; r < a.nlparts; r += 1024) x += a.lpart[r];
      if (has_valid)
        for (int r = 64 * w + lane; r < a.nvparts; r += 1024) y += a.vpart[r];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if (has_valid)
        for (int off = 32; off > 0; off >>= 1) y += __shfl_down(y, off);
      if (lane == 0) { scratch[w] = x; scratch[16 + w] = y; }
    }
    __syncthreads();
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += scratch[w];
    if (has_valid) {
#pragma unroll
      for (int w = 0; w < 16; ++w) t += scratch[16 + w];
    }
    loss = s * a.lscale;
    vloss = t * a.vscale;
  }
  const float best = a.best_loss[a.parity];
  const bool on_valid = has_valid && a.best_on_valid != 0;
  const float cmp = on_valid ? vloss : loss;
  bool track = false;
#pragma unroll
  for (int k = 0; k < kPullMaxNets; ++k) track = track || (k < a.n_nets && a.net[k].best_flat != nullptr);   // static indices
  const bool better = track && (cmp < best);
  if (writer && threadIdx.x == 0) {
    *a.loss_slot = loss;
    a.loss_hist[a.hist_index] = loss;
    if (has_valid) a.valid_hist[a.valid_index] = vloss;
    a.best_loss[a.parity ^ 1] = better ? cmp : best;
  }
  return better;
}

}  // namespace ndq
