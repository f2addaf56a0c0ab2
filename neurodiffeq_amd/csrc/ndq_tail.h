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

constexpr int kPullMaxRows = 32;        // partial rows (closure workgroups) the pull prologue can take
// ... and rows x networks up to which it is USED (ndq_fused_fit_run): every workgroup reads every row of every network.
// MI355X, us per fit() epoch, one launch against two: 1 row 12.1 / 14.4 (K = 1), 14.6 / 16.5 (K = 2); 16 rows 15.5 / 16.9
// (K = 1), 19.9 / 19.3 (K = 2, four tile slots per workgroup); 32 rows, K = 2 (C1): 23.8 / 18.8.
constexpr int kPullMaxWork = 16;
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

// ---- batched prologue ------------------------------------------------------------------------------------------------
// Everything the prologue reads was written by the previous launch, so every first touch is a far miss (measured on
// MI355X: ~2 us each when one depends on the other -- a prologue of five dependent rounds cost 10 - 12 us, twice the tail
// launch it replaces).  Here ALL loads of the prologue -- loss partials, best loss, and for every column a thread owns the
// partial rows, parameter and moments -- are issued before the first use, the
// arithmetic (and therefore every bit of the result) is the same as above.  NT: threads per workgroup, LEN: parameters
// per network (both static: the loads live in registers).  Two builds: up to 2 and up to 16 partial rows.

template <int NT> struct PullScalarLoads {
  static constexpr int VW = (1024 + NT - 1) / NT;          // "virtual" waves of the tail kernel each wave plays
  float x[VW], y[VW], best;
};
template <int NT>
__device__ __forceinline__ void pull_scalars_issue(const PullArgs& a, PullScalarLoads<NT>& s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool has_valid = a.vpart != nullptr;
  const float* vp = has_valid ? a.vpart : a.lpart;
  const int nv = has_valid ? a.nvparts : a.nlparts;
#pragma unroll
  for (int j = 0; j < PullScalarLoads<NT>::VW; ++j) {
    const int r = 64 * (wave + j * (NT / 64)) + lane;
    s.x[j] = a.lpart[r < a.nlparts ? r : a.nlparts - 1];
    s.y[j] = vp[r < nv ? r : nv - 1];
  }
  s.best = a.best_loss[a.parity];
}
// n{l,v}parts <= 1024 (one term per virtual thread of tail_loss_total).  scratch: 32 floats of LDS.
template <int NT>
__device__ __forceinline__ bool pull_scalars_finish(const PullArgs& a, const PullScalarLoads<NT>& s, float* scratch, bool writer) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool has_valid = a.vpart != nullptr;
#pragma unroll
  for (int j = 0; j < PullScalarLoads<NT>::VW; ++j) {
    const int w = wave + j * (NT / 64), r = 64 * w + lane;
    float x = 0.f, y = 0.f;
    if (r < a.nlparts) x += s.x[j];
    if (has_valid && r < a.nvparts) y += s.y[j];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    for (int off = 32; off > 0; off >>= 1) y += __shfl_down(y, off);
    if (lane == 0 && w < 16) { scratch[w] = x; scratch[16 + w] = y; }
  }
  __syncthreads();
  float sl = 0.f, sv = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) sl += scratch[w];
#pragma unroll
  for (int w = 0; w < 16; ++w) sv += scratch[16 + w];
  const float loss = sl * a.lscale, vloss = sv * a.vscale;
  const bool on_valid = has_valid && a.best_on_valid != 0;
  const float cmp = on_valid ? vloss : loss;
  bool track = false;
#pragma unroll
  for (int k = 0; k < kPullMaxNets; ++k) track = track || (k < a.n_nets && a.net[k].best_flat != nullptr);   // static indices
  const bool better = track && (cmp < s.best);
  if (writer && threadIdx.x == 0) {
    *a.loss_slot = loss;
    a.loss_hist[a.hist_index] = loss;
    if (has_valid) a.valid_hist[a.valid_index] = vloss;
    a.best_loss[a.parity ^ 1] = better ? cmp : s.best;
  }
  return better;
}

template <int NT, int LEN, int R> struct PullNetLoads {
  static constexpr int CPT = (LEN + NT - 1) / NT;          // columns per thread
  float a[CPT][R], p[CPT], m[CPT], v[CPT];
};
template <int NT, int LEN, int R>
__device__ __forceinline__ void pull_net_issue(const PullNet& n, int nparts, PullNetLoads<NT, LEN, R>& l) {
#pragma unroll
  for (int c = 0; c < PullNetLoads<NT, LEN, R>::CPT; ++c) {
    const int i = (int)threadIdx.x + c * NT < LEN ? (int)threadIdx.x + c * NT : LEN - 1;
#pragma unroll
    for (int r = 0; r < R; ++r) l.a[c][r] = n.part[(size_t)(r < nparts ? r : nparts - 1) * LEN + i];
    l.p[c] = n.p_in[i];
    l.m[c] = n.m_in[i];
    l.v[c] = n.v_in[i];
  }
}
template <int NT, int LEN, int R>
__device__ __forceinline__ void pull_net_finish(const PullNet& n, int nparts, const PullNetLoads<NT, LEN, R>& l, float* pnew,
                                                bool writer, bool better) {
#pragma unroll
  for (int c = 0; c < PullNetLoads<NT, LEN, R>::CPT; ++c) {
    const int i = (int)threadIdx.x + c * NT;
    if (i >= LEN) continue;
    // tail_column_total for nparts <= R <= 32: row group rg holds the chain (0 + a[rg]) + a[rg + 16], the other three
    // chains of the group are empty, the 16 groups are added in order
    float g = 0.f;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      float s0 = 0.f;
      const float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (rg < R && rg < nparts) s0 += l.a[c][rg < R ? rg : 0];
      if (rg + 16 < R && rg + 16 < nparts) s0 += l.a[c][rg + 16 < R ? rg + 16 : 0];
      g += (s0 + s1) + (s2 + s3);
    }
    float p, m, v;
    adam_value(n.adam, l.p[c], g, l.m[c], l.v[c], p, m, v);
    pnew[i] = p;
    if (writer) {
      n.p_out[i] = p; n.m_out[i] = m; n.v_out[i] = v;
      if (n.grad) n.grad[i] = g;
      if (better && n.best_flat) n.best_flat[i] = l.p[c];
    }
  }
}

template <int NT, int LEN, int K, int R>
__device__ __forceinline__ void pull_prologue_batched(const PullArgs& a, float* pnew, int stride, float* scratch, bool writer) {
  PullScalarLoads<NT> s;
  PullNetLoads<NT, LEN, R> l[K];
  pull_scalars_issue<NT>(a, s);
  pull_net_issue<NT, LEN, R>(a.net[0], a.nparts, l[0]);
  // two networks' loads in flight at a time where the registers allow it
  constexpr bool AHEAD = 2 * PullNetLoads<NT, LEN, R>::CPT * (R + 3) <= 224;
  if constexpr (K > 1 && AHEAD) pull_net_issue<NT, LEN, R>(a.net[1], a.nparts, l[1]);
  const bool better = pull_scalars_finish<NT>(a, s, scratch, writer);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if constexpr (AHEAD) {
      if (k + 2 < K) pull_net_issue<NT, LEN, R>(a.net[k + 2], a.nparts, l[k + 2]);
    }
    pull_net_finish<NT, LEN, R>(a.net[k], a.nparts, l[k], pnew + k * stride, writer, better);
    if constexpr (!AHEAD) {
      if (k + 1 < K) pull_net_issue<NT, LEN, R>(a.net[k + 1], a.nparts, l[k + 1]);
    }
  }
}

// The whole prologue for K networks of LEN parameters each: updated parameters of network k at pnew + k * stride (LDS),
// scratch: 32 floats of LDS.  The caller synchronises the workgroup afterwards.
template <int NT, int LEN, int K>
__device__ __forceinline__ void pull_prologue(const PullArgs& a, float* pnew, int stride, float* scratch, bool writer) {
  bool same = a.nlparts <= 1024 && a.nvparts <= 1024;
#pragma unroll
  for (int k = 0; k < K; ++k) same = same && a.net[k].len == LEN;
  if (same && a.nparts <= 2) {
    pull_prologue_batched<NT, LEN, K, 2>(a, pnew, stride, scratch, writer);
  } else if (same && a.nparts <= 16) {
    pull_prologue_batched<NT, LEN, K, 16>(a, pnew, stride, scratch, writer);
  } else {                     // 17 .. 32 rows: chunks of columns (the hosts of this package never ask for it, see kPullMaxWork)
    const bool better = pull_scalars(a, scratch, writer);
#pragma unroll
    for (int k = 0; k < K; ++k) pull_update_net(a.net[k], a.nparts, pnew + k * stride, writer, better, threadIdx.x, blockDim.x);
  }
}


// ---- loop mode: a whole run of epochs in ONE launch -----------------------------------------------------------------
// When the training grid and the validation grid are one workgroup each (the reference's default ODE solvers: 32 points),
// nothing of an epoch has to leave the CU: ONE workgroup runs launch after launch of the pull-mode sequence above inside
// a loop -- prologue (finish the previous epoch), validation closure, training closure -- with parameters, moments, the
// gradient row and the loss partials in LDS.  It reads the state the way a pull-mode launch would find it and leaves
// what the last launch of the run would have left (in the other buffer set), so ndq_fused_fit_run can mix the two and
// ends with the same ordinary tail.  Same device functions, same order of operations: bit-identical to both other routes.
constexpr int kLoopMaxLaunches = 64;    // launches one loop kernel stands for (Adam's bias corrections travel as arguments)
constexpr int kLoopMaxNets = 2;
struct LoopNet {
  const float* p_in; const float* m_in; const float* v_in; const float* part_in;     // state at entry (part_in: e0 >= 1)
  float* p_out; float* m_out; float* v_out; float* part_out;                         // ... and at exit
  float* grad; float* best_flat;
  float lr, b1, b2, eps, wd;
  float bc1[kLoopMaxLaunches], bc2s[kLoopMaxLaunches];     // of the epoch finished by the prologue of launch e0 + i
};
struct LoopArgs {
  int e0, e1;                  // launches [e0, e1) of the call's sequence (launch e: prologue for e >= 1, training closure
  int n_epochs;                // for e < n_epochs, validation closure for has_valid && e >= 1)
  int has_valid, track_best;
  int hist_index, valid_index, parity;        // as passed to ndq_fused_fit_run (values of the call's first launch)
  long long coord_stride;      // floats between the training batches of consecutive epochs
  const float* lp_in; const float* vp_in; float* lp_out; float* vp_out;
  float lscale, vscale;
  float* loss_hist; float* loss_slot; float* valid_hist; float* best_loss;
  LoopNet net[kLoopMaxNets];
};

// The PullArgs launch e of the sequence would have been given, with every buffer of the epoch being finished in LDS:
// state[k] = {p, m, v, row} of network k (PP floats each), misc = {training loss partial, validation loss partial,
// best-loss ping-pong (2)}.
template <int K>
__device__ __forceinline__ void loop_pull_args(const LoopArgs& L, int e, float* state, int pp, int len, float* misc, PullArgs& pa) {
  const int j = e - 1;
  const bool with_valid = L.has_valid != 0 && j >= 1;
  pa.enabled = 1; pa.n_nets = K; pa.nparts = 1;
  pa.lpart = misc; pa.nlparts = 1; pa.lscale = L.lscale;
  pa.loss_hist = L.loss_hist; pa.hist_index = L.hist_index + j; pa.loss_slot = L.loss_slot;
  pa.vpart = with_valid ? misc + 1 : nullptr; pa.nvparts = 1; pa.vscale = L.vscale;
  pa.valid_hist = L.valid_hist; pa.valid_index = L.valid_index + j - 1; pa.best_on_valid = L.track_best == 2 ? 1 : 0;
  pa.best_loss = misc + 2; pa.parity = L.parity ^ (j & 1);
  const bool track = L.track_best == 1 || (L.track_best == 2 && with_valid);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float* s = state + (size_t)k * 4 * pp;
    PullNet& n = pa.net[k];
    n.part = s + 3 * pp;
    n.p_in = s; n.m_in = s + pp; n.v_in = s + 2 * pp;
    n.p_out = s; n.m_out = s + pp; n.v_out = s + 2 * pp;
    n.grad = L.net[k].grad; n.best_flat = track ? L.net[k].best_flat : nullptr; n.len = len;
    n.adam = AdamConsts{L.net[k].lr, L.net[k].b1, L.net[k].b2, L.net[k].eps, L.net[k].wd, L.net[k].bc1[e - L.e0],
                        L.net[k].bc2s[e - L.e0]};
  }
}


}  // namespace ndq
