// C-ABI of libndq.so (declared in include/ndq.h): descriptor dispatch onto the templated gfx950 kernels of
// ndq_mlp.h, the second-stage reduction and the fused Adam step.
#include <cstdlib>
#include <vector>
#include "ndq_launch.h"
#include "ndq_sample.h"
#include "ndq_oneshot.h"
#include "ndq_tail.h"

extern "C" int ndq_oneshot_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream);

namespace ndq {

// ---------------------------------------------------------------------------------------------- kernel table
// X(D, FIRST, MASK2, NB, L, ACT, NOUT, LAP).  Stream sets are closed under "second order needs first order".
// BASELINE configs: C1 (1,1,0,2,2,SIN) | C2 (2,1,0b101,2,2,TANH) | C3 (2,1,0b001,4,3,TANH) | C5 u,v (2,1,0b101,4,3,TANH),
// p (2,1,0,4,3,TANH); value-only variants serve solution evaluation (solvers.py:682-720).
// sigmoid / swish: 1-D and 2-D stream sets at the default width.  d = 3: SolverSpherical's default FCNN(3,1,(32,32)) (solvers_spherical.py) -- diagonal (0b101001), Laplacian-merged and full Hessian.
#ifndef NDQ_CFG_TABLE
#define NDQ_CFG_TABLE(X)      \
  X(1, 0, 0, 2, 2, ACT_SIN, 1, 0)   \
  X(1, 1, 0, 2, 2, ACT_SIN, 1, 0)   \
  X(1, 1, 1, 2, 2, ACT_SIN, 1, 0)   \
  X(1, 0, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(1, 1, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(1, 1, 1, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 0, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 1, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 1, 1, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 1, 5, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 1, 7, 2, 2, ACT_TANH, 1, 0)  \
  X(2, 0, 0, 4, 3, ACT_TANH, 1, 0)  \
  X(2, 1, 0, 4, 3, ACT_TANH, 1, 0)  \
  X(2, 1, 1, 4, 3, ACT_TANH, 1, 0)  \
  X(2, 1, 5, 4, 3, ACT_TANH, 1, 0)  \
  X(1, 0, 0, 2, 2, ACT_TANH, 25, 0) \
  X(1, 1, 1, 2, 2, ACT_TANH, 25, 0) \
  X(2, 1, 5, 2, 2, ACT_TANH, 3, 0)  \
  X(2, 1, 5, 2, 2, ACT_TANH, 1, 1)  \
  X(2, 1, 5, 4, 3, ACT_TANH, 1, 1)  \
  X(3, 0, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(3, 1, 0, 2, 2, ACT_TANH, 1, 0)  \
  X(3, 1, 1, 2, 2, ACT_TANH, 1, 0)  \
  X(3, 1, 41, 2, 2, ACT_TANH, 1, 0) \
  X(3, 1, 41, 2, 2, ACT_TANH, 1, 1) \
  X(3, 1, 63, 2, 2, ACT_TANH, 1, 0)  \
  X(1, 0, 0, 2, 2, ACT_SIGMOID, 1, 0) \
  X(1, 1, 0, 2, 2, ACT_SIGMOID, 1, 0) \
  X(1, 1, 1, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 0, 0, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 1, 0, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 1, 1, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 1, 5, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 1, 7, 2, 2, ACT_SIGMOID, 1, 0) \
  X(2, 1, 5, 2, 2, ACT_SIGMOID, 1, 1) \
  X(1, 0, 0, 2, 2, ACT_SWISH, 1, 0) \
  X(1, 1, 0, 2, 2, ACT_SWISH, 1, 0) \
  X(1, 1, 1, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 0, 0, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 1, 0, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 1, 1, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 1, 5, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 1, 7, 2, 2, ACT_SWISH, 1, 0) \
  X(2, 1, 5, 2, 2, ACT_SWISH, 1, 1)
#endif

#define NDQ_ENTRY(D, F, M, NB, L, A, O, LP) make_kernels<Cfg<D, F, M, NB, L, A, O, LP>>(),

static const ndq_mlp_kernels kTable[] = {NDQ_CFG_TABLE(NDQ_ENTRY)};
static std::vector<const ndq_mlp_kernels*> g_registered;     // extension modules (ndq_mlp_register)

static bool same_desc(const ndq_mlp_desc& a, const ndq_mlp_desc& b) {
  return a.d == b.d && a.first == b.first && a.mask2 == b.mask2 && a.hidden == b.hidden && a.layers == b.layers &&
         a.act == b.act && a.n_out == b.n_out && a.lap == b.lap && a.skip == b.skip && a.mask3 == b.mask3 && a.mask4 == b.mask4 &&
         a.actp == b.actp && a.widths == b.widths && a.mono == b.mono;
}

static const ndq_mlp_kernels* find(const ndq_mlp_desc* d) {
  if (!d || d->hidden < 1 || d->hidden > NDQ_MAX_HIDDEN) return nullptr;
  for (const ndq_mlp_kernels& e : kTable)
    if (same_desc(e.desc, *d)) return &e;
  for (const ndq_mlp_kernels* e : g_registered)
    if (same_desc(e->desc, *d)) return e;
  return nullptr;
}

static int bwd_blocks(const ndq_mlp_kernels* e, int n) {
  const int tiles = (n + 15) / 16;
  int blocks = (tiles + e->bwd_waves - 1) / e->bwd_waves;
  if (blocks > NDQ_BWD_MAX_BLOCKS) blocks = NDQ_BWD_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return blocks;
}

// ---------------------------------------------------------------------------------------------- reduction
// out[i] = (acc ? out[i] : 0) + scale * sum_r partials[r*len + i];  one thread per 4 columns, rows in fixed order,
// 4 independent row chains per thread for memory-level parallelism (combined in a fixed order).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nparts, int len,
                                                              float* __restrict__ out, int accumulate, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  // fp64 accumulators: a loss over 1 M points arrives as 4 096 block sums of one sign -- a sequential fp32 sum of those
  // measured 5e-6 off at C5; the adds are nothing next to the loads
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  int r = 0;
#pragma unroll 4                      // (16 loads in flight per thread instead of 4: the chains stay in order)
  for (; r + 3 < nparts; r += 4) {
    s0 += (double)part[(size_t)r * len + i];
    s1 += (double)part[(size_t)(r + 1) * len + i];
    s2 += (double)part[(size_t)(r + 2) * len + i];
    s3 += (double)part[(size_t)(r + 3) * len + i];
  }
  for (; r < nparts; ++r) s0 += (double)part[(size_t)r * len + i];
  const float s = (float)(((s0 + s1) + (s2 + s3)) * (double)scale);
  out[i] = accumulate ? out[i] + s : s;
}

// both second-stage sums of one batch in one launch: blocks [0, gridDim.x-1) reduce 64 gradient columns each with 4
// row groups per column (fixed order: row groups sequentially, then the 4 group sums in order); the last block sums
// the loss partials (wave shuffle tree + the 4 waves in order).
struct Reduce2Args {
  const float* part; int nparts, len; float* out; int accumulate;
  const float* lpart; int nlparts; float* lout; float lscale;
};

// ---- the per-thread part of a column sum with EVERY load in flight before the first add (round 5).
// The loops "for (r = rg; r + 48 < nparts; r += 64) { s0 += ..; s1 += ..; s2 += ..; s3 += ..; }" below compile to four
// loads, s_waitcnt vmcnt(0), four adds, branch: with the closure kernel's 256 partial rows that is FOUR dependent round
// trips to memory written by the previous launch (i.e. HBM / another XCD's L2, ~0.7 us each) inside a kernel whose whole
// duration is 5 us -- and the loss-partial loop in front of it and the parameter / moment loads behind it add three more.
// ColumnRows issues the first 16 rows of a thread (rows rg, rg + 16, ..., rg + 240: all of them for nparts <= 256, the
// closure kernels' maximum) as straight-line clamped loads; finish() adds them in EXACTLY the order of the loops it
// replaces (rows beyond nparts contribute +0.0f), then walks any further rows the old way.
struct ColumnRows {
  float v[16];
  // UNCONDITIONAL clamped loads: a load whose value is only selected under "row < nparts" gets sunk into that branch by the
  // compiler (one load + s_waitcnt per row -- seen in the ISA), so the raw values are pinned by one empty asm statement in
  // finish(), where all sixteen must be live at once; the selection happens after it.
  __device__ __forceinline__ void issue(const float* __restrict__ part, int nparts, int len, int col, int rg) {
    const int cc = col < len ? col : len - 1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = rg + 16 * j;
      v[j] = part[(size_t)(r < nparts ? r : nparts - 1) * len + cc];
    }
  }
  // (s0 + s1) + (s2 + s3) of: for (r = rg; r + 48 < nparts; r += 64) {s0 += row r; s1 += row r+16; s2 += row r+32; s3 += row r+48;}
  //                           for (; r < nparts; r += 16) s0 += row r;
  __device__ __forceinline__ float finish(const float* __restrict__ part, int nparts, int len, int col, int rg) const {
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = v[j];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                      "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (rg + 16 * j < nparts && col < len) ? x[j] : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (rg + 64 * it + 48 < nparts) { s0 += x[4 * it]; s1 += x[4 * it + 1]; s2 += x[4 * it + 2]; s3 += x[4 * it + 3]; }
      else { s0 += x[4 * it]; s0 += x[4 * it + 1]; s0 += x[4 * it + 2]; s0 += x[4 * it + 3]; }
    }
    if (nparts > 256 && col < len) {                     // (larger partial sets: the remaining rows, same scheme)
      int r = rg + 256;
      for (; r + 48 < nparts; r += 64) {
        s0 += part[(size_t)r * len + col];
        s1 += part[(size_t)(r + 16) * len + col];
        s2 += part[(size_t)(r + 32) * len + col];
        s3 += part[(size_t)(r + 48) * len + col];
      }
      for (; r < nparts; r += 16) s0 += part[(size_t)r * len + col];
    }
    return (s0 + s1) + (s2 + s3);
  }
};

// column sums of partials[nparts][len] for the 64 columns of this workgroup: 16 row groups x 64 columns, every thread
// keeps 4 independent chains (rows rg, rg+16, ...), then the 16 group sums are added in fixed order.  Returns the
// column total in the threads of row group 0 (valid where col < len).
__device__ __forceinline__ float column_sum_1024(const float* __restrict__ part, int nparts, int len, int col, int rg,
                                                 float* sm /* [16*64] */) {
  ColumnRows rows;
  rows.issue(part, nparts, len, col, rg);
  sm[rg * 64 + (threadIdx.x & 63)] = rows.finish(part, nparts, len, col, rg);
  __syncthreads();
  float s = 0.f;
  if (rg == 0) {
#pragma unroll
    for (int g = 0; g < 16; ++g) s += sm[g * 64 + (threadIdx.x & 63)];
  }
  return s;
}

// sum of n floats by one 1024-thread workgroup, fixed order; result valid in thread 0
__device__ __forceinline__ float block_sum_1024(const float* __restrict__ v, int n, float* sm16) {
  float x = 0.f;
  for (int r = threadIdx.x; r < n; r += 1024) x += v[r];
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
  if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < 16; ++w) s += sm16[w];
  return s;
}

__global__ __launch_bounds__(1024) void reduce_grad_loss_kernel(Reduce2Args a) {
  __shared__ float sm[16 * 64];
  if (blockIdx.x + 1 < gridDim.x) {
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    const float s = column_sum_1024(a.part, a.nparts, a.len, i, rg, sm);
    if (rg == 0 && i < a.len) a.out[i] = a.accumulate ? a.out[i] + s : s;
  } else {
    const float s = block_sum_1024(a.lpart, a.nlparts, sm);
    if (threadIdx.x == 0) *a.lout = s * a.lscale;
  }
}

// ---------------------------------------------------------------------------------------------- epoch tail
struct TailArgs {
  float* p; const float* g; float* m; float* v; int len;
  float lr, b1, b2, eps, wd, bc1, bc2s;
  // where the parameters / moments this epoch started from live, when that is not p / m / v (fit() in pull mode keeps
  // two sets of buffers; the tail that closes a call brings the result home): nullptr = in place
  const float* p_in; const float* m_in; const float* v_in;
  const float* loss_slots; int nb; float* loss_hist; int hist_index; float* best_loss; int parity; float* best_flat;
  int write_scalars;
};
__global__ __launch_bounds__(256) void epoch_tail_kernel(TailArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float loss = 0.f;
  for (int k = 0; k < a.nb; ++k) loss += a.loss_slots[k];
  loss /= (float)a.nb;
  const float best = a.best_loss[a.parity];
  const bool better = (a.best_flat != nullptr) && (loss < best);   // false for NaN, like the reference's comparison
  if (i < a.len) {
    const float pi = a.p[i];
    if (better) a.best_flat[i] = pi;
    if (a.m != nullptr) {             // validation epochs pass no optimiser state: bookkeeping only
      float pn, mi, vi;
      ndq::adam_value(ndq::AdamConsts{a.lr, a.b1, a.b2, a.eps, a.wd, a.bc1, a.bc2s}, pi, a.g[i], a.m[i], a.v[i], pn, mi, vi);
      a.m[i] = mi;
      a.v[i] = vi;
      a.p[i] = pn;
    }
  }
  if (i == 0 && a.write_scalars) {
    a.loss_hist[a.hist_index] = loss;
    a.best_loss[a.parity ^ 1] = better ? loss : best;
  }
}

// second-stage sums AND the epoch tail in one launch (single batch per epoch, no all-reduce in between): every
// workgroup first adds up the loss partials itself (nlparts <= a few hundred floats), then reduces its 64 gradient
// columns and applies best-snapshot + Adam to them.
// loss of a validation batch evaluated by the same closure launch (ndq_fused_fit_run): block partials -> one scalar
struct ValidArgs {
  const float* part; int nparts; float scale; float* hist; int index;
  int best_on_valid;          // 1: the best-network snapshot follows the validation loss instead of the training loss
};
struct ReduceTailArgs {
  Reduce2Args r;
  TailArgs t;
  int tail_blocks;            // workgroups [tail_blocks, gridDim.x) draw the next batch (smp), if any
  ndq::SampleArgs smp;
  ValidArgs v;                // v.part == nullptr: no validation loss in this launch
};
// TC: gradient columns per workgroup (TC x 16 threads).  64 (1024 threads) is the layout of rounds 2 - 5; narrower workgroups
// spread the 1.2 MB of partial rows over more CUs / XCDs and launch faster (NDQ_TAIL_COLS; profiles/r06p_tail_cols_ab.txt).  The
// per-column order (16 row groups of 16 rows) and the loss order (64-lane tree per wave, waves in index order) do not depend on
// TC as long as the loss partials fit one per thread -- the launchers fall back to 64 columns otherwise.
#ifndef NDQ_TAIL_COLS
#define NDQ_TAIL_COLS 64
#endif
template <int TC>
__device__ __forceinline__ void reduce_tail_body(const ReduceTailArgs& a) {
  // Latency-bound (19 workgroups at C2): every global load -- loss partials, this thread's rows of the gradient
  // partials, the parameter / moment values it will update -- is issued before the first reduction step, and there is
  // ONE barrier.  Summation orders are fixed (per-thread chains, then LDS slots added in index order).
  constexpr int TT = TC * 16, WAVES = TT / 64;
  __shared__ float sm[16 * TC];
  __shared__ float smw[16];
  __shared__ float smv[16];
  const int tid = threadIdx.x, c = tid % TC, rg = tid / TC, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * TC + c;
  const bool col = i < a.r.len;
  const bool has_valid = a.v.part != nullptr;           // uniform over the launch
  const bool has_train = a.r.nparts > 0;                // false: stand-alone validation epoch (no sums, no Adam)
  // ---- issue: gradient-partial rows, first loss / validation partial, parameter + moments, best loss -- straight-line
  // (see ColumnRows: the loops these replace serialised seven memory round trips)
  ColumnRows rows;
  if (has_train) rows.issue(a.r.part, a.r.nparts, a.r.len, i, rg);
  const float lp0 = tid < a.r.nlparts ? a.r.lpart[tid] : 0.f;
  const float vp0 = has_valid && tid < a.v.nparts ? a.v.part[tid] : 0.f;
  const bool upd = (rg == 0) && col;
  const float* p_in = a.t.p_in ? a.t.p_in : a.t.p;
  const float* m_in = a.t.m_in ? a.t.m_in : a.t.m;
  const float* v_in = a.t.v_in ? a.t.v_in : a.t.v;
  float pi = 0.f, m0 = 0.f, v0 = 0.f;
  if (upd) {
    pi = p_in[i];
    if (has_train || a.t.p_in) { m0 = m_in[i]; v0 = v_in[i]; }
  }
  const float best = a.t.best_loss[a.t.parity];
  // ---- use
  float lp = lp0, vp = vp0;
  for (int r = tid + TT; r < a.r.nlparts; r += TT) lp += a.r.lpart[r];       // (TC < 64: never taken, see the launchers)
  if (has_valid)
    for (int r = tid + TT; r < a.v.nparts; r += TT) vp += a.v.part[r];
  const float colsum = has_train ? rows.finish(a.r.part, a.r.nparts, a.r.len, i, rg) : 0.f;
  for (int off = 32; off > 0; off >>= 1) lp += __shfl_down(lp, off);
  if (has_valid)
    for (int off = 32; off > 0; off >>= 1) vp += __shfl_down(vp, off);
  if (lane == 0) { smw[wave] = lp; smv[wave] = vp; }
  sm[rg * TC + c] = colsum;
  __syncthreads();
  float loss = 0.f, vloss = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) loss += smw[w];      // (the 16 - WAVES waves a 1024-thread workgroup would add hold exact zeros)
  loss *= a.r.lscale;
  if (has_valid) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) vloss += smv[w];
    vloss *= a.v.scale;
  }
  // what the snapshot follows: the validation loss of the parameters this epoch starts from (fit() with validation
  // epochs), else the training loss (solvers.py:414-415: n_batches_valid = 0)
  const bool on_valid = has_valid && a.v.best_on_valid != 0;
  const float cmp = on_valid ? vloss : loss;
  const bool better = (a.t.best_flat != nullptr) && (on_valid || has_train) && (cmp < best);
  if (upd) {
    if (better) a.t.best_flat[i] = pi;
    if (has_train) {
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) g += sm[k * TC + c];
      a.r.out[i] = g;
      float pn, mi, vi;
      ndq::adam_value(ndq::AdamConsts{a.t.lr, a.t.b1, a.t.b2, a.t.eps, a.t.wd, a.t.bc1, a.t.bc2s}, pi, g, m0, v0, pn, mi, vi);
      a.t.m[i] = mi;
      a.t.v[i] = vi;
      a.t.p[i] = pn;
    } else if (a.t.p_in) {             // a validation-only tail that also brings the parameters / moments home
      a.t.p[i] = pi; a.t.m[i] = m0; a.t.v[i] = v0;
    }
  }
  if (blockIdx.x == 0 && tid == 0 && a.t.write_scalars) {
    if (has_train) {
      *a.r.lout = loss;
      a.t.loss_hist[a.t.hist_index] = loss;
    }
    if (has_valid) a.v.hist[a.v.index] = vloss;
    a.t.best_loss[a.t.parity ^ 1] = better ? cmp : best;
  }
}
__global__ __launch_bounds__(1024) void reduce_tail_kernel(ReduceTailArgs a) {
  if ((int)blockIdx.x >= a.tail_blocks) {                  // prefetch of the next batch (ndq_fused_step.next_sampler)
    ndq::sample_point_store(a.smp, ((int)blockIdx.x - a.tail_blocks) * 1024 + (int)threadIdx.x);
    return;
  }
  reduce_tail_body<64>(a);
}

// Data parallel with the one-shot exchange (ndq_oneshot.h): the SAME launch also carries the all-reduce.  Every
// workgroup reduces its 64 gradient columns (and the loss) locally, pushes that slice [64 gradients | loss] into every
// rank's inbox, waits for the slices of all ranks, adds them in rank order and applies best-snapshot + Adam to its
// columns -- a data-parallel training epoch is two launches, like a single-GPU one.  Inbox slice of workgroup b:
// floats [65 b, 65 b + 65) of the parity's per-rank vector; flag (parity, source rank, b).
__global__ __launch_bounds__(1024) void reduce_tail_dp_kernel(ReduceTailArgs a, ndq::OneshotDev c, unsigned step) {
  __shared__ float sm[16 * 64];
  __shared__ float smw[16];
  __shared__ float sloss;
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6, blk = blockIdx.x;
  const int i = blk * 64 + col;
  const bool incol = i < a.r.len;
  ColumnRows rows;                                        // (every load in flight before the first add: see ColumnRows)
  rows.issue(a.r.part, a.r.nparts, a.r.len, i, rg);
  const float lp0 = tid < a.r.nlparts ? a.r.lpart[tid] : 0.f;
  const bool upd = (rg == 0) && incol;
  float pi = 0.f, m0 = 0.f, v0 = 0.f;
  if (upd) { pi = a.t.p[i]; m0 = a.t.m[i]; v0 = a.t.v[i]; }
  const float best = a.t.best_loss[a.t.parity];
  float lp = lp0;
  for (int r = tid + 1024; r < a.r.nlparts; r += 1024) lp += a.r.lpart[r];
  const float colsum = rows.finish(a.r.part, a.r.nparts, a.r.len, i, rg);
  for (int off = 32; off > 0; off >>= 1) lp += __shfl_down(lp, off);
  if (col == 0) smw[rg] = lp;
  sm[rg * 64 + col] = colsum;
  __syncthreads();
  // ---- push this workgroup's slice [64 local column sums | local loss] to every rank
  const int parity = step & 1u;
  const size_t slot = ((size_t)parity * c.world + c.rank) * c.max_len + (size_t)blk * 65;
  if (rg == 0) {
    float g = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) g += sm[k * 64 + col];
    for (int q = 0; q < c.world; ++q)
      __hip_atomic_store(c.inbox[q] + slot + col, incol ? g : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (col == 0) {
      float loss = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) loss += smw[w];
      loss *= a.r.lscale;
      for (int q = 0; q < c.world; ++q)
        __hip_atomic_store(c.inbox[q] + slot + 64, loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < c.world) {
    unsigned* f = c.flags[tid] + ((size_t)parity * c.world + c.rank) * c.max_blocks + blk;
    __hip_atomic_store(f, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = c.flags[c.rank] + ((size_t)parity * c.world + tid) * c.max_blocks + blk;
    if (!ndq::oneshot_wait(mine, step, c.spin_limit, c.status)) timed_out = 1;
  }
  __syncthreads();
  // a peer's slice never arrived (counted in the status word, which the solver turns into an error at its next history
  // flush): no parameter is updated with a stale inbox, and the epoch's loss is NaN
  const bool bad = timed_out != 0;
  // ---- fixed-order sum over the ranks, then the tail on the global values
  const float* inbox = c.inbox[c.rank] + (size_t)parity * c.world * c.max_len + (size_t)blk * 65;
  if (tid == 0) {
    float loss = 0.f;
    for (int q = 0; q < c.world; ++q)
      loss += __hip_atomic_load(inbox + (size_t)q * c.max_len + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    sloss = bad ? __builtin_nanf("") : loss;
  }
  __syncthreads();
  const float loss = sloss;
  const bool better = (a.t.best_flat != nullptr) && (loss < best);
  if (upd && !bad) {
    float g = 0.f;
    for (int q = 0; q < c.world; ++q)
      g += __hip_atomic_load(inbox + (size_t)q * c.max_len + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.r.out[i] = g;
    if (better) a.t.best_flat[i] = pi;
    float pn, mi, vi;
    ndq::adam_value(ndq::AdamConsts{a.t.lr, a.t.b1, a.t.b2, a.t.eps, a.t.wd, a.t.bc1, a.t.bc2s}, pi, g, m0, v0, pn, mi, vi);
    a.t.m[i] = mi;
    a.t.v[i] = vi;
    a.t.p[i] = pn;
  }
  if (blk == 0 && tid == 0 && a.t.write_scalars) {
    *a.r.lout = loss;
    a.t.loss_hist[a.t.hist_index] = loss;
    a.t.best_loss[a.t.parity ^ 1] = better ? loss : best;
  }
}

// the same for the 2..4 networks behind one multi-network closure launch, in ONE launch: blockIdx.y = network
struct ReduceTailMultiArgs {
  ReduceTailArgs net[4];
};
template <int TC>
__global__ __launch_bounds__(TC * 16) void reduce_tail_multi_kernel(ReduceTailMultiArgs a) {
  const ReduceTailArgs& mine = a.net[blockIdx.y];
  if ((int)blockIdx.x * TC >= mine.r.len) return;          // networks of one shape: never taken, kept for safety
  reduce_tail_body<TC>(mine);
}
// the sums / tail launch of all networks of a system: narrow workgroups where every loss partial has a thread of its own
static void launch_tail_multi(const ReduceTailMultiArgs& a, int max_params, int n_nets, int max_lparts, hipStream_t st) {
  if (NDQ_TAIL_COLS < 64 && max_lparts <= NDQ_TAIL_COLS * 16)
    hipLaunchKernelGGL(reduce_tail_multi_kernel<NDQ_TAIL_COLS>, dim3((max_params + NDQ_TAIL_COLS - 1) / NDQ_TAIL_COLS, n_nets),
                       dim3(NDQ_TAIL_COLS * 16), 0, st, a);
  else
    hipLaunchKernelGGL(reduce_tail_multi_kernel<64>, dim3((max_params + 63) / 64, n_nets), dim3(1024), 0, st, a);
}

// ---------------------------------------------------------------------------------------------- Adam
// torch.optim.Adam (amsgrad=False, maximize=False) single-tensor formula:
//   g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int len, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  float pn, mi, vi;
  ndq::adam_value(ndq::AdamConsts{lr, b1, b2, eps, wd, bc1, bc2s}, p[i], g[i], m[i], v[i], pn, mi, vi);
  m[i] = mi;
  v[i] = vi;
  p[i] = pn;
}

}  // namespace ndq

using namespace ndq;

extern "C" {

int ndq_mlp_supported(const ndq_mlp_desc* desc) { return find(desc) ? 1 : 0; }

int ndq_mlp_register(const ndq_mlp_kernels* k) {
  if (!k || !k->fwd || !k->bwd || k->desc.hidden < 1 || k->desc.hidden > NDQ_MAX_HIDDEN || k->n_streams < 1 || k->n_params < 1 || k->bwd_waves < 1 ||
      k->lds_bytes > 160 * 1024)
    return NDQ_EINVAL;
  if (!find(&k->desc)) g_registered.push_back(k);
  return 0;
}

int ndq_mlp_num_streams(const ndq_mlp_desc* desc) {
  const ndq_mlp_kernels* e = find(desc);
  return e ? e->n_streams : NDQ_EUNSUPPORTED;
}

int ndq_mlp_num_params(const ndq_mlp_desc* desc) {
  const ndq_mlp_kernels* e = find(desc);
  return e ? e->n_params : NDQ_EUNSUPPORTED;
}

int ndq_mlp_bwd_blocks(const ndq_mlp_desc* desc, int n) {
  const ndq_mlp_kernels* e = find(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (n <= 0) return NDQ_EINVAL;
  return bwd_blocks(e, n);
}

int ndq_mlp_jet_fwd(const ndq_mlp_desc* desc, const float* coords, int ldc, int n, const float* params, float* jets,
                    int ldj, void* stream) {
  const ndq_mlp_kernels* e = find(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (!coords || !params || !jets || n <= 0 || ldc < n || ldj < n) return NDQ_EINVAL;
  return e->fwd(coords, ldc, n, params, jets, ldj, stream);
}

int ndq_mlp_jet_bwd(const ndq_mlp_desc* desc, const float* coords, int ldc, int n, const float* params,
                    const float* gbar, int ldj, float* partials, void* stream) {
  const ndq_mlp_kernels* e = find(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (!coords || !params || !gbar || !partials || n <= 0 || ldc < n || ldj < n) return NDQ_EINVAL;
  return e->bwd(coords, ldc, n, params, gbar, ldj, partials, bwd_blocks(e, n), stream);
}

int ndq_reduce_partials(const float* partials, int nparts, int len, float* out, int accumulate, float scale,
                        void* stream) {
  if (!partials || !out || nparts <= 0 || len <= 0) return NDQ_EINVAL;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     partials, nparts, len, out, accumulate, scale);
  return (int)hipGetLastError();
}

int ndq_reduce_grad_loss(const float* partials, int nparts, int len, float* out, int accumulate,
                         const float* loss_partials, int n_loss_parts, float* loss_out, float loss_scale, void* stream) {
  if (!partials || !out || !loss_partials || !loss_out || nparts <= 0 || len <= 0 || n_loss_parts <= 0) return NDQ_EINVAL;
  Reduce2Args a{partials, nparts, len, out, accumulate, loss_partials, n_loss_parts, loss_out, loss_scale};
  hipLaunchKernelGGL(reduce_grad_loss_kernel, dim3((len + 63) / 64 + 1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

int ndq_epoch_tail(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int len, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, const float* loss_slots, int n_batches,
                   float* loss_hist, int hist_index, float* best_loss, int parity, float* best_flat, int write_scalars,
                   void* stream) {
  const bool adam = exp_avg != nullptr;
  if (!params || len <= 0 || !loss_slots || n_batches <= 0 || !loss_hist || !best_loss || hist_index < 0 ||
      (parity != 0 && parity != 1) || (adam && (!grad || !exp_avg_sq || step <= 0)))
    return NDQ_EINVAL;
  TailArgs a{};
  a.p = params; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.len = len;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = adam ? (float)(1.0 - pow((double)beta1, (double)step)) : 1.f;
  a.bc2s = adam ? (float)sqrt(1.0 - pow((double)beta2, (double)step)) : 1.f;
  a.loss_slots = loss_slots; a.nb = n_batches; a.loss_hist = loss_hist; a.hist_index = hist_index;
  a.best_loss = best_loss; a.parity = parity; a.best_flat = best_flat; a.write_scalars = write_scalars;
  hipLaunchKernelGGL(epoch_tail_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

static void fill_reduce_tail(ReduceTailArgs& a, const ndq_fused_step* s, const float* loss_partials, int blocks, float seed,
                             float* loss_hist, float* best_loss, int adam_step, int hist_index, int parity,
                             int write_scalars);

int ndq_fused_step_run(const ndq_fused_step* s, const float* coords, int adam_step, int hist_index, int parity,
                       void* stream) {
  if (!s || !s->launch || !coords) return NDQ_EINVAL;
  int rc = s->launch(coords, s->ldc, s->n, s->params, s->partials, s->loss_partials, nullptr, nullptr, s->ldj, s->seed,
                     1, stream);
  if (rc) return rc;
  if (!s->adam_m)
    return ndq_reduce_grad_loss(s->partials, s->blocks, s->n_params, s->grad, 0, s->loss_partials, s->blocks,
                                s->loss_slot, s->seed, stream);
  if (adam_step <= 0 || hist_index < 0 || (parity != 0 && parity != 1) || !s->loss_hist || !s->best_loss) return NDQ_EINVAL;
  if (s->allreduce == reinterpret_cast<ndq_allreduce_fn>(&ndq_oneshot_allreduce) && s->comm &&
      ((s->n_params + 63) / 64) * 65 <= static_cast<ndq::Oneshot*>(s->comm)->dev.max_len &&
      (s->n_params + 63) / 64 <= static_cast<ndq::Oneshot*>(s->comm)->dev.max_blocks) {
    // data parallel over the one-shot exchange: local sums + exchange + tail in ONE launch (reduce_tail_dp_kernel)
    ndq::Oneshot* c = static_cast<ndq::Oneshot*>(s->comm);
    ReduceTailArgs a{};
    fill_reduce_tail(a, s, s->loss_partials, s->blocks, s->seed, s->loss_hist, s->best_loss, adam_step, hist_index, parity, 1);
    const unsigned step = ++c->step;
    hipLaunchKernelGGL(reduce_tail_dp_kernel, dim3((s->n_params + 63) / 64), dim3(1024), 0, static_cast<hipStream_t>(stream), a,
                       c->dev, step);
    return (int)hipGetLastError();
  }
  if (s->allreduce) {
    // data parallel: local second-stage sums -> ONE all-reduce of [grad | loss] -> tail on the reduced vector
    if (s->loss_slot != s->grad + s->n_params) return NDQ_EINVAL;
    rc = ndq_reduce_grad_loss(s->partials, s->blocks, s->n_params, s->grad, 0, s->loss_partials, s->blocks, s->loss_slot,
                              s->seed, stream);
    if (rc) return rc;
    rc = s->allreduce(s->grad, s->grad, (size_t)s->n_params + 1, /*ncclFloat32*/ 7, /*ncclSum*/ 0, s->comm, stream);
    if (rc) return 1000 + rc;   // ncclResult_t, offset so that it cannot be taken for a hipError_t
    return ndq_epoch_tail(s->params, s->grad, s->adam_m, s->adam_v, s->n_params, s->lr, s->beta1, s->beta2, s->eps,
                          s->weight_decay, adam_step, s->loss_slot, 1, s->loss_hist, hist_index, s->best_loss, parity,
                          s->best_flat, 1, stream);
  }
  ReduceTailArgs a{};
  a.r = Reduce2Args{s->partials, s->blocks, s->n_params, s->grad, 0, s->loss_partials, s->blocks, s->loss_slot, s->seed};
  a.t.p = s->params; a.t.g = s->grad; a.t.m = s->adam_m; a.t.v = s->adam_v; a.t.len = s->n_params;
  a.t.lr = s->lr; a.t.b1 = s->beta1; a.t.b2 = s->beta2; a.t.eps = s->eps; a.t.wd = s->weight_decay;
  a.t.bc1 = (float)(1.0 - pow((double)s->beta1, (double)adam_step));
  a.t.bc2s = (float)sqrt(1.0 - pow((double)s->beta2, (double)adam_step));
  a.t.loss_slots = s->loss_slot; a.t.nb = 1; a.t.loss_hist = s->loss_hist; a.t.hist_index = hist_index;
  a.t.best_loss = s->best_loss; a.t.parity = parity; a.t.best_flat = s->best_flat; a.t.write_scalars = 1;
  a.tail_blocks = (s->n_params + 63) / 64;
  int blocks = a.tail_blocks;
  if (s->next_sampler) {
    rc = ndq::fill_sample_args(a.smp, s->next_sampler, s->next_seed, s->next_draw, s->next_stream, s->next_coords,
                               s->next_ldc);
    if (rc) return rc;
    blocks += (a.smp.total + 1023) / 1024;
  }
  hipLaunchKernelGGL(reduce_tail_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

static void fill_reduce_tail(ReduceTailArgs& a, const ndq_fused_step* s, const float* loss_partials, int blocks, float seed,
                             float* loss_hist, float* best_loss, int adam_step, int hist_index, int parity,
                             int write_scalars) {
  a.r = Reduce2Args{s->partials, blocks, s->n_params, s->grad, 0, loss_partials, blocks, s->loss_slot, seed};
  a.t.p = s->params; a.t.g = s->grad; a.t.m = s->adam_m; a.t.v = s->adam_v; a.t.len = s->n_params;
  a.t.lr = s->lr; a.t.b1 = s->beta1; a.t.b2 = s->beta2; a.t.eps = s->eps; a.t.wd = s->weight_decay;
  a.t.bc1 = (float)(1.0 - pow((double)s->beta1, (double)adam_step));
  a.t.bc2s = (float)sqrt(1.0 - pow((double)s->beta2, (double)adam_step));
  a.t.loss_slots = s->loss_slot; a.t.nb = 1; a.t.loss_hist = loss_hist; a.t.hist_index = hist_index;
  a.t.best_loss = best_loss; a.t.parity = parity; a.t.best_flat = s->best_flat; a.t.write_scalars = write_scalars;
  a.tail_blocks = 0x7fffffff;
  a.v = ValidArgs{};
}

int ndq_fused_multi_step_run(const ndq_fused_step* steps, int n_nets, ndq_fused_launch_multi_fn launch,
                             const float* coords, int adam_step, int hist_index, int parity, void* stream) {
  if (!steps || !launch || !coords || n_nets < 1 || n_nets > 4 || adam_step <= 0 || hist_index < 0 ||
      (parity != 0 && parity != 1))
    return NDQ_EINVAL;
  const ndq_fused_step& s0 = steps[0];
  if (!s0.loss_hist || !s0.best_loss || !s0.loss_partials) return NDQ_EINVAL;
  const float* params[4];
  float* partials[4];
  for (int k = 0; k < n_nets; ++k) {
    if (!steps[k].params || !steps[k].partials || !steps[k].grad || !steps[k].adam_m || !steps[k].adam_v ||
        !steps[k].loss_slot)
      return NDQ_EINVAL;
    params[k] = steps[k].params;
    partials[k] = steps[k].partials;
  }
  int rc = launch(coords, s0.ldc, s0.n, params, partials, s0.loss_partials, nullptr, nullptr, s0.ldj, s0.seed, 1, stream);
  if (rc) return rc;
  ReduceTailMultiArgs a{};
  int max_params = 0;
  for (int k = 0; k < n_nets; ++k) {
    fill_reduce_tail(a.net[k], &steps[k], s0.loss_partials, s0.blocks, s0.seed, s0.loss_hist, s0.best_loss, adam_step,
                     hist_index, parity, k == 0 ? 1 : 0);
    if (steps[k].n_params > max_params) max_params = steps[k].n_params;
  }
  for (int k = n_nets; k < 4; ++k) a.net[k] = a.net[0];
  launch_tail_multi(a, max_params, n_nets, s0.blocks, static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

// ndq_fused_fit_run: see include/ndq.h.  Per epoch ONE closure launch (training workgroups + validation workgroups) and
// ONE sums / tail launch (blockIdx.y = network); a trailing validation-only pair closes the call.
int ndq_fused_fit_run(const ndq_fused_fit* f, int n_epochs, const float* const* train_coords, int adam_step,
                      int hist_index, int valid_index, int parity, void* stream) {
  if (!f || !f->launch || f->n_nets < 1 || f->n_nets > 4 || n_epochs < 0 || (n_epochs > 0 && (!train_coords || adam_step <= 0)) ||
      hist_index < 0 || valid_index < 0 || (parity != 0 && parity != 1) || f->track_best < 0 || f->track_best > 2)
    return NDQ_EINVAL;
  const ndq_fused_step& s0 = f->net[0];
  const bool valid = f->valid_coords != nullptr;
  if (!valid && n_epochs == 0) return NDQ_EINVAL;
  if (!s0.best_loss || (n_epochs > 0 && (!s0.loss_hist || !s0.loss_partials || s0.n <= 0 || s0.blocks <= 0))) return NDQ_EINVAL;
  if (valid && (!f->valid_loss_partials || !f->valid_hist || f->valid_n <= 0 || f->valid_blocks <= 0)) return NDQ_EINVAL;
  if (f->track_best == 2 && !valid) return NDQ_EINVAL;
  const float* params[4];
  float* partials[4];
  int max_params = 0;
  for (int k = 0; k < f->n_nets; ++k) {
    const ndq_fused_step& s = f->net[k];
    if (!s.params || s.n_params <= 0 || (f->track_best && !s.best_flat)) return NDQ_EINVAL;
    if (n_epochs > 0 && (!s.partials || !s.grad || !s.adam_m || !s.adam_v || !s.loss_slot)) return NDQ_EINVAL;
    params[k] = s.params;
    partials[k] = s.partials;
    if (s.n_params > max_params) max_params = s.n_params;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int max_lparts = s0.blocks > f->valid_blocks ? s0.blocks : f->valid_blocks;
  // tail of one epoch: e < n_epochs: training epoch e (+ the validation loss of epoch e - 1 if with_valid);
  // e == n_epochs: the trailing validation epoch alone
  auto tail = [&](int e, bool with_train, bool with_valid, int vindex) {
    ReduceTailMultiArgs a{};
    for (int k = 0; k < f->n_nets; ++k) {
      const ndq_fused_step& s = f->net[k];
      ReduceTailArgs& t = a.net[k];
      t.r = Reduce2Args{s.partials, with_train ? s0.blocks : 0, s.n_params, s.grad, 0, s0.loss_partials,
                        with_train ? s0.blocks : 0, s.loss_slot, s0.seed};
      t.t.p = s.params; t.t.g = s.grad; t.t.m = s.adam_m; t.t.v = s.adam_v; t.t.len = s.n_params;
      t.t.lr = s.lr; t.t.b1 = s.beta1; t.t.b2 = s.beta2; t.t.eps = s.eps; t.t.wd = s.weight_decay;
      const int step = adam_step + e;
      t.t.bc1 = with_train ? (float)(1.0 - pow((double)s.beta1, (double)step)) : 1.f;
      t.t.bc2s = with_train ? (float)sqrt(1.0 - pow((double)s.beta2, (double)step)) : 1.f;
      t.t.loss_slots = s.loss_slot; t.t.nb = 1; t.t.loss_hist = s0.loss_hist; t.t.hist_index = hist_index + e;
      t.t.best_loss = s0.best_loss; t.t.parity = parity; t.t.write_scalars = (k == 0) ? 1 : 0;
      // the snapshot follows the training loss (track_best = 1) or the validation loss (2): a tail that does not carry
      // that loss only hands the best value on to the other slot of the ping-pong
      const bool track = (f->track_best == 1 && with_train) || (f->track_best == 2 && with_valid);
      t.t.best_flat = track ? s.best_flat : nullptr;
      t.tail_blocks = 0x7fffffff;
      if (with_valid)
        t.v = ValidArgs{f->valid_loss_partials, f->valid_blocks, f->valid_scale, f->valid_hist, vindex, f->track_best == 2 ? 1 : 0};
    }
    for (int k = f->n_nets; k < 4; ++k) a.net[k] = a.net[0];
    launch_tail_multi(a, max_params, f->n_nets, max_lparts, st);
    parity ^= 1;
    return (int)hipGetLastError();
  };
  bool pull = f->pull_ok != 0 && n_epochs >= 2 && s0.blocks * f->n_nets <= ndq::kPullMaxWork && f->alt_loss_partials &&
              (!valid || f->alt_valid_loss_partials);
  for (int k = 0; k < f->n_nets && pull; ++k)
    pull = f->alt_params[k] && f->alt_m[k] && f->alt_v[k] && f->alt_partials[k];
  if (pull) {
    // ---- pull mode: ONE launch per epoch.  State of epoch e (parameters, moments) lives in buffer set e & 1; launch e
    // writes its partial rows into set e & 1 and, in its prologue, finishes epoch e - 1 from set (e - 1) & 1.
    float* P[2][4]; float* M[2][4]; float* V[2][4]; float* PART[2][4];
    for (int k = 0; k < f->n_nets; ++k) {
      P[0][k] = f->net[k].params; M[0][k] = f->net[k].adam_m; V[0][k] = f->net[k].adam_v; PART[0][k] = f->net[k].partials;
      P[1][k] = f->alt_params[k]; M[1][k] = f->alt_m[k]; V[1][k] = f->alt_v[k]; PART[1][k] = f->alt_partials[k];
    }
    float* LP[2] = {s0.loss_partials, f->alt_loss_partials};
    float* VP[2] = {f->valid_loss_partials, f->alt_valid_loss_partials};
    const int last = valid ? n_epochs : n_epochs - 1;          // index of the last closure launch
    int fin = last & 1;                                        // buffer set holding the state after the last closure launch
    // ---- loop mode: training and validation grid are ONE workgroup each -> runs of up to kLoopMaxLaunches launches of the
    // sequence below become one launch of one workgroup that keeps the state in LDS (csrc/ndq_tail.h: LoopArgs)
    bool loop = f->loop_ok != 0 && f->launch_loop && s0.blocks == 1 && (!valid || f->valid_blocks == 1) &&
                f->n_nets <= ndq::kLoopMaxNets;
    long long stride = 0;
    if (loop) {
      if (!train_coords[0] || !train_coords[1]) return NDQ_EINVAL;
      stride = train_coords[1] - train_coords[0];
      for (int e = 2; e < n_epochs && loop; ++e) {
        if (!train_coords[e]) return NDQ_EINVAL;
        loop = train_coords[e] - train_coords[0] == stride * e;
      }
    }
    if (loop) {
      int seg = 0;
      for (int e0 = 0; e0 <= last; e0 += ndq::kLoopMaxLaunches, ++seg) {
        const int e1 = e0 + ndq::kLoopMaxLaunches <= last + 1 ? e0 + ndq::kLoopMaxLaunches : last + 1;
        const int in = seg & 1, out = in ^ 1;
        ndq::LoopArgs L{};
        L.e0 = e0; L.e1 = e1; L.n_epochs = n_epochs; L.has_valid = valid ? 1 : 0; L.track_best = f->track_best;
        L.hist_index = hist_index; L.valid_index = valid_index; L.parity = parity; L.coord_stride = stride;
        L.lp_in = LP[in]; L.vp_in = VP[in]; L.lp_out = LP[out]; L.vp_out = VP[out];
        L.lscale = s0.seed; L.vscale = f->valid_scale;
        L.loss_hist = s0.loss_hist; L.loss_slot = s0.loss_slot; L.valid_hist = f->valid_hist; L.best_loss = s0.best_loss;
        for (int k = 0; k < f->n_nets; ++k) {
          const ndq_fused_step& s = f->net[k];
          ndq::LoopNet& n = L.net[k];
          n.p_in = P[in][k]; n.m_in = M[in][k]; n.v_in = V[in][k]; n.part_in = PART[in][k];
          n.p_out = P[out][k]; n.m_out = M[out][k]; n.v_out = V[out][k]; n.part_out = PART[out][k];
          n.grad = s.grad; n.best_flat = s.best_flat;
          n.lr = s.lr; n.b1 = s.beta1; n.b2 = s.beta2; n.eps = s.eps; n.wd = s.weight_decay;
          for (int e = e0 > 1 ? e0 : 1; e < e1; ++e) {         // launch e finishes epoch e - 1: Adam step adam_step + e - 1
            const int step = adam_step + e - 1;
            n.bc1[e - e0] = (float)(1.0 - pow((double)s.beta1, (double)step));
            n.bc2s[e - e0] = (float)sqrt(1.0 - pow((double)s.beta2, (double)step));
          }
        }
        int rc = f->launch_loop(train_coords[0], s0.ldc, s0.n, s0.seed, valid ? f->valid_coords : nullptr, f->valid_ldc,
                                valid ? f->valid_n : 0, &L, stream);
        if (rc) return rc;
        fin = out;
      }
      parity ^= (last & 1);
    }
    for (int e = 0; e <= last && !loop; ++e) {
      ndq::PullArgs pa{};
      if (e >= 1) {                                            // prologue: finish training epoch j = e - 1
        const int j = e - 1, in = j & 1, out = e & 1;
        const bool with_valid = valid && j >= 1;                // validation of epoch j - 1, evaluated by launch j
        pa.enabled = 1; pa.n_nets = f->n_nets; pa.nparts = s0.blocks;
        pa.lpart = LP[in]; pa.nlparts = s0.blocks; pa.lscale = s0.seed;
        pa.loss_hist = s0.loss_hist; pa.hist_index = hist_index + j; pa.loss_slot = s0.loss_slot;
        pa.vpart = with_valid ? VP[in] : nullptr; pa.nvparts = f->valid_blocks; pa.vscale = f->valid_scale;
        pa.valid_hist = f->valid_hist; pa.valid_index = valid_index + j - 1; pa.best_on_valid = f->track_best == 2 ? 1 : 0;
        pa.best_loss = s0.best_loss; pa.parity = parity;
        const bool track = f->track_best == 1 || (f->track_best == 2 && with_valid);
        for (int k = 0; k < f->n_nets; ++k) {
          const ndq_fused_step& s = f->net[k];
          ndq::PullNet& n = pa.net[k];
          n.part = PART[in][k];
          n.p_in = P[in][k]; n.m_in = M[in][k]; n.v_in = V[in][k];
          n.p_out = P[out][k]; n.m_out = M[out][k]; n.v_out = V[out][k];
          n.grad = s.grad; n.best_flat = track ? s.best_flat : nullptr; n.len = s.n_params;
          const int step = adam_step + j;
          n.adam = ndq::AdamConsts{s.lr, s.beta1, s.beta2, s.eps, s.weight_decay,
                                   (float)(1.0 - pow((double)s.beta1, (double)step)),
                                   (float)sqrt(1.0 - pow((double)s.beta2, (double)step))};
        }
        parity ^= 1;
      }
      const bool train_part = e < n_epochs, valid_part = valid && e >= 1;
      if (train_part && !train_coords[e]) return NDQ_EINVAL;
      const float* pp[4]; float* qq[4];
      for (int k = 0; k < f->n_nets; ++k) { pp[k] = P[e & 1][k]; qq[k] = PART[e & 1][k]; }
      int rc = f->launch(train_part ? train_coords[e] : nullptr, s0.ldc, train_part ? s0.n : 0, pp, train_part ? qq : nullptr,
                         LP[e & 1], s0.seed, valid_part ? f->valid_coords : nullptr, f->valid_ldc, valid_part ? f->valid_n : 0,
                         VP[e & 1], &pa, stream);
      if (rc) return rc;
    }
    // ---- the call's last launch: an ordinary tail that leaves everything in the primary buffers
    ReduceTailMultiArgs a{};
    for (int k = 0; k < f->n_nets; ++k) {
      const ndq_fused_step& s = f->net[k];
      ReduceTailArgs& t = a.net[k];
      t.t.p = s.params; t.t.g = s.grad; t.t.m = s.adam_m; t.t.v = s.adam_v; t.t.len = s.n_params;
      t.t.lr = s.lr; t.t.b1 = s.beta1; t.t.b2 = s.beta2; t.t.eps = s.eps; t.t.wd = s.weight_decay;
      t.t.loss_slots = s.loss_slot; t.t.nb = 1; t.t.loss_hist = s0.loss_hist; t.t.best_loss = s0.best_loss; t.t.parity = parity;
      t.t.write_scalars = (k == 0) ? 1 : 0;
      t.tail_blocks = 0x7fffffff;
      t.t.p_in = fin ? P[fin][k] : nullptr; t.t.m_in = fin ? M[fin][k] : nullptr; t.t.v_in = fin ? V[fin][k] : nullptr;
      if (valid) {              // validation of the last epoch (evaluated by the trailing launch); no Adam
        t.r = Reduce2Args{PART[fin][k], 0, s.n_params, s.grad, 0, LP[fin], 0, s.loss_slot, s0.seed};
        t.t.bc1 = 1.f; t.t.bc2s = 1.f; t.t.hist_index = hist_index + n_epochs - 1;
        t.t.best_flat = f->track_best == 2 ? s.best_flat : nullptr;
        t.v = ValidArgs{VP[fin], f->valid_blocks, f->valid_scale, f->valid_hist, valid_index + n_epochs - 1,
                        f->track_best == 2 ? 1 : 0};
      } else {                  // the last training epoch's update
        const int step = adam_step + n_epochs - 1;
        t.r = Reduce2Args{PART[fin][k], s0.blocks, s.n_params, s.grad, 0, LP[fin], s0.blocks, s.loss_slot, s0.seed};
        t.t.bc1 = (float)(1.0 - pow((double)s.beta1, (double)step));
        t.t.bc2s = (float)sqrt(1.0 - pow((double)s.beta2, (double)step));
        t.t.hist_index = hist_index + n_epochs - 1;
        t.t.best_flat = f->track_best == 1 ? s.best_flat : nullptr;
      }
    }
    for (int k = f->n_nets; k < 4; ++k) a.net[k] = a.net[0];
    launch_tail_multi(a, max_params, f->n_nets, max_lparts, st);
    return (int)hipGetLastError();
  }
  for (int e = 0; e < n_epochs; ++e) {
    if (!train_coords[e]) return NDQ_EINVAL;
    const bool with_valid = valid && e > 0;
    int rc = f->launch(train_coords[e], s0.ldc, s0.n, params, partials, s0.loss_partials, s0.seed,
                       with_valid ? f->valid_coords : nullptr, f->valid_ldc, with_valid ? f->valid_n : 0,
                       f->valid_loss_partials, nullptr, stream);
    if (rc) return rc;
    rc = tail(e, true, with_valid, valid_index + e - 1);
    if (rc) return rc;
  }
  if (valid) {
    int rc = f->launch(nullptr, 0, 0, params, nullptr, nullptr, 0.f, f->valid_coords, f->valid_ldc, f->valid_n,
                       f->valid_loss_partials, nullptr, stream);
    if (rc) return rc;
    rc = tail(n_epochs, false, true, valid_index + (n_epochs > 0 ? n_epochs - 1 : 0));
    if (rc) return rc;
  }
  return 0;
}

int ndq_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int len, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || len <= 0 || step <= 0) return NDQ_EINVAL;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), params, grad,
                     exp_avg, exp_avg_sq, len, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
  return (int)hipGetLastError();
}

int ndq_sample(const ndq_sampler_desc* desc, unsigned long long seed, unsigned long long draw, unsigned stream_id,
               float* coords, int ldc, void* stream) {
  return ndq::launch_sample(desc, seed, draw, stream_id, coords, ldc, (hipStream_t)stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- one-shot all-reduce
extern "C" {

int ndq_oneshot_create(int rank, int world, int max_len, void** out, unsigned char* handle64) {
  if (!out || !handle64 || world < 1 || world > ndq::kOneshotMaxRanks || rank < 0 || rank >= world || max_len < 1)
    return NDQ_EINVAL;
  ndq::Oneshot* c = new (std::nothrow) ndq::Oneshot();
  if (!c) return NDQ_EINVAL;
  std::memset(c, 0, sizeof(*c));
  int max_blocks = 0;
  const size_t bytes = ndq::oneshot_layout(world, max_len, &c->inbox_bytes, &c->flag_bytes, &max_blocks);
  // fine-grained (uncached) device memory: remote stores of the peers become visible to this device's loads without a
  // kernel boundary
  hipError_t e = hipExtMallocWithFlags(&c->base, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) e = hipExtMallocWithFlags(&c->base, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { delete c; return (int)e; }
  e = hipMemset(c->base, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c->base);
  if (e != hipSuccess) { (void)hipFree(c->base); delete c; return (int)e; }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
  std::memcpy(handle64, &h, 64);
  c->dev.rank = rank; c->dev.world = world; c->dev.max_len = max_len; c->dev.max_blocks = max_blocks;
  c->dev.spin_limit = ndq::kOneshotSpinLimit;
  if (const char* e = std::getenv("NDQ_ONESHOT_SPIN_LIMIT")) c->dev.spin_limit = std::strtoull(e, nullptr, 10);
  c->dev.status = reinterpret_cast<unsigned*>(static_cast<char*>(c->base) + c->inbox_bytes + c->flag_bytes);
  c->step = 0;
  *out = c;
  return 0;
}

// handles: world x 64 bytes, rank-major (every rank's ndq_oneshot_create handle, its own included)
int ndq_oneshot_connect(void* ctx, const unsigned char* handles) {
  ndq::Oneshot* c = static_cast<ndq::Oneshot*>(ctx);
  if (!c || !handles) return NDQ_EINVAL;
  for (int q = 0; q < c->dev.world; ++q) {
    void* p = c->base;
    if (q != c->dev.rank) {
      hipIpcMemHandle_t h;
      std::memcpy(&h, handles + 64 * q, 64);
      hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return (int)e;
      c->peer_base[q] = p;
    }
    c->dev.inbox[q] = static_cast<float*>(p);
    c->dev.flags[q] = reinterpret_cast<unsigned*>(static_cast<char*>(p) + c->inbox_bytes);
  }
  return 0;
}

// ncclAllReduce's signature (sum of fp32 only): what ndq_fused_step.allreduce points at, with .comm = the context
int ndq_oneshot_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
  ndq::Oneshot* c = static_cast<ndq::Oneshot*>(comm);
  if (!c || !send || !recv || dtype != 7 || op != 0 || count == 0 || (long)count > c->dev.max_len) return NDQ_EINVAL;
  const unsigned step = ++c->step;
  const int blocks = ((int)count + ndq::kOneshotChunk - 1) / ndq::kOneshotChunk;
  hipLaunchKernelGGL(ndq::oneshot_allreduce_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), c->dev,
                     static_cast<const float*>(send), static_cast<float*>(recv), (int)count, step);
  return (int)hipGetLastError();
}

// number of flag waits that ran into their spin limit since creation (synchronises the device); 0 = healthy
int ndq_oneshot_status(void* ctx) {
  ndq::Oneshot* c = static_cast<ndq::Oneshot*>(ctx);
  if (!c) return NDQ_EINVAL;
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(&v, c->dev.status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)v;
}

int ndq_oneshot_destroy(void* ctx) {
  ndq::Oneshot* c = static_cast<ndq::Oneshot*>(ctx);
  if (!c) return NDQ_EINVAL;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < c->dev.world; ++q)
    if (q != c->dev.rank && c->peer_base[q]) (void)hipIpcCloseMemHandle(c->peer_base[q]);
  (void)hipFree(c->base);
  delete c;
  return 0;
}

}  // extern "C"
