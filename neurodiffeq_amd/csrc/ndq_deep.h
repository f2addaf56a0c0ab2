// MI355X (gfx950) device + host code for DEEP networks wider than 64 hidden units: L >= 2 hidden layers of one width
// W in 65 .. 512 -- FCNN(hidden_units=(128, 128, 128)), tests/test_pde.py:377's (100, 100), and what the lid-driven-cavity
// notebooks' FCNN(n_hidden_units=256 | 512, n_hidden_layers=1) really builds (networks.py:41: n_hidden_layers + 1 layers).
//
// A W x W weight matrix does not fit a workgroup's LDS as bf16x3 fragment images beyond W = 64 (128 x 128: 2 x 96 KB), and the
// per-wave layer state (16 points x W units x NS streams) leaves the register file at W = 256.  So these shapes run LAYER BY
// LAYER with the pre-activation streams of every hidden layer in HBM (the MI355X has 288 GB and a 256 MB Infinity Cache in
// front of it; 65 536 points x 5 streams x 128 units x 4 B = 168 MB per layer):
//
//   Z_l [NS][NP][HP] fp32, point-major rows of HP = ceil16(W) units -- a lane's 4 consecutive units of one point are ONE
//   16-byte load / store, which is exactly the C/D fragment of the 16x16x4 MFMA (units = rows, points = columns).
//
//   forward   deep_gemm_bf<SRC 0 | 1, EPI 0>   Z_l = W_l sigma-jet(Z_{l-1}) + b_l   (l = 2: sigma-jet of the first layer, from the
//                                               coordinates)
//             deep_head_fwd                     u_s = Wout sigma-jet(Z_L)_s + bout -> output streams
//   reverse   deep_head_bwd                     seeds -> Zbar_L, dWout, db_L, dbout   (one unit per thread, seeds are wave-uniform)
//             deep_wgrad_bf                     dW_l = sum_{s, n} Zbar_l^T sigma-jet(Z_{l-1})   (split over points, partial tiles)
//             deep_gemm_bf<SRC 2, EPI 1 | 2>    Hbar_{l-1} = W_l^T Zbar_l, act-backward in the epilogue -> Zbar_{l-1}, db_{l-1}
//                                               (l = 2: the first layer's dW1 / db1 instead of a store)
//             deep_reduce_all                   every second-stage sum of the sweep in ONE launch (job table, fixed order)
//
// The two per-point GEMMs run on the bf16 matrix core as bf16x3 products (v_mfma_f32_16x16x32_bf16, six significant plane
// products, fp32-class accuracy: csrc/ndq_mlp.h), the layer's weight planes resident in LDS for the whole launch; at W = 128
// they are bound by the instructions they issue (jets, operand splits, products: profiles/r05b_deep_ab.md).  The weight-gradient
// GEMM contracts over (point, stream) PAIRS, since round 5 on the same bf16x3 products (deep_wgrad_bf: a lane's 8 slots of a
// chunk are the next 8 pair values it produces, no transposes); deep_wgrad_gemm (exact-f32 v_mfma_f32_16x16x4_f32, 4 points
// per product) and deep_fwd_gemm / deep_bwd_gemm (exact-f32 per-point GEMMs) are kept behind -DNDQ_DEEP_WGRAD_BF=0 /
// -DNDQ_DEEP_BF16X3=0 as the A/B baselines.
// Reductions are fixed-order everywhere (per-wave partial rows / partial tiles -> deep_reduce_all): bit-reproducible results.
// Reference restated: networks.py:59-70 (forward), neurodiffeq.py:21-34 (diff sweeps), solvers.py:393 (backward).
#pragma once
#include "ndq_mlp.h"

#ifndef NDQ_DEEP_XCD_REMAP
#define NDQ_DEEP_XCD_REMAP 1       // 0: plain blockIdx (A/B runs of the XCD-aware workgroup mapping, xcd_block_id below)
#endif
#ifndef NDQ_DEEP_WGRAD_BF
#define NDQ_DEEP_WGRAD_BF 1        // 0: weight gradients as exact-f32 MFMA products (deep_wgrad_gemm, rounds 4 - 5a); 1: bf16x3 (deep_wgrad_bf)
#endif
#ifndef NDQ_DEEP_HEAD_FUSED
#define NDQ_DEEP_HEAD_FUSED 1      // 0: deep_head_bwd materialises Zbar_L (round 4); 1: its consumers form it from Z_L and the seeds
#endif

namespace ndq {

// WP_: per-layer widths (hidden_units = (128, 64), (256, 128, 64) ...: networks.py:26-66 takes any list), 10 bits per layer,
// layer 1 lowest, for 2 .. 3 layers; 0 = every layer W_ wide.  Everything is laid out for the WIDEST layer (W_ = max): a
// narrower layer's missing units are padding -- zero rows / columns in the weight planes, zero bias, so their
// pre-activations are 0, whatever sigma(0) is meets zero weights downstream and nothing flows back into them -- and only
// the real rows / columns of a gradient are copied out (the flat vector holds the real shapes, torch order).
template <int D_, int FIRST_, unsigned M2_, int LAP_, unsigned M3_, int W_, int L_, int ACT_, int NOUT_, unsigned WP_ = 0u>
struct DeepCfg {
  using SS = Streams<D_, FIRST_, M2_, LAP_, M3_>;
  static_assert(M3_ == 0 || act_has_s4(ACT_), "third-order streams: activations with a stated fourth derivative");
  static_assert(W_ >= 1 && W_ <= 512 && L_ >= 2 && L_ <= 8, "2 .. 8 hidden layers of up to 512 units");
  static_assert(WP_ == 0 || L_ <= 3, "per-layer widths: up to three hidden layers");
  static constexpr int D = D_, W = W_, L = L_, ACT = ACT_, NOUT = NOUT_, NS = SS::NS, NC = NS * NOUT_;
  static constexpr unsigned WP = WP_;
  static constexpr int HP = (W_ + 15) & ~15, NB = HP / 16;
  static constexpr int THREADS = 256, WAVES = 4;
  // real width of hidden layer l (1 .. L); wl(0) = the network's inputs
  static constexpr int wl(int l) { return l == 0 ? D_ : (WP_ == 0 ? W_ : (int)((WP_ >> (10 * (l - 1))) & 1023u)); }
  static constexpr bool widths_ok() {
    int mx = 0;
    for (int l = 1; l <= L_; ++l) { if (wl(l) < 1 || wl(l) > W_) return false; mx = wl(l) > mx ? wl(l) : mx; }
    return mx == W_;
  }
  static_assert(widths_ok(), "per-layer widths: 1 .. W each, the widest equal to W");
  // flat parameter vector, torch order: W1 (w1, D) b1 (w1) | W_l (w_l, w_{l-1}) b_l (w_l), l = 2..L | Wout (NOUT, w_L) bout (NOUT)
  static constexpr int offW1 = 0, offb1 = wl(1) * D_;
  static constexpr int offW(int l) {                                                           // l in 2 .. L + 1
    int o = wl(1) * D_ + wl(1);
    for (int k = 2; k < l; ++k) o += wl(k) * wl(k - 1) + wl(k);
    return o;
  }
  static constexpr int offb(int l) { return offW(l) + wl(l) * wl(l - 1); }
  static constexpr int offWout = offW(L_ + 1), offbout = offWout + NOUT_ * wl(L_);
  static constexpr int P = offbout + NOUT_;
  // output blocks (16 units) a wave accumulates per pass of the per-point GEMMs: JBC * NS fragments of 4 registers
  // (the NB blocks are spread evenly over the passes: balanced(8 blocks, at most 6 per pass) = 4 + 4, not 6 + 2)
  static constexpr int balanced(int most) { const int passes = (NB + most - 1) / most; return (NB + passes - 1) / passes; }
  static constexpr int jbc() { int j = 48 / NS; j = j < 1 ? 1 : j; j = j > 8 ? 8 : j; return balanced(j > NB ? NB : j); }
  static constexpr int JBC = jbc(), NCH = (NB + JBC - 1) / JBC;
  // first-layer gradient accumulators of the last reverse GEMM: JBF blocks per pass
  static constexpr int jbf() { int j = 12 / (D_ + 1); j = j < 1 ? 1 : j; return balanced(j > JBC ? JBC : j); }
  static constexpr int JBF = jbf(), NCHF = (NB + JBF - 1) / JBF;
  // reverse GEMM between hidden layers: its epilogue operands (Z_{l-1}) are prefetched for all JBB blocks
  static constexpr int jbb() { int j = 24 / NS; j = j < 1 ? 1 : j; return balanced(j > JBC ? JBC : j); }
  static constexpr int JBB = jbb(), NCHB = (NB + JBB - 1) / JBB;
  static constexpr int TJ = 4;                                       // weight-gradient GEMM: TJ x TJ blocks per wave
  static constexpr int NT = (NB + TJ - 1) / TJ;
};

// ------------------------------------------------------------------------------------------------ one unit's jets
// h streams of ONE hidden unit from its pre-activation streams z (z[0]: value), csrc/ndq_mlp.h act_forward for a scalar
template <class C>
__device__ __forceinline__ void jet_unit_forward(const real (&z)[C::NS], real (&h)[C::NS], real& t, real& c) {
  using SS = typename C::SS;
  using A = Act<C::ACT>;
  A::fwd(z[0], t, c);
  h[0] = t;
  if constexpr (SS::FIRST) {
    const real s1 = A::s1(t, c);
#pragma unroll
    for (int a = 0; a < C::D; ++a) h[1 + a] = s1 * z[1 + a];
    if constexpr (SS::LAP) {
      const real s2 = A::s2(t, c, s1);
      real q2 = 0.f;
      sfor<C::D>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        if constexpr (SS::in_lap(a)) q2 = rfma(z[1 + a], z[1 + a], q2);
      });
      h[SS::S2] = rfma(s2, q2, s1 * z[SS::S2]);
    } else if constexpr (SS::N2 > 0) {
      const real s2 = A::s2(t, c, s1);
      sfor<SS::N2>([&](auto k_) {
        constexpr int s = SS::S2 + decltype(k_)::value;
        h[s] = rfma(s2 * z[1 + SS::A(s)], z[1 + SS::B(s)], s1 * z[s]);
      });
      if constexpr (SS::N3 > 0) {
        const real s3 = A::s3(t, c, s1);
        sfor<SS::N3>([&](auto k_) {
          constexpr int s = SS::S3 + decltype(k_)::value;
          constexpr int a = SS::T(s, 0), bb = SS::T(s, 1), cc = SS::T(s, 2);
          constexpr int sab = SS::pair_stream(a, bb), sac = SS::pair_stream(a, cc), sbc = SS::pair_stream(bb, cc);
          const real za = z[1 + a], zb = z[1 + bb], zc = z[1 + cc];
          const real mix = rfma(z[sab], zc, rfma(z[sac], zb, z[sbc] * za));
          h[s] = rfma(s3 * za, zb * zc, rfma(s2, mix, s1 * z[s]));
        });
      }
    }
  }
}

// adjoint of jet_unit_forward: g holds hbar on entry, zbar on return (csrc/ndq_mlp.h act_backward for a scalar)
template <class C>
__device__ __forceinline__ void jet_unit_backward(const real (&z)[C::NS], real t, real c, real (&g)[C::NS]) {
  using SS = typename C::SS;
  using A = Act<C::ACT>;
  const real s1 = A::s1(t, c);
  real z0 = s1 * g[0];
  if constexpr (SS::FIRST) {
    const real s2 = A::s2(t, c, s1);
    real za[C::D];
#pragma unroll
    for (int a = 0; a < C::D; ++a) {
      z0 = rfma(s2 * z[1 + a], g[1 + a], z0);
      za[a] = s1 * g[1 + a];
    }
    if constexpr (SS::LAP) {
      const real s3 = A::s3(t, c, s1);
      const real hb = g[SS::S2];
      real q2 = 0.f;
      sfor<C::D>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        if constexpr (SS::in_lap(a)) {
          q2 = rfma(z[1 + a], z[1 + a], q2);
          za[a] = rfma(2.f * s2 * z[1 + a], hb, za[a]);
        }
      });
      z0 = rfma(rfma(s3, q2, s2 * z[SS::S2]), hb, z0);
      g[SS::S2] = s1 * hb;
    } else if constexpr (SS::N2 > 0) {
      const real s3 = A::s3(t, c, s1);
      real zb2[SS::N2];
#pragma unroll
      for (int k = 0; k < SS::N2; ++k) zb2[k] = 0.f;
      if constexpr (SS::N3 > 0) {
        const real s4 = A::s4(t, c, s1);
        sfor<SS::N3>([&](auto k_) {
          constexpr int s = SS::S3 + decltype(k_)::value;
          constexpr int a = SS::T(s, 0), bb = SS::T(s, 1), cc = SS::T(s, 2);
          constexpr int sab = SS::pair_stream(a, bb), sac = SS::pair_stream(a, cc), sbc = SS::pair_stream(bb, cc);
          const real hb = g[s];
          const real zA = z[1 + a], zB = z[1 + bb], zC = z[1 + cc];
          const real zab = z[sab], zac = z[sac], zbc = z[sbc];
          const real mix = rfma(zab, zC, rfma(zac, zB, zbc * zA));
          z0 = rfma(rfma(s4 * zA, zB * zC, rfma(s3, mix, s2 * z[s])), hb, z0);
          za[a] = rfma(rfma(s3 * zB, zC, s2 * zbc), hb, za[a]);
          za[bb] = rfma(rfma(s3 * zA, zC, s2 * zac), hb, za[bb]);
          za[cc] = rfma(rfma(s3 * zA, zB, s2 * zab), hb, za[cc]);
          zb2[sab - SS::S2] = rfma(s2 * zC, hb, zb2[sab - SS::S2]);
          zb2[sac - SS::S2] = rfma(s2 * zB, hb, zb2[sac - SS::S2]);
          zb2[sbc - SS::S2] = rfma(s2 * zA, hb, zb2[sbc - SS::S2]);
          g[s] = s1 * hb;
        });
      }
      sfor<SS::N2>([&](auto k_) {
        constexpr int s = SS::S2 + decltype(k_)::value;
        constexpr int a = SS::A(s), bb = SS::B(s);
        const real hb = g[s];
        z0 = rfma(rfma(s3 * z[1 + a], z[1 + bb], s2 * z[s]), hb, z0);
        za[a] = rfma(s2 * z[1 + bb], hb, za[a]);
        za[bb] = rfma(s2 * z[1 + a], hb, za[bb]);
        g[s] = rfma(s1, hb, zb2[decltype(k_)::value]);
      });
    }
#pragma unroll
    for (int a = 0; a < C::D; ++a) g[1 + a] = za[a];
  }
  g[0] = z0;
}

// pre-activation streams of first-layer unit k at the point x (k >= W: a padding unit, all zero)
template <class C>
__device__ __forceinline__ void first_unit_streams(const real* __restrict__ prm, int k, const real (&x)[C::D], real (&z)[C::NS]) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s) z[s] = 0.f;
  if (k < C::wl(1)) {
    real v = prm[C::offb1 + k];
#pragma unroll
    for (int a = 0; a < C::D; ++a) {
      const real w = prm[C::offW1 + k * C::D + a];
      v = rfma(w, x[a], v);
      if constexpr (C::SS::FIRST) z[1 + a] = w;
    }
    z[0] = v;
  }
}

// XCD-aware workgroup id.  The dispatcher places block b on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"; a
// pure speed assumption: any placement computes the same thing) and every XCD has its own 4 MiB L2.  The kernels below have
// GROUPS of workgroups that read the same rows of a Z tensor at the same time -- the NT x NT output tiles of one point slice
// in the weight-gradient GEMM, the NCH output chunks of one stripe of tiles in the per-point GEMMs.  With the plain id those
// neighbours sit on different XCDs and every one of them pulls the rows from HBM (measured, round 4: 545 MB per
// weight-gradient launch at 128 x 3 against 268 MB of operands).  The remapped id hands every XCD a CONTIGUOUS range of
// virtual ids, so a group lands on one XCD, is dispatched within a few ids of each other, and its common rows come out of
// that XCD's L2.  Bijective for any grid size (q = n / 8, r = n % 8: XCD x owns q + (x < r) ids).
__device__ __forceinline__ int xcd_block_id() {
#if NDQ_DEEP_XCD_REMAP
  const int b = (int)blockIdx.x, n = (int)gridDim.x, x = b & 7, q = n >> 3, r = n & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
#else
  return (int)blockIdx.x;
#endif
}

struct DeepArgs {
  const real* coords;     // [D][ldc]
  const real* prm;        // [P]
  int n, np, ldc;         // points, points rounded up to whole tiles, leading dimension of coords
  const real* wmat;       // padded weight matrix of the layer [HP][HP] (forward: W_l, reverse: W_l^T)
  const real* bias;       // forward: b_l [W]
  const real* zin;        // forward: Z_{l-1};  reverse: Zbar_l
  const real* zprev;      // reverse: Z_{l-1} (act-backward in the epilogue);  weight gradient: Z_{l-1}
  real* zout;             // forward: Z_l;  reverse: Zbar_{l-1}
  real* pb;               // reverse: partial rows of db_{l-1} [stripes][HP]
  real* pw1;              // reverse into the first layer: partial rows of dW1 [stripes][HP][D]
  real* pw;               // weight gradient: partial tiles [KS][HP][HP]
  const void* wpl;        // bf16x3 route: the layer's weight planes in MFMA A-operand order (deep_prep_planes)
  // head fused into its consumers (NDQ_DEEP_HEAD_FUSED): Zbar_L = act-backward(Z_L, Wout^T seeds) is never written -- zin is
  // Z_L, and the kernels form Zbar_L from it and the seeds as they load
  const real* gbar;       // [NS][NOUT][ldj] seeds of the output streams
  int ldj;
  real* pwo;              // partial rows of dWout [rows][NOUT][HP]   (weight-gradient GEMM of layer L, tiles of column block 0)
  real* pbh;              // partial rows of db_L  [rows][HP]
  real* pbo;              // partial rows of dbout [rows][NOUT]
  real* jets;             // EPI 3 (last forward GEMM with the head folded in): output streams [NS][NOUT][ldj]
  int ow;                 // forward GEMMs: real width of the layer being computed (its bias has that many entries; DeepCfg::WP)
};

// ------------------------------------------------------------------------------------------------ per-point GEMMs
// FIRSTIN: the layer input is the first layer's sigma-jet, evaluated from the coordinates; else sigma-jet(zin).
// Wave w of the grid owns output chunk w % NCH (JBC blocks of 16 units) and walks the 16-point tiles w / NCH, + stripes.
#ifndef NDQ_DEEP_OCC
#define NDQ_DEEP_OCC 1             // workgroups per CU the per-point GEMMs are compiled for (experiments: 2 = 256 registers per wave)
#endif
template <class C, bool FIRSTIN>
__global__ __launch_bounds__(C::THREADS, NDQ_DEEP_OCC) void deep_fwd_gemm(DeepArgs a) {
  const int lane = threadIdx.x & 63, p = lane & 15, kg = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ch = gw % C::NCH, stripe = gw / C::NCH, nstripes = nw / C::NCH;
  if (stripe >= nstripes) return;
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
  // this wave's output units never change: their biases are loaded once (rows 4 kg + r of block b, column p)
  real4 bs[C::JBC];
#pragma unroll
  for (int jb = 0; jb < C::JBC; ++jb) {
    const int j0 = 16 * (ch * C::JBC + jb) + 4 * kg;
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[jb][r] = (j0 + r < a.ow) ? a.bias[j0 + r < a.ow ? j0 + r : 0] : 0.f;
  }
  for (int tile = stripe; tile < ntiles; tile += nstripes) {
    const int n = tile * 16 + p;
    const int nn = n < a.n ? n : a.n - 1;
    real x[C::D];
    if constexpr (FIRSTIN) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) x[d] = a.coords[(size_t)d * a.ldc + nn];
    }
    real4 acc[C::NS][C::JBC];
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int jb = 0; jb < C::JBC; ++jb) acc[s][jb] = real4{0.f, 0.f, 0.f, 0.f};
    // the next contraction step's operands are fetched before the current step's MFMAs are issued
    real4 zn[FIRSTIN ? 1 : C::NS], wn[C::JBC];
    real f1w[FIRSTIN ? 4 : 1][C::D], f1b[FIRSTIN ? 4 : 1];       // FIRSTIN: first-layer rows of the step's 4 contraction units
    auto fetch = [&](int c16) {
      const int k0 = 16 * c16 + 4 * kg;
      if constexpr (!FIRSTIN) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) zn[s] = *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + k0);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool ok = k0 + t < C::wl(1);
          f1b[t] = ok ? a.prm[C::offb1 + (ok ? k0 + t : 0)] : 0.f;
#pragma unroll
          for (int d = 0; d < C::D; ++d) f1w[t][d] = ok ? a.prm[C::offW1 + (ok ? k0 + t : 0) * C::D + d] : 0.f;
        }
      }
#pragma unroll
      for (int jb = 0; jb < C::JBC; ++jb) {
        const int b = ch * C::JBC + jb;
        wn[jb] = *reinterpret_cast<const real4*>(a.wmat + (size_t)(16 * (b < C::NB ? b : 0) + p) * C::HP + k0);
      }
    };
    fetch(0);
    for (int c16 = 0; c16 < C::NB; ++c16) {
      real4 hh[C::NS], w4[C::JBC];
#pragma unroll
      for (int jb = 0; jb < C::JBC; ++jb) w4[jb] = wn[jb];
      {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          real z[C::NS], h[C::NS], tt, cc;
          if constexpr (FIRSTIN) {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = 0.f;
            real zv = f1b[t];
#pragma unroll
            for (int d = 0; d < C::D; ++d) {
              zv = rfma(f1w[t][d], x[d], zv);
              if constexpr (C::SS::FIRST) z[1 + d] = f1w[t][d];
            }
            z[0] = zv;
          } else {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = zn[s][t];
          }
          jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
          for (int s = 0; s < C::NS; ++s) hh[s][t] = h[s];
        }
      }
      if (c16 + 1 < C::NB) fetch(c16 + 1);
      __builtin_amdgcn_sched_barrier(0);       // the loads stay ABOVE the MFMAs (the scheduler sinks them below otherwise)
#pragma unroll
      for (int jb = 0; jb < C::JBC; ++jb) {
        const int b = ch * C::JBC + jb;
        if (b < C::NB) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][jb] = mfma16x16x4(w4[jb][t], hh[s][t], acc[s][jb]);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < C::JBC; ++jb) {
      const int b = ch * C::JBC + jb;
      if (b < C::NB) {
        const int j0 = 16 * b + 4 * kg;                    // rows 4 kg + r of the block, column p
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][jb][r] += bs[jb][r];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) *reinterpret_cast<real4*>(a.zout + s * sstride + (size_t)n * C::HP + j0) = acc[s][jb];
      }
    }
  }
}

// Hbar_{l-1} = W_l^T Zbar_l, then the act-backward of layer l - 1 in the epilogue.  TOFIRST (l == 2): layer 1's streams come
// from the coordinates and what leaves is dW1 / db1 (per-wave partial rows); else Zbar_{l-1} is stored and db_{l-1} summed.
template <class C, bool TOFIRST>
__global__ __launch_bounds__(C::THREADS, NDQ_DEEP_OCC) void deep_bwd_gemm(DeepArgs a) {
  constexpr int JB = TOFIRST ? C::JBF : C::JBB, NCH = TOFIRST ? C::NCHF : C::NCHB;
  const int lane = threadIdx.x & 63, p = lane & 15, kg = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ch = gw % NCH, stripe = gw / NCH, nstripes = nw / NCH;
  if (stripe >= nstripes) return;
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
  // TOFIRST: the first layer's rows of this wave's output units (they never change), loaded once
  real u1w[TOFIRST ? JB : 1][4][C::D], u1b[TOFIRST ? JB : 1][4];
  if constexpr (TOFIRST) {
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * (ch * JB + jb) + 4 * kg + r;
        const bool ok = k < C::wl(1);
        u1b[jb][r] = ok ? a.prm[C::offb1 + (ok ? k : 0)] : 0.f;
#pragma unroll
        for (int d = 0; d < C::D; ++d) u1w[jb][r][d] = ok ? a.prm[C::offW1 + (ok ? k : 0) * C::D + d] : 0.f;
      }
  }
  real gb[JB][4], gw1[TOFIRST ? JB : 1][4][C::D];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gb[jb][r] = 0.f;
      if constexpr (TOFIRST) {
#pragma unroll
        for (int d = 0; d < C::D; ++d) gw1[jb][r][d] = 0.f;
      }
    }
  for (int tile = stripe; tile < ntiles; tile += nstripes) {
    const int n = tile * 16 + p;
    const int nn = n < a.n ? n : a.n - 1;
    real x[C::D];
    if constexpr (TOFIRST) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) x[d] = a.coords[(size_t)d * a.ldc + nn];
    }
    real4 acc[C::NS][JB];
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) acc[s][jb] = real4{0.f, 0.f, 0.f, 0.f};
    // the epilogue's operands (Z_{l-1} of the tile's output units) are requested up front: they arrive under the MFMAs
    real4 zpre[TOFIRST ? 1 : JB][C::NS];
    if constexpr (!TOFIRST) {
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int b = ch * JB + jb;
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
          zpre[jb][s] = *reinterpret_cast<const real4*>(a.zprev + s * sstride + (size_t)n * C::HP + 16 * (b < C::NB ? b : 0) + 4 * kg);
      }
    }
    real4 zn[C::NS], wn[JB];
    auto fetch = [&](int c16) {
      const int k0 = 16 * c16 + 4 * kg;
#pragma unroll
      for (int s = 0; s < C::NS; ++s) zn[s] = *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + k0);
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int b = ch * JB + jb;
        wn[jb] = *reinterpret_cast<const real4*>(a.wmat + (size_t)(16 * (b < C::NB ? b : 0) + p) * C::HP + k0);
      }
    };
    fetch(0);
    for (int c16 = 0; c16 < C::NB; ++c16) {
      real4 zb[C::NS], w4[JB];
#pragma unroll
      for (int s = 0; s < C::NS; ++s) zb[s] = zn[s];
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) w4[jb] = wn[jb];
      if (c16 + 1 < C::NB) fetch(c16 + 1);
      __builtin_amdgcn_sched_barrier(0);       // the loads stay ABOVE the MFMAs (the scheduler sinks them below otherwise)
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int b = ch * JB + jb;
        if (b < C::NB) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][jb] = mfma16x16x4(w4[jb][t], zb[s][t], acc[s][jb]);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const int b = ch * JB + jb;
      if (b < C::NB) {
        const int j0 = 16 * b + 4 * kg;
        real4 out[C::NS];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real z[C::NS], g[C::NS], tt, cc;
          if constexpr (TOFIRST) {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = 0.f;
            real zv = u1b[jb][r];
#pragma unroll
            for (int d = 0; d < C::D; ++d) {
              zv = rfma(u1w[jb][r][d], x[d], zv);
              if constexpr (C::SS::FIRST) z[1 + d] = u1w[jb][r][d];
            }
            z[0] = zv;
          } else {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = zpre[jb][s][r];
          }
          Act<C::ACT>::fwd(z[0], tt, cc);
#pragma unroll
          for (int s = 0; s < C::NS; ++s) g[s] = acc[s][jb][r];
          jet_unit_backward<C>(z, tt, cc, g);
          gb[jb][r] += g[0];
          if constexpr (TOFIRST) {
#pragma unroll
            for (int d = 0; d < C::D; ++d) gw1[jb][r][d] += C::SS::FIRST ? rfma(g[0], x[d], g[C::SS::FIRST ? 1 + d : 0]) : g[0] * x[d];
          } else {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) out[s][r] = g[s];
          }
        }
        if constexpr (!TOFIRST) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s) *reinterpret_cast<real4*>(a.zout + s * sstride + (size_t)n * C::HP + j0) = out[s];
        }
      }
    }
  }
  // the 16 points of a tile are one DPP row: fixed-order sums, lanes p == 0 write this wave's partial row
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) {
    const int b = ch * JB + jb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real v = point_sum(gb[jb][r]);
      if (b < C::NB && p == 0) a.pb[(size_t)stripe * C::HP + 16 * b + 4 * kg + r] = v;
      if constexpr (TOFIRST) {
#pragma unroll
        for (int d = 0; d < C::D; ++d) {
          const real u = point_sum(gw1[jb][r][d]);
          if (b < C::NB && p == 0) a.pw1[((size_t)stripe * C::HP + 16 * b + 4 * kg + r) * C::D + d] = u;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ per-point GEMMs, bf16x3
// The same products as deep_fwd_gemm / deep_bwd_gemm on the REAL matrix core: v_mfma_f32_16x16x32_bf16 with 3-way split
// operands (x = x0 + x1 + x2, six significant plane products accumulated in fp32, smallest first: fp32-class accuracy,
// csrc/ndq_mlp.h) -- 6 MFMAs of 16 cycles per 32-deep contraction chunk against 8 exact-f32 MFMAs of 32 cycles, and they
// overlap with the VALU work of the operand prologue (the f32 MFMA shares the VALU datapath).
//   weights   deep_prep writes every hidden matrix (and its transpose) as bf16x3 planes in MFMA A-operand order:
//             element ((b * NCK + c) * 3 + plane) * 64 + lane = the 8 bf16 of row 16 b + (lane & 15), columns
//             32 c + 8 (lane >> 4) ... + 7.  A workgroup keeps the planes of ITS output chunk (JBR blocks x all NCK
//             contraction chunks, <= 96 KB) resident in LDS for the whole launch: staged once, read by all four waves.
//   operand   a lane's 8 contraction units of its point are two 16-byte loads per stream; sigma-jet (SRC 0 / 1) and the
//             3-way split of step c + 1 are computed while the MFMAs of step c are in flight.
// SRC: 0 = sigma-jet of the first layer, from the coordinates; 1 = sigma-jet(zin); 2 = zin as it is (reverse GEMM);
//      3 = reverse GEMM of the LAST hidden layer with the head folded in: zin is Z_L, the operand Zbar_L =
//          act-backward(Z_L, Wout^T seeds) is formed in registers (no deep_head_bwd pass, no Zbar_L tensor in HBM).
// EPI: 0 = + bias, store Z_l; 1 = act-backward with Z_{l-1}, store Zbar_{l-1}, sum db_{l-1}; 2 = act-backward into the
//      first layer (dW1, db1 partial rows); 3 = EPI 0 of the LAST hidden layer with the output layer folded in (one chunk
//      holds all units, NCH == 1: W <= 128): u_s = Wout sigma-jet(Z_L)_s + bout from the accumulators -- no deep_head_fwd
//      pass re-reading Z_L.
#ifndef NDQ_DEEP_BF_LDS_KB
#define NDQ_DEEP_BF_LDS_KB 96      // LDS a workgroup spends on its resident weight planes (48: two workgroups per CU)
#endif
constexpr int kDeepBfOcc = NDQ_DEEP_BF_LDS_KB <= 48 ? 2 : 1;
// Waves per workgroup of the per-point GEMMs.  The resident weight planes allow ONE workgroup per CU, i.e. one wave per SIMD
// with 4 waves.  8 waves sharing the planes (two per SIMD: better VALU throughput on the dependent chains of the jets, as
// the weight-gradient kernel shows between one and two workgroups per CU) were measured in round 5 and lose: the variants
// need 312 - 472 registers, the 256 of an 8-wave workgroup cost 184 - 556 bytes of scratch per lane (128 x 3: 611 against
// 502 us per step, 256 x 2: 1 021 against 1 003; profiles/r05n_bf_waves_ab.jsonl).  Kept as a build option.
#ifndef NDQ_DEEP_BF_WAVES
#define NDQ_DEEP_BF_WAVES 4
#endif
constexpr int kDeepBfWaves = NDQ_DEEP_BF_WAVES, kDeepBfThreads = 64 * kDeepBfWaves;
template <class C, int EPI> constexpr int deep_bf_jb() {
  constexpr int NCK = (C::HP + 31) / 32;
  int j = (NDQ_DEEP_BF_LDS_KB / 3) / NCK;                  // planes of the chunk: 3 KB per (block, contraction chunk)
  const int most = (EPI == 0 || EPI == 3) ? C::JBC : (EPI == 1 ? C::JBB : C::JBF);
  j = j > most ? most : j;
  j = j < 1 ? 1 : j;
  return C::balanced(j);
}

template <class C, int SRC, int EPI>
__global__ __launch_bounds__(kDeepBfThreads, kDeepBfOcc) void deep_gemm_bf(DeepArgs a) {
  constexpr int JB = deep_bf_jb<C, EPI>(), NCH = (C::NB + JB - 1) / JB, NCK = (C::HP + 31) / 32, NS = C::NS;
  constexpr int kPlanes = SRC >= 2 ? (NDQ_HBAR_NPROD == 6 ? 3 : 2) : (NDQ_FWD_NPROD == 6 ? 3 : 2);    // planes of the per-point operand
  extern __shared__ __attribute__((aligned(16))) bf16x8 wl[];          // [JB][NCK][3][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kg = lane >> 4;
  const int vid = xcd_block_id();                          // (the NCH chunks of a stripe share its operand rows: one XCD)
  const int ch = vid % NCH, bstripe = vid / NCH, nbstripes = gridDim.x / NCH;
  if (bstripe >= nbstripes) return;                        // (whole workgroups: no barrier is missed)
  // ---- this workgroup's weight planes -> LDS, once.  Eight 16-byte loads of a thread in flight per pass (round 5): the plain
  // copy loop compiled to "global_load_dwordx4, s_waitcnt vmcnt(0), ds_write_b128" per element -- 24 dependent round trips for
  // the 96 KB of a 128-unit layer, ~10 us of every launch of this kernel.
  {
    const real4* src = static_cast<const real4*>(a.wpl);            // (a bf16x8 element is 16 bytes)
    real4* dst = reinterpret_cast<real4*>(wl);
    constexpr int TOT = JB * NCK * 3 * 64, PER = NCK * 3 * 64, KP = 8;
    for (int e0 = 0; e0 < TOT; e0 += KP * kDeepBfThreads) {
      real4 v[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int e = e0 + threadIdx.x + k * kDeepBfThreads, ec = e < TOT ? e : TOT - 1;
        const int b = ch * JB + ec / PER;
        v[k] = src[(size_t)(b < C::NB ? b : C::NB - 1) * PER + ec % PER];
      }
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int e = e0 + threadIdx.x + k * kDeepBfThreads;
        if (e < TOT) dst[e] = (ch * JB + e / PER < C::NB) ? v[k] : real4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
  // per-wave constants of the epilogue
  constexpr bool STORE = EPI == 0 || EPI == 3;             // forward GEMM: + bias, store Z_l
  static_assert(EPI != 3 || NCH == 1, "the folded output layer needs all units of a point in one workgroup");
  real4 bs[STORE ? JB : 1];
  real u1w[EPI == 2 ? JB : 1][4][C::D], u1b[EPI == 2 ? JB : 1][4];
  real gb[!STORE ? JB : 1][4], gw1[EPI == 2 ? JB : 1][4][C::D];
  real wout[EPI == 3 ? JB : 1][4][C::NOUT];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) {
    const int j0 = 16 * (ch * JB + jb) + 4 * kg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (real widths: STORE / EPI 3 -- the layer being computed, a.ow (EPI 3: the last one); EPI 2 -- the first layer)
      const bool ok = j0 + r < (STORE ? a.ow : C::wl(1));
      if constexpr (STORE) bs[jb][r] = ok ? a.bias[ok ? j0 + r : 0] : 0.f;
      if constexpr (EPI == 3) {
#pragma unroll
        for (int o = 0; o < C::NOUT; ++o) wout[jb][r][o] = ok ? a.prm[C::offWout + o * C::wl(C::L) + (ok ? j0 + r : 0)] : 0.f;
      }
      if constexpr (!STORE) gb[jb][r] = 0.f;
      if constexpr (EPI == 2) {
        u1b[jb][r] = ok ? a.prm[C::offb1 + (ok ? j0 + r : 0)] : 0.f;
#pragma unroll
        for (int d = 0; d < C::D; ++d) {
          u1w[jb][r][d] = ok ? a.prm[C::offW1 + (ok ? j0 + r : 0) * C::D + d] : 0.f;
          gw1[jb][r][d] = 0.f;
        }
      }
    }
  }
  __syncthreads();
  // ---- carried from one tile to the next (round 5): the rows of contraction step 0, the point's coordinates and seeds of the
  // NEXT tile are requested during the last contraction step of the current one -- the loads at the top of the tile loop were one
  // exposed HBM round trip per tile (a wave has its SIMD to itself)
  real4 vlon[SRC == 0 ? 1 : NS], vhin[SRC == 0 ? 1 : NS];
  real xn[(SRC == 0 || EPI == 2) ? C::D : 1], gsn[SRC == 3 ? C::NC : 1];
  auto fetch_next = [&](int t) {
    const int nr = t * 16 + p, nnr = nr < a.n ? nr : a.n - 1;
    if constexpr (SRC == 0 || EPI == 2) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nnr];
    }
    if constexpr (SRC == 3) {
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) {
        const real v = a.gbar[(size_t)cc * a.ldj + nnr];
        gsn[cc] = nr < a.n ? v : 0.f;                       // (padding points: zero -> Zbar_L = 0)
      }
    }
    if constexpr (SRC != 0) {
      const real4 zero = real4{0.f, 0.f, 0.f, 0.f};
      const int k0 = 8 * kg;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const real* row = a.zin + s * sstride + (size_t)nr * C::HP;
        vlon[s] = k0 < C::HP ? *reinterpret_cast<const real4*>(row + k0) : zero;
        vhin[s] = k0 + 4 < C::HP ? *reinterpret_cast<const real4*>(row + k0 + 4) : zero;
      }
    }
  };
  {
    const int t0 = bstripe * kDeepBfWaves + wave;
    if (t0 < ntiles) fetch_next(t0);
  }
  for (int tile = bstripe * kDeepBfWaves + wave; tile < ntiles; tile += nbstripes * kDeepBfWaves) {
    const int n = tile * 16 + p;
    real x[(SRC == 0 || EPI == 2) ? C::D : 1];
    if constexpr (SRC == 0 || EPI == 2) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) x[d] = xn[d];
    }
    real4 acc[NS][JB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) acc[s][jb] = real4{0.f, 0.f, 0.f, 0.f};
    real4 zpre[EPI == 1 ? JB : 1][NS];
    if constexpr (EPI == 1) {
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int b = ch * JB + jb;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          zpre[jb][s] = *reinterpret_cast<const real4*>(a.zprev + s * sstride + (size_t)n * C::HP + 16 * (b < C::NB ? b : 0) + 4 * kg);
      }
    }
    // operand of contraction step c: this lane's units 32 c + 8 kg ... + 7 of its point
    real4 vlo[SRC == 0 ? 1 : NS], vhi[SRC == 0 ? 1 : NS];
    real f1w[SRC == 0 ? 8 : 1][C::D], f1b[SRC == 0 ? 8 : 1];
    real wo8[SRC == 3 ? 8 : 1][C::NOUT], gs[SRC == 3 ? C::NC : 1];
    if constexpr (SRC == 3) {                              // the point's seeds
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) gs[cc] = gsn[cc];
    }
    // rows: also load this lane's rows of Z (false for step 0: they were prefetched into vlon / vhin)
    auto fetch = [&](int c, bool rows) {
      const int k0 = 32 * c + 8 * kg;
      if constexpr (SRC == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = k0 + e < C::wl(C::L);
#pragma unroll
          for (int o = 0; o < C::NOUT; ++o) wo8[e][o] = ok ? a.prm[C::offWout + o * C::wl(C::L) + (ok ? k0 + e : 0)] : 0.f;
        }
      }
      if constexpr (SRC == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = k0 + e < C::wl(1);
          f1b[e] = ok ? a.prm[C::offb1 + (ok ? k0 + e : 0)] : 0.f;
#pragma unroll
          for (int d = 0; d < C::D; ++d) f1w[e][d] = ok ? a.prm[C::offW1 + (ok ? k0 + e : 0) * C::D + d] : 0.f;
        }
      } else if (rows) {
        const real4 zero = real4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const real* row = a.zin + s * sstride + (size_t)n * C::HP;
          vlo[s] = k0 < C::HP ? *reinterpret_cast<const real4*>(row + k0) : zero;
          vhi[s] = k0 + 4 < C::HP ? *reinterpret_cast<const real4*>(row + k0 + 4) : zero;
        }
      }
    };
    bf16x8 pl[NS][3];
    auto planes = [&]() {
      real4 hlo[NS], hhi[NS];
      if constexpr (SRC == 2) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { hlo[s] = vlo[s]; hhi[s] = vhi[s]; }
      } else if constexpr (SRC == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          real z[NS], g[NS], tt, cc;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            z[s] = e < 4 ? vlo[s][e & 3] : vhi[s][e & 3];
            real v = 0.f;
#pragma unroll
            for (int o = 0; o < C::NOUT; ++o) v = rfma(wo8[e][o], gs[s * C::NOUT + o], v);
            g[s] = v;
          }
          Act<C::ACT>::fwd(z[0], tt, cc);
          jet_unit_backward<C>(z, tt, cc, g);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (e < 4) hlo[s][e & 3] = g[s];
            else hhi[s][e & 3] = g[s];
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          real z[NS], h[NS], tt, cc;
          if constexpr (SRC == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) z[s] = 0.f;
            real zv = f1b[e];
#pragma unroll
            for (int d = 0; d < C::D; ++d) {
              zv = rfma(f1w[e][d], x[d], zv);
              if constexpr (C::SS::FIRST) z[1 + d] = f1w[e][d];
            }
            z[0] = zv;
          } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) z[s] = e < 4 ? vlo[s][e & 3] : vhi[s][e & 3];
          }
          jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (e < 4) hlo[s][e & 3] = h[s];
            else hhi[s][e & 3] = h[s];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) split3<kPlanes>(hlo[s], hhi[s], pl[s]);
    };
    fetch(0, false);
    if constexpr (SRC != 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { vlo[s] = vlon[s]; vhi[s] = vhin[s]; }
    }
    planes();
    for (int c = 0; c < NCK; ++c) {
      if (c + 1 < NCK) fetch(c + 1, true);
      else {                                               // last step: the next tile's first rows / point data
        const int tn = tile + nbstripes * kDeepBfWaves;
        fetch_next(tn < ntiles ? tn : tile);
      }
      __builtin_amdgcn_sched_barrier(0);                   // the loads stay above the MFMAs
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const bf16x8* w = wl + ((jb * NCK + c) * 3) * 64 + lane;
        const bf16x8 a0 = w[0], a1 = w[64], a2 = w[128];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < NS; ++s)                                                             \
      acc[s][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][K], acc[s][jb], 0, 0, 0);
        // forward layers: all six plane products (they make the derivative streams); reverse layers: NDQ_HBAR_NPROD of them,
        // their per-point operand split into two planes (csrc/ndq_mlp.h, profiles/r06_headline_ab.md)
        if constexpr (SRC >= 2) { NDQ_PRODUCTS_BWD(NDQ_T) } else { NDQ_PRODUCTS_FWD(NDQ_T) }
#undef NDQ_T
      }
      if (c + 1 < NCK) planes();                           // VALU work of the next step, under the MFMAs in flight
    }
    // ---- epilogue
    real uo[EPI == 3 ? C::NC : 1];
    if constexpr (EPI == 3) {
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) uo[cc] = 0.f;
    }
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const int b = ch * JB + jb;
      if (b < C::NB) {
        const int j0 = 16 * b + 4 * kg;
        if constexpr (STORE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[0][jb][r] += bs[jb][r];
#pragma unroll
          for (int s = 0; s < NS; ++s) *reinterpret_cast<real4*>(a.zout + s * sstride + (size_t)n * C::HP + j0) = acc[s][jb];
          if constexpr (EPI == 3) {                        // output layer on the accumulators (deep_head_fwd's arithmetic)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              real z[NS], h[NS], tt, cc;
#pragma unroll
              for (int s = 0; s < NS; ++s) z[s] = acc[s][jb][r];
              jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
              for (int o = 0; o < C::NOUT; ++o)
#pragma unroll
                for (int s = 0; s < NS; ++s) uo[s * C::NOUT + o] = rfma(wout[jb][r][o], h[s], uo[s * C::NOUT + o]);
            }
          }
        } else {
          real4 out[NS];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            real z[NS], g[NS], tt, cc;
            if constexpr (EPI == 2) {
#pragma unroll
              for (int s = 0; s < NS; ++s) z[s] = 0.f;
              real zv = u1b[jb][r];
#pragma unroll
              for (int d = 0; d < C::D; ++d) {
                zv = rfma(u1w[jb][r][d], x[d], zv);
                if constexpr (C::SS::FIRST) z[1 + d] = u1w[jb][r][d];
              }
              z[0] = zv;
            } else {
#pragma unroll
              for (int s = 0; s < NS; ++s) z[s] = zpre[jb][s][r];
            }
            Act<C::ACT>::fwd(z[0], tt, cc);
#pragma unroll
            for (int s = 0; s < NS; ++s) g[s] = acc[s][jb][r];
            jet_unit_backward<C>(z, tt, cc, g);
            gb[jb][r] += g[0];
            if constexpr (EPI == 2) {
#pragma unroll
              for (int d = 0; d < C::D; ++d) gw1[jb][r][d] += C::SS::FIRST ? rfma(g[0], x[d], g[C::SS::FIRST ? 1 + d : 0]) : g[0] * x[d];
            } else {
#pragma unroll
              for (int s = 0; s < NS; ++s) out[s][r] = g[s];
            }
          }
          if constexpr (EPI == 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<real4*>(a.zout + s * sstride + (size_t)n * C::HP + j0) = out[s];
          }
        }
      }
    }
    if constexpr (EPI == 3) {
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) {
        real v = quad_sum(uo[cc]);
        if (cc < C::NOUT) v += a.prm[C::offbout + cc];
        if (n < a.n && (cc & 3) == kg) a.jets[(size_t)cc * a.ldj + n] = v;
      }
    }
  }
  if constexpr (!STORE) {
    // partial rows: one per WAVE (row index = workgroup stripe x 4 + wave), every chunk fills its own units
    const int row = bstripe * kDeepBfWaves + wave;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const int b = ch * JB + jb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real v = point_sum(gb[jb][r]);
        if (b < C::NB && p == 0) a.pb[(size_t)row * C::HP + 16 * b + 4 * kg + r] = v;
        if constexpr (EPI == 2) {
#pragma unroll
          for (int d = 0; d < C::D; ++d) {
            const real u = point_sum(gw1[jb][r][d]);
            if (b < C::NB && p == 0) a.pw1[((size_t)row * C::HP + 16 * b + 4 * kg + r) * C::D + d] = u;
          }
        }
      }
    }
  }
}
template <class C, int EPI> constexpr size_t deep_bf_lds_bytes() { return (size_t)deep_bf_jb<C, EPI>() * ((C::HP + 31) / 32) * 3 * 64 * 16; }

// bf16x3 planes of the padded hidden matrices in MFMA A-operand order (see deep_gemm_bf): wpl[l - 2] = W_l, wtl[l - 2] = W_l^T
template <class C>
__global__ __launch_bounds__(256) void deep_prep_planes(const real* __restrict__ prm, bf16x8* __restrict__ wpl, bf16x8* __restrict__ wtl) {
  constexpr int NCK = (C::HP + 31) / 32, PER = C::NB * NCK * 64;     // lane-elements per matrix and plane set
  const int l = 2 + blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * PER; e += gridDim.x * blockDim.x) {
    const bool tr = e >= PER;
    const int f = tr ? e - PER : e;
    const int lane = f & 63, c = (f >> 6) % NCK, b = (f >> 6) / NCK;
    const int row = 16 * b + (lane & 15), k0 = 32 * c + 8 * (lane >> 4);
    real4 lo, hi;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = k0 + t;
      // element (row, k) of W_l, or of its transpose
      const int j = tr ? k : row, kk = tr ? row : k;
      const real v = (j < C::wl(l) && kk < C::wl(l - 1)) ? prm[C::offW(l) + j * C::wl(l - 1) + kk] : 0.f;
      if (t < 4) lo[t] = v; else hi[t - 4] = v;
    }
    bf16x8 pl[3];
    split3(lo, hi, pl);
    bf16x8* dst = (tr ? wtl : wpl) + (size_t)(l - 2) * (C::NB * NCK * 3 * 64);
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[((size_t)(b * NCK + c) * 3 + q) * 64 + lane] = pl[q];
  }
}

// dW_l[j][k] = sum_{s, n} Zbar_l[s][n][j] sigma-jet(Z_{l-1})[s][n][k]: the contraction runs over points, 4 per MFMA
// (contraction slot kg of a lane = point 4 g + kg of point group g).
// A workgroup owns a 64 x 64 tile of dW_l (tile blockIdx % NT^2) and the slice blockIdx / NT^2 of the point groups, which
// its four waves walk interleaved; their accumulators are added in wave order through LDS, so one partial tile leaves per
// workgroup.  A lane's 4 rows (columns) of the tile are the 4 CONSECUTIVE units it loads as one 16-byte value: MFMA block u
// of the tile = units {4 i + u}, a relabelling of rows that costs nothing.  The next group's loads are issued before the
// current group's MFMAs (the kernel has ~60 registers: 4 workgroups per CU hide the rest of the latency).
// HEAD (layer L with the head folded in, NDQ_DEEP_HEAD_FUSED): zin is Z_L; the row operand Zbar_L = act-backward(Z_L, Wout^T
// seeds) is formed per lane from its 4 units of its point, and the tiles of column block 0 also leave what deep_head_bwd used
// to: partial rows of dWout (seeds x sigma-jet(Z_L)), db_L (value stream of Zbar_L) and -- tile 0 -- dbout, one per wave.
template <class C, bool FIRSTIN, bool HEAD = false>
__global__ __launch_bounds__(C::THREADS, 2) void deep_wgrad_gemm(DeepArgs a) {
  __shared__ real comb[3][64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
  constexpr int NT2 = C::NT * C::NT;
  const int vid = xcd_block_id();                          // (the NT x NT tiles of a point slice share its rows: one XCD)
  const int tl = vid % NT2, ks = vid / NT2, KS = gridDim.x / NT2;
  if (ks >= KS) return;
  const int j0 = 64 * (tl / C::NT) + 4 * i, k0 = 64 * (tl % C::NT) + 4 * i;      // this lane's 4 row / column units
  const bool jok = j0 < C::HP, kok = k0 < C::HP;
  const size_t sstride = (size_t)a.np * C::HP;
  real4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = real4{0.f, 0.f, 0.f, 0.f};
  // FIRSTIN: the first layer's rows of this lane's 4 column units
  real w1v[FIRSTIN ? 4 : 1][C::D], b1v[FIRSTIN ? 4 : 1];
  if constexpr (FIRSTIN) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool ok = k0 + v < C::wl(1);
      b1v[v] = ok ? a.prm[C::offb1 + (ok ? k0 + v : 0)] : 0.f;
#pragma unroll
      for (int d = 0; d < C::D; ++d) w1v[v][d] = ok ? a.prm[C::offW1 + (ok ? k0 + v : 0) * C::D + d] : 0.f;
    }
  }
  const int ngroups = a.np >> 2, g0 = ks * C::WAVES + wave, gstep = KS * C::WAVES;
  real4 za[C::NS], zk[FIRSTIN ? 1 : C::NS];
  real x[FIRSTIN ? C::D : 1];
  // HEAD: Wout columns of this lane's 4 row units, the point's seeds (prefetched with the rows), the by-products
  const bool side = HEAD && (tl % C::NT) == 0;
  real wov[HEAD ? 4 : 1][C::NOUT], gsn[HEAD ? C::NC : 1], dwo[HEAD ? 4 : 1][C::NOUT], dbl[HEAD ? 4 : 1], gbo[HEAD ? C::NOUT : 1];
  if constexpr (HEAD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool ok = j0 + v < C::wl(C::L);
      dbl[v] = 0.f;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) {
        wov[v][o] = ok ? a.prm[C::offWout + o * C::wl(C::L) + (ok ? j0 + v : 0)] : 0.f;
        dwo[v][o] = 0.f;
      }
    }
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) gbo[o] = 0.f;
  }
  auto fetch = [&](int g4) {
    const int n = 4 * g4 + kg;
    if constexpr (HEAD) {
      const int ns = n < a.n ? n : a.n - 1;
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) gsn[cc] = n < a.n ? a.gbar[(size_t)cc * a.ldj + ns] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
      za[s] = jok ? *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + j0) : real4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FIRSTIN) {
      const int nn = n < a.n ? n : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) x[d] = a.coords[(size_t)d * a.ldc + nn];
    } else {
#pragma unroll
      for (int s = 0; s < C::NS; ++s)
        zk[s] = kok ? *reinterpret_cast<const real4*>(a.zprev + s * sstride + (size_t)n * C::HP + k0) : real4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (g0 < ngroups) fetch(g0);
  for (int g4 = g0; g4 < ngroups; g4 += gstep) {
    real4 av[C::NS], hv[C::NS];
    if constexpr (HEAD) {
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) gbo[o] += gsn[o];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        real z[C::NS], g[C::NS], h[C::NS], tt, cc;
#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
          z[s] = za[s][v];
          real w = 0.f;
#pragma unroll
          for (int o = 0; o < C::NOUT; ++o) w = rfma(wov[v][o], gsn[s * C::NOUT + o], w);
          g[s] = w;
        }
        if (side) {                                        // (workgroup-uniform)
          jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int o = 0; o < C::NOUT; ++o) dwo[v][o] = rfma(gsn[s * C::NOUT + o], h[s], dwo[v][o]);
        } else {
          Act<C::ACT>::fwd(z[0], tt, cc);
        }
        jet_unit_backward<C>(z, tt, cc, g);
        dbl[v] += g[0];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) av[s][v] = g[s];
      }
    } else {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) av[s] = za[s];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      real z[C::NS], h[C::NS], tt, cc;
      if constexpr (FIRSTIN) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s] = 0.f;
        real zv = b1v[v];
#pragma unroll
        for (int d = 0; d < C::D; ++d) {
          zv = rfma(w1v[v][d], x[d], zv);
          if constexpr (C::SS::FIRST) z[1 + d] = w1v[v][d];
        }
        z[0] = zv;
      } else {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s] = zk[s][v];
      }
      jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
      for (int s = 0; s < C::NS; ++s) hv[s][v] = h[s];
    }
    if (g4 + gstep < ngroups) fetch(g4 + gstep);           // in flight under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = mfma16x16x4(av[s][u], hv[s][v], acc[u][v]);
  }
  if constexpr (HEAD) {
    // by-products: one partial row per WAVE; a lane's sums run over the points with contraction slot kg -> + the 4 slots
    const int row = ks * C::WAVES + wave;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const real db = quad_sum(dbl[v]);
      if (side && kg == 0 && j0 + v < C::HP) a.pbh[(size_t)row * C::HP + j0 + v] = db;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) {
        const real dw = quad_sum(dwo[v][o]);
        if (side && kg == 0 && j0 + v < C::HP) a.pwo[((size_t)row * C::NOUT + o) * C::HP + j0 + v] = dw;
      }
    }
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) {
      const real gb = quad_sum(gbo[o]);
      if (tl == 0 && lane == 0) a.pbo[(size_t)row * C::NOUT + o] = gb;
    }
  }
  // waves 1 .. 3 -> LDS, wave 0 adds them in order and stores the workgroup's partial tile
  if (wave > 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) comb[wave - 1][(u * 4 + v) * 4 + r][lane] = acc[u][v][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll 1
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][v][r] += comb[w][(u * 4 + v) * 4 + r][lane];
    // D layout: register r of lane (i, kg) = row 4 kg + r, column i of the 16 x 16 block (u, v):
    // row unit = 64 tj + 4 (4 kg + r) + u, column units 64 tk + 4 i + v, v = 0 .. 3 -- one 16-byte store
    real* out = a.pw + (size_t)ks * C::HP * C::HP;
    const int jbase = 64 * (tl / C::NT);
    if (kok) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = jbase + 4 * (4 * kg + r) + u;
          if (row < C::HP)
            *reinterpret_cast<real4*>(out + (size_t)row * C::HP + k0) = real4{acc[u][0][r], acc[u][1][r], acc[u][2][r], acc[u][3][r]};
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradients, bf16x3
// dW_l on the bf16 matrix core (round 5).  Same tiling, same loads and the same per-lane arithmetic in front of the products as
// deep_wgrad_gemm; what changes is the product: v_mfma_f32_16x16x32_bf16 contracts 32 slots per instruction, 8 per lane.  The
// contraction index of dW_l = sum_{s, n} Zbar_l[s][n][j] H_{l-1}[s][n][k] is the PAIR (point, stream), and any order of it
// serves as long as both operands agree -- so a lane's 8 slots of a chunk are simply the next 8 (group, stream) values it
// produces: point group g hands every lane (i, kg) the NS stream values of point 4 g + kg for its 4 row units and its 4 column
// units (two 16-byte loads per stream, as before), they are split two at a time into three bf16 planes (x = x0 + x1 + x2) and
// written into the chunk's next two slots; a full chunk issues the six significant plane products for the 4 x 4 blocks of the
// tile.  U = 8 / gcd(8, NS) groups (at least 2) fill whole chunks: a ROUND of U groups is straight-line code, so every slot
// index is a compile-time constant and nothing is transposed anywhere.  6 products of 16 cycles per 8 slots against 8
// exact-f32 products of 32 cycles (product and VALU time add up for either format: DESIGN.md 4.0).  A wave's last round is
// padded with groups whose row operand is zero.
// Accuracy: fp32-class, as for the per-point GEMMs (csrc/ndq_mlp.h); sums in fixed order (per wave, then waves in order).
template <int E, int NP = NDQ_HTR_PLANES>
__device__ __forceinline__ void split3_pair_into(real x0, real x1, bf16x8 (&pl)[3]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x0, x1};
  const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h0, f32x2);
  const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
  pl[0][E] = h0[0]; pl[0][E + 1] = h0[1];
  pl[1][E] = h1[0]; pl[1][E + 1] = h1[1];
  if constexpr (NP == 3) {              // (two planes where no product of the weight-gradient GEMM reads the third: NDQ_WG_NPROD < 6)
    const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    pl[2][E] = h2[0]; pl[2][E + 1] = h2[1];
  }
}
constexpr int deep_wgbf_groups(int ns) {
  int g = 8, a = ns;
  while (a) { const int t = g % a; g = a; a = t; }         // gcd(8, ns)
  const int u = 8 / g;
  return u < 2 ? 2 : u;
}
// workgroups per CU: two where a round is ONE chunk (NS = 1, 2, 4: <= 256 registers; measured at 128 x 3: 477 against 512 us
// per step with one), else one (NS = 5: 320 - 340 registers, forcing 256 spills)
#ifndef NDQ_DEEP_WGBF_OCC
#define NDQ_DEEP_WGBF_OCC 0        // 0: by stream count (above); 1 / 2: forced (A/B runs)
#endif
constexpr int deep_wgbf_occ(int ns) {
  return NDQ_DEEP_WGBF_OCC ? NDQ_DEEP_WGBF_OCC : (ns * deep_wgbf_groups(ns) == 8 ? 2 : 1);
}
// groups whose rows are in flight per wave (a group is 2 NS 16-byte loads per lane).  Measured at 256 x 2 (one workgroup per
// CU): 1 / 2 / 4 groups ahead 1 006 / 1 004 / 1 015 us per step -- the kernel is not waiting for memory; 2 costs nothing.
#ifndef NDQ_DEEP_WGBF_PF
#define NDQ_DEEP_WGBF_PF 0         // 0: 1 with two workgroups per CU, else 2; 1 / 2 / 4: forced (A/B runs)
#endif
constexpr int deep_wgbf_pf(int ns) { return NDQ_DEEP_WGBF_PF ? NDQ_DEEP_WGBF_PF : (deep_wgbf_occ(ns) == 2 ? 1 : 2); }

template <class C, bool FIRSTIN, bool HEAD = false>
__global__ __launch_bounds__(C::THREADS, deep_wgbf_occ(C::NS)) void deep_wgrad_bf(DeepArgs a) {
  __shared__ real comb[3][64][64];
  constexpr int NS = C::NS, U = deep_wgbf_groups(NS), PF = deep_wgbf_pf(NS), G = U > PF ? U : PF;     // groups per loop iteration
  static_assert((NS * U) % 8 == 0 && G % U == 0 && G % PF == 0, "whole chunks per round, whole rounds and buffer turns per iteration");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
  constexpr int NT2 = C::NT * C::NT;
  const int vid = xcd_block_id();
  const int tl = vid % NT2, ks = vid / NT2, KS = gridDim.x / NT2;
  if (ks >= KS) return;
  const int j0 = 64 * (tl / C::NT) + 4 * i, k0 = 64 * (tl % C::NT) + 4 * i;
  const bool jok = j0 < C::HP, kok = k0 < C::HP;
  const int j0c = jok ? j0 : 0, k0c = kok ? k0 : 0;        // (clamped: every load below is unconditional)
  const size_t sstride = (size_t)a.np * C::HP;
  real4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = real4{0.f, 0.f, 0.f, 0.f};
  real w1v[FIRSTIN ? 4 : 1][C::D], b1v[FIRSTIN ? 4 : 1];
  if constexpr (FIRSTIN) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool ok = k0 + v < C::wl(1);
      b1v[v] = ok ? a.prm[C::offb1 + (ok ? k0 + v : 0)] : 0.f;
#pragma unroll
      for (int d = 0; d < C::D; ++d) w1v[v][d] = ok ? a.prm[C::offW1 + (ok ? k0 + v : 0) * C::D + d] : 0.f;
    }
  }
  const int ngroups = a.np >> 2, g0 = ks * C::WAVES + wave, gstep = KS * C::WAVES;
  const bool side = HEAD && (tl % C::NT) == 0;
  real wov[HEAD ? 4 : 1][C::NOUT], dwo[HEAD ? 4 : 1][C::NOUT], dbl[HEAD ? 4 : 1], gbo[HEAD ? C::NOUT : 1];
  if constexpr (HEAD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool ok = j0 + v < C::wl(C::L);
      dbl[v] = 0.f;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) {
        wov[v][o] = ok ? a.prm[C::offWout + o * C::wl(C::L) + (ok ? j0 + v : 0)] : 0.f;
        dwo[v][o] = 0.f;
      }
    }
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) gbo[o] = 0.f;
  }
  // the rows of the NEXT PF groups, requested while the current group is being worked on (PF buffers taken in turn; group index
  // clamped, `live` = the group exists)
  real4 zab[PF][NS], zkb[PF][FIRSTIN ? 1 : NS];
  real xb[PF][FIRSTIN ? C::D : 1], gsb[PF][HEAD ? C::NC : 1];
  bool liveb[PF];
  auto fetch = [&](auto b_, int g4) {
    constexpr int b = decltype(b_)::value;
    const bool live = g4 < ngroups;
    liveb[b] = live;
    const int n = 4 * (live ? g4 : ngroups - 1) + kg, ns = n < a.n ? n : a.n - 1;
    if constexpr (HEAD) {
#pragma unroll
      for (int cc = 0; cc < C::NC; ++cc) {
        const real v = a.gbar[(size_t)cc * a.ldj + ns];
        gsb[b][cc] = (live && n < a.n) ? v : 0.f;          // (padding points and padding groups: Zbar_L = 0)
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) zab[b][s] = *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + j0c);
    if constexpr (FIRSTIN) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) xb[b][d] = a.coords[(size_t)d * a.ldc + ns];
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) zkb[b][s] = *reinterpret_cast<const real4*>(a.zprev + s * sstride + (size_t)n * C::HP + k0c);
    }
  };
  sfor<PF>([&](auto b_) { fetch(b_, g0 + decltype(b_)::value * gstep); });
  bf16x8 cA[4][3], cB[4][3];
  real pa[4], pb[4];
  for (int gr = g0; gr < ngroups; gr += G * gstep) {
    sfor<G>([&](auto gi_) {
      constexpr int gi = decltype(gi_)::value, bi = gi % PF;
      // ---- this group's values: rows av = Zbar_l (HEAD: formed here), columns hv = sigma-jet(Z_{l-1})
      real4 av[NS], hv[NS];
      const bool rows = liveb[bi] && jok;
      real4 (&za)[NS] = zab[bi];
      real4 (&zk)[FIRSTIN ? 1 : NS] = zkb[bi];
      real (&x)[FIRSTIN ? C::D : 1] = xb[bi];
      real (&gsn)[HEAD ? C::NC : 1] = gsb[bi];
      if constexpr (HEAD) {
#pragma unroll
        for (int o = 0; o < C::NOUT; ++o) gbo[o] += gsn[o];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          real z[NS], g[NS], h[NS], tt, cc;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            z[s] = jok ? za[s][v] : 0.f;
            real w = 0.f;
#pragma unroll
            for (int o = 0; o < C::NOUT; ++o) w = rfma(wov[v][o], gsn[s * C::NOUT + o], w);
            g[s] = w;
          }
          if (side) {                                        // (workgroup-uniform)
            jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
              for (int o = 0; o < C::NOUT; ++o) dwo[v][o] = rfma(gsn[s * C::NOUT + o], h[s], dwo[v][o]);
          } else {
            Act<C::ACT>::fwd(z[0], tt, cc);
          }
          jet_unit_backward<C>(z, tt, cc, g);
          dbl[v] += g[0];
#pragma unroll
          for (int s = 0; s < NS; ++s) av[s][v] = g[s];
        }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) av[s] = rows ? za[s] : real4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        real z[NS], h[NS], tt, cc;
        if constexpr (FIRSTIN) {
#pragma unroll
          for (int s = 0; s < NS; ++s) z[s] = 0.f;
          real zv = b1v[v];
#pragma unroll
          for (int d = 0; d < C::D; ++d) {
            zv = rfma(w1v[v][d], x[d], zv);
            if constexpr (C::SS::FIRST) z[1 + d] = w1v[v][d];
          }
          z[0] = zv;
        } else {
#pragma unroll
          for (int s = 0; s < NS; ++s) z[s] = kok ? zk[s][v] : 0.f;
        }
        jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
        for (int s = 0; s < NS; ++s) hv[s][v] = h[s];
      }
      fetch(std::integral_constant<int, bi>{}, gr + (gi + PF) * gstep);      // this buffer's next group, PF groups ahead
      __builtin_amdgcn_sched_barrier(0);                     // (the loads are issued before the splits / products below)
      // ---- the group's NS values per unit -> the next NS slots of the chunk
      sfor<NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value, q = (gi % U) * NS + s, e = q % 8;
        if constexpr (e % 2 == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) { pa[u] = av[s][u]; pb[u] = hv[s][u]; }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            split3_pair_into<e - 1>(pa[u], av[s][u], cA[u]);
            split3_pair_into<e - 1>(pb[u], hv[s][u], cB[u]);
          }
        }
        if constexpr (e == 7) {
#define NDQ_WG(QA, QB)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                              \
  _Pragma("unroll") for (int v = 0; v < 4; ++v)                                                              \
      acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cA[u][QA], cB[v][QB], acc[u][v], 0, 0, 0);
          NDQ_WPRODUCTS(NDQ_WG)
#undef NDQ_WG
        }
      });
    });
  }
  if constexpr (HEAD) {
    const int row = ks * C::WAVES + wave;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const real db = quad_sum(dbl[v]);
      if (side && kg == 0 && j0 + v < C::HP) a.pbh[(size_t)row * C::HP + j0 + v] = db;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) {
        const real dw = quad_sum(dwo[v][o]);
        if (side && kg == 0 && j0 + v < C::HP) a.pwo[((size_t)row * C::NOUT + o) * C::HP + j0 + v] = dw;
      }
    }
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) {
      const real gb = quad_sum(gbo[o]);
      if (tl == 0 && lane == 0) a.pbo[(size_t)row * C::NOUT + o] = gb;
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) comb[wave - 1][(u * 4 + v) * 4 + r][lane] = acc[u][v][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll 1
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][v][r] += comb[w][(u * 4 + v) * 4 + r][lane];
    real* out = a.pw + (size_t)ks * C::HP * C::HP;
    const int jbase = 64 * (tl / C::NT);
    if (kok) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = jbase + 4 * (4 * kg + r) + u;
          if (row < C::HP)
            *reinterpret_cast<real4*>(out + (size_t)row * C::HP + k0) = real4{acc[u][0][r], acc[u][1][r], acc[u][2][r], acc[u][3][r]};
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ output layer
struct DeepHeadArgs {
  const real* prm;
  const real* z;          // Z_L [NS][np][HP]
  const real* gbar;       // reverse: [NS][NOUT][ldj] seeds
  real* jets;             // forward: [NS][NOUT][ldj]
  real* zbar;             // reverse: Zbar_L
  real* pwo;              // reverse: partial rows of dWout [stripes][NOUT][HP]
  real* pb;               // reverse: partial rows of db_L [stripes][HP]
  real* pbo;              // reverse: partial rows of dbout [stripes][NOUT]
  int n, np, ldj;
};

// u_s[o] = Wout[o] . sigma-jet(Z_L)_s + bout[o]: a wave per 16-point tile, lane (p, q) walks units 16 b + 4 q + r
template <class C>
__global__ __launch_bounds__(C::THREADS) void deep_head_fwd(DeepHeadArgs a) {
  const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
  for (int tile = gw; tile < ntiles; tile += nw) {
    const int n = tile * 16 + p;
    real acc[C::NC];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) acc[c] = 0.f;
    for (int b = 0; b < C::NB; ++b) {
      const int j0 = 16 * b + 4 * q;
      real4 zz[C::NS];
#pragma unroll
      for (int s = 0; s < C::NS; ++s) zz[s] = *reinterpret_cast<const real4*>(a.z + s * sstride + (size_t)n * C::HP + j0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real z[C::NS], h[C::NS], tt, cc;
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s] = zz[s][r];
        jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
        for (int o = 0; o < C::NOUT; ++o) {
          const real wo = (j0 + r < C::wl(C::L)) ? a.prm[C::offWout + o * C::wl(C::L) + j0 + r] : 0.f;
#pragma unroll
          for (int s = 0; s < C::NS; ++s) acc[s * C::NOUT + o] = rfma(wo, h[s], acc[s * C::NOUT + o]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
      real v = quad_sum(acc[c]);
      if (c < C::NOUT) v += a.prm[C::offbout + c];
      if (n < a.n && (c & 3) == q) a.jets[(size_t)c * a.ldj + n] = v;
    }
  }
}

// seeds -> Zbar_L and the gradients of the output layer.  One hidden unit per thread (64 consecutive units per wave), the
// wave walks points: the point's seeds are wave-uniform, the sums over points are per-thread registers.
template <class C>
__global__ __launch_bounds__(C::THREADS) void deep_head_bwd(DeepHeadArgs a) {
  constexpr int UG = (C::HP + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ug = gw % UG, stripe = gw / UG, nstripes = nw / UG;
  if (stripe >= nstripes) return;
  const int j = ug * 64 + lane;
  const bool live = j < C::HP;
  const int jj = live ? j : 0;
  real wo[C::NOUT], dwo[C::NOUT], gbo[C::NOUT];
#pragma unroll
  for (int o = 0; o < C::NOUT; ++o) {
    wo[o] = (j < C::wl(C::L)) ? a.prm[C::offWout + o * C::wl(C::L) + (j < C::wl(C::L) ? jj : 0)] : 0.f;
    dwo[o] = 0.f;
    gbo[o] = 0.f;
  }
  real db = 0.f;
  const size_t sstride = (size_t)a.np * C::HP;
  for (int n = stripe; n < a.np; n += nstripes) {
    const bool valid = n < a.n;
    const int nn = valid ? n : a.n - 1;
    real gs[C::NC];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) gs[c] = valid ? a.gbar[(size_t)c * a.ldj + nn] : 0.f;
    real z[C::NS], h[C::NS], tt, cc;
#pragma unroll
    for (int s = 0; s < C::NS; ++s) z[s] = a.z[s * sstride + (size_t)n * C::HP + jj];
    jet_unit_forward<C>(z, h, tt, cc);
    real g[C::NS];
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
      real v = 0.f;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) {
        dwo[o] = rfma(gs[s * C::NOUT + o], h[s], dwo[o]);
        v = rfma(wo[o], gs[s * C::NOUT + o], v);
      }
      g[s] = v;
    }
    jet_unit_backward<C>(z, tt, cc, g);
    db += g[0];
    if (live) {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) a.zbar[s * sstride + (size_t)n * C::HP + j] = g[s];
    }
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) gbo[o] += gs[o];
  }
  if (live) {
    a.pb[(size_t)stripe * C::HP + j] = db;
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) a.pwo[((size_t)stripe * C::NOUT + o) * C::HP + j] = dwo[o];
  }
  if (ug == 0 && lane == 0) {
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) a.pbo[(size_t)stripe * C::NOUT + o] = gbo[o];
  }
}

// ------------------------------------------------------------------------------------------------ helpers
// padded copies of the hidden weight matrices: wp[l - 2] = W_l (HP x HP, zero padding), wt[l - 2] = W_l^T
template <class C>
__global__ __launch_bounds__(256) void deep_prep(const real* __restrict__ prm, real* __restrict__ wp, real* __restrict__ wt) {
  const int l = 2 + blockIdx.y;
  const size_t base = (size_t)(l - 2) * C::HP * C::HP;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < C::HP * C::HP; e += gridDim.x * blockDim.x) {
    const int j = e / C::HP, k = e % C::HP;
    const real v = (j < C::wl(l) && k < C::wl(l - 1)) ? prm[C::offW(l) + j * C::wl(l - 1) + k] : 0.f;
    wp[base + e] = v;
    wt[base + (size_t)k * C::HP + j] = v;
  }
}

// Second stage of every reduction of a reverse sweep in ONE launch.  Job k: dst[r * cols + c] = sum_{i < nparts}
// src[(i * rows_p + r) * cols_p + c] (partial rows / tiles written by the kernels above; the padding of rows_p x cols_p is
// dropped).  Fixed order: thread (x, y) of a 64 x 16 block adds the parts i = y, y + 16, ... of element x in four
// independent chains (fp64 accumulators, as reduce_partials_kernel of csrc/ndq_api.hip), the 16 slice sums are combined
// in order through LDS.
struct DeepReduceJob {
  const real* src;
  real* dst;
  int nparts, rows_p, cols_p, rows, cols, block0;      // block0: first workgroup of the job
};
struct DeepReduceJobs {
  int njobs, nblocks;
  DeepReduceJob job[24];
};
__global__ __launch_bounds__(1024) void deep_reduce_all(DeepReduceJobs J) {
  __shared__ double part[16][64];
  int k = 0;
  while (k + 1 < J.njobs && (int)blockIdx.x >= J.job[k + 1].block0) ++k;
  const DeepReduceJob& jb = J.job[k];
  const int e = ((int)blockIdx.x - jb.block0) * 64 + threadIdx.x, y = threadIdx.y;
  const bool live = e < jb.rows * jb.cols;
  const int r = live ? e / jb.cols : 0, c = live ? e % jb.cols : 0;
  const real* src = jb.src + (size_t)r * jb.cols_p + c;
  const size_t step = (size_t)jb.rows_p * jb.cols_p;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  // the first 256 parts of this thread (parts y, y + 16, ..., y + 240) as straight-line clamped loads, all in flight before the
  // first add -- the loop form below compiles to "four loads, s_waitcnt vmcnt(0), four adds, branch", one memory round trip per
  // 64 parts (csrc/ndq_api.hip ColumnRows has the story); same order of additions
  real pv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = y + 16 * j;
    pv[j] = src[(size_t)(q < jb.nparts ? q : jb.nparts - 1) * step];
  }
  asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]),
                    "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15]));
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = y + 64 * it;
    if (q + 48 < jb.nparts) {
      s0 += (double)pv[4 * it]; s1 += (double)pv[4 * it + 1]; s2 += (double)pv[4 * it + 2]; s3 += (double)pv[4 * it + 3];
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        if (q + 16 * kk < jb.nparts) s0 += (double)pv[4 * it + kk];
    }
  }
  int i = y + 256;
  for (; i + 48 < jb.nparts; i += 64) {
    const real v0 = src[(size_t)i * step], v1 = src[(size_t)(i + 16) * step], v2 = src[(size_t)(i + 32) * step],
               v3 = src[(size_t)(i + 48) * step];
    s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
  }
  for (; i < jb.nparts; i += 16) s0 += (double)src[(size_t)i * step];
  part[y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (y == 0 && live) {
    double v = part[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 16; ++w) v += part[w][threadIdx.x];
    jb.dst[e] = (real)v;
  }
}

}  // namespace ndq
