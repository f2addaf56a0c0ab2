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
//   forward   deep_fwd_gemm   Z_l = W_l sigma-jet(Z_{l-1}) + b_l      (l = 2: sigma-jet of the first layer, from the coordinates)
//             deep_head_fwd   u_s = Wout sigma-jet(Z_L)_s + bout -> output streams
//   reverse   deep_head_bwd   seeds -> Zbar_L, dWout, db_L, dbout     (one unit per thread, seeds are wave-uniform)
//             deep_wgrad_gemm dW_l = sum_{s, n} Zbar_l^T sigma-jet(Z_{l-1})   (split over points, partials summed in fixed order)
//             deep_bwd_gemm   Hbar_{l-1} = W_l^T Zbar_l, act-backward in the epilogue -> Zbar_{l-1}, db_{l-1}
//                             (l = 2: the first layer's dW1 / db1 instead of a store)
//
// All three GEMMs run on v_mfma_f32_16x16x4_f32 -- EXACT fp32 products, no operand splitting, and both operands of every
// product load straight from the point-major layout (the weight-gradient GEMM contracts over points: its A and B operands
// are "one value per lane" of 4 consecutive points, no transposes).  That pipe peaks at 157 TFLOP/s on MI355X; the bf16x3
// route of csrc/ndq_mlp.h (417 TFLOP/s effective) is the next step for these kernels, not taken yet -- at W = 128 the
// layer round trips through HBM cost as much as the arithmetic.
// Reductions are fixed-order everywhere (per-wave partial rows -> deep_reduce2d), results are bit-reproducible.
// Reference restated: networks.py:59-70 (forward), neurodiffeq.py:21-34 (diff sweeps), solvers.py:393 (backward).
#pragma once
#include "ndq_mlp.h"

namespace ndq {

template <int D_, int FIRST_, unsigned M2_, int LAP_, unsigned M3_, int W_, int L_, int ACT_, int NOUT_>
struct DeepCfg {
  using SS = Streams<D_, FIRST_, M2_, LAP_, M3_>;
  static_assert(M3_ == 0 || ACT_ == ACT_TANH || ACT_ == ACT_SIN || ACT_ == ACT_SIGMOID,
                "third-order streams: tanh / sin / sigmoid networks");
  static_assert(W_ >= 1 && W_ <= 512 && L_ >= 2 && L_ <= 8, "2 .. 8 hidden layers of up to 512 units");
  static constexpr int D = D_, W = W_, L = L_, ACT = ACT_, NOUT = NOUT_, NS = SS::NS, NC = NS * NOUT_;
  static constexpr int HP = (W_ + 15) & ~15, NB = HP / 16;
  static constexpr int THREADS = 256, WAVES = 4;
  // flat parameter vector, torch order: W1 (W, D) b1 (W) | W_l (W, W) b_l (W), l = 2..L | Wout (NOUT, W) bout (NOUT)
  static constexpr int offW1 = 0, offb1 = W_ * D_;
  static constexpr int offW(int l) { return W_ * D_ + W_ + (l - 2) * (W_ * W_ + W_); }     // l in 2 .. L + 1
  static constexpr int offb(int l) { return offW(l) + W_ * W_; }
  static constexpr int offWout = offW(L_ + 1), offbout = offWout + NOUT_ * W_;
  static constexpr int P = offbout + NOUT_;
  // output blocks (16 units) a wave accumulates per pass of the per-point GEMMs: JBC * NS fragments of 4 registers
  static constexpr int jbc() { int j = 48 / NS; j = j < 1 ? 1 : j; j = j > 8 ? 8 : j; return j > NB ? NB : j; }
  static constexpr int JBC = jbc(), NCH = (NB + JBC - 1) / JBC;
  // first-layer gradient accumulators of the last reverse GEMM: JBF blocks per pass
  static constexpr int jbf() { int j = 12 / (D_ + 1); j = j < 1 ? 1 : j; return j > JBC ? JBC : j; }
  static constexpr int JBF = jbf(), NCHF = (NB + JBF - 1) / JBF;
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
  if (k < C::W) {
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
};

// ------------------------------------------------------------------------------------------------ per-point GEMMs
// FIRSTIN: the layer input is the first layer's sigma-jet, evaluated from the coordinates; else sigma-jet(zin).
// Wave w of the grid owns output chunk w % NCH (JBC blocks of 16 units) and walks the 16-point tiles w / NCH, + stripes.
template <class C, bool FIRSTIN>
__global__ __launch_bounds__(C::THREADS) void deep_fwd_gemm(DeepArgs a) {
  const int lane = threadIdx.x & 63, p = lane & 15, kg = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ch = gw % C::NCH, stripe = gw / C::NCH, nstripes = nw / C::NCH;
  if (stripe >= nstripes) return;
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
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
    for (int c16 = 0; c16 < C::NB; ++c16) {
      const int k0 = 16 * c16 + 4 * kg;                    // this lane's 4 contraction units
      real4 hh[C::NS];
      {
        real4 zz[C::NS];
        if constexpr (!FIRSTIN) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s) zz[s] = *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + k0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          real z[C::NS], h[C::NS], tt, cc;
          if constexpr (FIRSTIN) first_unit_streams<C>(a.prm, k0 + t, x, z);
          else {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = zz[s][t];
          }
          jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
          for (int s = 0; s < C::NS; ++s) hh[s][t] = h[s];
        }
      }
#pragma unroll
      for (int jb = 0; jb < C::JBC; ++jb) {
        const int b = ch * C::JBC + jb;
        if (b < C::NB) {
          const real4 w4 = *reinterpret_cast<const real4*>(a.wmat + (size_t)(16 * b + p) * C::HP + k0);
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][jb] = mfma16x16x4(w4[t], hh[s][t], acc[s][jb]);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < C::JBC; ++jb) {
      const int b = ch * C::JBC + jb;
      if (b < C::NB) {
        const int j0 = 16 * b + 4 * kg;                    // rows 4 kg + r of the block, column p
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][jb][r] += (j0 + r < C::W) ? a.bias[j0 + r] : 0.f;
#pragma unroll
        for (int s = 0; s < C::NS; ++s) *reinterpret_cast<real4*>(a.zout + s * sstride + (size_t)n * C::HP + j0) = acc[s][jb];
      }
    }
  }
}

// Hbar_{l-1} = W_l^T Zbar_l, then the act-backward of layer l - 1 in the epilogue.  TOFIRST (l == 2): layer 1's streams come
// from the coordinates and what leaves is dW1 / db1 (per-wave partial rows); else Zbar_{l-1} is stored and db_{l-1} summed.
template <class C, bool TOFIRST>
__global__ __launch_bounds__(C::THREADS) void deep_bwd_gemm(DeepArgs a) {
  constexpr int JB = TOFIRST ? C::JBF : C::JBC, NCH = TOFIRST ? C::NCHF : C::NCH;
  const int lane = threadIdx.x & 63, p = lane & 15, kg = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  const int ch = gw % NCH, stripe = gw / NCH, nstripes = nw / NCH;
  if (stripe >= nstripes) return;
  const int ntiles = a.np >> 4;
  const size_t sstride = (size_t)a.np * C::HP;
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
    for (int c16 = 0; c16 < C::NB; ++c16) {
      const int k0 = 16 * c16 + 4 * kg;
      real4 zb[C::NS];
#pragma unroll
      for (int s = 0; s < C::NS; ++s) zb[s] = *reinterpret_cast<const real4*>(a.zin + s * sstride + (size_t)n * C::HP + k0);
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int b = ch * JB + jb;
        if (b < C::NB) {
          const real4 w4 = *reinterpret_cast<const real4*>(a.wmat + (size_t)(16 * b + p) * C::HP + k0);
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][jb] = mfma16x16x4(w4[t], zb[s][t], acc[s][jb]);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const int b = ch * JB + jb;
      if (b < C::NB) {
        const int j0 = 16 * b + 4 * kg;
        real4 zz[C::NS], out[C::NS];
        if constexpr (!TOFIRST) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s) zz[s] = *reinterpret_cast<const real4*>(a.zprev + s * sstride + (size_t)n * C::HP + j0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real z[C::NS], g[C::NS], tt, cc;
          if constexpr (TOFIRST) first_unit_streams<C>(a.prm, j0 + r, x, z);
          else {
#pragma unroll
            for (int s = 0; s < C::NS; ++s) z[s] = zz[s][r];
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

// dW_l[j][k] = sum_{s, n} Zbar_l[s][n][j] sigma-jet(Z_{l-1})[s][n][k]: the contraction runs over points, 4 per MFMA.
// Wave w owns the TJ x TJ block tile w % (NT * NT) of dW_l and the 4-point groups w / (NT * NT), + KS.
template <class C, bool FIRSTIN>
__global__ __launch_bounds__(C::THREADS) void deep_wgrad_gemm(DeepArgs a) {
  const int lane = threadIdx.x & 63, i = lane & 15, kg = lane >> 4;
  const int gw = blockIdx.x * C::WAVES + (threadIdx.x >> 6), nw = gridDim.x * C::WAVES;
  constexpr int NT2 = C::NT * C::NT;
  const int tl = gw % NT2, ks = gw / NT2, KS = nw / NT2;
  if (ks >= KS) return;
  const int tj = tl / C::NT, tk = tl % C::NT;
  const size_t sstride = (size_t)a.np * C::HP;
  real4 acc[C::TJ][C::TJ];
#pragma unroll
  for (int u = 0; u < C::TJ; ++u)
#pragma unroll
    for (int v = 0; v < C::TJ; ++v) acc[u][v] = real4{0.f, 0.f, 0.f, 0.f};
  const int ngroups = a.np >> 2;
  for (int g4 = ks; g4 < ngroups; g4 += KS) {
    const int n = 4 * g4 + kg;                             // this lane's point = MFMA contraction slot kg
    const int nn = n < a.n ? n : a.n - 1;
    real x[C::D];
    if constexpr (FIRSTIN) {
#pragma unroll
      for (int d = 0; d < C::D; ++d) x[d] = a.coords[(size_t)d * a.ldc + nn];
    }
    real av[C::NS][C::TJ], hv[C::NS][C::TJ];
#pragma unroll
    for (int u = 0; u < C::TJ; ++u) {
      const int bj = tj * C::TJ + u, bk = tk * C::TJ + u;
      const int jj = 16 * (bj < C::NB ? bj : 0) + i, kk = 16 * (bk < C::NB ? bk : 0) + i;
#pragma unroll
      for (int s = 0; s < C::NS; ++s) av[s][u] = (bj < C::NB) ? a.zin[s * sstride + (size_t)n * C::HP + jj] : 0.f;
      real z[C::NS], h[C::NS], tt, cc;
      if constexpr (FIRSTIN) first_unit_streams<C>(a.prm, bk < C::NB ? kk : C::W, x, z);
      else {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s] = a.zprev[s * sstride + (size_t)n * C::HP + kk];
      }
      jet_unit_forward<C>(z, h, tt, cc);
#pragma unroll
      for (int s = 0; s < C::NS; ++s) hv[s][u] = h[s];
    }
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int u = 0; u < C::TJ; ++u)
#pragma unroll
        for (int v = 0; v < C::TJ; ++v) acc[u][v] = mfma16x16x4(av[s][u], hv[s][v], acc[u][v]);
  }
  real* out = a.pw + (size_t)ks * C::HP * C::HP;
#pragma unroll
  for (int u = 0; u < C::TJ; ++u)
#pragma unroll
    for (int v = 0; v < C::TJ; ++v) {
      const int bj = tj * C::TJ + u, bk = tk * C::TJ + v;
      if (bj < C::NB && bk < C::NB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * bj + 4 * kg + r) * C::HP + 16 * bk + i] = acc[u][v][r];
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
          const real wo = (j0 + r < C::W) ? a.prm[C::offWout + o * C::W + j0 + r] : 0.f;
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
    wo[o] = (j < C::W) ? a.prm[C::offWout + o * C::W + jj] : 0.f;
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
    const real v = (j < C::W && k < C::W) ? prm[C::offW(l) + j * C::W + k] : 0.f;
    wp[base + e] = v;
    wt[base + (size_t)k * C::HP + j] = v;
  }
}

// dst[r * cols + c] = sum_{i < nparts} src[(i * rows_p + r) * cols_p + c], fixed order (fp64 accumulators, as
// reduce_partials_kernel of csrc/ndq_api.hip)
__global__ __launch_bounds__(256) void deep_reduce2d(const real* __restrict__ src, int nparts, int rows_p, int cols_p, int rows, int cols,
                                                     real* __restrict__ dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const int r = e / cols, c = e % cols;
  const size_t off = (size_t)r * cols_p + c, step = (size_t)rows_p * cols_p;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  int i = 0;
  for (; i + 3 < nparts; i += 4) {
    s0 += (double)src[off + (size_t)i * step];
    s1 += (double)src[off + (size_t)(i + 1) * step];
    s2 += (double)src[off + (size_t)(i + 2) * step];
    s3 += (double)src[off + (size_t)(i + 3) * step];
  }
  for (; i < nparts; ++i) s0 += (double)src[off + (size_t)i * step];
  dst[e] = (real)((s0 + s1) + (s2 + s3));
}

}  // namespace ndq
