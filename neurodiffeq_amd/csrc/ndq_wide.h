// MI355X (gfx950) device code for networks with ONE hidden layer of 65 .. 512 units -- the reference's own headline shape
// (README.md:125: FCNN(n_input_units=2, n_output_units=1, hidden_units=(512,)); networks.py:26-66 takes any width).
//
// Such a network has no hidden-to-hidden GEMM: z = W1 x + b1 (K = d <= 6), h = sigma(z), u = Wout h + bout.  Its derivative
// streams collapse to  h_s = c_s * sigma^(ord s)(z)  with PER-UNIT constants c_s (products of W1 entries: d/dx_a -> W1[j][a],
// d2/dx_a dx_b -> W1[j][a] W1[j][b], Laplacian stream -> sum_a W1[j][a]^2, ...), so a (point, unit) pair costs one
// activation evaluation, its derivatives and NS fused multiply-adds -- elementwise VALU work, nothing for the matrix core
// (an MFMA formulation would have to split sigma^(k) into bf16x3 planes first, which costs more VALU than the FMAs it
// replaces; the f32 MFMA shares the VALU datapath on gfx950).
//
// Execution model ("units over lanes"): a wave owns a tile of 16 points.  Its 64 lanes are UL unit lanes x PL point lanes
// (W > 256: 64 x 1, W > 128: 32 x 2, else 16 x 4); a lane keeps the weights of its U = ceil(W / UL) <= 8 units
// (W1 row, b1, Wout column, the constants c_s) and their gradient accumulators in REGISTERS for the whole kernel:
//   forward   per round (one point per point lane): z, sigma and its derivatives for the lane's U units, partial output
//             sums -> per-wave LDS rows; every (point, stream, output) row is then added up by ONE lane in fixed order
//             (no shuffles, no atomics) and lands in the tile's output block;
//   per-point the generated function PW (conditions + residuals + loss seeds, codegen.py) on lanes 0 .. 15, seeds -> LDS;
//   reverse   per round: the point's seeds are an LDS broadcast read; every lane updates the gradient accumulators of its
//             own units -- NO cross-lane reduction inside the tile loop (that is the reason for units-over-lanes: the
//             weight gradients are sums over POINTS).
// sigma(z) of the 16 x U (point, unit) pairs of a tile stays in registers between the two passes (128 VGPRs at W = 512;
// one wave per SIMD has 512).  Reductions at the end of the kernel: point lanes -> waves (LDS, fixed order) ->
// partials[block][P], second stage as everywhere (ndq_reduce_partials / reduce_tail_kernel).
//
// Three kernels share the tile code: wide_jet_fwd_kernel / wide_jet_bwd_kernel (the C-ABI's ndq_mlp_jet_fwd / _bwd for
// these shapes, joined through ndq_mlp_register) and wide_closure_kernel (single launch: forward + PW + reverse).
// Reference restated: networks.py:59-70 (forward), neurodiffeq.py:21-34 (diff sweeps), solvers.py:369-395 (closure).
#pragma once
#include "ndq_mlp.h"

#ifndef NDQ_WIDE_THREADS
#define NDQ_WIDE_THREADS 256       // one wave per SIMD.  512 (two, 256 registers each: 80 - 220 B of scratch per lane) measured in
                                   // round 5: README (512,) 51.9 -> 51.9 us per step, 2 -> 512 -> 3 110.6 -> 105.0 (profiles/r05o_wide_threads_ab.jsonl)
#endif

namespace ndq {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int D_, int FIRST_, unsigned M2_, int LAP_, unsigned M3_, int W_, int ACT_, int NOUT_>
struct WideCfg {
  using SS = Streams<D_, FIRST_, M2_, LAP_, M3_>;
  static_assert(M3_ == 0 || act_has_s4(ACT_), "third-order streams: activations with a stated fourth derivative");
  static_assert(W_ >= 1 && W_ <= 512, "one hidden layer of up to 512 units");
  static constexpr int D = D_, W = W_, ACT = ACT_, NOUT = NOUT_, NS = SS::NS, NC = NS * NOUT_;
  static constexpr int L = 1, HR = W_, SKIP = 0, ACTP = 0;
  static constexpr bool NO_PULL = true;          // no pull prologue / loop mode (csrc/ndq_tail.h) for these kernels
  static constexpr unsigned HRP = 0, MONO = 0;
  static constexpr int UL = W_ > 256 ? 64 : (W_ > 128 ? 32 : 16);     // unit lanes
  static constexpr int PL = 64 / UL;                                  // point lanes
  static constexpr int U = (W_ + UL - 1) / UL;                        // units per lane
  static constexpr int ROUNDS = 16 / PL;                              // rounds per 16-point tile
  static constexpr int THREADS = NDQ_WIDE_THREADS, BWD_THREADS = THREADS, FWD_THREADS = THREADS, WAVES = THREADS / 64;
  static constexpr int MAXORD = SS::N3 > 0 ? 3 : (SS::N2 > 0 ? 2 : (SS::FIRST ? 1 : 0));
  static constexpr int ord(int s) { return s == 0 ? 0 : (s < SS::S2 ? 1 : (s < SS::S3 ? 2 : 3)); }
  static constexpr int NX = SS::N2 + SS::N3;                          // constants c_s kept per unit (s >= S2)
  // flat parameter vector, torch order: W1 (W, D) | b1 (W) | Wout (NOUT, W) | bout (NOUT)
  static constexpr int offW1 = 0, offb1 = W_ * D_, offWout = W_ * D_ + W_, offbout = W_ * D_ + W_ + NOUT_ * W_;
  static constexpr int P = offbout + NOUT_;
  // per-wave LDS (floats): reduction rows [RP points][NC][RS] | tile outputs [16][NCP] | seeds [16][NCP]
  static constexpr int RS = UL + 4;
  static constexpr int NCP = (NC + 3) & ~3;
  static constexpr int rp() {
    int r = 16;
    while (r > PL && r * NC * RS * 4 > 20 * 1024 * 4 / WAVES) r >>= 1;
    return r;
  }
  static constexpr int RP = rp();                                     // points per reduction pass (PL <= RP <= 16)
  static constexpr int ldsRed = 0, ldsOut = RP * NC * RS, ldsSeed = ldsOut + 16 * NCP;
  static constexpr int waveFloats = (ldsSeed + 16 * NCP + 3) & ~3;
};

// ------------------------------------------------------------------------------------------------ a lane's units
template <class C>
struct WideUnits {
  real w[C::U][C::D];                          // W1 rows
  real b[C::U];                                // b1
  real cs[C::U][C::NX > 0 ? C::NX : 1];        // c_s, s >= S2
  real wo[C::U][C::NOUT];                      // Wout column
  real k[C::U][C::NS];                         // NOUT == 1: Wout * c_s (forward coefficients); unused otherwise
};

// c_s of stream S for one unit (S == 0: 1, first order: the W1 entry)
template <class C, int S>
__device__ __forceinline__ real wide_cs(const real (&w)[C::D], const real (&cs)[C::NX > 0 ? C::NX : 1]) {
  using SS = typename C::SS;
  if constexpr (S == 0) return 1.f;
  else if constexpr (S < SS::S2) return w[S - 1];
  else return cs[S - SS::S2];
}

template <class C>
__device__ __forceinline__ void wide_load_units(const real* __restrict__ prm, int ul, WideUnits<C>& un) {
  using SS = typename C::SS;
#pragma unroll
  for (int i = 0; i < C::U; ++i) {
    const int j = ul + C::UL * i;
    const bool ok = j < C::W;                  // padding units: zero weights, nothing downstream sees them
    const int jj = ok ? j : 0;
#pragma unroll
    for (int a = 0; a < C::D; ++a) un.w[i][a] = ok ? prm[C::offW1 + jj * C::D + a] : 0.f;
    un.b[i] = ok ? prm[C::offb1 + jj] : 0.f;
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) un.wo[i][o] = ok ? prm[C::offWout + o * C::W + jj] : 0.f;
    if constexpr (SS::LAP) {
      real q2 = 0.f;
      sfor<C::D>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        if constexpr (SS::in_lap(a)) q2 = rfma(un.w[i][a], un.w[i][a], q2);
      });
      un.cs[i][0] = q2;
    } else {
      sfor<SS::N2>([&](auto k_) {
        constexpr int s = SS::S2 + decltype(k_)::value;
        un.cs[i][s - SS::S2] = un.w[i][SS::A(s)] * un.w[i][SS::B(s)];
      });
      sfor<SS::N3>([&](auto k_) {
        constexpr int s = SS::S3 + decltype(k_)::value;
        un.cs[i][s - SS::S2] = un.w[i][SS::T(s, 0)] * un.w[i][SS::T(s, 1)] * un.w[i][SS::T(s, 2)];
      });
    }
    if constexpr (C::NOUT == 1) {
      sfor<C::NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        un.k[i][s] = un.wo[i][0] * wide_cs<C, s>(un.w[i], un.cs[i]);
      });
    }
  }
}

// sigma and its derivatives up to order NORD from the kept state (t, c)
template <class C, int NORD>
__device__ __forceinline__ void wide_sigmas(real t, real c, real (&sg)[NORD + 1]) {
  using A = Act<C::ACT>;
  sg[0] = t;
  if constexpr (NORD >= 1) sg[1] = A::s1(t, c);
  if constexpr (NORD >= 2) sg[2] = A::s2(t, c, sg[1]);
  if constexpr (NORD >= 3) sg[3] = A::s3(t, c, sg[1]);
  if constexpr (NORD >= 4) sg[4] = A::s4(t, c, sg[1]);
}

// kept activation state of a tile: sigma(z) (and the second state value of sin / swish / aptx) of ROUNDS x U pairs
template <class C>
struct WideKept {
  real t[C::ROUNDS][C::U];
  real c[C::ROUNDS][C::U];
};

// one round of the forward pass: this lane's point x, its U units -> partial output sums acc[s * NOUT + o]
template <class C, bool OUT>
__device__ __forceinline__ void wide_round_forward(const WideUnits<C>& un, const real (&x)[C::D], real (&kt)[C::U], real (&kc)[C::U],
                                                   real (&acc)[C::NC]) {
  using A = Act<C::ACT>;
#pragma unroll
  for (int c = 0; c < C::NC; ++c) acc[c] = 0.f;
#pragma unroll
  for (int i = 0; i < C::U; ++i) {
    real z = un.b[i];
#pragma unroll
    for (int a = 0; a < C::D; ++a) z = rfma(un.w[i][a], x[a], z);
    real t, c;
    A::fwd(z, t, c);
    kt[i] = t;
    kc[i] = c;
    if constexpr (OUT) {
      real sg[C::MAXORD + 1];
      wide_sigmas<C, C::MAXORD>(t, c, sg);
      sfor<C::NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        if constexpr (C::NOUT == 1) {
          acc[s] = rfma(un.k[i][s], sg[C::ord(s)], acc[s]);
        } else {
          const real hs = wide_cs<C, s>(un.w[i], un.cs[i]) * sg[C::ord(s)];
#pragma unroll
          for (int o = 0; o < C::NOUT; ++o) acc[s * C::NOUT + o] = rfma(un.wo[i][o], hs, acc[s * C::NOUT + o]);
        }
      });
    }
  }
}

// gradient accumulators of a lane's units.  NOUT == 1: b1 / w1 are kept in units of Wout (multiplied in at the end).
template <class C>
struct WideGrad {
  real w1[C::U][C::D];
  real b1[C::U];
  real wo[C::U][C::NOUT];
};

template <class C>
__device__ __forceinline__ void wide_grad_zero(WideGrad<C>& g) {
#pragma unroll
  for (int i = 0; i < C::U; ++i) {
    g.b1[i] = 0.f;
#pragma unroll
    for (int a = 0; a < C::D; ++a) g.w1[i][a] = 0.f;
#pragma unroll
    for (int o = 0; o < C::NOUT; ++o) g.wo[i][o] = 0.f;
  }
}

// one round of the reverse pass: seeds gs[s * NOUT + o] of this lane's point (zero for padding points)
//   dWout[o][j] += sum_ord G_ord[o] sigma^(ord),            G_ord[o] = sum_{s of order ord} gs[s][o] c_s
//   zbar         = sum_ord Gh_ord sigma^(ord + 1),           Gh_ord = sum_o Wout[o][j] G_ord[o]
//   db1[j] += zbar,  dW1[j][a] += zbar x_a + sum_s gh_s sigma^(ord s) d c_s / d W1[j][a],   gh_s = sum_o Wout[o][j] gs[s][o]
template <class C>
__device__ __forceinline__ void wide_round_backward(const WideUnits<C>& un, const real (&x)[C::D], const real (&kt)[C::U],
                                                    const real (&kc)[C::U], const real (&gs)[C::NC], WideGrad<C>& g) {
  using SS = typename C::SS;
  constexpr int NO = C::NOUT;
#pragma unroll
  for (int i = 0; i < C::U; ++i) {
    real sg[C::MAXORD + 2];
    wide_sigmas<C, C::MAXORD + 1>(kt[i], kc[i], sg);
    // G_ord[o]
    real G[C::MAXORD + 1][NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      G[0][o] = gs[o];
      if constexpr (C::MAXORD >= 1) {
        real v = 0.f;
#pragma unroll
        for (int a = 0; a < C::D; ++a) v = rfma(gs[(1 + a) * NO + o], un.w[i][a], v);
        G[1][o] = v;
      }
      if constexpr (C::MAXORD >= 2) {
        real v = 0.f;
#pragma unroll
        for (int s = SS::S2; s < SS::S3; ++s) v = rfma(gs[s * NO + o], un.cs[i][s - SS::S2], v);
        G[2][o] = v;
      }
      if constexpr (C::MAXORD >= 3) {
        real v = 0.f;
#pragma unroll
        for (int s = SS::S3; s < C::NS; ++s) v = rfma(gs[s * NO + o], un.cs[i][s - SS::S2], v);
        G[3][o] = v;
      }
    }
    // dWout
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      real v = g.wo[i][o];
#pragma unroll
      for (int k = 0; k <= C::MAXORD; ++k) v = rfma(G[k][o], sg[k], v);
      g.wo[i][o] = v;
    }
    // gh_s (s >= 1) and Gh_ord: contracted with the Wout column (NOUT == 1: the factor is applied once, at the end)
    real gh[C::NS], Gh[C::MAXORD + 1];
    if constexpr (NO == 1) {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) gh[s] = gs[s];
#pragma unroll
      for (int k = 0; k <= C::MAXORD; ++k) Gh[k] = G[k][0];
    } else {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        real v = 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) v = rfma(un.wo[i][o], gs[s * NO + o], v);
        gh[s] = v;
      }
#pragma unroll
      for (int k = 0; k <= C::MAXORD; ++k) {
        real v = 0.f;
#pragma unroll
        for (int o = 0; o < NO; ++o) v = rfma(un.wo[i][o], G[k][o], v);
        Gh[k] = v;
      }
    }
    real zb = 0.f;
#pragma unroll
    for (int k = 0; k <= C::MAXORD; ++k) zb = rfma(Gh[k], sg[k + 1], zb);
    // explicit dependence of the constants c_s on the row of W1 (product rule, position by position)
    real e[C::D];
#pragma unroll
    for (int a = 0; a < C::D; ++a) e[a] = 0.f;
    if constexpr (SS::FIRST) {
#pragma unroll
      for (int a = 0; a < C::D; ++a) e[a] = gh[1 + a] * sg[1];
      if constexpr (SS::LAP) {
        const real f = 2.f * gh[SS::S2] * sg[2];
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          if constexpr (SS::in_lap(a)) e[a] = rfma(f, un.w[i][a], e[a]);
        });
      } else {
        sfor<SS::N2>([&](auto k_) {
          constexpr int s = SS::S2 + decltype(k_)::value;
          constexpr int a = SS::A(s), bb = SS::B(s);
          const real f = gh[s] * sg[2];
          e[a] = rfma(f, un.w[i][bb], e[a]);
          e[bb] = rfma(f, un.w[i][a], e[bb]);
        });
        sfor<SS::N3>([&](auto k_) {
          constexpr int s = SS::S3 + decltype(k_)::value;
          constexpr int a = SS::T(s, 0), bb = SS::T(s, 1), cc = SS::T(s, 2);
          const real f = gh[s] * sg[3];
          e[a] = rfma(f, un.w[i][bb] * un.w[i][cc], e[a]);
          e[bb] = rfma(f, un.w[i][a] * un.w[i][cc], e[bb]);
          e[cc] = rfma(f, un.w[i][a] * un.w[i][bb], e[cc]);
        });
      }
    }
    g.b1[i] += zb;
#pragma unroll
    for (int a = 0; a < C::D; ++a) g.w1[i][a] += rfma(zb, x[a], e[a]);
  }
}

// (wave-uniform read of one lane's value; fp64: two 32-bit halves)
__device__ __forceinline__ real readlane_real(real v, int lane) {
#if NDQ_F64
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
#else
  return __builtin_bit_cast(real, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
#endif
}

// coordinates of this lane's point in round r: lane l < 16 of xv holds point l of the tile
template <class C>
__device__ __forceinline__ void wide_round_coords(const real (&xv)[C::D], int r, int pl, real (&x)[C::D]) {
#pragma unroll
  for (int a = 0; a < C::D; ++a) {
    if constexpr (C::PL == 1) x[a] = readlane_real(xv[a], r);
    else x[a] = __shfl(xv[a], r * C::PL + pl);
  }
}

// forward pass of a tile.  KEEP: sigma states stay in `kept`; OUT: output streams -> wl[ldsOut + pt * NCP + c] (bias added).
template <class C, bool KEEP, bool OUT>
__device__ __forceinline__ void wide_tile_forward(const WideUnits<C>& un, const real (&xv)[C::D], real* wl, const real* __restrict__ bout,
                                                  int lane, int ul, int pl, WideKept<C>& kept) {
#pragma unroll
  for (int r = 0; r < C::ROUNDS; ++r) {
    real x[C::D], acc[C::NC], kt[C::U], kc[C::U];
    wide_round_coords<C>(xv, r, pl, x);
    wide_round_forward<C, OUT>(un, x, kt, kc, acc);
    if constexpr (KEEP) {
#pragma unroll
      for (int i = 0; i < C::U; ++i) { kept.t[r][i] = kt[i]; kept.c[r][i] = kc[i]; }
    }
    if constexpr (OUT) {
      const int pt = r * C::PL + pl;                       // point of the tile
      const int ptl = pt & (C::RP - 1);                    // ... of the reduction pass
#pragma unroll
      for (int c = 0; c < C::NC; ++c) wl[C::ldsRed + (ptl * C::NC + c) * C::RS + ul] = acc[c];
      if (((r + 1) * C::PL) % C::RP == 0) {                // a pass is complete (compile-time per unrolled round)
        wave_lds_sync();
        const int base = (r + 1) * C::PL - C::RP;          // first point of the pass
        for (int row = lane; row < C::RP * C::NC; row += 64) {
          const real* rr = wl + C::ldsRed + row * C::RS;
          real4 s4 = lds4(rr);
#pragma unroll
          for (int k = 1; k < C::UL / 4; ++k) {
            const real4 v = lds4(rr + 4 * k);
            s4[0] += v[0]; s4[1] += v[1]; s4[2] += v[2]; s4[3] += v[3];
          }
          const int c = row % C::NC, p_ = row / C::NC;
          real v = (s4[0] + s4[1]) + (s4[2] + s4[3]);
          if (c < C::NOUT) v += bout[c];                   // value stream: + output bias
          wl[C::ldsOut + (base + p_) * C::NCP + c] = v;
        }
        wave_lds_sync();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// reverse pass of a tile: seeds in wl[ldsSeed + pt * NCP + c]
template <class C>
__device__ __forceinline__ void wide_tile_backward(const WideUnits<C>& un, const real (&xv)[C::D], const real* wl, int pl,
                                                   const WideKept<C>& kept, WideGrad<C>& g) {
#pragma unroll
  for (int r = 0; r < C::ROUNDS; ++r) {
    real x[C::D], gs[C::NCP];
    wide_round_coords<C>(xv, r, pl, x);
    const real* sd = wl + C::ldsSeed + (r * C::PL + pl) * C::NCP;
#pragma unroll
    for (int k = 0; k < C::NCP / 4; ++k) {
      const real4 v = lds4(sd + 4 * k);
      gs[4 * k] = v[0]; gs[4 * k + 1] = v[1]; gs[4 * k + 2] = v[2]; gs[4 * k + 3] = v[3];
    }
    real gsc[C::NC];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) gsc[c] = gs[c];
    // the reverse pass re-derives sigma', sigma'' ... from the kept sigma: behind an opaque copy, or the compiler keeps the
    // forward pass's derivative values of all 16 x U pairs alive instead (common subexpressions: +384 registers)
    real kt[C::U], kc[C::U];
#pragma unroll
    for (int i = 0; i < C::U; ++i) {
      kt[i] = kept.t[r][i];
      kc[i] = kept.c[r][i];
      asm volatile("" : "+v"(kt[i]));
      asm volatile("" : "+v"(kc[i]));
    }
    wide_round_backward<C>(un, x, kt, kc, gsc, g);
    __builtin_amdgcn_sched_barrier(0);       // rounds are independent: keep the scheduler from interleaving all 16 (registers)
  }
}

// lanes -> waves (fixed order) -> out[P].  gbo: this lane's sum of the value-stream seeds (output bias gradient; lanes
// that carried no point hold zeros).  `red`: P + 16 floats of LDS nobody else uses any more.
template <class C>
__device__ __forceinline__ void wide_block_reduce_store(const WideUnits<C>& un, WideGrad<C>& g, real (&gbo)[C::NOUT], real* red,
                                                        int wave, int lane, int ul, int pl, real* __restrict__ out) {
  // point lanes of a unit lane (lanes ul, ul + UL, ...) in fixed order
#pragma unroll
  for (int i = 0; i < C::U; ++i) {
    if constexpr (C::NOUT == 1) {
      g.b1[i] *= un.wo[i][0];
#pragma unroll
      for (int a = 0; a < C::D; ++a) g.w1[i][a] *= un.wo[i][0];
    }
#pragma unroll
    for (int m = C::UL; m < 64; m <<= 1) {
      g.b1[i] += __shfl_xor(g.b1[i], m);
#pragma unroll
      for (int a = 0; a < C::D; ++a) g.w1[i][a] += __shfl_xor(g.w1[i][a], m);
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) g.wo[i][o] += __shfl_xor(g.wo[i][o], m);
    }
  }
#pragma unroll
  for (int o = 0; o < C::NOUT; ++o) gbo[o] = point_sum(quad_sum(gbo[o]));
  __syncthreads();                             // every wave is done with its tile regions
  for (int k = 0; k < C::WAVES; ++k) {
    if (wave == k && pl == 0) {
      auto put = [&](int idx, real v) {
        if (k == 0) red[idx] = v;
        else red[idx] += v;
      };
#pragma unroll
      for (int i = 0; i < C::U; ++i) {
        const int j = ul + C::UL * i;
        if (j < C::W) {
          put(C::offb1 + j, g.b1[i]);
#pragma unroll
          for (int a = 0; a < C::D; ++a) put(C::offW1 + j * C::D + a, g.w1[i][a]);
#pragma unroll
          for (int o = 0; o < C::NOUT; ++o) put(C::offWout + o * C::W + j, g.wo[i][o]);
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int o = 0; o < C::NOUT; ++o) put(C::offbout + o, gbo[o]);
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < C::P; i += blockDim.x) out[i] = red[i];
}

template <class C> constexpr size_t wide_lds_bytes() {
  const int tiles = C::WAVES * C::waveFloats;
  const int red = C::P + 16;
  return sizeof(real) * ((tiles > red ? tiles : red) + 64);
}

// this lane's point of the tile for the per-point stages (lane = (p, q), all four q groups hold the same point)
// ------------------------------------------------------------------------------------------------ stream kernels
template <class C>
__global__ __launch_bounds__(C::THREADS) void wide_jet_fwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int ul = lane & (C::UL - 1), pl = lane / C::UL;
  real* wl = lds + wave * C::waveFloats;
  WideUnits<C> un;
  wide_load_units<C>(a.params, ul, un);
  const int ntiles = (a.n + 15) >> 4;
  real xvn[C::D];                              // coordinates one tile ahead (see wide_closure_body)
  {
    const int n0 = (blockIdx.x * C::WAVES + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xvn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  for (int tile = blockIdx.x * C::WAVES + wave; tile < ntiles; tile += gridDim.x * C::WAVES) {
    const int n = tile * 16 + p;
    const bool valid = n < a.n;
    real xv[C::D];
#pragma unroll
    for (int d = 0; d < C::D; ++d) xv[d] = xvn[d];
    {
      const int n1 = n + gridDim.x * C::WAVES * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xvn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
    WideKept<C> kept;
    wide_tile_forward<C, false, true>(un, xv, wl, a.params + C::offbout, lane, ul, pl, kept);
    if (valid) {
      for (int c = q; c < C::NC; c += 4) a.jets[(size_t)c * a.ldj + n] = wl[C::ldsOut + p * C::NCP + c];
    }
    wave_lds_sync();
  }
}

template <class C>
__global__ __launch_bounds__(C::THREADS) void wide_jet_bwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int ul = lane & (C::UL - 1), pl = lane / C::UL;
  real* wl = lds + wave * C::waveFloats;
  WideUnits<C> un;
  wide_load_units<C>(a.params, ul, un);
  WideGrad<C> g;
  wide_grad_zero<C>(g);
  real gbo[C::NOUT];
#pragma unroll
  for (int o = 0; o < C::NOUT; ++o) gbo[o] = 0.f;
  const int ntiles = (a.n + 15) >> 4;
  real xvn[C::D];                              // coordinates one tile ahead (see wide_closure_body)
  {
    const int n0 = (blockIdx.x * C::WAVES + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xvn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  for (int tile = blockIdx.x * C::WAVES + wave; tile < ntiles; tile += gridDim.x * C::WAVES) {
    const int n = tile * 16 + p;
    const bool valid = n < a.n;
    const int nn = valid ? n : a.n - 1;
    real xv[C::D];
#pragma unroll
    for (int d = 0; d < C::D; ++d) xv[d] = xvn[d];
    {
      const int n1 = n + gridDim.x * C::WAVES * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xvn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
    // seeds of point p: lane group q stores components q, q + 4, ...
#pragma unroll
    for (int k = 0; k < C::NCP / 4; ++k) {
      const int c = q + 4 * k;
      const real v = (valid && c < C::NC) ? a.gbar[(size_t)(c < C::NC ? c : 0) * a.ldj + nn] : 0.f;
      wl[C::ldsSeed + p * C::NCP + c] = v;
#pragma unroll
      for (int o = 0; o < C::NOUT; ++o) gbo[o] += (c == o) ? v : 0.f;
    }
    WideKept<C> kept;
    wide_tile_forward<C, true, false>(un, xv, wl, nullptr, lane, ul, pl, kept);
    wave_lds_sync();
    wide_tile_backward<C>(un, xv, wl, pl, kept, g);
    wave_lds_sync();
  }
  wide_block_reduce_store<C>(un, g, gbo, lds, wave, lane, ul, pl, a.partials + (size_t)blockIdx.x * C::P);
}

// ------------------------------------------------------------------------------------------------ single-launch closure
// PW (generated, codegen.py: the grouped closure's interface): apply(cc[NC], theta, srow, seed, want_adj, r, f, grow, gth) reads
// the point's stream row srow[s * NOUT + o] and leaves the adjoint seeds in grow; dep(d) = batch coordinate fed to network
// input d; NC coordinate-block rows per point (coordinates, then per-point data columns).
template <class C, class PW, bool TRAIN>
__device__ __forceinline__ void wide_closure_body(const FusedArgs& a, real* lds, const int blk, const int nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int ul = lane & (C::UL - 1), pl = lane / C::UL;
  real* wl = lds + wave * C::waveFloats;
  WideUnits<C> un;
  wide_load_units<C>(a.params, ul, un);
  WideGrad<C> g;
  if constexpr (TRAIN) wide_grad_zero<C>(g);
  real gbo[C::NOUT];
#pragma unroll
  for (int o = 0; o < C::NOUT; ++o) gbo[o] = 0.f;
  real lsum = 0.f;
  real th[PW::NT > 0 ? PW::NT : 1], tsum[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
  for (int j = 0; j < PW::NT; ++j) { th[j] = a.theta[j]; tsum[j] = 0.f; }
  const int ntiles = (a.n + 15) >> 4;
  // coordinates one tile ahead (round 5: the load at the top of the tile loop was one exposed HBM round trip per tile -- a wave
  // has the SIMD to itself; the first tile's load overlaps what is left of the prologue)
  real ccn[PW::NC];
  {
    const int n0 = (blk * C::WAVES + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < PW::NC; ++d) ccn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  for (int tile = blk * C::WAVES + wave; tile < ntiles; tile += nblk * C::WAVES) {
    const int n = tile * 16 + p;
    const bool valid = n < a.n;
    real cc[PW::NC];
#pragma unroll
    for (int d = 0; d < PW::NC; ++d) cc[d] = ccn[d];
    {
      const int n1 = n + nblk * C::WAVES * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < PW::NC; ++d) ccn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
    real xv[C::D];
    sfor<C::D>([&](auto d_) {
      constexpr int d = decltype(d_)::value;
      xv[d] = cc[PW::dep(d)];
    });
    WideKept<C> kept;
    wide_tile_forward<C, TRAIN, true>(un, xv, wl, a.params + C::offbout, lane, ul, pl, kept);
    // ---- per-point stage on lanes 0 .. 15 (one point each); seeds -> LDS
    if (q == 0) {
      real r[PW::NR], f[PW::NF > 0 ? PW::NF : 1], gth[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
      for (int j = 0; j < PW::NT; ++j) gth[j] = 0.f;
      real* srow = wl + C::ldsOut + p * C::NCP;
      real* grow = wl + C::ldsSeed + p * C::NCP;
      if constexpr (TRAIN) {
#pragma unroll
        for (int c = 0; c < C::NCP; ++c) grow[c] = 0.f;
      }
      PW::apply(cc, th, srow, valid ? a.seed : 0.f, TRAIN ? 1 : 0, r, f, grow, gth);
      if (valid) {
        lsum += PW::loss(r);
        if constexpr (TRAIN) {
#pragma unroll
          for (int j = 0; j < PW::NT; ++j) tsum[j] += gth[j];
#pragma unroll
          for (int o = 0; o < C::NOUT; ++o) gbo[o] += grow[o];
        }
        if (a.resid) {
#pragma unroll
          for (int e = 0; e < PW::NEQ; ++e) a.resid[(size_t)e * a.ldj + n] = r[e];
        }
        if (a.funcs) {
#pragma unroll
          for (int m = 0; m < PW::NF; ++m) a.funcs[(size_t)m * a.ldj + n] = f[m];
        }
      }
    }
    if constexpr (TRAIN) {
      wave_lds_sync();
      wide_tile_backward<C>(un, xv, wl, pl, kept, g);
    }
    wave_lds_sync();
  }
  if constexpr (TRAIN) wide_block_reduce_store<C>(un, g, gbo, lds, wave, lane, ul, pl, a.partials + (size_t)blk * C::P);
  // loss: lanes (q == 0 lanes carry the points) -> wave -> workgroup, fixed order
  lsum = point_sum(quad_sum(lsum));
  __syncthreads();
  real* ws = lds + ((C::P + 16 + 3) & ~3);
  if (lane == 0) ws[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    real v = 0.f;
    for (int w = 0; w < C::WAVES; ++w) v += ws[w];
    a.loss_partials[blk] = v;
  }
  if constexpr (TRAIN) theta_block_sum<PW::NT>(tsum, ws + 16, C::WAVES, a.theta_partials ? a.theta_partials + (size_t)blk * PW::NT : nullptr);
}

template <class C, class PW> constexpr size_t wide_closure_lds_bytes() {
  return wide_lds_bytes<C>() + sizeof(real) * (32 + C::WAVES * (PW::NT > 0 ? PW::NT : 1));
}

template <class C, class PW, bool TRAIN>
__global__ __launch_bounds__(C::THREADS) void wide_closure_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  wide_closure_body<C, PW, TRAIN>(a, lds, blockIdx.x, gridDim.x);
}

// training batch and validation batch in one launch (fit(): solvers.py:443-497), as fused_closure_tv_kernel; no pull mode
template <class C, class PW>
__global__ __launch_bounds__(C::THREADS) void wide_closure_tv_kernel(FusedArgs t, FusedArgs v, int train_blocks, PullArgs) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  if ((int)blockIdx.x < train_blocks) wide_closure_body<C, PW, true>(t, lds, blockIdx.x, train_blocks);
  else wide_closure_body<C, PW, false>(v, lds, (int)blockIdx.x - train_blocks, (int)gridDim.x - train_blocks);
}

}  // namespace ndq
