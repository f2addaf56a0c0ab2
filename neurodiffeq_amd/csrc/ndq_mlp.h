// MI355X (gfx950) device code for the PINN training hot path: fused FCNN "jet" forward and its hand-written
// adjoint.  Replaces, for one collocation batch, what the reference does with
//   FCNN.forward                      (neurodiffeq/networks.py:59-70)
//   k reverse sweeps per diff() call  (neurodiffeq/neurodiffeq.py:21-34, operators.py:15-33)
//   loss.backward() through that twice-differentiated graph (solvers.py:393)
// by ONE forward kernel that propagates value / first / second partial-derivative "streams" of every hidden
// unit through the MLP, and ONE backward kernel that recomputes the streams and reverses the recurrences
// (math: SURVEY.md App. A.1/A.2; numpy statement of the same recurrences: oracle/jet_ref.py).
//
// Execution model (CDNA4), details in DESIGN.md section 4:
//  * a wave (64 lanes) owns a tile of 16 collocation points; lane = (p = lane&15 : point, q = lane>>4); ONE wave per
//    SIMD with the whole 512-entry register file (the f32 MFMA shares the VALU datapath on gfx950 --
//    scripts/ubench_mfma_valu.hip -- so a second wave per SIMD only buys the overlap of its bf16 MFMAs and stalls with
//    the other's VALU work: ~3 % where the state fits twice, see NDQ_BWD_THREADS).
//  * a "fragment" real4 frag[NB] holds, for point p, hidden units 16*b + 4*q + r (b < NB, r < 4) -- exactly the C/D
//    layout of the 16x16 MFMAs with units as rows and points as columns.  The 8 values a lane holds per 32 units are
//    ALSO a valid B operand of the next layer's MFMA if the contraction runs in the permuted order
//    slot(kg, e) <-> unit 16*(2c + (e>>2)) + 4*kg + (e&3), so hidden activations never leave registers between
//    layers: the weights are pre-permuted into that "fragment order" in LDS once per workgroup.
//  * per-point GEMMs (z = W h, hbar = W^T zbar; K = H) run on the bf16 matrix core with 3-way split operands
//    ("bf16x3", fp32-class accuracy, overlaps with the activation math); widths that are not a multiple of 32 fall
//    back to the exact f32 MFMA.
//  * weight gradients of hidden layers contract over POINTS, so both operands need the point index on the MFMA k
//    axis, i.e. a transpose of the fragments.  Closure / adjoint kernels of H = 32 and H = 64 networks (Cfg::WG_TR,
//    Cfg::WG_TR64; round 3): the bf16x3 planes the forward / hbar GEMMs split anyway go to LDS and come back through
//    ds_read_b64_tr_b16, six plane products on the bf16 matrix core.  Everything else (other widths, the grouped
//    closure, fp64): fp32 tiles through a padded (ld = H+4) per-wave LDS tile into the exact f32 MFMA.
//  * "Laplacian stream" (LAP = 1): when the residual needs second derivatives only through their sum, ONE stream
//    carries sum_a d2/dx_a^2 instead of one stream per coordinate.
//  * all reductions are fixed-order: DPP row rotations / lane shuffles -> per-wave LDS regions -> workgroup ->
//    partials[block][P] in HBM -> second-stage kernel.  No real atomics across waves.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>
#include "ndq_tail.h"

namespace ndq {

// The scalar type of the kernels: fp32 (the product path, north_star) or -- NDQ_F64, built into libndq64.so for the
// torch custom-op seam -- fp64, the reference's default precision (neurodiffeq/__init__.py:22).  In the fp64 build
// every per-point GEMM runs on v_mfma_f64_16x16x4_f64 (same operand layout as the f32 16x16x4), the bf16x3 paths are
// compiled out, transcendental functions come from libm.
#ifndef NDQ_F64
#define NDQ_F64 0
#endif
#if NDQ_F64
typedef double real;
#else
typedef float real;
#endif
typedef real real4 __attribute__((ext_vector_type(4)));

#if NDQ_F64
__device__ __forceinline__ real rfma(real a, real b, real c) { return fma(a, b, c); }
#else
__device__ __forceinline__ real rfma(real a, real b, real c) { return fmaf(a, b, c); }
#endif
// D (16 x 16, 4 values per lane) += A (16 x 4) B (4 x 16): one value of A and of B per lane
__device__ __forceinline__ real4 mfma16x16x4(real a, real b, real4 c) {
#if NDQ_F64
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
// Row of the A operand whose products land in "slot (q, r) = unit 4q + r" of the 16-row block a lane holds.  The f32
// 16x16x4 MFMA puts row 4q + r of D into register r of lane group q -- the fragment layout of this file; the f64 one puts
// row q + 4r there (measured: scripts/ubench_mfma_f64_layout.hip).  Hidden units may be numbered any way inside a block
// as long as every use agrees, so the fp64 build keeps the fragment layout and feeds the MFMA its A rows permuted:
// operand row i' carries unit mrow(i') = 4 (i' mod 4) + i' / 4.
__device__ __forceinline__ constexpr int mrow(int i) { return NDQ_F64 ? 4 * (i & 3) + (i >> 2) : i; }

// ------------------------------------------------------------------------------------------------ phase timestamps
// Experiments only (-DNDQ_PHASE_TS via NDQ_JIT_FLAGS; scripts/phase_ts.py): thread 0 of every workgroup records the
// shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) at up to 4 points of a closure kernel.
#ifdef NDQ_PHASE_TS
__device__ unsigned long long ndq_phase_ts[256 * 8];
#define NDQ_TS(k)                                                            \
  do {                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 256) {                              \
      ndq_phase_ts[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter();     \
      ndq_phase_ts[blockIdx.x * 8 + 4 + (k)] = wall_clock64();               \
    }                                                                        \
  } while (0)
// NDQ_TT(k): shader clock of wave 0 of workgroup 0 at point k (< 24) inside a tile (the last tile's stamps survive)
__device__ unsigned long long ndq_tile_ts[48];
__device__ int ndq_tile_iter;      // 0 while wave 0 of workgroup 0 is in its first tile, 1 afterwards
#define NDQ_TT(k)                                                                                   \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0)                                                        \
      ndq_tile_ts[24 * ndq_tile_iter + (k)] = __builtin_readcyclecounter();                         \
  } while (0)
// NDQ_PT(k): wall clock of thread 0 of workgroup 0 at point k (< 8) of the pull prologue (scripts/pull_ts.py)
__device__ unsigned long long ndq_pull_ts[8];
#define NDQ_PT(k)                                                                                   \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0) ndq_pull_ts[(k)] = wall_clock64();                     \
  } while (0)
#else
#define NDQ_TS(k)
#define NDQ_TT(k)
#define NDQ_PT(k)
#endif

// ------------------------------------------------------------------------------------------------ static for
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------------------------------------ streams
// A derivative stream is () value, (a) d/dx_a, (a,b) d2/dx_a dx_b with a <= b.  Stream order:
//   0 | 1..D (if FIRST) | the second-order pairs selected by M2 in the order (0,0),(0,1)..(0,D-1),(1,1)...
// LAP = 1 ("Laplacian stream"): the diagonal pairs (a,a) selected by M2 are NOT propagated separately; ONE stream
// carries their sum  L = sum_a d2/dx_a^2  (closed under the recurrences: h_L = s2 sum_a z_a^2 + s1 z_L, z_L = W h_L).
// Usable whenever the residual depends on those second derivatives only through that sum (Laplace, Poisson, heat,
// Navier-Stokes ...); the pointwise code generator proves this symbolically before asking for it.
// Third order (M3, bit k <-> k-th triple a <= b <= c in lexicographic order): d3/dx_a dx_b dx_c, needed by the Sobolev
// losses of second-order PDEs (losses.py:17-26) and by diff(u, t, order=3).  Recurrence (Faa di Bruno):
//   h_abc = s3 z_a z_b z_c + s2 (z_ab z_c + z_ac z_b + z_bc z_a) + s1 z_abc,   z_abc = W h_abc
// so a triple needs its three pairs in M2.
// Fourth order (M4, bit k <-> k-th quadruple a <= b <= c <= d in lexicographic order; round 6): d4/dx_a dx_b dx_c dx_d, what
// diff(u, x, order=4) asks for (beam / biharmonic / Kuramoto-Sivashinsky equations; neurodiffeq.py:21-34 has no order limit).
// Recurrence = Faa di Bruno over the set partitions of the four index POSITIONS (repeated indices count by themselves):
//   h_abcd = s4 z_a z_b z_c z_d + s3 (6 terms z_pair z z) + s2 (3 terms z_pair z_pair + 4 terms z_triple z) + s1 z_abcd
// so a quadruple needs its six pairs in M2 and its four triples in M3.
template <int D_, int FIRST_, unsigned M2_, int LAP_ = 0, unsigned M3_ = 0, unsigned M4_ = 0>
struct Streams {
  static constexpr int D = D_;
  static constexpr int FIRST = FIRST_;
  static constexpr unsigned M2 = M2_;
  static constexpr unsigned M3 = M3_;
  static constexpr unsigned M4 = M4_;
  static constexpr int LAP = LAP_;
  static constexpr int NPAIR = D * (D + 1) / 2;
  static constexpr int NTRIP = D * (D + 1) * (D + 2) / 6;
  static constexpr int count3() {
    int c = 0;
    for (int k = 0; k < NTRIP && k < 32; ++k) c += (M3 >> k) & 1u;     // (D >= 5 has more triples than mask bits: the first 32)
    return c;
  }
  static constexpr int N3 = count3();
  static constexpr int NQUAD = D * (D + 1) * (D + 2) * (D + 3) / 24;
  static constexpr int count4() {
    int c = 0;
    for (int k = 0; k < NQUAD && k < 32; ++k) c += (M4 >> k) & 1u;
    return c;
  }
  static constexpr int N4 = count4();
  static constexpr int quad_x(int k, int pos) {     // pos-th index of the k-th quadruple
    int idx = 0;
    for (int a = 0; a < D; ++a)
      for (int b = a; b < D; ++b)
        for (int c = b; c < D; ++c)
          for (int d = c; d < D; ++d) {
            if (idx == k) return pos == 0 ? a : (pos == 1 ? b : (pos == 2 ? c : d));
            ++idx;
          }
    return -1;
  }
  static constexpr int tri_x(int k, int pos) {      // pos-th index of the k-th triple
    int idx = 0;
    for (int a = 0; a < D; ++a)
      for (int b = a; b < D; ++b)
        for (int c = b; c < D; ++c) {
          if (idx == k) return pos == 0 ? a : (pos == 1 ? b : c);
          ++idx;
        }
    return -1;
  }
  static constexpr int count2() {
    int c = 0;
    for (int k = 0; k < NPAIR; ++k) c += (M2 >> k) & 1u;
    return c;
  }
  static constexpr int N2 = LAP ? 1 : count2();   // number of second-order streams
  static constexpr bool in_lap(int a) {           // coordinate a contributes to the Laplacian stream
    int idx = 0;
    for (int x = 0; x < D; ++x)
      for (int y = x; y < D; ++y) {
        if (x == a && y == a) return (M2 >> idx) & 1u;
        ++idx;
      }
    return false;
  }
  static constexpr int NS = 1 + FIRST * D + N2 + N3 + N4;
  static constexpr int S2 = 1 + FIRST * D;  // index of the first second-order stream
  static constexpr int S3 = S2 + N2;        // index of the first third-order stream
  static constexpr int S4 = S3 + N3;        // index of the first fourth-order stream
  static_assert(M4 == 0 || LAP == 0, "fourth-order streams and the Laplacian stream do not combine");
  static_assert(FIRST == 1 || M2 == 0, "second-order streams need the first-order ones");
  static_assert(M3 == 0 || LAP == 0, "third-order streams and the Laplacian stream do not combine");
  static constexpr int pair_of(int s) {  // s >= S2 -> pair index
    int c = S2;
    for (int k = 0; k < NPAIR; ++k)
      if ((M2 >> k) & 1u) {
        if (c == s) return k;
        ++c;
      }
    return -1;
  }
  static constexpr int pair_a(int k) {
    int idx = 0;
    for (int a = 0; a < D; ++a)
      for (int b = a; b < D; ++b) {
        if (idx == k) return a;
        ++idx;
      }
    return -1;
  }
  static constexpr int pair_b(int k) {
    int idx = 0;
    for (int a = 0; a < D; ++a)
      for (int b = a; b < D; ++b) {
        if (idx == k) return b;
        ++idx;
      }
    return -1;
  }
  static constexpr int A(int s) { return pair_a(pair_of(s)); }  // coordinate indices of second-order stream s
  static constexpr int B(int s) { return pair_b(pair_of(s)); }
  static constexpr int tri_of(int s) {      // s >= S3 -> triple index
    int c = S3;
    for (int k = 0; k < NTRIP && k < 32; ++k)
      if ((M3 >> k) & 1u) {
        if (c == s) return k;
        ++c;
      }
    return -1;
  }
  static constexpr int T(int s, int pos) { return tri_x(tri_of(s), pos); }   // coordinate indices of third-order stream s
  static constexpr int pair_stream(int a, int b) {   // stream index of d2/dx_a dx_b, -1 if not carried
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    int idx = 0, c = S2;
    for (int x = 0; x < D; ++x)
      for (int y = x; y < D; ++y) {
        if ((M2 >> idx) & 1u) {
          if (x == lo && y == hi) return c;
          ++c;
        }
        ++idx;
      }
    return -1;
  }
  static constexpr bool closed3() {
    for (int k = 0; k < NTRIP && k < 32; ++k)
      if ((M3 >> k) & 1u) {
        const int a = tri_x(k, 0), b = tri_x(k, 1), c = tri_x(k, 2);
        if (pair_stream(a, b) < 0 || pair_stream(a, c) < 0 || pair_stream(b, c) < 0) return false;
      }
    return true;
  }
  static_assert(closed3(), "a third-order stream needs its three second-order sub-streams");
  static constexpr int quad_of(int s) {     // s >= S4 -> quadruple index
    int c = S4;
    for (int k = 0; k < NQUAD && k < 32; ++k)
      if ((M4 >> k) & 1u) {
        if (c == s) return k;
        ++c;
      }
    return -1;
  }
  static constexpr int Q(int s, int pos) { return quad_x(quad_of(s), pos); }   // coordinate indices of fourth-order stream s
  static constexpr int tri_stream(int a, int b, int c) {   // stream index of d3/dx_a dx_b dx_c (any order of a, b, c), -1 if not carried
    int lo = a < b ? (a < c ? a : c) : (b < c ? b : c);
    int hi = a > b ? (a > c ? a : c) : (b > c ? b : c);
    int mid = a + b + c - lo - hi;
    int idx = 0, cnt = S3;
    for (int x = 0; x < D; ++x)
      for (int y = x; y < D; ++y)
        for (int z = y; z < D; ++z) {
          if (idx < 32 && ((M3 >> idx) & 1u)) {
            if (x == lo && y == mid && z == hi) return cnt;
            ++cnt;
          }
          ++idx;
        }
    return -1;
  }
  static constexpr bool closed4() {
    for (int k = 0; k < NQUAD && k < 32; ++k)
      if ((M4 >> k) & 1u) {
        int x[4] = {quad_x(k, 0), quad_x(k, 1), quad_x(k, 2), quad_x(k, 3)};
        for (int i = 0; i < 4; ++i)
          for (int j = i + 1; j < 4; ++j)
            if (pair_stream(x[i], x[j]) < 0) return false;
        for (int i = 0; i < 4; ++i)
          if (tri_stream(x[(i + 1) & 3], x[(i + 2) & 3], x[(i + 3) & 3]) < 0) return false;
      }
    return true;
  }
  static_assert(closed4(), "a fourth-order stream needs its six second-order and four third-order sub-streams");
};

// ------------------------------------------------------------------------------------------------ activations
// state kept per hidden unit: t = sigma(z) and c (only where sigma' is not a function of t).
enum { ACT_TANH = 0, ACT_SIN = 1, ACT_SIGMOID = 2, ACT_SWISH = 3, ACT_APTX = 4, ACT_ELU = 5, ACT_SOFTPLUS = 6, ACT_GELU = 7 };
// activations with a stated fourth derivative (third-order streams need it in the reverse pass)
constexpr bool act_has_s5(int act) { return act == 0 || act == 1 || act == 2; }      // tanh, sin, sigmoid (fourth-order streams)
constexpr bool act_has_s4(int act) {
  return act == ACT_TANH || act == ACT_SIN || act == ACT_SIGMOID || act == ACT_ELU || act == ACT_SOFTPLUS || act == ACT_GELU;
}

#ifndef NDQ_FAST_TANH
#define NDQ_FAST_TANH 1
#endif
#ifndef NDQ_BWD_THREADS
// 512 threads (2 waves per SIMD, 256 registers each) measured ~3 % faster on the C2 closure kernel, but every kernel
// that then spills to scratch (mlp_jet_bwd<2,1,5,..>: 28 VGPRs, <2,1,7,..>: 64) came back with a few corrupted
// workgroup rows per launch, different ones each time (tests: test_bwd_launches_are_bit_reproducible; the spill-free 1-D kernels and all
// one-wave-per-SIMD kernels, spilling or not, are bit-reproducible over thousands of launches).  Cause not found --
// two scratch-using waves on one SIMD is the common factor -- so: one wave per SIMD.
#define NDQ_BWD_THREADS 256
#endif

// Hidden-layer GEMMs on the bf16 matrix core with 3-way split operands ("bf16x3"): x = x0 + x1 + x2 (three bf16
// chunks = 24 mantissa bits), products a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1 accumulated in fp32 -> relative error
// ~2^-23, i.e. fp32-class accuracy.  Why: the f32-input MFMA shares the VALU datapath on gfx950 (it does NOT overlap
// with VALU work -- scripts/ubench_mfma_valu.hip, profiles/archive/r01/r01e_ubench_*), while v_mfma_f32_16x16x32_bf16 runs on the
// real matrix core, 6 of them replace 8 f32 MFMAs at ~1/3 of the cycles, and they overlap with the activation math.
#ifndef NDQ_BF16X3
#define NDQ_BF16X3 1
#endif
// Weight gradients: a first bf16x3 variant in round 1 (fragments transposed by MFMAs against 0/1 selection operands,
// six split products) came out 9e-5 off (rel-L2, C2 closure) and was rejected -- the transposing products themselves
// went through the bf16 matrix core.  The route of round 3 (Cfg::WG_TR / WG_TR64) transposes the bf16 PLANES through
// LDS (ds_read_b64_tr_b16: exact) and passes the gradient goldens at the level of the f32 MFMA route (1e-7 ... 6e-6,
// incl. the trained states of tests/golden/c2_trained.npz, c3_trained.npz).
// How many of the bf16 partial products of a "weights (planes a0, a1, a2) x per-point operand (planes 0, 1, 2)" GEMM are
// issued, smallest first (profiles/r06_headline_ab.md).  6 = bf16x3 (fp32 class); 5 drops a0 x plane 2 (the per-point operand
// is then exact to 2^-18 instead of 2^-27), 4 drops a1 x plane 1 as well, 3 is "bf16x2": hi*hi + hi*lo + lo*hi.
//   NDQ_FWD_NPROD   the forward GEMMs z = W h (and the multi-output layer): what the derivative STREAMS are made of.  Stays 6:
//                   with 5 the reference's trained C3 state has u_xx off by 3.2e-5 (1e-5 contract; 3.2e-6 with 6).
//   NDQ_HBAR_NPROD  the reverse GEMMs hbar = W^T zbar: gradients only.
//   NDQ_WG_NPROD    the weight-gradient GEMMs dW = sum over points of Zbar (planes I) x H (planes J): 6, 4 (no third planes), 3.
// An operand none of whose consumers reads its third plane is split into TWO planes (a third less splitting work and, for
// the transposed LDS images of the weight-gradient GEMM, a third less LDS traffic): NDQ_H_PLANES (forward activations),
// NDQ_HTR_PLANES (their transposed images), NDQ_Z_PLANES (the adjoint operand).
#ifdef NDQ_BF16_NPROD              // (round-6 experiments: one number for forward and reverse)
#define NDQ_FWD_NPROD NDQ_BF16_NPROD
#define NDQ_HBAR_NPROD NDQ_BF16_NPROD
#endif
#ifndef NDQ_FWD_NPROD
#define NDQ_FWD_NPROD 6
#endif
// Defaults: all six everywhere -- what the C-ABI's generic adjoint entry points (ndq_mlp_jet_bwd: libndq.so's table, the
// extension modules, the three-kernel pipeline, torch.ops.ndq.mlp_jet_bwd) are built with: they take ARBITRARY adjoint
// seeds, and with seeds that cancel across points (the kernel tests draw them at random) a two-plane operand's 2^-18 shows
// as 1.7e-5 .. 3.8e-5 in the weight gradients (profiles/r06_headline_ab.md, third table).  The single-launch CLOSURE modules
// (codegen.py: fused_closure / fused_multi_closure / fused_group_closure kernels), whose seeds are those of the training
// loss, define forward 6 / reverse 4 / weight gradients 3 for themselves: C2 closure 18.36 -> 16.25 us, C3 375 -> 322 us;
// gradient rel-L2 against the fp64 reference at the stated sizes 4.7e-8 / 5.5e-8 (6 / 6 / 6: 4.4e-8 / 5.2e-8), 2.5e-7 /
// 1.1e-6 on the small golden batches, 2.6e-6 (2.4e-6) at the reference's trained C3 state; streams, residuals, loss unchanged.
#ifndef NDQ_HBAR_NPROD
#define NDQ_HBAR_NPROD 6
#endif
#ifndef NDQ_WG_NPROD
#define NDQ_WG_NPROD 6
#endif
#define NDQ_PRODUCTS_6(T) T(a1, 1) T(a2, 0) T(a0, 2) T(a1, 0) T(a0, 1) T(a0, 0)
#define NDQ_PRODUCTS_5(T) T(a1, 1) T(a2, 0) T(a1, 0) T(a0, 1) T(a0, 0)
#define NDQ_PRODUCTS_4(T) T(a2, 0) T(a1, 0) T(a0, 1) T(a0, 0)
#define NDQ_PRODUCTS_3(T) T(a1, 0) T(a0, 1) T(a0, 0)
#define NDQ_CAT_(a, b) a##b
#define NDQ_CAT(a, b) NDQ_CAT_(a, b)
#define NDQ_PRODUCTS_FWD(T) NDQ_CAT(NDQ_PRODUCTS_, NDQ_FWD_NPROD)(T)
#define NDQ_PRODUCTS_BWD(T) NDQ_CAT(NDQ_PRODUCTS_, NDQ_HBAR_NPROD)(T)
#if NDQ_FWD_NPROD < 3 || NDQ_FWD_NPROD > 6 || NDQ_HBAR_NPROD < 3 || NDQ_HBAR_NPROD > 6
#error "NDQ_FWD_NPROD / NDQ_HBAR_NPROD must be 3, 4, 5 or 6"
#endif
#if NDQ_WG_NPROD == 6
#define NDQ_WPRODUCTS(W) W(1, 1) W(2, 0) W(0, 2) W(1, 0) W(0, 1) W(0, 0)
#elif NDQ_WG_NPROD == 4
#define NDQ_WPRODUCTS(W) W(1, 1) W(1, 0) W(0, 1) W(0, 0)
#elif NDQ_WG_NPROD == 3
#define NDQ_WPRODUCTS(W) W(1, 0) W(0, 1) W(0, 0)
#else
#error "NDQ_WG_NPROD must be 3, 4 or 6"
#endif
#define NDQ_HTR_PLANES ((NDQ_WG_NPROD == 6) ? 3 : 2)
#define NDQ_H_PLANES ((NDQ_FWD_NPROD == 6 || NDQ_WG_NPROD == 6) ? 3 : 2)
#define NDQ_Z_PLANES ((NDQ_HBAR_NPROD == 6 || NDQ_WG_NPROD == 6) ? 3 : 2)
#ifndef NDQ_WIDE_LOWREG
#define NDQ_WIDE_LOWREG 1
#endif
#ifndef NDQ_PIN_WEIGHT_READS
#define NDQ_PIN_WEIGHT_READS 1
#endif
#ifndef NDQ_WIDE_SG
#define NDQ_WIDE_SG 2    // streams whose bf16 planes are live at a time in the wide-net GEMMs
#endif
#ifndef NDQ_KEEP_H
#define NDQ_KEEP_H 1
#endif
#ifndef NDQ_FWD_THREADS
#define NDQ_FWD_THREADS 256
#endif
#ifndef NDQ_WG_SCHED_BARRIER
#define NDQ_WG_SCHED_BARRIER 1
#endif
// Ablation switches (experiments only, wrong results by design): NDQ_ABL bit 0 = no weight-gradient GEMMs, bit 1 = no
// hbar GEMM, bit 2 = no forward hidden GEMM, bit 3 = no act_backward, bit 4 = no LDS transposes (MFMAs on stale data),
// bit 5 = no operand splitting (planes reused).  scripts/ablate.py times the closure kernel with each of them.
#ifndef NDQ_WIDE_LAUNDER
#define NDQ_WIDE_LAUNDER 1
#endif
#ifndef NDQ_WG_BF16
#define NDQ_WG_BF16 0     // narrow nets: weight-gradient GEMMs on the bf16 matrix core (split operands) instead of f32 MFMAs
#endif
#ifndef NDQ_WG_TR
#define NDQ_WG_TR 0       // narrow nets, single-launch closure kernels (set by codegen.py for those modules): hidden-layer
                          // weight gradients on the bf16 matrix core from the bf16x3 planes the forward / hbar GEMMs split
                          // anyway, transposed by ds_read_b64_tr_b16 (Cfg::WG_TR)
#endif
#ifndef NDQ_MULTI_G2
#define NDQ_MULTI_G2 2      // tile slots per workgroup for K = 2 (experiments: 4 = two waves per SIMD)
#endif
#ifndef NDQ_WG_TR_K
#define NDQ_WG_TR_K 1     // networks of the closure kernel the module is built for (multi-network closure: K weight images
                          // and K x G staging regions share the workgroup's LDS)
#endif
#ifndef NDQ_FWD_BF16X1
#define NDQ_FWD_BF16X1 0  // opt-in (a library of its own: NDQ_LIB_FLAGS=-DNDQ_FWD_BF16X1=1): the hidden-layer GEMMs of the forward
                          // STREAM kernel (mlp_jet_fwd_kernel, three-kernel pipeline) on single bf16 operands -- BASELINE config 5's
                          // "bf16 fwd / fp32 grad"; the adjoint kernels recompute their forward pass in bf16x3 as always
#endif
#ifndef NDQ_QUAD_SWAP
#define NDQ_QUAD_SWAP 1   // quad_sum through v_permlane16/32_swap instead of ds_bpermute: same bits.  Round 5 measured it NOT faster
                          // (C2 closure 19.5 vs 19.2 us: the LDS round trip was not on the critical path then); with the reverse
                          // pass's products and planes cut down (round 6) it is: 17.52 -> 17.03 us on one box, 16.23 -> 16.09 on
                          // another (profiles/r06h_flags_ab_c2.txt, r06j_own_tile_ab.txt) -- on.  0: the ds_bpermute route.
#endif
#ifndef NDQ_SPLIT_PAIRS
#define NDQ_SPLIT_PAIRS 0   // split3 on 2-wide vectors: 3.5 % fewer VALU instructions in C3's closure kernel, no time gained
                            // (C3 418.4 -> 419.8 us, C2 8-wave 19.9 -> 19.7 us on MI355X): off
#endif
#ifndef NDQ_WG32
#define NDQ_WG32 1        // wide nets: weight-gradient GEMMs as split-operand v_mfma_f32_32x32x16_bf16 (one stream per instruction)
#endif
#ifndef NDQ_ABL
#define NDQ_ABL 0
#endif
#ifndef NDQ_STAGE_INFLIGHT
#define NDQ_STAGE_INFLIGHT 1   // 0: weight staging array by array (rounds 1 - 4; A/B of stage_weights' in-flight pass)
#endif
#ifndef NDQ_GROUP_PREFETCH
#define NDQ_GROUP_PREFETCH 1   // 0: grouped closure kernel loads a group's coordinates at the top of the group loop (A/B)
#endif
#ifndef NDQ_STAGGER
#define NDQ_STAGGER 0       // s_sleep argument (x 64 cycles) by which waves WAVES/2.. start their tile loop late
#endif

// tanh z = 1 - 2 / (2^(2 z log2 e) + 1): one v_exp_f32 + one v_rcp_f32 (both ~1 ulp).  Absolute error <= ~1.5e-7
// over the whole range (saturates correctly: e -> inf gives 1, e -> 0 gives -1); the libm tanhf costs ~10x the
// VALU issue slots, and the VALU is what competes with the MFMA pipe in these kernels.
__device__ __forceinline__ real tanh_fast(real z) {
#if NDQ_F64
  return tanh(z);
#else
  const real e = __builtin_amdgcn_exp2f(z * 2.88539008177792681472f);
  return rfma(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
#endif
}

template <int ACT> struct Act;
template <> struct Act<ACT_TANH> {  // nn.Tanh, networks.py:27 default
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) {
#if NDQ_FAST_TANH || NDQ_F64
    t = tanh_fast(z);
#else
    t = tanhf(z);
#endif
    c = 0.f;
  }
  static __device__ __forceinline__ real s1(real t, real, real = 1.f) { return rfma(-t, t, 1.f); }
  static __device__ __forceinline__ real s2(real t, real, real s1v) { return -2.f * t * s1v; }
  // third derivative -2 s1 (1 - 3 t^2); with t^2 = 1 - s1 that is s1 (4 - 6 s1): two instructions instead of four
  static __device__ __forceinline__ real s3(real, real, real s1v) { return s1v * rfma(-6.f, s1v, 4.f); }
  // fourth derivative: -2 s2 (1 - 3 t^2) + 12 t s1^2 with s2 = -2 t s1, i.e. t s1 (24 s1 - 8)
  static __device__ __forceinline__ real s4(real t, real, real s1v) { return t * s1v * rfma(24.f, s1v, -8.f); }
  // d/dz = (1 - t^2) d/dt: sigma^(5) = s1 (16 - 120 t^2 + 120 t^4) = s1 (16 - 120 s1 + 120 s1^2)
  static __device__ __forceinline__ real s5(real, real, real s1v) { return s1v * rfma(s1v, rfma(120.f, s1v, -120.f), 16.f); }
};
template <> struct Act<ACT_SIN> {  // SinActv, networks.py:142-152
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) {
#if NDQ_F64
    sincos(z, &t, &c);
#else
    sincosf(z, &t, &c);
#endif
  }
  static __device__ __forceinline__ real s1(real, real c, real = 1.f) { return c; }
  static __device__ __forceinline__ real s2(real t, real, real) { return -t; }
  static __device__ __forceinline__ real s3(real, real c, real) { return -c; }
  static __device__ __forceinline__ real s4(real t, real, real) { return t; }
  static __device__ __forceinline__ real s5(real, real c, real) { return c; }
};

__device__ __forceinline__ real sigmoid_fast(real z) {   // 1 / (1 + 2^(-z log2 e)); saturates cleanly to 0 / 1
#if NDQ_F64
  return 1.0 / (1.0 + exp(-z));
#else
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
#endif
}
template <> struct Act<ACT_SIGMOID> {  // torch.nn.Sigmoid as FCNN(actv=nn.Sigmoid): everything is a polynomial in t
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) { t = sigmoid_fast(z); c = 0.f; }
  static __device__ __forceinline__ real s1(real t, real, real = 1.f) { return t * (1.f - t); }
  static __device__ __forceinline__ real s2(real t, real, real s1v) { return s1v * rfma(-2.f, t, 1.f); }
  static __device__ __forceinline__ real s3(real, real, real s1v) { return s1v * rfma(-6.f, s1v, 1.f); }
  static __device__ __forceinline__ real s4(real t, real, real s1v) {      // s2 (1 - 12 s1)
    return s1v * rfma(-2.f, t, 1.f) * rfma(-12.f, s1v, 1.f);
  }
  // d/dz [s2 (1 - 12 s1)] = s3 (1 - 12 s1) - 12 s2^2, (1 - 2 t)^2 = 1 - 4 s1:  s1 (1 - 30 s1 + 120 s1^2)
  static __device__ __forceinline__ real s5(real, real, real s1v) { return s1v * rfma(s1v, rfma(120.f, s1v, -30.f), 1.f); }
};
// Swish with the default fixed beta = 1 (networks.py:155-175): f = z sigma(z).  State: t = f, c = sigma(z); since
// z sigma = t the derivatives need no z:  f1 = c + t(1-c),  f2 = (1-c)(2c + t(1-2c)),  f3 = (1-c)(3c(1-2c) + t(1-6c+6c^2))
template <> struct Act<ACT_SWISH> {
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) { c = sigmoid_fast(z); t = z * c; }
  static __device__ __forceinline__ real s1(real t, real c, real = 1.f) { return rfma(t, 1.f - c, c); }
  static __device__ __forceinline__ real s2(real t, real c, real) {
    return (1.f - c) * rfma(t, rfma(-2.f, c, 1.f), 2.f * c);
  }
  static __device__ __forceinline__ real s3(real t, real c, real) {
    return (1.f - c) * rfma(t, rfma(6.f * c, c - 1.f, 1.f), 3.f * c * rfma(-2.f, c, 1.f));
  }
};

// APTx with its default fixed parameters alpha = 1, beta = 1, gamma = 1/2 (networks.py:177-209): f = z (1 + tanh z) / 2.
// State: t = f, c = z (z cannot be recovered from f and tanh z where 1 + tanh z underflows); with T = tanh z
//   f1 = (1 + T)/2 + z (1 - T^2)/2,  f2 = (1 - T^2)(1 - z T),  f3 = (1 - T^2)(3 z T^2 - 3 T - z)
//
// Trainable parameters (Cfg::ACTP): the tile loop always evaluates the UNIT-SCALE function G -- swish: u sigma(u);
// APTx: u (al + tanh u) / 2 -- on u = beta z; the scales live in the weights.  f(z) = o G(beta z) with o = 1 / beta
// (swish) or 2 gamma / beta (APTx), so stage_weights() loads W_l' = beta_l o_{l-1} W_l, b_l' = beta_l b_l,
// Wout' = o_L Wout and block_reduce_store() scales the weight gradients back (dW = f dW').  alpha is a shape
// parameter, the one value the activation code reads at run time (LayerState::al).  The parameters' own gradients
// come out of act_backward() as three per-layer sums over units and points of the layer's unit-scale streams h = G(u),
// their adjoints g and the pre-activation adjoints ubar it produces (GradAcc::ap):
//   d alpha = sum_s g_s u_s / 2        (alpha enters h through alpha u / 2 only, every stream passes straight through)
//   d gamma = sum_s g_s h_s / gamma    (f is linear in gamma)
//   d beta  = (sum_t ubar_t u_t - sum_s g_s h_s) / beta      (Euler: f(z; beta) = G(beta z) / beta is homogeneous)
// accumulated element by element, so what cancels are the small per-element differences, not the layer totals.
template <> struct Act<ACT_APTX> {
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real al = 1.f) { c = z; t = 0.5f * z * (al + tanh_fast(z)); }
  static __device__ __forceinline__ real s1(real, real z, real al = 1.f) {
    const real T = tanh_fast(z);
    return 0.5f * rfma(z, rfma(-T, T, 1.f), al + T);
  }
  static __device__ __forceinline__ real s2(real, real z, real) {
    const real T = tanh_fast(z);
    return rfma(-T, T, 1.f) * rfma(-z, T, 1.f);
  }
  static __device__ __forceinline__ real s3(real, real z, real) {
    const real T = tanh_fast(z);
    return rfma(-T, T, 1.f) * rfma(3.f * z * T, T, rfma(-3.f, T, -z));
  }
};

// torch.nn.ELU (alpha = 1; tests/test_pde.py:377 of the reference trains FCNN(hidden_units=(100, 100), actv=nn.ELU)):
// f = z (z > 0), e^z - 1 (z <= 0); every derivative is 1, 0, 0 ... for z > 0 and e^z = f + 1 for z <= 0.  State: t = f
// alone (t > 0 <=> z > 0).  (The second derivative jumps at 0: that is the activation the user chose, the reference
// differentiates it too.)
template <> struct Act<ACT_ELU> {
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) {
#if NDQ_F64
    t = z > 0. ? z : expm1(z);
#else
    t = z > 0.f ? z : expm1f(z);
#endif
    c = 0.f;
  }
  static __device__ __forceinline__ real s1(real t, real, real = 1.f) { return t > 0.f ? (real)1.f : t + 1.f; }
  static __device__ __forceinline__ real s2(real t, real, real) { return t > 0.f ? (real)0.f : t + 1.f; }
  static __device__ __forceinline__ real s3(real t, real, real) { return t > 0.f ? (real)0.f : t + 1.f; }
  static __device__ __forceinline__ real s4(real t, real, real) { return t > 0.f ? (real)0.f : t + 1.f; }
};
// torch.nn.Softplus (beta = 1, threshold = 20): f = log(1 + e^z), f1 = sigmoid(z) =: c, f2 = c (1 - c),
// f3 = c (1 - c)(1 - 2c), f4 = f2 (1 - 6 f2).  State: t = f, c = sigmoid(z).  (Above the threshold torch returns z and the
// derivative 1; sigmoid(20) = 1 - 2e-9 is 1 in fp32.)
template <> struct Act<ACT_SOFTPLUS> {
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) {
    c = sigmoid_fast(z);
#if NDQ_F64
    t = z > 20. ? z : log1p(exp(z));
#else
    t = z > 20.f ? z : log1pf(expf(z));
#endif
  }
  static __device__ __forceinline__ real s1(real, real c, real = 1.f) { return c; }
  static __device__ __forceinline__ real s2(real, real c, real) { return c * (1.f - c); }
  static __device__ __forceinline__ real s3(real, real c, real) { return c * (1.f - c) * rfma(-2.f, c, 1.f); }
  static __device__ __forceinline__ real s4(real, real c, real) {
    const real d = c * (1.f - c);
    return d * rfma(-6.f, d, 1.f);
  }
};
// torch.nn.GELU (approximate = 'none'): f = z Phi(z) with Phi the normal CDF, phi its density:
// f1 = Phi + z phi, f2 = phi (2 - z^2), f3 = phi (z^3 - 4 z), f4 = phi (-z^4 + 7 z^2 - 4).  State: t = f, c = z.
template <> struct Act<ACT_GELU> {
  static __device__ __forceinline__ real cdf(real z) {
#if NDQ_F64
    return 0.5 * (1.0 + erf(z * 0.70710678118654752440));
#else
    return 0.5f * (1.f + erff(z * 0.70710678118654752440f));
#endif
  }
  static __device__ __forceinline__ real pdf(real z) {
#if NDQ_F64
    return 0.39894228040143267794 * exp(-0.5 * z * z);
#else
    return 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);
#endif
  }
  static __device__ __forceinline__ void fwd(real z, real& t, real& c, real = 1.f) { c = z; t = z * cdf(z); }
  static __device__ __forceinline__ real s1(real, real z, real = 1.f) { return rfma(z, pdf(z), cdf(z)); }
  static __device__ __forceinline__ real s2(real, real z, real) { return pdf(z) * rfma(-z, z, 2.f); }
  static __device__ __forceinline__ real s3(real, real z, real) { return pdf(z) * z * rfma(z, z, -4.f); }
  static __device__ __forceinline__ real s4(real, real z, real) {
    const real z2 = z * z;
    return pdf(z) * rfma(z2, 7.f - z2, -4.f);
  }
};

// ------------------------------------------------------------------------------------------------ config
template <int D_, int FIRST_, unsigned M2_, int NB_, int L_, int ACT_, int NOUT_ = 1, int LAP_ = 0, int SKIP_ = 0,
          unsigned M3_ = 0, int ACTP_ = 0, int HR_ = 0, unsigned HRP_ = 0, unsigned MONO_ = 0, unsigned M4_ = 0>
struct Cfg {
  using SS = Streams<D_, FIRST_, M2_, LAP_, M3_, M4_>;
  static_assert(M3_ == 0 || act_has_s4(ACT_), "third-order streams: activations with a stated fourth derivative");
  static_assert(M4_ == 0 || (act_has_s5(ACT_) && ACTP_ == 0 && MONO_ == 0),
                "fourth-order streams: tanh / sin / sigmoid, no trainable activation parameters, no monomial features");
  static constexpr int D = D_, NB = NB_, H = 16 * NB_, L = L_, ACT = ACT_, NS = SS::NS;
  // MONO: a networks.MonomialNN (networks.py:109-139) in front of the first linear layer -- the D coordinates are
  // expanded to the features x_a^deg, degree after degree (bit k of MONO <-> degree k + 1, ascending), so the first
  // layer takes NIN = D * NDEG inputs and its derivative streams are no longer constant columns of W1:
  //   z = b1 + sum_f W1[:, f] x_a^deg,  z_a = sum_deg W1[:, (deg, a)] deg x_a^(deg-1),  z_aa = ... deg (deg-1) x_a^(deg-2),
  // mixed second derivatives stay zero.
  static constexpr unsigned MONO = MONO_;
  static constexpr int mono_count() { int c = 0; for (int k = 0; k < 8; ++k) c += (MONO_ >> k) & 1u; return c; }
  static constexpr int NDEG = mono_count();
  static constexpr int NIN = MONO_ != 0 ? D_ * NDEG : D_;          // inputs of the first linear layer
  static constexpr int mono_deg(int i) {                           // i-th degree, ascending
    int c = 0;
    for (int k = 0; k < 8; ++k)
      if ((MONO_ >> k) & 1u) { if (c == i) return k + 1; ++c; }
    return 0;
  }
  static constexpr int MAXDEG = MONO_ != 0 ? mono_deg(NDEG - 1) : 1;
  static_assert(MONO_ < 256u && (MONO_ == 0 || (M3_ == 0 && SKIP_ == 0 && NB_ <= 3)),
                "monomial features: degrees 1..8, up to second-order streams, no skip connection, H <= 48");
  // HR: the network's real hidden width when it is no multiple of 16 (HR_ = 0: H).  Registers, fragments and LDS images
  // are laid out for the padded width H; the padding units have zero weights in LDS (whatever their activation value,
  // nothing downstream sees it) and no slot in the flat parameter / gradient vectors, which are indexed with HR.
  // HRP: hidden layers of DIFFERENT widths, 8 bits per layer (layer 1 in the low byte); 0: every layer HR wide.  All
  // layers are laid out for the widest one.
  static constexpr int HR = HR_ > 0 ? HR_ : H;
  static constexpr unsigned HRP = HRP_;
  static constexpr int hr(int l) { return HRP_ != 0 ? (int)((HRP_ >> (8 * (l - 1))) & 255u) : HR; }    // l in 1..L
  static constexpr int hr_max() { int m = 0; for (int l = 1; l <= L_; ++l) m = hr(l) > m ? hr(l) : m; return m; }
  static constexpr int hr_min() { int m = 1 << 20; for (int l = 1; l <= L_; ++l) m = hr(l) < m ? hr(l) : m; return m; }
  static_assert(hr_max() <= H && hr_max() > H - 16 && hr_min() >= 1, "real widths and padded width disagree");
  static_assert(HRP_ == 0 || HR == hr_max(), "packed widths: HR is the widest layer");
  static constexpr bool RAGGED = hr_min() != H;
  static constexpr int NOUT = NOUT_;               // output units; > 1: the output layer is an MFMA layer too
  static constexpr int NBO = (NOUT_ + 15) / 16;    // 16-row blocks of the (zero-padded) output layer
  static constexpr int HO = 16 * NBO;
  static constexpr int HP = (H > HO ? H : HO) + 4;  // padded leading dimension of the transpose staging tiles
  // workgroup sizes: 8 waves (2 per SIMD, <= 256 registers each) when the per-wave state fits, else 4 waves with the
  // whole 512-entry register file per wave
  static constexpr int BWD_THREADS =
      (NDQ_F64 || NB_ * NB_ * (L_ - 1) + NB_ * SS::NS * L_ + (NOUT_ > 1 ? NB_ * NBO : 0) > 40) ? 256 : NDQ_BWD_THREADS;
  static constexpr int FWD_THREADS = (NB_ >= 4) ? 256 : NDQ_FWD_THREADS;
  // flat parameter offsets, torch order: W1 (H,D) b1 (H) | W_l (H,H) b_l (H), l = 2..L | Wout (1,H) bout (1)
  static constexpr int offW1 = 0, offb1 = hr(1) * NIN;
  static constexpr int offW(int l) {       // l in 2..L + 1 (L + 1: the output matrix)
    int o = hr(1) * NIN + hr(1);
    for (int k = 2; k < l; ++k) o += hr(k) * hr(k - 1) + hr(k);
    return o;
  }
  static constexpr int offb(int l) { return offW(l) + hr(l) * hr(l - 1); }
  static constexpr int offWout = offW(L + 1);
  static constexpr int offbout = offWout + NOUT * hr(L);
  // SKIP: a trainable bias-free linear map from the inputs straight to the output, out += S x (networks.Resnet,
  // networks.py:73-106); its weights S (n_out x d) follow the output bias in the flat parameter vector
  static constexpr int SKIP = SKIP_;
  static constexpr int offS = offbout + NOUT;
  // ACTP = 1: trainable activation parameters (networks.Swish(trainable=True): beta; networks.APTx(trainable=True):
  // alpha, beta, gamma -- networks.py:155-209), one set per hidden layer, behind everything else in the flat vector.
  // ACTP = 2: the same scalars as FIXED non-default values (Swish(beta=2.0)): they follow the P trainable entries in
  // the parameter buffer and have no gradient slots.
  static constexpr int ACTP = ACTP_;
  static constexpr int AK = (ACT_ == ACT_SWISH) ? 1 : (ACT_ == ACT_APTX) ? 3 : 0;
  static_assert(ACTP_ == 0 || AK > 0, "trainable activation parameters: Swish / APTx");
  static constexpr bool ALPHA = (ACTP_ != 0) && (ACT_ == ACT_APTX);
  static constexpr int offA = offS + SKIP * NOUT * D;
  static constexpr int P = offA + (ACTP_ == 1 ? AK * L : 0);
  // LDS carve (floats): W1T [D][H] | b1 [H] | per hidden-hidden layer: Wf [H*H] (+ Wt [H*H] for bwd) | b_l | Wout | bout
  static constexpr int ldsW1T = 0, ldsb1 = NIN * H;
  static constexpr int ldsLayer0 = NIN * H + H;
  // hidden GEMM operand format: bf16x3 planes (3 x 2 B per weight) when the width is a multiple of 32, else f32
  static constexpr bool BF16 = (NDQ_BF16X3 != 0) && (NB_ % 2 == 0) && (NDQ_F64 == 0);
  static constexpr int NC = NB_ / 2;                       // K-chunks of 32 contraction slots (bf16 path)
  static constexpr int WEL = BF16 ? (H * H * 3) / 2 : H * H;   // floats of LDS per weight matrix image
  // multi-output networks: the output layer (HO x H, zero-padded rows) runs on the bf16 matrix core as well when its
  // padded height is a multiple of 32 (its rows are the contraction axis of hbar = Wout^T gout)
  static constexpr bool BF16O = BF16 && (NOUT_ > 1) && (NBO % 2 == 0);
  static constexpr int NCO = NBO / 2;
  static constexpr int WOEL = BF16O ? (HO * H * 3) / 2 : HO * H;   // floats of LDS per output-layer image
  // (two-waves-per-SIMD builds have no registers to keep them in: they recompute, and tile_backward hides the layer states
  // behind an opaque copy so that the compiler does not quietly keep the forward pass's values alive instead)
  static constexpr bool KEEP_H = (NB_ == 2) && (NDQ_KEEP_H != 0) && (SS::NS <= 6) && (BWD_THREADS == 256);
  static constexpr bool LAUNDER = ((NB_ == 2) && (BWD_THREADS != 256)) || (NB_ >= 4 && NDQ_WIDE_LAUNDER);
  // wide nets (H >= 64): the reverse pass is register-bound, so (a) the per-point GEMMs go through their bf16 planes
  // SG streams at a time instead of all at once, (b) the bias-type gradient sums (db_l, dW1, dWout: one value per
  // unit) live in a per-wave LDS region instead of registers, (c) the first layer's derivative streams (columns of
  // W1) are re-read from LDS for the reverse pass instead of being kept.  (Recomputing the middle layer's state in
  // the reverse pass instead of keeping it was tried too: no fewer spills, 16 % slower -- rejected.)
  static constexpr bool WIDE = (NB_ >= 4) && (NDQ_WIDE_LOWREG != 0);
  // ... and the 8-wave builds of narrow nets (256 registers per wave) run the reverse GEMM two streams at a time as well
  static constexpr bool GROUP_HBAR = WIDE || (BWD_THREADS != 256 && SS::NS > 2);
  static constexpr int SG = GROUP_HBAR ? NDQ_WIDE_SG : SS::NS;
  static constexpr bool ACC_LDS = WIDE && (NOUT_ == 1);
  static constexpr int biasFloats = ACC_LDS ? H * (D_ + L_ + 1) : 0;      // b1 | W1 [D] | b_2..b_L | Wout
  static constexpr int biasB1 = 0, biasW1 = H, biasBl = H * (1 + D_), biasWout = H * (D_ + L_);
  static constexpr int layerStride(bool bwd) { return (bwd ? 2 : 1) * WEL + H; }
  static constexpr int ldsWf(int l, bool bwd) { return ldsLayer0 + (l - 2) * layerStride(bwd); }
  static constexpr int ldsWt(int l) { return ldsWf(l, true) + WEL; }
  static constexpr int ldsb(int l, bool bwd) { return ldsWf(l, bwd) + (bwd ? 2 : 1) * WEL; }
  // output layer: NOUT == 1: Wout [H] | bout [1];  NOUT > 1: fragment-ordered Wo [HO*H] (+ transposed [HO*H]) | bout [HO]
  static constexpr int ldsWout(bool bwd) { return ldsLayer0 + (L - 1) * layerStride(bwd); }
  static constexpr int ldsWoutT() { return ldsWout(true) + WOEL; }
  static constexpr int ldsbout(bool bwd) { return ldsWout(bwd) + (NOUT == 1 ? H : (bwd ? 2 : 1) * WOEL); }
  static constexpr int ldsSkip(bool bwd) { return ldsbout(bwd) + (NOUT == 1 ? 1 : HO); }
  // skip weights: NOUT == 1: S [D];  NOUT > 1: transposed and zero-padded, St [D][HO];  then APTx's alpha per layer [L]
  static constexpr int ldsAlpha(bool bwd) { return ldsSkip(bwd) + SKIP * D * (NOUT == 1 ? 1 : HO); }
  static constexpr int ldsWeightsEnd(bool bwd) { return (ldsAlpha(bwd) + (ALPHA ? L : 0) + 3) & ~3; }
  // weight-gradient transposes: streams staged per barrier round (narrow nets: two at a time, so that the LDS round
  // trip of one stream hides behind the MFMAs of the other; wide nets: no LDS to spare)
  // WG_BF16 (narrow nets on the bf16x3 path): the weight-gradient GEMM dW += sum_s Zbar_s H_s^T runs on
  // v_mfma_f32_16x16x32_bf16 as well -- one instruction contracts over 32 slots = 2 streams x 16 points, so two streams
  // are staged per round; 6 split products of 16 cycles replace 16 exact-f32 MFMAs of 32 cycles per pair of 16x16 blocks
  static constexpr bool WG_BF16 = BF16 && (NB_ <= 2) && (NDQ_WG_BF16 != 0);
  // WG32 (wide nets, H = 64): the hidden-layer weight gradients on v_mfma_f32_32x32x16_bf16 -- K = 16 is exactly the 16
  // points of ONE stream, so no second stream has to be staged (there is no LDS left for it); per stream 4 macro-blocks
  // x 6 split products of 32 cycles (768) replace 64 exact-f32 MFMAs of 32 cycles (2 048); the accumulators are 32x32
  // blocks (GradAcc::w32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
  static constexpr bool WG32 = BF16 && (NB_ == 4) && (NDQ_WG32 != 0);
  static constexpr int WG_SB = (NB_ <= 2 && SS::NS >= 2 && (BWD_THREADS == 256 || WG_BF16)) ? 2 : 1;
  static constexpr int stageFloatsPlain = WG_SB * 2 * 16 * HP;    // Zt and Ht tiles of WG_SB streams
  // WG_TR (H = 32 on the bf16x3 path, closure kernels built with NDQ_WG_TR): dW_l += sum_s Zbar_s H_s^T contracts over
  // POINTS, which the fragment layout keeps in lanes -- both operands have to be transposed.  The exact-f32 route stages
  // fp32 tiles in LDS and feeds 64 v_mfma_f32_16x16x4_f32 per tile (2 048 cycles that overlap with nothing, 4.0).  But
  // both operands exist as bf16x3 planes already: H_s was split for the forward GEMM z = W h, Zbar_s is split for
  // hbar = W^T zbar.  The planes go to LDS as [point][unit] images -- a lane's bf16x8 is one ds_write_b128 -- and come
  // back through ds_read_b64_tr_b16, which hands lane i the 4 points of unit i (a 4 x 4 transpose per read, no VALU):
  // two reads make the 8 contraction slots of a lane, one v_mfma_f32_16x16x32_bf16 contracts 2 streams x 16 points, and
  // the six significant plane products of 16 cycles (48 per tile, hidden under VALU work like every bf16 MFMA) replace
  // the 64 f32 MFMAs, the fp32 staging reads and -- in the 8-wave build -- the recomputation of H_s in the reverse pass.
  // Per wave: (L - 1) H images of NS streams, kept from the forward pass of the tile, + one Zbar image of 2 streams.
  static constexpr int trPlane = 256;                      // floats of one plane image: 16 points x 32 units x 2 B
  static constexpr int trHimg = (L_ - 1) * SS::NS * 3 * trPlane;
  static constexpr int trStage = trHimg + 2 * 3 * trPlane;
  static constexpr int trWaves = NDQ_WG_TR_K == 1 ? BWD_THREADS / 64 : (NDQ_WG_TR_K == 2 ? NDQ_MULTI_G2 : 1);   // per network
  static constexpr bool WG_TR = BF16 && (NB_ == 2) && (L_ >= 2) && (NOUT_ == 1) && (NDQ_WG_TR != 0) && !WG_BF16 &&
                                (NDQ_WG_TR_K * (ldsWeightsEnd(true) + trWaves * trStage) +
                                 (NDQ_WG_TR_K > 1 ? 2 * trWaves * NDQ_WG_TR_K * SS::NS * 16 : 0) + 64 <= 40 * 1024);
  // WG_TR64 (H = 64, same switch): the 32x32x16 weight-gradient GEMM of WG32 fed the same way.  There is no LDS to keep
  // the forward pass's H planes, so per stream the layer input is recomputed and split in the forward GEMM's operand
  // layout (16 values per lane instead of the 2 x 16 both operands cost when they are read back as fp32 tiles), the
  // Zbar planes are the ones hbar = W^T zbar needs anyway; one Zbar and one H image of one stream per wave.
  static constexpr int trPlane64 = 512;                    // 16 points x 64 units x 2 B
  static constexpr int trStage64 = 2 * 3 * trPlane64;
  static constexpr bool WG_TR64 = WG32 && (NOUT_ == 1) && (NDQ_WG_TR != 0) && (NDQ_WG_TR_K == 1) &&
                                  (ldsWeightsEnd(true) + (BWD_THREADS / 64) * (trStage64 + biasFloats) + 64 <= 40 * 1024);
  static constexpr int stageFloatsPerWave = WG_TR ? trStage : WG_TR64 ? trStage64 : stageFloatsPlain;
};

struct MlpArgs {
  const real* coords;   // [D][ldc]  SoA collocation coordinates
  const real* params;   // [P] flat, torch parameter order
  const real* gbar;     // bwd: [NS][NOUT][ldj] adjoint of every output stream
  real* jets;           // fwd: [NS][NOUT][ldj] output streams of the raw network
  real* partials;       // bwd: [gridDim.x][P] per-workgroup parameter-gradient partial sums
  int n;                 // number of points
  int ldc;               // leading dimension of coords
  int ldj;               // leading dimension of jets / gbar
};

// ------------------------------------------------------------------------------------------------ weight staging
// Cfg::ACTP, scale factors of the trainable activation parameters (see Act<ACT_APTX>): the pre-activation of layer l
// (1..L) is multiplied by act_pre(l) = beta_l, its activations by act_post(l) = 1 / beta_l (swish), 2 gamma_l / beta_l
// (APTx); act_post(0) = 1 (the inputs).
template <class C>
__device__ __forceinline__ real act_pre(const real* __restrict__ prm, int l) {
  if constexpr (C::ACTP == 0) return 1.f;
  else return prm[C::offA + (l - 1) * C::AK + (C::AK == 3 ? 1 : 0)];
}
template <class C>
__device__ __forceinline__ real act_post(const real* __restrict__ prm, int l) {
  if constexpr (C::ACTP == 0) return 1.f;
  else {
    if (l == 0) return 1.f;
    const real beta = act_pre<C>(prm, l);
    if constexpr (C::AK == 3) return 2.f * prm[C::offA + (l - 1) * 3 + 2] / beta;
    else return 1.f / beta;
  }
}
template <class C>
__device__ __forceinline__ real actp_mul(real v, real f) {
  if constexpr (C::ACTP != 0) return v * f;
  else return v;
}

// tid / nt: this thread's index among the nt threads that stage the image together (default: the whole workgroup; the
// multi-network closure lets every network's own waves stage that network's image, all images at once)
template <class C, bool BWD>
__device__ __forceinline__ void stage_weights(real* lds, const real* __restrict__ prm, int tid = -1, int nt = 0) {
  constexpr int H = C::H, D = C::D, NB = C::NB;
  constexpr int HL = C::hr(C::L);          // width of the last hidden layer (input of the output layer)
  if (tid < 0) { tid = threadIdx.x; nt = blockDim.x; }
  const real f1 = act_pre<C>(prm, 1), fo = act_post<C>(prm, C::L);
  // padding units (RAGGED: j >= the layer's real width hw) read as zero
  auto unit = [&](int j, int hw, int idx, real f) {
    if constexpr (C::RAGGED) return j < hw ? actp_mul<C>(prm[idx], f) : (real)0.f;
    else return actp_mul<C>(prm[idx], f);
  };
  auto real_unit = [](int j, int hw) { return !C::RAGGED || j < hw; };
#if NDQ_STAGE_INFLIGHT
  // ---- plain FCNNs on the bf16x3 path (the BASELINE shapes): ONE pass over the flat parameter vector in torch order with the
  // loads of a pass all in flight (round 5).  The loops below stage array by array -- W1, b1, Wout, bout, W_l, b_l -- and every
  // loop compiles to "load, s_waitcnt vmcnt(0), ds_write, branch": five (C2) dependent round trips to memory that the previous
  // step's tail launch wrote on other XCDs, 1.35 us of every launch.  Here a thread loads its elements e = base + tid + k nt,
  // k < KP, unconditionally (clamped), one empty asm statement keeps all KP values live at once, and place() puts element e
  // where the loops below would have put it (same LDS images, bit for bit).
  if constexpr (C::ACTP == 0 && !C::RAGGED && C::MONO == 0 && C::SKIP == 0 && C::BF16 && (C::NOUT == 1 || C::BF16O)) {
    constexpr int KP = 8, P = C::P, LS = H * H + H;          // LS: flat stride of one hidden layer (matrix + bias)
    // multi-output networks on the bf16 matrix core (BF16O): planes of the zero-padded output matrix Wo [HO][H], forward and
    // transposed image, index scheme of the loop further down
    auto place_wout = [&](int j, int k, real w) {
      __bf16* wf = reinterpret_cast<__bf16*>(lds + C::ldsWout(BWD));
      __bf16* wt = reinterpret_cast<__bf16*>(lds + C::ldsWoutT());
      const __bf16 w0 = (__bf16)w; const real r1 = w - (real)w0;
      const __bf16 w1 = (__bf16)r1; const __bf16 w2 = (__bf16)(r1 - (real)w1);
      {
        const int blk = (j >> 4) * C::NC + (k >> 5);
        const int base = ((blk * 3) * 64 + (j & 15) + 16 * ((k & 15) >> 2)) * 8 + 4 * ((k & 31) >> 4) + (k & 3);
        wf[base] = w0; wf[base + 512] = w1; wf[base + 1024] = w2;
      }
      if (BWD) {
        const int blk = (k >> 4) * C::NCO + (j >> 5);
        const int base = ((blk * 3) * 64 + (k & 15) + 16 * ((j & 15) >> 2)) * 8 + 4 * ((j & 31) >> 4) + (j & 3);
        wt[base] = w0; wt[base + 512] = w1; wt[base + 1024] = w2;
      }
    };
    if constexpr (C::NOUT > 1) {                             // padding rows NOUT .. HO - 1 of Wo and of bout: exact zeros
      for (int i = C::NOUT * H + tid; i < C::HO * H; i += nt) place_wout(i / H, i % H, (real)0.f);
      for (int i = C::NOUT + tid; i < C::HO; i += nt) lds[C::ldsbout(BWD) + i] = 0.f;
    }
    auto place = [&](int e, real w) {
      if (e < C::offb1) {                                    // W1[j][a] -> W1T[a][j]
        const int j = e / C::NIN, a = e - j * C::NIN;
        lds[C::ldsW1T + a * H + j] = w;
      } else if (e < C::offW(2)) {
        lds[C::ldsb1 + (e - C::offb1)] = w;
      } else if (e < C::offWout) {
        const int l = 2 + (e - C::offW(2)) / LS, r = (e - C::offW(2)) % LS;
        if (r >= H * H) { lds[C::ldsb(l, BWD) + (r - H * H)] = w; return; }
        const int j = r / H, k = r - j * H;
        __bf16* wf = reinterpret_cast<__bf16*>(lds + C::ldsWf(l, BWD));
        __bf16* wt = reinterpret_cast<__bf16*>(lds + C::ldsWt(l));
        const __bf16 w0 = (__bf16)w; const real r1 = w - (real)w0;
        const __bf16 w1 = (__bf16)r1; const __bf16 w2 = (__bf16)(r1 - (real)w1);
        {
          const int blk = (j >> 4) * C::NC + (k >> 5);
          const int base = ((blk * 3) * 64 + (j & 15) + 16 * ((k & 15) >> 2)) * 8 + 4 * ((k & 31) >> 4) + (k & 3);
          wf[base] = w0; wf[base + 512] = w1; wf[base + 1024] = w2;
        }
        if (BWD) {
          const int blk = (k >> 4) * C::NC + (j >> 5);
          const int base = ((blk * 3) * 64 + (k & 15) + 16 * ((j & 15) >> 2)) * 8 + 4 * ((j & 31) >> 4) + (j & 3);
          wt[base] = w0; wt[base + 512] = w1; wt[base + 1024] = w2;
        }
      } else if (e < C::offbout) {
        if constexpr (C::NOUT == 1) lds[C::ldsWout(BWD) + (e - C::offWout)] = w;
        else place_wout((e - C::offWout) / H, (e - C::offWout) % H, w);
      } else {
        lds[C::ldsbout(BWD) + (e - C::offbout)] = w;
      }
    };
    for (int base = 0; base < P; base += KP * nt) {
      real v[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int e = base + tid + k * nt;
        v[k] = prm[e < P ? e : P - 1];
      }
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int e = base + tid + k * nt;
        if (e < P) place(e, v[k]);
      }
    }
    return;
  }
#endif
  for (int i = tid; i < C::NIN * H; i += nt) {  // W1T[a][j] = W1[j][a]  (MONO: a runs over the NIN features)
    const int a = i / H, j = i - a * H;
    lds[C::ldsW1T + i] = unit(j, C::hr(1), C::offW1 + j * C::NIN + a, f1);
  }
  for (int i = tid; i < H; i += nt) lds[C::ldsb1 + i] = unit(i, C::hr(1), C::offb1 + i, f1);
  if constexpr (C::ALPHA) {
    if (tid < C::L) lds[C::ldsAlpha(BWD) + tid] = prm[C::offA + 3 * tid];
  }
  if constexpr (C::SKIP != 0 && C::NOUT > 1) {   // St[a][u] = S[u][a], rows >= NOUT zero
    for (int i = tid; i < D * C::HO; i += nt) {
      const int a = i / C::HO, u = i - a * C::HO;
      lds[C::ldsSkip(BWD) + i] = u < C::NOUT ? prm[C::offS + u * D + a] : 0.f;
    }
  }
  if constexpr (C::NOUT == 1) {
    for (int i = tid; i < H; i += nt) lds[C::ldsWout(BWD) + i] = unit(i, HL, C::offWout + i, fo);
    if (tid == 0) lds[C::ldsbout(BWD)] = prm[C::offbout];
    if constexpr (C::SKIP != 0) {
      if (tid < D) lds[C::ldsSkip(BWD) + tid] = prm[C::offS + tid];
    }
  } else if constexpr (C::BF16O) {
    // bf16x3 planes of the zero-padded output matrix Wo [HO][H] in bf16 fragment order (same index scheme as the
    // hidden layers below): forward image = A operand of (ob = 16-row output block, c = chunk of 32 hidden units),
    // transposed image = A operand of (kb = 16 hidden units, c = chunk of 32 output rows)
    const real* Wo = prm + C::offWout;
    __bf16* wf = reinterpret_cast<__bf16*>(lds + C::ldsWout(BWD));
    __bf16* wt = reinterpret_cast<__bf16*>(lds + C::ldsWoutT());
    for (int i = tid; i < C::HO * H; i += nt) {
      const int j = i / H, k = i - j * H;
      real w;
      if constexpr (C::RAGGED) w = (j < C::NOUT && k < HL) ? actp_mul<C>(Wo[j * HL + k], fo) : 0.f;
      else w = j < C::NOUT ? actp_mul<C>(Wo[i], fo) : 0.f;
      const __bf16 w0 = (__bf16)w; const real r1 = w - (real)w0;
      const __bf16 w1 = (__bf16)r1; const __bf16 w2 = (__bf16)(r1 - (real)w1);
      {
        const int blk = (j >> 4) * C::NC + (k >> 5);
        const int base = ((blk * 3) * 64 + (j & 15) + 16 * ((k & 15) >> 2)) * 8 + 4 * ((k & 31) >> 4) + (k & 3);
        wf[base] = w0; wf[base + 512] = w1; wf[base + 1024] = w2;
      }
      if (BWD) {
        const int blk = (k >> 4) * C::NCO + (j >> 5);
        const int base = ((blk * 3) * 64 + (k & 15) + 16 * ((j & 15) >> 2)) * 8 + 4 * ((j & 31) >> 4) + (j & 3);
        wt[base] = w0; wt[base + 512] = w1; wt[base + 1024] = w2;
      }
    }
    for (int i = tid; i < C::HO; i += nt) lds[C::ldsbout(BWD) + i] = i < C::NOUT ? prm[C::offbout + i] : 0.f;
  } else {
    constexpr int NBO = C::NBO;
    const real* Wo = prm + C::offWout;  // [NOUT][H], rows >= NOUT are zero padding
    for (int i = tid; i < C::HO * H; i += nt) {
      const int lane = i & 63, t = (i >> 6) & 3, blk = i >> 8;
      {  // forward A operand of block (ob, kb): blk = ob*NB + kb:  A[i'][q'] = Wo[16 ob + i'][16 kb + 4 q' + t]
        const int ob = blk / NB, kb = blk - ob * NB;
        const int o = 16 * ob + mrow(lane & 15);
        const int k = 16 * kb + 4 * (lane >> 4) + t;
        lds[C::ldsWout(BWD) + i] = (o < C::NOUT && real_unit(k, HL)) ? actp_mul<C>(Wo[o * HL + k], fo) : 0.f;
      }
      if (BWD) {  // transposed A operand of block (kb, ob): blk = kb*NBO + ob:  A[i'][q'] = Wo[16 ob + 4 q' + t][16 kb + i']
        const int kb = blk / NBO, ob = blk - kb * NBO;
        const int o = 16 * ob + 4 * (lane >> 4) + t;
        const int k = 16 * kb + mrow(lane & 15);
        lds[C::ldsWoutT() + i] = (o < C::NOUT && real_unit(k, HL)) ? actp_mul<C>(Wo[o * HL + k], fo) : 0.f;
      }
    }
    for (int i = tid; i < C::HO; i += nt) lds[C::ldsbout(BWD) + i] = i < C::NOUT ? prm[C::offbout + i] : 0.f;
  }
  if constexpr (C::BF16) {
    // bf16x3 planes in "bf16 fragment order": A operand of (ob = 16-row output block, c = chunk of 32 contraction
    // slots), plane pl: lane (i = lane&15, kg = lane>>4) holds 8 bf16, slot e <-> unit 16*(2c + (e>>2)) + 4*kg + (e&3).
    // index in bf16 units: ((((ob*NC + c)*3 + pl)*64 + lane)*8 + e
#pragma unroll
    for (int l = 2; l <= C::L; ++l) {
      const real* W = prm + C::offW(l);
      __bf16* wf = reinterpret_cast<__bf16*>(lds + C::ldsWf(l, BWD));
      __bf16* wt = reinterpret_cast<__bf16*>(lds + C::ldsWt(l));
      const real fb = act_pre<C>(prm, l), fw = actp_mul<C>(fb, act_post<C>(prm, l - 1));
      for (int i = tid; i < H * H; i += nt) {   // one coalesced pass over W[j][k] (out j, in k): split once, scatter twice
        const int j = i / H, k = i - j * H;
        real w;
        if constexpr (C::RAGGED) w = (j < C::hr(l) && k < C::hr(l - 1)) ? actp_mul<C>(W[j * C::hr(l - 1) + k], fw) : 0.f;
        else w = actp_mul<C>(W[i], fw);
        const __bf16 w0 = (__bf16)w; const real r1 = w - (real)w0;
        const __bf16 w1 = (__bf16)r1; const __bf16 w2 = (__bf16)(r1 - (real)w1);
        {  // forward image: block (ob = j/16, c = k/32), lane (j%16, kg = (k%16)/4), slot e = 4*((k%32)/16) + k%4
          const int blk = (j >> 4) * C::NC + (k >> 5);
          const int base = ((blk * 3) * 64 + (j & 15) + 16 * ((k & 15) >> 2)) * 8 + 4 * ((k & 31) >> 4) + (k & 3);
          wf[base] = w0; wf[base + 512] = w1; wf[base + 1024] = w2;
        }
        if (BWD) {  // transposed image: block (ob = k/16, c = j/32), lane (k%16, kg = (j%16)/4), slot e = 4*((j%32)/16) + j%4
          const int blk = (k >> 4) * C::NC + (j >> 5);
          const int base = ((blk * 3) * 64 + (k & 15) + 16 * ((j & 15) >> 2)) * 8 + 4 * ((j & 31) >> 4) + (j & 3);
          wt[base] = w0; wt[base + 512] = w1; wt[base + 1024] = w2;
        }
      }
      for (int i = tid; i < H; i += nt) lds[C::ldsb(l, BWD) + i] = unit(i, C::hr(l), C::offb(l) + i, fb);
    }
  } else
#pragma unroll
  for (int l = 2; l <= C::L; ++l) {
    const real* W = prm + C::offW(l);
    const real fb = act_pre<C>(prm, l), fw = actp_mul<C>(fb, act_post<C>(prm, l - 1));
    for (int i = tid; i < H * H; i += nt) {
      const int lane = i & 63, t = (i >> 6) & 3, blk = i >> 8;  // blk = first*NB + second
      const int b0 = blk / NB, b1 = blk - b0 * NB;
      // forward A operand of block (ib=b0, kb=b1), step t:  A[i'][k=q'] = W[16 ib + i'][16 kb + 4 q' + t]
      auto weight = [&](int j, int k) {
        if constexpr (C::RAGGED) return (j < C::hr(l) && k < C::hr(l - 1)) ? actp_mul<C>(W[j * C::hr(l - 1) + k], fw) : (real)0.f;
        else return actp_mul<C>(W[j * H + k], fw);
      };
      lds[C::ldsWf(l, BWD) + i] = weight(16 * b0 + mrow(lane & 15), 16 * b1 + 4 * (lane >> 4) + t);
      if (BWD)  // transposed A operand of block (kb=b0, ib=b1): A[i'][k=q'] = W[16 ib + 4 q' + t][16 kb + i']
        lds[C::ldsWt(l) + i] = weight(16 * b1 + 4 * (lane >> 4) + t, 16 * b0 + mrow(lane & 15));
    }
    for (int i = tid; i < H; i += nt) lds[C::ldsb(l, BWD) + i] = unit(i, C::hr(l), C::offb(l) + i, fb);
  }
}

__device__ __forceinline__ real4 lds4(const real* p) { return *reinterpret_cast<const real4*>(p); }

// An offset of zero the optimiser cannot see through.  Added to the LDS base once per tile it keeps the (loop-invariant)
// weight reads inside the tile loop.  Wide nets only: there the hoisted reads would occupy hundreds of registers for
// the whole kernel (C3 closure kernel: 548 -> 483 us with the reads pinned); at H = 32 hoisting is harmless (+-1 %).
template <class C>
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  if constexpr ((C::WIDE || C::BWD_THREADS != 256) && (NDQ_PIN_WEIGHT_READS != 0)) asm volatile("" : "+v"(z));
  return z;
}

// ------------------------------------------------------------------------------------------------ per-layer pieces
// hidden-unit state of one layer for one tile
template <class C>
struct LayerState {
  real t[C::NB][4];                 // sigma(z)
  real c[C::NB][4];                 // second state value: cos(z) for sin, sigma(z) for swish, z for aptx; unused (dead) otherwise
  real4 z[C::NS][C::NB];             // pre-activation derivative streams (index 0 unused: value is in t)
  real al;                          // Cfg::ALPHA: the layer's APTx alpha (wave-uniform); never touched otherwise
};
template <class C>
__device__ __forceinline__ real layer_alpha(const LayerState<C>& st) {
  if constexpr (C::ALPHA) return st.al;
  else return 1.f;
}
template <class C, bool BWD>
__device__ __forceinline__ void load_alpha(const real* lds, int l, LayerState<C>& st) {   // l = 1..L
  if constexpr (C::ALPHA) st.al = lds[C::ldsAlpha(BWD) + l - 1];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// bf16x3 planes of all streams of one fragment set: pl[s][c][k], k = 0 (high) .. 2 (low)
template <class C>
struct Planes {
  bf16x8 pl[C::NS][C::NC > 0 ? C::NC : 1][3];
};

// streams of h = sigma(z) from the layer state:  h0 = t, h_a = s1 z_a, h_ab = s2 z_a z_b + s1 z_ab
// Fourth-order stream S (S >= SS::S4) of one hidden unit: the Faa di Bruno terms over the set partitions of the four index
// positions, grouped by the order of sigma they multiply.  zv(s) reads pre-activation stream s of the unit.
//   q4 = z_a z_b z_c z_d;  q3 = sum over the 6 pairs (i, j) of z_ij z_k z_l;  q2 = sum over the 3 pairings of z_ij z_kl
//   + sum over the 4 triples of z_jkl z_i;  h = s4 q4 + s3 q3 + s2 q2 + s1 z_abcd
template <class SS, int S, class ZV>
__device__ __forceinline__ void quad_terms(ZV&& zv, real& q4, real& q3, real& q2) {
  constexpr int x0 = SS::Q(S, 0), x1 = SS::Q(S, 1), x2 = SS::Q(S, 2), x3 = SS::Q(S, 3);
  const real z0 = zv(1 + x0), z1 = zv(1 + x1), z2 = zv(1 + x2), z3 = zv(1 + x3);
  const real p01 = zv(SS::pair_stream(x0, x1)), p02 = zv(SS::pair_stream(x0, x2)), p03 = zv(SS::pair_stream(x0, x3));
  const real p12 = zv(SS::pair_stream(x1, x2)), p13 = zv(SS::pair_stream(x1, x3)), p23 = zv(SS::pair_stream(x2, x3));
  const real t0 = zv(SS::tri_stream(x1, x2, x3)), t1 = zv(SS::tri_stream(x0, x2, x3));
  const real t2 = zv(SS::tri_stream(x0, x1, x3)), t3 = zv(SS::tri_stream(x0, x1, x2));
  q4 = (z0 * z1) * (z2 * z3);
  q3 = rfma(p01, z2 * z3, rfma(p02, z1 * z3, rfma(p03, z1 * z2, rfma(p12, z0 * z3, rfma(p13, z0 * z2, p23 * (z0 * z1))))));
  q2 = rfma(p01, p23, rfma(p02, p13, rfma(p03, p12, rfma(t0, z0, rfma(t1, z1, rfma(t2, z2, t3 * z3))))));
}

template <class C>
__device__ __forceinline__ void act_forward(const LayerState<C>& st, real4 (&h)[C::NS][C::NB]) {
  using SS = typename C::SS;
  using A = Act<C::ACT>;
#pragma unroll
  for (int b = 0; b < C::NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real t = st.t[b][r], c = st.c[b][r];
      const real s1 = A::s1(t, c, layer_alpha<C>(st));
      h[0][b][r] = t;
      if constexpr (SS::FIRST) {
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          h[1 + a][b][r] = s1 * st.z[1 + a][b][r];
        });
        if constexpr (SS::LAP) {
          const real s2 = A::s2(t, c, s1);
          real q2 = 0.f;
          sfor<C::D>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            if constexpr (SS::in_lap(a)) q2 = rfma(st.z[1 + a][b][r], st.z[1 + a][b][r], q2);
          });
          h[SS::S2][b][r] = rfma(s2, q2, s1 * st.z[SS::S2][b][r]);
        } else if constexpr (SS::N2 > 0) {
          const real s2 = A::s2(t, c, s1);
          sfor<SS::N2>([&](auto k_) {
            constexpr int s = SS::S2 + decltype(k_)::value;
            constexpr int a = SS::A(s), bb = SS::B(s);
            h[s][b][r] = rfma(s2 * st.z[1 + a][b][r], st.z[1 + bb][b][r], s1 * st.z[s][b][r]);
          });
          if constexpr (SS::N3 > 0) {
            const real s3 = A::s3(t, c, s1);
            sfor<SS::N3>([&](auto k_) {
              constexpr int s = SS::S3 + decltype(k_)::value;
              constexpr int a = SS::T(s, 0), bb = SS::T(s, 1), cc = SS::T(s, 2);
              constexpr int sab = SS::pair_stream(a, bb), sac = SS::pair_stream(a, cc), sbc = SS::pair_stream(bb, cc);
              const real za = st.z[1 + a][b][r], zb = st.z[1 + bb][b][r], zc = st.z[1 + cc][b][r];
              const real mix = rfma(st.z[sab][b][r], zc, rfma(st.z[sac][b][r], zb, st.z[sbc][b][r] * za));
              h[s][b][r] = rfma(s3 * za, zb * zc, rfma(s2, mix, s1 * st.z[s][b][r]));
            });
            if constexpr (SS::N4 > 0) {
              const real s4 = A::s4(t, c, s1);
              sfor<SS::N4>([&](auto k_) {
                constexpr int s = SS::S4 + decltype(k_)::value;
                real q4, q3, q2;
                quad_terms<SS, s>([&](int i) { return st.z[i][b][r]; }, q4, q3, q2);
                h[s][b][r] = rfma(s4, q4, rfma(s3, q3, rfma(s2, q2, s1 * st.z[s][b][r])));
              });
            }
          }
        }
      }
    }
}

// one stream of act_forward (compile-time stream index S)
template <class C, int S>
__device__ __forceinline__ void act_forward_stream(const LayerState<C>& st, real4 (&hs)[C::NB]) {
  using SS = typename C::SS;
  using A = Act<C::ACT>;
#pragma unroll
  for (int b = 0; b < C::NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real t = st.t[b][r], c = st.c[b][r];
      if constexpr (S == 0) {
        hs[b][r] = t;
      } else if constexpr (S < SS::S2) {
        hs[b][r] = A::s1(t, c, layer_alpha<C>(st)) * st.z[S][b][r];
      } else if constexpr (SS::LAP) {
        const real s1 = A::s1(t, c, layer_alpha<C>(st));
        real q2 = 0.f;
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          if constexpr (SS::in_lap(a)) q2 = rfma(st.z[1 + a][b][r], st.z[1 + a][b][r], q2);
        });
        hs[b][r] = rfma(A::s2(t, c, s1), q2, s1 * st.z[S][b][r]);
      } else if constexpr (S < SS::S3) {
        constexpr int a = SS::A(S), bb = SS::B(S);
        const real s1 = A::s1(t, c, layer_alpha<C>(st));
        hs[b][r] = rfma(A::s2(t, c, s1) * st.z[1 + a][b][r], st.z[1 + bb][b][r], s1 * st.z[S][b][r]);
      } else if constexpr (S < SS::S4) {
        constexpr int a = SS::T(S, 0), bb = SS::T(S, 1), cc = SS::T(S, 2);
        constexpr int sab = SS::pair_stream(a, bb), sac = SS::pair_stream(a, cc), sbc = SS::pair_stream(bb, cc);
        const real s1 = A::s1(t, c, layer_alpha<C>(st)), s2 = A::s2(t, c, s1), s3 = A::s3(t, c, s1);
        const real za = st.z[1 + a][b][r], zb = st.z[1 + bb][b][r], zc = st.z[1 + cc][b][r];
        const real mix = rfma(st.z[sab][b][r], zc, rfma(st.z[sac][b][r], zb, st.z[sbc][b][r] * za));
        hs[b][r] = rfma(s3 * za, zb * zc, rfma(s2, mix, s1 * st.z[S][b][r]));
      } else {
        const real s1 = A::s1(t, c, layer_alpha<C>(st)), s2 = A::s2(t, c, s1), s3 = A::s3(t, c, s1), s4 = A::s4(t, c, s1);
        real q4, q3, q2;
        quad_terms<SS, S>([&](int i) { return st.z[i][b][r]; }, q4, q3, q2);
        hs[b][r] = rfma(s4, q4, rfma(s3, q3, rfma(s2, q2, s1 * st.z[S][b][r])));
      }
    }
}

// adjoint of act_forward: given hbar (overwritten in place with zbar).  Cfg::ACTP: ap[0] += sum_s g_s u_s,
// ap[1] += sum_t ubar_t u_t - sum_s g_s h_s, ap[2] += sum_s g_s h_s over this lane's units (see Act<ACT_APTX>)
template <class C>
__device__ __forceinline__ void act_backward(const LayerState<C>& st, real4 (&g)[C::NS][C::NB], real (&ap)[3]) {
  using SS = typename C::SS;
  using A = Act<C::ACT>;
#pragma unroll
  for (int b = 0; b < C::NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const real t = st.t[b][r], c = st.c[b][r];
      const real s1 = A::s1(t, c, layer_alpha<C>(st));
      real u0 = 0.f, gh = 0.f, gu = 0.f;     // ACTP: pre-activation value, sum_s g_s h_s, sum_s g_s u_s of this unit
      if constexpr (C::ACTP == 1) {
        // swish keeps t = u sigma(u) and c = sigma(u): u = t / c (c underflows to 0 only where the contribution does too)
        if constexpr (C::ACT == ACT_SWISH) u0 = (c > 0.f) ? t / c : 0.f;
        else u0 = c;
        gh = g[0][b][r] * t;
        gu = g[0][b][r] * u0;
      }
      real z0 = s1 * g[0][b][r];
      if constexpr (SS::FIRST) {
        const real s2 = A::s2(t, c, s1);
        if constexpr (C::ACTP == 1) {        // h_a = s1 u_a; h_L = s2 sum u_a^2 + s1 u_L; h_ab = s2 u_a u_b + s1 u_ab
          sfor<C::D>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            gh = rfma(g[1 + a][b][r], s1 * st.z[1 + a][b][r], gh);
            gu = rfma(g[1 + a][b][r], st.z[1 + a][b][r], gu);
          });
          if constexpr (SS::LAP) {
            real q2 = 0.f;
            sfor<C::D>([&](auto a_) {
              constexpr int a = decltype(a_)::value;
              if constexpr (SS::in_lap(a)) q2 = rfma(st.z[1 + a][b][r], st.z[1 + a][b][r], q2);
            });
            gh = rfma(g[SS::S2][b][r], rfma(s2, q2, s1 * st.z[SS::S2][b][r]), gh);
            gu = rfma(g[SS::S2][b][r], st.z[SS::S2][b][r], gu);
          } else {
            sfor<SS::N2>([&](auto k_) {
              constexpr int s = SS::S2 + decltype(k_)::value;
              constexpr int a = SS::A(s), bb = SS::B(s);
              gh = rfma(g[s][b][r], rfma(s2 * st.z[1 + a][b][r], st.z[1 + bb][b][r], s1 * st.z[s][b][r]), gh);
              gu = rfma(g[s][b][r], st.z[s][b][r], gu);
            });
          }
        }
        real za[C::D];
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          z0 = rfma(s2 * st.z[1 + a][b][r], g[1 + a][b][r], z0);
          za[a] = s1 * g[1 + a][b][r];
        });
        if constexpr (SS::LAP) {
          // h_L = s2 * sum_a z_a^2 + s1 * z_L
          const real s3 = A::s3(t, c, s1);
          const real hb = g[SS::S2][b][r];
          real q2 = 0.f;
          sfor<C::D>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            if constexpr (SS::in_lap(a)) {
              const real zA = st.z[1 + a][b][r];
              q2 = rfma(zA, zA, q2);
              za[a] = rfma(2.f * s2 * zA, hb, za[a]);
            }
          });
          z0 = rfma(rfma(s3, q2, s2 * st.z[SS::S2][b][r]), hb, z0);
          g[SS::S2][b][r] = s1 * hb;
        } else if constexpr (SS::N2 > 0) {
          const real s3 = A::s3(t, c, s1);
          real zb2[SS::N2];                 // what the third- / fourth-order streams add to the second-order adjoints
#pragma unroll
          for (int k = 0; k < SS::N2; ++k) zb2[k] = 0.f;
          real zb3[SS::N3 > 0 ? SS::N3 : 1];   // what the fourth-order streams add to the third-order adjoints
#pragma unroll
          for (int k = 0; k < (SS::N3 > 0 ? SS::N3 : 1); ++k) zb3[k] = 0.f;
          if constexpr (SS::N4 > 0) {
            // adjoint of h = s4 q4 + s3 q3 + s2 q2 + s1 z_abcd, partition by partition: a term sigma^(r) prod_B z_B gives
            // sigma^(r+1) prod_B z_B to the value's adjoint and sigma^(r) prod_{B' != B} z_B' to block B's (oracle/jet_ref.py)
            const real s4 = A::s4(t, c, s1), s5 = A::s5(t, c, s1);
            sfor<SS::N4>([&](auto k_) {
              constexpr int s = SS::S4 + decltype(k_)::value;
              constexpr int x[4] = {SS::Q(s, 0), SS::Q(s, 1), SS::Q(s, 2), SS::Q(s, 3)};
              const real hb = g[s][b][r];
              real q4, q3, q2;
              quad_terms<SS, s>([&](int i) { return st.z[i][b][r]; }, q4, q3, q2);
              z0 = rfma(rfma(s5, q4, rfma(s4, q3, rfma(s3, q2, s2 * st.z[s][b][r]))), hb, z0);
              const real z1v[4] = {st.z[1 + x[0]][b][r], st.z[1 + x[1]][b][r], st.z[1 + x[2]][b][r], st.z[1 + x[3]][b][r]};
              sfor<4>([&](auto i_) {        // first-order adjoints: position i against the other three (j, k, l)
                constexpr int i = decltype(i_)::value, j = (i + 1) & 3, k = (i + 2) & 3, l = (i + 3) & 3;
                const real pjk = st.z[SS::pair_stream(x[j], x[k])][b][r], pjl = st.z[SS::pair_stream(x[j], x[l])][b][r];
                const real pkl = st.z[SS::pair_stream(x[k], x[l])][b][r];
                const real tjkl = st.z[SS::tri_stream(x[j], x[k], x[l])][b][r];
                const real d1 = rfma(s4 * z1v[j], z1v[k] * z1v[l],
                                     rfma(s3, rfma(pjk, z1v[l], rfma(pjl, z1v[k], pkl * z1v[j])), s2 * tjkl));
                za[x[i]] = rfma(d1, hb, za[x[i]]);
                // third-order adjoints: the triple without position i gets s2 z_i
                zb3[SS::tri_stream(x[j], x[k], x[l]) - SS::S3] = rfma(s2 * z1v[i], hb, zb3[SS::tri_stream(x[j], x[k], x[l]) - SS::S3]);
              });
              // second-order adjoints: pair (i, j) gets s3 z_k z_l + s2 z_kl
              auto pair_adj = [&](auto i_, auto j_, auto k_2, auto l_) {
                constexpr int i = decltype(i_)::value, j = decltype(j_)::value, k = decltype(k_2)::value, l = decltype(l_)::value;
                constexpr int sij = SS::pair_stream(x[i], x[j]), skl = SS::pair_stream(x[k], x[l]);
                zb2[sij - SS::S2] = rfma(rfma(s3 * z1v[k], z1v[l], s2 * st.z[skl][b][r]), hb, zb2[sij - SS::S2]);
              };
              using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
              using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
              pair_adj(I0{}, I1{}, I2{}, I3{}); pair_adj(I0{}, I2{}, I1{}, I3{}); pair_adj(I0{}, I3{}, I1{}, I2{});
              pair_adj(I1{}, I2{}, I0{}, I3{}); pair_adj(I1{}, I3{}, I0{}, I2{}); pair_adj(I2{}, I3{}, I0{}, I1{});
              g[s][b][r] = s1 * hb;
            });
          }
          if constexpr (SS::N3 > 0) {
            const real s4 = A::s4(t, c, s1);
            sfor<SS::N3>([&](auto k_) {
              constexpr int s = SS::S3 + decltype(k_)::value;
              constexpr int a = SS::T(s, 0), bb = SS::T(s, 1), cc = SS::T(s, 2);
              constexpr int sab = SS::pair_stream(a, bb), sac = SS::pair_stream(a, cc), sbc = SS::pair_stream(bb, cc);
              const real hb = g[s][b][r];
              const real zA = st.z[1 + a][b][r], zB = st.z[1 + bb][b][r], zC = st.z[1 + cc][b][r];
              const real zab = st.z[sab][b][r], zac = st.z[sac][b][r], zbc = st.z[sbc][b][r];
              const real mix = rfma(zab, zC, rfma(zac, zB, zbc * zA));
              z0 = rfma(rfma(s4 * zA, zB * zC, rfma(s3, mix, s2 * st.z[s][b][r])), hb, z0);
              za[a] = rfma(rfma(s3 * zB, zC, s2 * zbc), hb, za[a]);
              za[bb] = rfma(rfma(s3 * zA, zC, s2 * zac), hb, za[bb]);
              za[cc] = rfma(rfma(s3 * zA, zB, s2 * zab), hb, za[cc]);
              zb2[sab - SS::S2] = rfma(s2 * zC, hb, zb2[sab - SS::S2]);
              zb2[sac - SS::S2] = rfma(s2 * zB, hb, zb2[sac - SS::S2]);
              zb2[sbc - SS::S2] = rfma(s2 * zA, hb, zb2[sbc - SS::S2]);
              g[s][b][r] = rfma(s1, hb, zb3[decltype(k_)::value]);
            });
          }
          sfor<SS::N2>([&](auto k_) {
            constexpr int s = SS::S2 + decltype(k_)::value;
            constexpr int a = SS::A(s), bb = SS::B(s);
            const real hb = g[s][b][r];
            const real zA = st.z[1 + a][b][r], zB = st.z[1 + bb][b][r];
            z0 = rfma(rfma(s3 * zA, zB, s2 * st.z[s][b][r]), hb, z0);
            za[a] = rfma(s2 * zB, hb, za[a]);
            za[bb] = rfma(s2 * zA, hb, za[bb]);
            g[s][b][r] = rfma(s1, hb, zb2[decltype(k_)::value]);
          });
        }
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          g[1 + a][b][r] = za[a];
        });
      }
      g[0][b][r] = z0;
      if constexpr (C::ACTP == 1) {
        real uu = z0 * u0;
#pragma unroll
        for (int s = 1; s < C::NS; ++s) uu = rfma(g[s][b][r], st.z[s][b][r], uu);
        ap[0] += gu;
        ap[1] += uu - gh;
        ap[2] += gh;
      }
    }
}

template <class C> __device__ __forceinline__ void zero_frag(real4 (&z)[C::NS][C::NB]);

// split the 8 fp32 values a lane holds for one K-chunk (blocks 2c, 2c+1) into three bf16x8 operands
// (NP = 2: the third plane is left unset -- for operands nobody reads it of, see NDQ_H_PLANES / NDQ_Z_PLANES above)
template <int NP = 3>
__device__ __forceinline__ void split3(const real4 a, const real4 b, bf16x8 (&pl)[3]) {
  const real x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  if constexpr ((NDQ_ABL & 32) != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { pl[0][e] = (__bf16)x[e]; pl[1][e] = pl[0][e]; pl[2][e] = pl[0][e]; }
    return;
  }
#if NDQ_SPLIT_PAIRS && !NDQ_F64
  // two values at a time on 2-wide vector types: one v_cvt_pk_bf16_f32 per pair and plane, packed subtractions
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 v = {x[2 * k], x[2 * k + 1]};
    const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h0, f32x2);
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    pl[0][2 * k] = h0[0]; pl[0][2 * k + 1] = h0[1];
    pl[1][2 * k] = h1[0]; pl[1][2 * k + 1] = h1[1];
    pl[2][2 * k] = h2[0]; pl[2][2 * k + 1] = h2[1];
  }
#else
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h0 = (__bf16)x[e];
    const real r1 = x[e] - (real)h0;
    const __bf16 h1 = (__bf16)r1;
    pl[0][e] = h0; pl[1][e] = h1;
    if constexpr (NP == 3) pl[2][e] = (__bf16)(r1 - (real)h1);
  }
#endif
}

// NDQ_FWD_BF16X1: plane 0 alone (round to nearest even), the other two stay unset and unread
__device__ __forceinline__ void split1(const real4 a, const real4 b, bf16x8 (&pl)[3]) {
  const real x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e) pl[0][e] = (__bf16)x[e];
}

template <class C, int NP = NDQ_H_PLANES>
__device__ __forceinline__ void split_all(const real4 (&h)[C::NS][C::NB], Planes<C>& P) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int c = 0; c < C::NC; ++c) split3<NP>(h[s][2 * c], h[s][2 * c + 1], P.pl[s][c]);
}

// z[s][ob] += W h[s] with h given as bf16x3 planes
template <class C, bool X1 = false, bool REV = false>
__device__ __forceinline__ void gemm_planes(const real* __restrict__ wl, int lane, const Planes<C>& P,
                                            real4 (&z)[C::NS][C::NB]) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
#pragma unroll
  for (int c = 0; c < C::NC; ++c)
#pragma unroll
    for (int ob = 0; ob < C::NB; ++ob) {
      const bf16x8 a0 = w[((ob * C::NC + c) * 3 + 0) * 64 + lane];
      if constexpr (X1) {                  // NDQ_FWD_BF16X1: the leading planes only
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, P.pl[s][c][0], z[s][ob], 0, 0, 0);
        continue;
      }
      const bf16x8 a1 = w[((ob * C::NC + c) * 3 + 1) * 64 + lane];
      const bf16x8 a2 = w[((ob * C::NC + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < C::NS; ++s)                                                          \
      z[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, P.pl[s][c][K], z[s][ob], 0, 0, 0);
      if constexpr (REV) { NDQ_PRODUCTS_BWD(NDQ_T) } else { NDQ_PRODUCTS_FWD(NDQ_T) }
#undef NDQ_T
    }
}

// hbar = W^T zbar in place (all streams are split first, then overwritten)
template <class C>
__device__ __forceinline__ void gemm_bf16x3_inplace(const real* __restrict__ wl, int lane, real4 (&g)[C::NS][C::NB]) {
  if constexpr (C::GROUP_HBAR) {     // group by group: planes and outputs of SG streams live at a time
    constexpr int NG = (C::NS + C::SG - 1) / C::SG;
    sfor<NG>([&](auto g_) {
      constexpr int s0 = decltype(g_)::value * C::SG;
      constexpr int sn = (C::NS - s0 < C::SG) ? C::NS - s0 : C::SG;
      const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
      bf16x8 pl[sn][C::NC][3];
      real4 o[sn][C::NB];
#pragma unroll
      for (int s = 0; s < sn; ++s) {
#pragma unroll
        for (int c = 0; c < C::NC; ++c) split3<NDQ_Z_PLANES>(g[s0 + s][2 * c], g[s0 + s][2 * c + 1], pl[s][c]);
#pragma unroll
        for (int b = 0; b < C::NB; ++b) o[s][b] = real4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c = 0; c < C::NC; ++c)
#pragma unroll
        for (int ob = 0; ob < C::NB; ++ob) {
          const bf16x8 a0 = w[((ob * C::NC + c) * 3 + 0) * 64 + lane];
          const bf16x8 a1 = w[((ob * C::NC + c) * 3 + 1) * 64 + lane];
          const bf16x8 a2 = w[((ob * C::NC + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < sn; ++s)                                                             \
      o[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][c][K], o[s][ob], 0, 0, 0);
          NDQ_PRODUCTS_BWD(NDQ_T)
#undef NDQ_T
        }
#pragma unroll
      for (int s = 0; s < sn; ++s)
#pragma unroll
        for (int b = 0; b < C::NB; ++b) g[s0 + s][b] = o[s][b];
    });
    return;
  }
  real4 o[C::NS][C::NB];
  zero_frag<C>(o);
  Planes<C> P;
  split_all<C, NDQ_Z_PLANES>(g, P);
  gemm_planes<C, false, true>(wl, lane, P, o);
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int b = 0; b < C::NB; ++b) g[s][b] = o[s][b];
}

// z[s][ib] (+)= sum_kb sum_t A(w[(ib*NB+kb)*4+t]) * B(h[s][kb][t]);  w points at a fragment-ordered H x H matrix
template <class C>
__device__ __forceinline__ void gemm_frag(const real* __restrict__ w, int lane, const real4 (&h)[C::NS][C::NB],
                                          real4 (&z)[C::NS][C::NB]) {
#pragma unroll
  for (int ib = 0; ib < C::NB; ++ib)
#pragma unroll
    for (int kb = 0; kb < C::NB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const real a = w[((ib * C::NB + kb) * 4 + t) * 64 + lane];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) z[s][ib] = mfma16x16x4(a, h[s][kb][t], z[s][ib]);
      }
}

template <class C>
__device__ __forceinline__ void zero_frag(real4 (&z)[C::NS][C::NB]) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int b = 0; b < C::NB; ++b) z[s][b] = real4{0.f, 0.f, 0.f, 0.f};
}

// MONO: pw[a][k] = x_a^k, k = 0..MAXDEG
template <class C>
__device__ __forceinline__ void mono_powers(const real (&x)[C::D], real (&pw)[C::D][C::MAXDEG + 1]) {
#pragma unroll
  for (int a = 0; a < C::D; ++a) {
    pw[a][0] = 1.f;
#pragma unroll
    for (int k = 1; k <= C::MAXDEG; ++k) pw[a][k] = pw[a][k - 1] * x[a];
  }
}

// first layer (VALU): z = W1 x + b1; derivative streams of z are columns of W1 (second order: zero)
template <class C, bool BWD>
__device__ __forceinline__ void first_layer(const real* lds, int q, const real (&x)[C::D], LayerState<C>& st) {
  using SS = typename C::SS;
  load_alpha<C, BWD>(lds, 1, st);
  if constexpr (C::MONO != 0) {
    real pw[C::D][C::MAXDEG + 1];
    mono_powers<C>(x, pw);
#pragma unroll
    for (int b = 0; b < C::NB; ++b) {
      const int j0 = 16 * b + 4 * q;
      real4 z = lds4(lds + C::ldsb1 + j0);
      real4 z1[C::D], z2[C::D];
#pragma unroll
      for (int a = 0; a < C::D; ++a) z1[a] = z2[a] = real4{0.f, 0.f, 0.f, 0.f};
      sfor<C::NDEG>([&](auto i_) {
        constexpr int i = decltype(i_)::value, deg = C::mono_deg(i);
#pragma unroll
        for (int a = 0; a < C::D; ++a) {
          const real4 w = lds4(lds + C::ldsW1T + (i * C::D + a) * C::H + j0);
          const real p1 = (real)deg * pw[a][deg - 1];
          const real p2 = deg >= 2 ? (real)(deg * (deg - 1)) * pw[a][deg >= 2 ? deg - 2 : 0] : (real)0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            z[r] = rfma(w[r], pw[a][deg], z[r]);
            z1[a][r] = rfma(w[r], p1, z1[a][r]);
            if constexpr (deg >= 2) z2[a][r] = rfma(w[r], p2, z2[a][r]);
          }
        }
      });
#pragma unroll
      for (int r = 0; r < 4; ++r) Act<C::ACT>::fwd(z[r], st.t[b][r], st.c[b][r], layer_alpha<C>(st));
      if constexpr (SS::FIRST) {
#pragma unroll
        for (int a = 0; a < C::D; ++a) st.z[1 + a][b] = z1[a];
        if constexpr (SS::LAP) {
          real4 zl = real4{0.f, 0.f, 0.f, 0.f};
          sfor<C::D>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            if constexpr (SS::in_lap(a)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) zl[r] += z2[a][r];
            }
          });
          st.z[SS::S2][b] = zl;
        } else {
          sfor<SS::N2>([&](auto k_) {
            constexpr int s = SS::S2 + decltype(k_)::value;
            if constexpr (SS::A(s) == SS::B(s)) st.z[s][b] = z2[SS::A(s)];
            else st.z[s][b] = real4{0.f, 0.f, 0.f, 0.f};
          });
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < C::NB; ++b) {
    const int j0 = 16 * b + 4 * q;
    real4 z = lds4(lds + C::ldsb1 + j0);
    real4 w[C::D];
#pragma unroll
    for (int a = 0; a < C::D; ++a) {
      w[a] = lds4(lds + C::ldsW1T + a * C::H + j0);
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = rfma(w[a][r], x[a], z[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Act<C::ACT>::fwd(z[r], st.t[b][r], st.c[b][r], layer_alpha<C>(st));
    if constexpr (SS::FIRST) {
#pragma unroll
      for (int a = 0; a < C::D; ++a) st.z[1 + a][b] = w[a];
#pragma unroll
      for (int s = SS::S2; s < C::NS; ++s) st.z[s][b] = real4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// hidden layer l (2..L): z = W_l h + b_l on every stream, then activation state
template <class C, bool BWD>
__device__ __forceinline__ void hidden_layer(const real* lds, int l, int lane, int q, const real4 (&h)[C::NS][C::NB],
                                             LayerState<C>& st) {
  real4 z[C::NS][C::NB];
  zero_frag<C>(z);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) z[0][b] = lds4(lds + C::ldsb(l, BWD) + 16 * b + 4 * q);
  gemm_frag<C>(lds + C::ldsWf(l, BWD), lane, h, z);   // exact-f32 MFMA path (widths that are not a multiple of 32)
  load_alpha<C, BWD>(lds, l, st);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Act<C::ACT>::fwd(z[0][b][r], st.t[b][r], st.c[b][r], layer_alpha<C>(st));
#pragma unroll
    for (int s = 1; s < C::NS; ++s) st.z[s][b] = z[s][b];
  }
}

// same with the layer input given as bf16x3 planes
template <class C, bool BWD, bool X1 = false>
__device__ __forceinline__ void hidden_layer_planes(const real* lds, int l, int lane, int q, const Planes<C>& P,
                                                    LayerState<C>& st) {
  real4 z[C::NS][C::NB];
  zero_frag<C>(z);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) z[0][b] = lds4(lds + C::ldsb(l, BWD) + 16 * b + 4 * q);
  if constexpr ((NDQ_ABL & 4) == 0) gemm_planes<C, X1>(lds + C::ldsWf(l, BWD), lane, P, z);
  load_alpha<C, BWD>(lds, l, st);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Act<C::ACT>::fwd(z[0][b][r], st.t[b][r], st.c[b][r], layer_alpha<C>(st));
#pragma unroll
    for (int s = 1; s < C::NS; ++s) st.z[s][b] = z[s][b];
  }
}

// cross-lane sums: the 4 lane groups are combined through ds_bpermute (NDQ_QUAD_SWAP: v_permlane16/32_swap, an
// experiment), the 16 points of a tile -- exactly one DPP row -- by DPP row rotations fused into the adds.  Every lane
// ends up with the total.
__device__ __forceinline__ real quad_sum(real v) {  // sum over the 4 lane groups q (lanes p, p+16, p+32, p+48)
#if NDQ_QUAD_SWAP && !NDQ_F64
  // v_permlane16_swap / v_permlane32_swap exchange rows between TWO registers; swapping a value with a copy of itself
  // leaves (own row, partner row) in the pair, whose sum is what v + shfl_xor(v, 16 / 32) computes -- the same two
  // operands, so the same bits (scripts/ubench_permlane_swap.hip) -- on the VALU, without the LDS round trip
  // (inline asm: ROCm 7.2's clang folds the two results of __builtin_amdgcn_permlane16/32_swap into one register when
  // they are added -- v_add v4, v5, v5 after the swap; the s_nop covers the VALU-write -> permlane-read hazard, which
  // nobody pads inside an asm statement)
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = __builtin_bit_cast(real, a) + __builtin_bit_cast(real, b);
  a = __builtin_bit_cast(unsigned, v); b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(real, a) + __builtin_bit_cast(real, b);
#else
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
#endif
}
template <int CTRL>
__device__ __forceinline__ real dpp_add(real v) {
#if NDQ_F64
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
#else
  return v + __builtin_bit_cast(real, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ real point_sum(real v) {  // sum over the 16 points of the tile (lanes with equal q)
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  return v;
}

// output layer (n_out = 1) on the VALU: out[s] = Wout . h[s] (+ bout on the value stream), identical in all 4 lane groups
template <class C, bool BWD>
__device__ __forceinline__ void tile_output(const real* lds, int q, const real (&x)[C::D], const real4 (&h)[C::NS][C::NB],
                                            real (&out)[C::NS]) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s) out[s] = 0.f;
#pragma unroll
  for (int b = 0; b < C::NB; ++b) {
    const real4 wo = lds4(lds + C::ldsWout(BWD) + 16 * b + 4 * q);
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[s] = rfma(wo[r], h[s][b][r], out[s]);
  }
#pragma unroll
  for (int s = 0; s < C::NS; ++s) out[s] = quad_sum(out[s]);
  out[0] += lds[C::ldsbout(BWD)];
  if constexpr (C::SKIP != 0) {            // + S x: value and first-order streams (second order: nothing)
#pragma unroll
    for (int a = 0; a < C::D; ++a) {
      const real sa = lds[C::ldsSkip(BWD) + a];
      out[0] = rfma(sa, x[a], out[0]);
      if constexpr (C::SS::FIRST) out[1 + a] += sa;
    }
  }
}

// + S x on a multi-output network (networks.Resnet): value stream += S x, first-order stream a += S[:, a]
template <class C, bool BWD>
__device__ __forceinline__ void output_skip_multi(const real* lds, int q, const real (&x)[C::D], real4 (&o)[C::NS][C::NBO]) {
  if constexpr (C::SKIP != 0 && C::NOUT > 1) {
#pragma unroll
    for (int a = 0; a < C::D; ++a)
#pragma unroll
      for (int ob = 0; ob < C::NBO; ++ob) {
        const real4 sa = lds4(lds + C::ldsSkip(BWD) + a * C::HO + 16 * ob + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[0][ob][r] = rfma(sa[r], x[a], o[0][ob][r]);
          if constexpr (C::SS::FIRST) o[1 + a][ob][r] += sa[r];
        }
      }
  }
}

// output layer with NOUT > 1 as an MFMA layer: o[s][ob] = Wo h[s] (+ bout on the value stream); rows >= NOUT are zero
template <class C, bool BWD>
__device__ __forceinline__ void output_layer_mfma(const real* lds, int lane, int q, const real4 (&h)[C::NS][C::NB],
                                                  real4 (&o)[C::NS][C::NBO]) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int ob = 0; ob < C::NBO; ++ob) o[s][ob] = real4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ob = 0; ob < C::NBO; ++ob) o[0][ob] = lds4(lds + C::ldsbout(BWD) + 16 * ob + 4 * q);
  if constexpr (C::BF16O) {
    Planes<C> P;
    split_all<C>(h, P);
    const bf16x8* wb = reinterpret_cast<const bf16x8*>(lds + C::ldsWout(BWD));
#pragma unroll
    for (int c = 0; c < C::NC; ++c)
#pragma unroll
      for (int ob = 0; ob < C::NBO; ++ob) {
        const bf16x8 a0 = wb[((ob * C::NC + c) * 3 + 0) * 64 + lane];
        const bf16x8 a1 = wb[((ob * C::NC + c) * 3 + 1) * 64 + lane];
        const bf16x8 a2 = wb[((ob * C::NC + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < C::NS; ++s)                                                          \
      o[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, P.pl[s][c][K], o[s][ob], 0, 0, 0);
        NDQ_PRODUCTS_FWD(NDQ_T)
#undef NDQ_T
      }
    return;
  }
  const real* w = lds + C::ldsWout(BWD);
#pragma unroll
  for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
    for (int kb = 0; kb < C::NB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const real a = w[((ob * C::NB + kb) * 4 + t) * 64 + lane];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) o[s][ob] = mfma16x16x4(a, h[s][kb][t], o[s][ob]);
      }
}

// forward pass of one tile keeping every layer's state; h = streams of the last hidden layer's activations
// forward-only hidden layer on the bf16x3 path (planes are transient)
template <class C, bool X1 = false>
__device__ __forceinline__ void gemm_layer_bf16(const real* lds, int l, int lane, int q, const real4 (&h)[C::NS][C::NB],
                                                LayerState<C>& st) {
  Planes<C> P;
  if constexpr (X1) {
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int c = 0; c < C::NC; ++c) split1(h[s][2 * c], h[s][2 * c + 1], P.pl[s][c]);
  } else {
    split_all<C>(h, P);
  }
  hidden_layer_planes<C, false, X1>(lds, l, lane, q, P, st);
}

// hidden layer on the bf16x3 path for wide nets: SG streams at a time, each group's activations h[s] computed from the
// input layer's state right before they are split into planes (never more than SG streams of h and of planes live)
template <class C, bool BWD, bool X1 = false>
__device__ __forceinline__ void hidden_layer_grouped(const real* lds, int l, int lane, int q, const LayerState<C>& st_in,
                                                     LayerState<C>& st) {
  real4 z[C::NS][C::NB];
  zero_frag<C>(z);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) z[0][b] = lds4(lds + C::ldsb(l, BWD) + 16 * b + 4 * q);
  const bf16x8* w = reinterpret_cast<const bf16x8*>(lds + C::ldsWf(l, BWD));
  constexpr int NG = (C::NS + C::SG - 1) / C::SG;
  sfor<NG>([&](auto g_) {
    constexpr int s0 = decltype(g_)::value * C::SG;
    constexpr int sn = (C::NS - s0 < C::SG) ? C::NS - s0 : C::SG;
    bf16x8 pl[sn][C::NC][3];
    sfor<sn>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      real4 hs[C::NB];
      act_forward_stream<C, s0 + s>(st_in, hs);
#pragma unroll
      for (int c = 0; c < C::NC; ++c) {
        if constexpr (X1) split1(hs[2 * c], hs[2 * c + 1], pl[s][c]);
        else split3<NDQ_H_PLANES>(hs[2 * c], hs[2 * c + 1], pl[s][c]);
      }
    });
#pragma unroll
    for (int c = 0; c < C::NC; ++c)
#pragma unroll
      for (int ob = 0; ob < C::NB; ++ob) {
        const bf16x8 a0 = w[((ob * C::NC + c) * 3 + 0) * 64 + lane];
        if constexpr (X1) {                // NDQ_FWD_BF16X1: the leading planes only
#pragma unroll
          for (int s = 0; s < sn; ++s)
            z[s0 + s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, pl[s][c][0], z[s0 + s][ob], 0, 0, 0);
          continue;
        }
        const bf16x8 a1 = w[((ob * C::NC + c) * 3 + 1) * 64 + lane];
        const bf16x8 a2 = w[((ob * C::NC + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < sn; ++s)                                                             \
      z[s0 + s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][c][K], z[s0 + s][ob], 0, 0, 0);
        NDQ_PRODUCTS_FWD(NDQ_T)
#undef NDQ_T
      }
  });
  // st may alias st_in (forward-only kernel): everything read from st_in is consumed above
  load_alpha<C, BWD>(lds, l, st);
#pragma unroll
  for (int b = 0; b < C::NB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Act<C::ACT>::fwd(z[0][b][r], st.t[b][r], st.c[b][r], layer_alpha<C>(st));
#pragma unroll
    for (int s = 1; s < C::NS; ++s) st.z[s][b] = z[s][b];
  }
}

// Cfg::WG_TR: the three planes of one stream of one tile as LDS images.  A lane (point p, lane group q) holds units
// 4q .. 4q+3 of block 0 and of block 1 of its point: 16 bytes, stored at chunk q (256 B apart), slot p ^ 4q (16 B each) --
// eight consecutive lanes write 128 contiguous bytes (rows of 64 B per point 4-way conflict on the 32-bank write path:
// measured +30 cycles per ds_write_b128), and the 4 rows x 4 chunks a 16-lane group of the transposing read gathers
// lie in 16 different slots.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ int tr_slot(int row, int chunk) { return chunk * 64 + ((row ^ (4 * chunk)) * 4); }   // floats
template <class C, int NP>
__device__ __forceinline__ void tr_store(real* img, int woff, const bf16x8 (&pl)[3]) {     // woff = tr_slot(p, q)
#pragma unroll
  for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(img + k * C::trPlane + woff) = pl[k];
}
// 8 contraction slots of one lane: two transposing reads (4 points each) of the same unit
__device__ __forceinline__ bf16x8 tr_read8(const real* a0, const real* a1) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <class C> struct KeptPlanes {
  // with one wave per SIMD there are registers to spare: the activation streams of every layer are kept from the
  // forward pass instead of being recomputed for the weight-gradient GEMMs and the output-layer gradient
  real4 h[C::KEEP_H ? C::L : 1][C::NS][C::NB];
};

template <class C, bool BWD>
__device__ __forceinline__ void tile_forward(const real* lds, int lane, int q, const real (&x)[C::D],
                                             LayerState<C> (&st)[C::L], real4 (&h)[C::NS][C::NB], KeptPlanes<C>& kp,
                                             real* stage = nullptr) {
  first_layer<C, BWD>(lds, q, x, st[0]);
  sfor<C::L - 1>([&](auto li_) {
    constexpr int li = decltype(li_)::value;  // computes layer l = li + 2 from layer li + 1
    if constexpr (C::BF16 && C::WIDE) {
      hidden_layer_grouped<C, BWD>(lds, li + 2, lane, q, st[li], st[li + 1]);
    } else {
      act_forward<C>(st[li], h);
      if constexpr (BWD && C::KEEP_H && !C::WG_TR) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
          for (int b = 0; b < C::NB; ++b) kp.h[li][s][b] = h[s][b];
      }
      if constexpr (C::BF16) {
        Planes<C> P;
        split_all<C>(h, P);
        if constexpr (BWD && C::WG_TR) {       // the reverse pass reads the planes back transposed (hbar_wgrad_tr)
#pragma unroll
          for (int s = 0; s < C::NS; ++s) tr_store<C, NDQ_HTR_PLANES>(stage + (li * C::NS + s) * 3 * C::trPlane, tr_slot(lane & 15, q), P.pl[s][0]);
        }
        hidden_layer_planes<C, BWD>(lds, li + 2, lane, q, P, st[li + 1]);
      } else {
        hidden_layer<C, BWD>(lds, li + 2, lane, q, h, st[li + 1]);
      }
    }
  });
  act_forward<C>(st[C::L - 1], h);
  if constexpr (BWD && C::KEEP_H) {
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int b = 0; b < C::NB; ++b) kp.h[C::L - 1][s][b] = h[s][b];
  }
}

// ------------------------------------------------------------------------------------------------ forward kernel
template <class C>
__global__ __launch_bounds__(C::FWD_THREADS) void mlp_jet_fwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int wavesPerBlock = blockDim.x >> 6;
  const int ntiles = (a.n + 15) >> 4;
  // the first tile's coordinates are requested before the weights are staged, every later tile's one tile ahead (round 5:
  // the load sat at the top of the tile loop -- one exposed HBM round trip per tile on a SIMD that runs a single wave; the
  // closure kernels always prefetched)
  real xn[C::D];
  {
    const int n0 = (blockIdx.x * wavesPerBlock + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  stage_weights<C, false>(lds, a.params);
  __syncthreads();
  for (int tile = blockIdx.x * wavesPerBlock + wave; tile < ntiles; tile += gridDim.x * wavesPerBlock) {
    const int n = tile * 16 + p;
    real x[C::D];
#pragma unroll
    for (int d = 0; d < C::D; ++d) x[d] = xn[d];
    {
      const int n1 = n + gridDim.x * wavesPerBlock * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
    const real* ldsw = lds + opaque_zero<C>();
    LayerState<C> st;                      // one state, reused layer after layer (nothing is kept for a reverse pass)
    first_layer<C, false>(ldsw, q, x, st);
    real4 h[C::NS][C::NB];
#pragma unroll
    for (int l = 2; l <= C::L; ++l) {
      constexpr bool X1 = NDQ_FWD_BF16X1 != 0;
      if constexpr (C::BF16 && C::WIDE) {
        hidden_layer_grouped<C, false, X1>(ldsw, l, lane, q, st, st);
      } else {
        act_forward<C>(st, h);
        if constexpr (C::BF16) gemm_layer_bf16<C, X1>(ldsw, l, lane, q, h, st);
        else hidden_layer<C, false>(ldsw, l, lane, q, h, st);
      }
    }
    act_forward<C>(st, h);
    if constexpr (C::NOUT == 1) {
      real out[C::NS];
      tile_output<C, false>(ldsw, q, x, h, out);
      if (q == 0 && n < a.n) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) a.jets[(size_t)s * a.ldj + n] = out[s];
      }
    } else {
      real4 o[C::NS][C::NBO];
      output_layer_mfma<C, false>(ldsw, lane, q, h, o);
      output_skip_multi<C, false>(ldsw, q, x, o);
      if (n < a.n) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
          for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = 16 * ob + 4 * q + r;
              if (u < C::NOUT) a.jets[((size_t)s * C::NOUT + u) * a.ldj + n] = o[s][ob][r];
            }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward kernel
// number of per-wave LDS reduction regions that fit next to the weights (workgroup LDS budget 160 KiB - margin)
template <class C>
constexpr int bwd_regions(int waves) {
  const int pp = (C::P + 3) & ~3;
  const int budget = (38 * 1024 * 4) / (int)sizeof(real) - C::ldsWeightsEnd(true) - waves * C::biasFloats;   // reals (152 KB)
  int r = budget / pp;
  if (r < 1) r = 1;
  return r > waves ? waves : r;
}

// per-wave gradient accumulators (registers), summed over all tiles the wave processes
template <class C>
struct GradAcc {
  real w1[C::NIN][C::NB][4];        // dW1[j][a], j = 16b+4q+r   (needs point_sum; MONO: a runs over the NIN features)
  real b1[C::NB][4];                // db1[j]                    (needs point_sum)
  real4 w[C::L > 1 ? C::L - 1 : 1][C::NB][C::NB];  // dW_l[16jb+4q+r][16kb+p], MFMA accumulators (already summed)
  f32x16 w32[C::L > 1 ? C::L - 1 : 1][2][2];       // WG32: the same as 32x32 blocks (w is then unused)
  real b[C::L > 1 ? C::L - 1 : 1][C::NB][4];      // db_l[j]     (needs point_sum)
  real wout[C::NB][4];              // NOUT == 1: dWout[j]       (needs point_sum)
  real bout;                        // NOUT == 1: dbout          (needs full wave sum)
  real skip[C::D];                  // SKIP, NOUT == 1: dS[a]    (needs full wave sum)
  real so[C::D][C::NBO][4];         // SKIP, NOUT > 1: dS[16ob+4q+r][a] (needs point_sum)
  real ap[C::L][3];                 // ACTP: per layer, see act_backward (needs full wave sum)
  real4 wo[C::NBO][C::NB];           // NOUT > 1: dWout[16ob+4q+r][16kb+p], MFMA accumulators
  real bo[C::NBO][4];               // NOUT > 1: dbout[16ob+4q+r] (needs point_sum)
  real* bias;                       // ACC_LDS: this wave's LDS region holding b1 / w1 / b / wout instead (already point-summed)
};

// ACC_LDS: add the tile's point-sum of v (one value per unit j = 16b+4q+r, per lane group) to this wave's LDS sums.
// Only this wave touches its region and LDS operations of a wave are issued in order, so the order of additions --
// tile after tile -- is fixed.
template <class C>
__device__ __forceinline__ void bias_accum(GradAcc<C>& acc, int off, int p, int q, int b, const real (&v)[4]) {
  real s[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = point_sum(v[r]);
  if (p == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __hip_atomic_fetch_add(acc.bias + off + 16 * b + 4 * q + r, s[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}

// first-layer derivative streams (columns of W1) re-read from LDS
template <class C>
__device__ __forceinline__ void reload_first_layer_streams(const real* lds, int q, LayerState<C>& st) {
  if constexpr (C::SS::FIRST) {
#pragma unroll
    for (int b = 0; b < C::NB; ++b)
#pragma unroll
      for (int a = 0; a < C::D; ++a) st.z[1 + a][b] = lds4(lds + C::ldsW1T + a * C::H + 16 * b + 4 * q);
  }
}

// dW_l += sum_s Zbar[s] H[s]^T over the 16 points of the tile, stream by stream through the LDS transpose tile.
// stage: per-wave region of 2*16*HP floats.  Point <-> MFMA k mapping: k = q at step st  <->  point 4*q + st,
// which makes both the b128 writes (8-lane groups: bank stride 4*(HP mod 8) ... HP = H+4 -> 16 B apart) and the b32
// reads (32-lane groups: q*4*HP = 16 banks apart) conflict-free.
template <class C, int NBA, bool HIDDEN = false>
__device__ __forceinline__ void weight_grad(real* stage, int lane, int p, int q, const real4 (&zb)[C::NS][NBA],
                                            const LayerState<C>& st_in, real4 (&acc)[NBA][C::NB],
                                            const real4 (*hkept)[C::NB] = nullptr, f32x16 (*acc32)[2] = nullptr) {
  constexpr int HP = C::HP, SB = C::WG_SB;
  constexpr int NR = (C::NS + SB - 1) / SB;           // barrier rounds
  if constexpr ((NDQ_ABL & 1) != 0) return;
  sfor<NR>([&](auto r_) {
    constexpr int s0 = decltype(r_)::value * SB;
    constexpr int sn = (C::NS - s0 < SB) ? C::NS - s0 : SB;
    sfor<sn>([&](auto k_) {
      constexpr int s = s0 + decltype(k_)::value;
      real* Zt = stage + decltype(k_)::value * 2 * 16 * HP;
      real* Ht = Zt + 16 * HP;
      real4 hs[C::NB];
      if constexpr (C::KEEP_H) {              // stream s of the layer's input activations: kept by the forward pass ...
#pragma unroll
        for (int b = 0; b < C::NB; ++b) hs[b] = hkept[s][b];
      } else {
        act_forward_stream<C, s>(st_in, hs);  // ... or recomputed from the layer state
      }
      if constexpr ((NDQ_ABL & 16) == 0) {
#pragma unroll
        for (int b = 0; b < NBA; ++b) *reinterpret_cast<real4*>(Zt + p * HP + 16 * b + 4 * q) = zb[s][b];
#pragma unroll
        for (int b = 0; b < C::NB; ++b) *reinterpret_cast<real4*>(Ht + p * HP + 16 * b + 4 * q) = hs[b];
      }
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (C::WG32 && HIDDEN) {
      // one stream is staged (SB = 1).  Lane (i = lane & 31, g = lane >> 5) of an operand holds unit 32 blk + i at the 8
      // points 8 g + e: slot 8 g + e of the 16-wide contraction on both operands.
      static_assert(SB == 1, "WG32 stages one stream per round");
      const int i32 = lane & 31, g8 = 8 * (lane >> 5);
      const real* Zt = stage + g8 * HP + i32;
      const real* Ht = Zt + 16 * HP;
      auto planes_of = [&](const real* t, int blk, bf16x8 (&pl)[3]) {
        real x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = t[e * HP + 32 * blk];
        split3<NDQ_HTR_PLANES>(real4{x[0], x[1], x[2], x[3]}, real4{x[4], x[5], x[6], x[7]}, pl);
      };
      bf16x8 pb[2][3];
#pragma unroll
      for (int b = 0; b < 2; ++b) planes_of(Ht, b, pb[b]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 pa[3];
        planes_of(Zt, jb, pa);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#define NDQ_W(I, J) acc32[jb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[I], pb[kb][J], acc32[jb][kb], 0, 0, 0);
          NDQ_WPRODUCTS(NDQ_W)
#undef NDQ_W
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      return;
    }
    if constexpr (C::WG_BF16) {
      // contraction slot 8 kg + e of the 32-wide bf16 MFMA <-> stream s0 + (kg >> 1), point (e & 3) + 8 (e >> 2) + 4 (kg & 1)
      // (the same map on both operands; this point order keeps the b32 reads of the two lane groups of a half-wave
      // 16 banks apart: 4 HP = 16 mod 32).  A lane therefore reads 8 points of ONE unit of ONE stream per block,
      // splits them into three bf16 planes, and the six significant plane products go through the matrix core.
      const int kg = lane >> 4;
      const bool live = sn == 2 || (kg >> 1) < sn; // a round with one stream: the upper half of the slots is zero
      const real* Zt = stage + (live ? (kg >> 1) : 0) * 2 * 16 * HP + (4 * (kg & 1)) * HP + (lane & 15);
      const real* Ht = Zt + 16 * HP;
      auto planes_of = [&](const real* t, int b, bf16x8 (&pl)[3]) {
        real x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const real v = t[((e & 3) + 8 * (e >> 2)) * HP + 16 * b];
          if constexpr (sn == 2) x[e] = v;
          else x[e] = live ? v : 0.f;
        }
        split3<NDQ_HTR_PLANES>(real4{x[0], x[1], x[2], x[3]}, real4{x[4], x[5], x[6], x[7]}, pl);
      };
      bf16x8 pb[C::NB][3];
#pragma unroll
      for (int b = 0; b < C::NB; ++b) planes_of(Ht, b, pb[b]);
#pragma unroll
      for (int jb = 0; jb < NBA; ++jb) {
        bf16x8 pa[3];
        planes_of(Zt, jb, pa);
#pragma unroll
        for (int kb = 0; kb < C::NB; ++kb) {
#define NDQ_W(I, J) acc[jb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[I], pb[kb][J], acc[jb][kb], 0, 0, 0);
          NDQ_WPRODUCTS(NDQ_W)
#undef NDQ_W
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      return;
    }
    // all operands of the round are read into registers of their own BEFORE the first MFMA: one LDS round trip per
    // round instead of one per k-step (the compiler otherwise recycles four registers and waits 16 times per stream)
    if constexpr (C::BWD_THREADS != 256) {
      // two waves per SIMD, 256 registers each: no room for a round's operands at once -- step by step
      sfor<sn>([&](auto k_) {
        const real* Zt = stage + decltype(k_)::value * 2 * 16 * HP;
        const real* Ht = Zt + 16 * HP;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          real a1[NBA], b1[C::NB];
#pragma unroll
          for (int b = 0; b < NBA; ++b) a1[b] = Zt[(4 * q + st) * HP + 16 * b + mrow(p)];
#pragma unroll
          for (int b = 0; b < C::NB; ++b) b1[b] = Ht[(4 * q + st) * HP + 16 * b + p];
#pragma unroll
          for (int jb = 0; jb < NBA; ++jb)
#pragma unroll
            for (int kb = 0; kb < C::NB; ++kb)
              acc[jb][kb] = mfma16x16x4(a1[jb], b1[kb], acc[jb][kb]);
        }
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      return;
    }
    real av[sn][4][NBA], bv[sn][4][C::NB];
    sfor<sn>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      const real* Zt = stage + k * 2 * 16 * HP;
      const real* Ht = Zt + 16 * HP;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int b = 0; b < NBA; ++b) av[k][st][b] = Zt[(4 * q + st) * HP + 16 * b + mrow(p)];
#pragma unroll
        for (int b = 0; b < C::NB; ++b) bv[k][st][b] = Ht[(4 * q + st) * HP + 16 * b + p];
      }
    });
#if NDQ_WG_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);          // keep the reads above the MFMAs
#endif
    sfor<sn>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int jb = 0; jb < NBA; ++jb)
#pragma unroll
          for (int kb = 0; kb < C::NB; ++kb)
            acc[jb][kb] = mfma16x16x4(av[k][st][jb], bv[k][st][kb], acc[jb][kb]);
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  });
}

// Cfg::WG_TR: hbar = W_l^T zbar and dW_l += sum_s Zbar_s H_s^T of one hidden layer, two streams per round.
//   ds_read_b64_tr_b16 (lane map measured by scripts/ubench_tr16.hip): result j of lane l is element (l & 3) of the 8
//   bytes addressed by lane (l & ~15) + 4 j + ((l & 15) >> 2).  Lane (i = l & 15, kg = l >> 4) points at the row of
//   point 4 (kg & 1) + 8 h + (i >> 2), chunk i & 3, block b's half of the 16 bytes, of stream kg >> 1 of the round: it
//   receives unit 16 b + i at the points 4 (kg & 1) + 8 h + j -- slot 8 kg + 4 h + j of the 32-wide contraction, the
//   same map on both operands.  A round with one stream has zero planes in the second Zbar slot (its lanes read H of
//   the first).  LDS operations of a wave execute in order, so the "barriers" below only pin the compiler's order.
template <class C, int LI>
__device__ __forceinline__ void hbar_wgrad_tr(const real* __restrict__ wl, real* stage, int lane, int p, int q,
                                              real4 (&g)[C::NS][C::NB], real4 (&acc)[C::NB][C::NB]) {
  static_assert(C::NB == 2 && C::NC == 1, "WG_TR: H = 32");
  constexpr int NS = C::NS, PL = C::trPlane, NR = (NS + 1) / 2;
  real* zimg = stage + C::trHimg;
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
  const int kg = lane >> 4, i = lane & 15, strm = kg >> 1;
  const int woff = tr_slot(p, q);
  const int r0 = tr_slot(4 * (kg & 1) + (i >> 2), i & 3), r1 = tr_slot(4 * (kg & 1) + 8 + (i >> 2), i & 3);
  sfor<NR>([&](auto r_) {
    constexpr int s0 = 2 * decltype(r_)::value;
    constexpr int sn = (NS - s0 < 2) ? NS - s0 : 2;
    // H operand of the round: written by the forward pass, independent of everything below -- with the whole register
    // file (one wave per SIMD) it is in flight first, with 256 registers only once the split planes are stored
    constexpr bool ROOMY = C::BWD_THREADS == 256;
    bf16x8 pb[2][3];
    auto read_h = [&]() {
      const real* hr = stage + (LI * NS + s0 + (sn == 2 ? strm : 0)) * 3 * PL;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k = 0; k < NDQ_HTR_PLANES; ++k) pb[kb][k] = tr_read8(hr + k * PL + kb * 2 + r0, hr + k * PL + kb * 2 + r1);
    };
    if constexpr ((NDQ_ABL & 1) == 0 && ROOMY) read_h();
    {
      bf16x8 pl[sn][3];
      real4 o[sn][2];
#pragma unroll
      for (int s = 0; s < sn; ++s) {
        split3<NDQ_Z_PLANES>(g[s0 + s][0], g[s0 + s][1], pl[s]);
        tr_store<C, NDQ_HTR_PLANES>(zimg + s * 3 * PL, woff, pl[s]);
        o[s][0] = real4{0.f, 0.f, 0.f, 0.f};
        o[s][1] = real4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (sn == 1) {
        bf16x8 zero[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) zero[k][e] = (__bf16)0.f;
        tr_store<C, NDQ_HTR_PLANES>(zimg + 3 * PL, woff, zero);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // Zbar operand: read back right away, the hbar MFMAs run while it is on its way
      bf16x8 pa[2][3];
      const real* zr = zimg + strm * 3 * PL;
      if constexpr ((NDQ_ABL & 1) == 0) {
        if constexpr (!ROOMY) read_h();
#pragma unroll
        for (int jb = 0; jb < (ROOMY ? 2 : 1); ++jb)
#pragma unroll
          for (int k = 0; k < NDQ_HTR_PLANES; ++k) pa[jb][k] = tr_read8(zr + k * PL + jb * 2 + r0, zr + k * PL + jb * 2 + r1);
      }
      if constexpr ((NDQ_ABL & 2) == 0) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          const bf16x8 a0 = w[(ob * 3 + 0) * 64 + lane];
          const bf16x8 a1 = w[(ob * 3 + 1) * 64 + lane];
          const bf16x8 a2 = w[(ob * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < sn; ++s)                                                             \
      o[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][K], o[s][ob], 0, 0, 0);
          NDQ_PRODUCTS_BWD(NDQ_T)
#undef NDQ_T
        }
      }
#pragma unroll
      for (int s = 0; s < sn; ++s) { g[s0 + s][0] = o[s][0]; g[s0 + s][1] = o[s][1]; }
      if constexpr ((NDQ_ABL & 1) == 0 && ROOMY) {
        // product by product over the four accumulator blocks: consecutive MFMAs never wait for one another
#define NDQ_W(I, J)                                                                                          \
  _Pragma("unroll") for (int jb = 0; jb < 2; ++jb) _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)          \
      acc[jb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[jb][I], pb[kb][J], acc[jb][kb], 0, 0, 0);
        NDQ_WPRODUCTS(NDQ_W)
#undef NDQ_W
      } else if constexpr ((NDQ_ABL & 1) == 0) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          if (jb == 1) {
#pragma unroll
            for (int k = 0; k < NDQ_HTR_PLANES; ++k) pa[0][k] = tr_read8(zr + k * PL + 2 + r0, zr + k * PL + 2 + r1);
          }
#define NDQ_W(I, J)                                                                                          \
  _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                           \
      acc[jb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0][I], pb[kb][J], acc[jb][kb], 0, 0, 0);
          NDQ_WPRODUCTS(NDQ_W)
#undef NDQ_W
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  });
}

// Cfg::WG_TR64: hbar = W_l^T zbar and dW_l += sum_s Zbar_s H_s^T of one hidden layer of an H = 64 network, SG streams
// per group (the weight fragments of the hbar GEMM are read once per group).  A plane image of one stream is two
// H = 32 images side by side (units 32 c .. 32 c + 31); the 32x32x16 operand of macro-block c: lane (i = l & 31,
// kg = l >> 5) receives unit 32 c + i at the points 8 kg + 4 h + j from the two transposing reads h = 0, 1 -- lanes
// 0..15 and 16..31 of a half-wave gather the two 8-byte halves of the same 16-byte slots.
template <class C>
__device__ __forceinline__ void hbar_wgrad_tr64(const real* __restrict__ wl, real* stage, int lane, int p, int q,
                                                const LayerState<C>& st_in, real4 (&g)[C::NS][C::NB], f32x16 (&acc32)[2][2]) {
  static_assert(C::NB == 4 && C::NC == 2, "WG_TR64: H = 64");
  constexpr int NS = C::NS, PL = C::trPlane64, NG = (NS + C::SG - 1) / C::SG;
  real* zimg = stage;
  real* himg = stage + 3 * PL;
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
  const int i16 = lane & 15, kg = lane >> 5;
  const int woff = tr_slot(p, q);
  const int r0 = tr_slot(8 * kg + (i16 >> 2), i16 & 3) + 2 * ((lane >> 4) & 1), r1 = tr_slot(8 * kg + 4 + (i16 >> 2), i16 & 3) + 2 * ((lane >> 4) & 1);
  sfor<NG>([&](auto g_) {
    constexpr int s0 = decltype(g_)::value * C::SG;
    constexpr int sn = (NS - s0 < C::SG) ? NS - s0 : C::SG;
    bf16x8 pl[sn][2][3];
#pragma unroll
    for (int s = 0; s < sn; ++s)
#pragma unroll
      for (int c = 0; c < 2; ++c) split3<NDQ_Z_PLANES>(g[s0 + s][2 * c], g[s0 + s][2 * c + 1], pl[s][c]);
    if constexpr ((NDQ_ABL & 1) == 0) {
      sfor<sn>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < NDQ_HTR_PLANES; ++k) *reinterpret_cast<bf16x8*>(zimg + k * PL + c * 256 + woff) = pl[s][c][k];
        {
          real4 hs[C::NB];
          act_forward_stream<C, s0 + s>(st_in, hs);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            bf16x8 ph[3];
            split3<NDQ_HTR_PLANES>(hs[2 * c], hs[2 * c + 1], ph);
#pragma unroll
            for (int k = 0; k < NDQ_HTR_PLANES; ++k) *reinterpret_cast<bf16x8*>(himg + k * PL + c * 256 + woff) = ph[k];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bf16x8 pb[2][3];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int k = 0; k < NDQ_HTR_PLANES; ++k) pb[kb][k] = tr_read8(himg + k * PL + kb * 256 + r0, himg + k * PL + kb * 256 + r1);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          bf16x8 pa[3];
#pragma unroll
          for (int k = 0; k < NDQ_HTR_PLANES; ++k) pa[k] = tr_read8(zimg + k * PL + jb * 256 + r0, zimg + k * PL + jb * 256 + r1);
#define NDQ_W(I, J)                                                                                          \
  _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                           \
      acc32[jb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[I], pb[kb][J], acc32[jb][kb], 0, 0, 0);
          NDQ_WPRODUCTS(NDQ_W)
#undef NDQ_W
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      });
    }
    real4 o[sn][C::NB];
#pragma unroll
    for (int s = 0; s < sn; ++s)
#pragma unroll
      for (int b = 0; b < C::NB; ++b) o[s][b] = real4{0.f, 0.f, 0.f, 0.f};
    if constexpr ((NDQ_ABL & 2) == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ob = 0; ob < C::NB; ++ob) {
          const bf16x8 a0 = w[((ob * 2 + c) * 3 + 0) * 64 + lane];
          const bf16x8 a1 = w[((ob * 2 + c) * 3 + 1) * 64 + lane];
          const bf16x8 a2 = w[((ob * 2 + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < sn; ++s)                                                             \
      o[s][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][c][K], o[s][ob], 0, 0, 0);
          NDQ_PRODUCTS_BWD(NDQ_T)
#undef NDQ_T
        }
    }
#pragma unroll
    for (int s = 0; s < sn; ++s)
#pragma unroll
      for (int b = 0; b < C::NB; ++b) g[s0 + s][b] = o[s][b];
  });
}

// one stream of hbar = W^T zbar in place: g[s] <- sum over (ib, t) of A(w) * B(g[s][ib][t]); stream by stream so that only
// one extra fragment is live (the two output blocks alternate as MFMA accumulators, 64 cycles apart > 40 latency)
template <class C>
__device__ __forceinline__ void gemm_frag_inplace(const real* __restrict__ w, int lane, real4 (&g)[C::NS][C::NB]) {
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    real4 o[C::NB];
#pragma unroll
    for (int b = 0; b < C::NB; ++b) o[b] = real4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < C::NB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ib = 0; ib < C::NB; ++ib)
          o[ib] = mfma16x16x4(w[((ib * C::NB + kb) * 4 + t) * 64 + lane], g[s][kb][t], o[ib]);
#pragma unroll
    for (int b = 0; b < C::NB; ++b) g[s][b] = o[b];
  }
}

template <class C>
__device__ __forceinline__ void acc_zero(GradAcc<C>& acc) {
#pragma unroll
  for (int b = 0; b < C::NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int d = 0; d < C::NIN; ++d) acc.w1[d][b][r] = 0.f;
      acc.b1[b][r] = 0.f;
      acc.wout[b][r] = 0.f;
#pragma unroll
      for (int l = 0; l < C::L - 1; ++l) acc.b[l][b][r] = 0.f;
    }
#pragma unroll
  for (int l = 0; l < C::L - 1; ++l)
#pragma unroll
    for (int jb = 0; jb < C::NB; ++jb)
#pragma unroll
      for (int kb = 0; kb < C::NB; ++kb) acc.w[l][jb][kb] = real4{0.f, 0.f, 0.f, 0.f};
  if constexpr (C::WG32) {
#pragma unroll
    for (int l = 0; l < C::L - 1; ++l)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc.w32[l][jb][kb][e] = 0.f;
  }
  acc.bout = 0.f;
#pragma unroll
  for (int d = 0; d < C::D; ++d) acc.skip[d] = 0.f;
#pragma unroll
  for (int ob = 0; ob < C::NBO; ++ob) {
#pragma unroll
    for (int kb = 0; kb < C::NB; ++kb) acc.wo[ob][kb] = real4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.bo[ob][r] = 0.f;
#pragma unroll
    for (int d = 0; d < C::D; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.so[d][ob][r] = 0.f;
  }
#pragma unroll
  for (int l = 0; l < C::L; ++l) acc.ap[l][0] = acc.ap[l][1] = acc.ap[l][2] = 0.f;
}

// start of the per-wave bias-sum regions (ACC_LDS), behind the staging tiles / reduction regions
template <class C> __device__ __forceinline__ real* bias_region(real* lds, int waves, int wave) {
  const int pp = (C::P + 3) & ~3;
  const int stage = waves * C::stageFloatsPerWave;
  const int red = bwd_regions<C>(waves) * pp;
  return lds + C::ldsWeightsEnd(true) + (stage > red ? stage : red) + wave * C::biasFloats;
}
template <class C> __device__ __forceinline__ void acc_init(GradAcc<C>& acc, real* lds, int waves, int wave, int lane) {
  acc_zero<C>(acc);
  acc.bias = nullptr;
  if constexpr (C::ACC_LDS) {
    acc.bias = bias_region<C>(lds, waves, wave);
    for (int i = lane; i < C::biasFloats; i += 64) acc.bias[i] = 0.f;
  }
}


template <class C>
__device__ __forceinline__ void tile_backward_hidden(const real* lds, real* stage, int lane, int p, int q,
                                                     const real (&x)[C::D], LayerState<C> (&st)[C::L],
                                                     real4 (&g)[C::NS][C::NB], GradAcc<C>& acc, KeptPlanes<C>& kp);

// reverse pass of one tile, multi-output network: go[s][ob] = dLoss/d out[s][16ob+4q+r] for the tile's points
template <class C>
__device__ __forceinline__ void tile_backward_multi(const real* lds, real* stage, int lane, int p, int q,
                                                    const real (&x)[C::D], const real4 (&go)[C::NS][C::NBO],
                                                    LayerState<C> (&st)[C::L], GradAcc<C>& acc, KeptPlanes<C>& kp) {
#pragma unroll
  for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.bo[ob][r] += go[0][ob][r];
  if constexpr (C::SKIP != 0) {
#pragma unroll
    for (int a = 0; a < C::D; ++a)
#pragma unroll
      for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real v = go[0][ob][r] * x[a];
          if constexpr (C::SS::FIRST) v += go[1 + a][ob][r];
          acc.so[a][ob][r] += v;
        }
  }
  weight_grad<C, C::NBO>(stage, lane, p, q, go, st[C::L - 1], acc.wo, kp.h[C::KEEP_H ? C::L - 1 : 0]);   // dWout += sum_s Gout[s] H_L[s]^T
  real4 g[C::NS][C::NB];
  zero_frag<C>(g);
  if constexpr (C::BF16O) {                                                // hbar_L = Wout^T gout on the bf16 matrix core
    bf16x8 pl[C::NS][C::NCO][3];
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int c = 0; c < C::NCO; ++c) split3<(NDQ_HBAR_NPROD == 6 ? 3 : 2)>(go[s][2 * c], go[s][2 * c + 1], pl[s][c]);
    const bf16x8* wb = reinterpret_cast<const bf16x8*>(lds + C::ldsWoutT());
#pragma unroll
    for (int c = 0; c < C::NCO; ++c)
#pragma unroll
      for (int kb = 0; kb < C::NB; ++kb) {
        const bf16x8 a0 = wb[((kb * C::NCO + c) * 3 + 0) * 64 + lane];
        const bf16x8 a1 = wb[((kb * C::NCO + c) * 3 + 1) * 64 + lane];
        const bf16x8 a2 = wb[((kb * C::NCO + c) * 3 + 2) * 64 + lane];
#define NDQ_T(A, K)                                                                                          \
  _Pragma("unroll") for (int s = 0; s < C::NS; ++s)                                                          \
      g[s][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, pl[s][c][K], g[s][kb], 0, 0, 0);
        NDQ_PRODUCTS_BWD(NDQ_T)
#undef NDQ_T
      }
    tile_backward_hidden<C>(lds, stage, lane, p, q, x, st, g, acc, kp);
    return;
  }
  const real* w = lds + C::ldsWoutT();
#pragma unroll
  for (int kb = 0; kb < C::NB; ++kb)                                       // hbar_L = Wout^T gout
#pragma unroll
    for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const real a = w[((kb * C::NBO + ob) * 4 + t) * 64 + lane];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) g[s][kb] = mfma16x16x4(a, go[s][ob][t], g[s][kb]);
      }
  tile_backward_hidden<C>(lds, stage, lane, p, q, x, st, g, acc, kp);
}

// reverse pass of one tile: gout[s] = dLoss/d out[s] for the tile's points (0 for padding lanes)
template <class C>
__device__ __forceinline__ void tile_backward(const real* lds, real* stage, int lane, int p, int q,
                                              const real (&x)[C::D], const real (&gout)[C::NS],
                                              LayerState<C> (&st)[C::L], GradAcc<C>& acc, KeptPlanes<C>& kp) {
  // ---------------- output layer adjoint (n_out = 1): hbar = Wout * gout; dWout += sum_s gout_s h_s; dbout += gout_0
  // (h_s of the last hidden layer is recomputed per stream from its state instead of being kept live)
  if constexpr (C::LAUNDER) {
#pragma unroll
    for (int l = 0; l < C::L; ++l)
#pragma unroll
      for (int b = 0; b < C::NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(st[l].t[b][r]));
  }
  real4 g[C::NS][C::NB];
  {
    real4 dw[C::NB];
#pragma unroll
    for (int b = 0; b < C::NB; ++b) dw[b] = real4{0.f, 0.f, 0.f, 0.f};
    sfor<C::NS>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      real4 hs[C::NB];
      if constexpr (C::KEEP_H) {
#pragma unroll
        for (int b = 0; b < C::NB; ++b) hs[b] = kp.h[C::L - 1][s][b];
      } else {
        act_forward_stream<C, s>(st[C::L - 1], hs);
      }
#pragma unroll
      for (int b = 0; b < C::NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) dw[b][r] = rfma(gout[s], hs[b][r], dw[b][r]);
    });
#pragma unroll
    for (int b = 0; b < C::NB; ++b) {
      const real4 wo = lds4(lds + C::ldsWout(true) + 16 * b + 4 * q);
      if constexpr (C::ACC_LDS) {
        const real v[4] = {dw[b][0], dw[b][1], dw[b][2], dw[b][3]};
        bias_accum<C>(acc, C::biasWout, p, q, b, v);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (!C::ACC_LDS) acc.wout[b][r] += dw[b][r];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) g[s][b][r] = wo[r] * gout[s];
      }
    }
  }
  acc.bout += (q == 0) ? gout[0] : 0.f;
  NDQ_TT(4);
  if constexpr (C::SKIP != 0) {
#pragma unroll
    for (int a = 0; a < C::D; ++a) {
      real v = gout[0] * x[a];
      if constexpr (C::SS::FIRST) v += gout[1 + a];
      acc.skip[a] += (q == 0) ? v : 0.f;
    }
  }
  tile_backward_hidden<C>(lds, stage, lane, p, q, x, st, g, acc, kp);
}

// hidden layers L .. 2 and the first layer, given g = hbar of the last hidden layer
template <class C>
__device__ __forceinline__ void tile_backward_hidden(const real* lds, real* stage, int lane, int p, int q,
                                                     const real (&x)[C::D], LayerState<C> (&st)[C::L],
                                                     real4 (&g)[C::NS][C::NB], GradAcc<C>& acc, KeptPlanes<C>& kp) {
  using SS = typename C::SS;
  // ---------------- hidden layers L .. 2
  sfor<C::L - 1>([&](auto k_) {
    constexpr int l = C::L - decltype(k_)::value;          // layer whose weights W_l (H x H) map h_{l-1} -> z_l
    constexpr int li = l - 1;             // state index of layer l
    if constexpr ((NDQ_ABL & 8) == 0) act_backward<C>(st[li], g, acc.ap[li]);           // g: hbar_l -> zbar_l
    NDQ_TT(5 + 3 * (C::L - l));
#pragma unroll
    for (int b = 0; b < C::NB; ++b) {
      if constexpr (C::ACC_LDS) {
        const real v[4] = {g[0][b][0], g[0][b][1], g[0][b][2], g[0][b][3]};
        bias_accum<C>(acc, C::biasBl + (l - 2) * C::H, p, q, b, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc.b[l - 2][b][r] += g[0][b][r];
      }
    }
    if constexpr (C::WIDE && li == 1) reload_first_layer_streams<C>(lds, q, st[0]);   // needed from here on again
    if constexpr (C::WG_TR) {
      hbar_wgrad_tr<C, li - 1>(lds + C::ldsWt(l), stage, lane, p, q, g, acc.w[l - 2]);
      NDQ_TT(6 + 3 * (C::L - l));
      NDQ_TT(7 + 3 * (C::L - l));
    } else if constexpr (C::WG_TR64) {
      hbar_wgrad_tr64<C>(lds + C::ldsWt(l), stage, lane, p, q, st[li - 1], g, acc.w32[l - 2]);
      NDQ_TT(6 + 3 * (C::L - l));
      NDQ_TT(7 + 3 * (C::L - l));
    } else if constexpr (C::BF16) {
      weight_grad<C, C::NB, true>(stage, lane, p, q, g, st[li - 1], acc.w[l - 2], kp.h[C::KEEP_H ? li - 1 : 0], acc.w32[l - 2]);
      NDQ_TT(6 + 3 * (C::L - l));
      if constexpr ((NDQ_ABL & 2) == 0) gemm_bf16x3_inplace<C>(lds + C::ldsWt(l), lane, g);
      NDQ_TT(7 + 3 * (C::L - l));
    } else {
      weight_grad<C, C::NB, true>(stage, lane, p, q, g, st[li - 1], acc.w[l - 2], kp.h[C::KEEP_H ? li - 1 : 0], acc.w32[l - 2]);   // inputs of layer l
      gemm_frag_inplace<C>(lds + C::ldsWt(l), lane, g);
    }
  });

  // ---------------- first layer: z_a = W1[:,a] (constant), z_ab = 0
  if constexpr (C::WIDE && C::L == 1) reload_first_layer_streams<C>(lds, q, st[0]);
  act_backward<C>(st[0], g, acc.ap[0]);  // g[0] = zbar, g[1+a] = zbar_a
  if constexpr (C::ACC_LDS) {
#pragma unroll
    for (int b = 0; b < C::NB; ++b) {
      const real v0[4] = {g[0][b][0], g[0][b][1], g[0][b][2], g[0][b][3]};
      bias_accum<C>(acc, C::biasB1, p, q, b, v0);
#pragma unroll
      for (int d = 0; d < C::D; ++d) {
        real v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = g[0][b][r] * x[d];
          if constexpr (SS::FIRST) v[r] += g[1 + d][b][r];
        }
        bias_accum<C>(acc, C::biasW1 + d * C::H, p, q, b, v);
      }
    }
  } else if constexpr (C::MONO != 0) {
    // dW1[j][(deg, a)] += zbar x_a^deg + zbar_a deg x_a^(deg-1) + zbar_aa deg (deg-1) x_a^(deg-2)
    real pw[C::D][C::MAXDEG + 1];
    mono_powers<C>(x, pw);
#pragma unroll
    for (int b = 0; b < C::NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real z0 = g[0][b][r];
        acc.b1[b][r] += z0;
        sfor<C::D>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          real g1 = 0.f, g2 = 0.f;
          if constexpr (SS::FIRST) {
            g1 = g[1 + a][b][r];
            if constexpr (SS::LAP) {
              if constexpr (SS::in_lap(a)) g2 = g[SS::S2][b][r];
            } else if constexpr (SS::pair_stream(a, a) >= 0) {
              g2 = g[SS::pair_stream(a, a)][b][r];
            }
          }
          sfor<C::NDEG>([&](auto i_) {
            constexpr int i = decltype(i_)::value, deg = C::mono_deg(i);
            real v = rfma(z0, pw[a][deg], g1 * ((real)deg * pw[a][deg - 1]));
            if constexpr (deg >= 2) v = rfma(g2, (real)(deg * (deg - 1)) * pw[a][deg - 2], v);
            acc.w1[i * C::D + a][b][r] += v;
          });
        });
      }
  } else {
#pragma unroll
    for (int b = 0; b < C::NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real z0 = g[0][b][r];
        acc.b1[b][r] += z0;
#pragma unroll
        for (int d = 0; d < C::D; ++d) {
          real v = z0 * x[d];
          if constexpr (SS::FIRST) v += g[1 + d][b][r];
          acc.w1[d][b][r] += v;
        }
      }
  }
}

// lanes -> wave -> workgroup (fixed order) -> out_row[P].
// Waves deposit their sums into R per-wave LDS regions (R = as many as fit; they overlay the transpose staging).
// Round k handles waves [k*R, (k+1)*R): round 0 stores, later rounds use no-return ds_add_f32 (each address is
// touched once per wave and rounds are separated by barriers, so the summation order is fixed); finally every thread
// adds the R regions in order and writes the workgroup's row of partials.
// RR != 0: number of regions given by the caller; tid / nt: this thread's index among the nt threads that reduce
// together (default: the whole workgroup) -- the multi-network closure reduces network by network, all at once.
template <class C, int WAVES, int RR = 0>
__device__ __forceinline__ void block_reduce_store(real* lds, GradAcc<C>& acc, int wave, int lane, int p, int q,
                                                   real* __restrict__ out, const real* __restrict__ prm = nullptr,
                                                   int tid = -1, int nt = 0) {
  constexpr int PP = (C::P + 3) & ~3;
  constexpr int R = RR != 0 ? RR : bwd_regions<C>(WAVES);
  if (tid < 0) { tid = threadIdx.x; nt = blockDim.x; }
  real* red0 = lds + C::ldsWeightsEnd(true);
  const real bsum = point_sum(quad_sum(acc.bout));
  real apsum[C::L][3];
#pragma unroll
  for (int l = 0; l < C::L; ++l)
#pragma unroll
    for (int k = 0; k < 3; ++k) apsum[l][k] = (C::ACTP == 1) ? point_sum(quad_sum(acc.ap[l][k])) : 0.f;
  real ssum[C::D];
#pragma unroll
  for (int a = 0; a < C::D; ++a) ssum[a] = (C::SKIP != 0) ? point_sum(quad_sum(acc.skip[a])) : 0.f;
#pragma unroll
  for (int b = 0; b < C::NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (C::ACC_LDS) {             // the per-unit sums were kept in this wave's LDS region
        const int j = 16 * b + 4 * q + r;
        acc.b1[b][r] = acc.bias[C::biasB1 + j];
        acc.wout[b][r] = acc.bias[C::biasWout + j];
#pragma unroll
        for (int d = 0; d < C::D; ++d) acc.w1[d][b][r] = acc.bias[C::biasW1 + d * C::H + j];
#pragma unroll
        for (int l = 0; l < C::L - 1; ++l) acc.b[l][b][r] = acc.bias[C::biasBl + l * C::H + j];
        continue;
      }
      acc.b1[b][r] = point_sum(acc.b1[b][r]);
      if constexpr (C::NOUT == 1) acc.wout[b][r] = point_sum(acc.wout[b][r]);
#pragma unroll
      for (int d = 0; d < C::NIN; ++d) acc.w1[d][b][r] = point_sum(acc.w1[d][b][r]);
#pragma unroll
      for (int l = 0; l < C::L - 1; ++l) acc.b[l][b][r] = point_sum(acc.b[l][b][r]);
    }
  if constexpr (C::NOUT > 1) {
#pragma unroll
    for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.bo[ob][r] = point_sum(acc.bo[ob][r]);
    if constexpr (C::SKIP != 0) {
#pragma unroll
      for (int a = 0; a < C::D; ++a)
#pragma unroll
        for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc.so[a][ob][r] = point_sum(acc.so[a][ob][r]);
    }
  }
  __syncthreads();  // every wave is done with its staging tile
  real* red = red0 + (wave % R) * PP;
  for (int k = 0; k * R < WAVES; ++k) {
    if (wave / R == k) {
      auto put = [&](int idx, real v) {
        if (k == 0) red[idx] = v;
        else __hip_atomic_fetch_add(&red[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
#pragma unroll
      for (int b = 0; b < C::NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * b + 4 * q + r;
          if (p == 0) {
            if (!C::RAGGED || j < C::hr(1)) {
              put(C::offb1 + j, acc.b1[b][r]);
#pragma unroll
              for (int d = 0; d < C::NIN; ++d) put(C::offW1 + j * C::NIN + d, acc.w1[d][b][r]);
            }
            if constexpr (C::NOUT == 1) {
              if (!C::RAGGED || j < C::hr(C::L)) put(C::offWout + j, acc.wout[b][r]);
            }
#pragma unroll
            for (int l = 0; l < C::L - 1; ++l) {
              if (!C::RAGGED || j < C::hr(l + 2)) put(C::offb(l + 2) + j, acc.b[l][b][r]);
            }
          }
        }
      if constexpr (C::WG32) {
#pragma unroll
        for (int l = 0; l < C::L - 1; ++l)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int j = 32 * jb + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), k = 32 * kb + (lane & 31);
                if (!C::RAGGED || (j < C::hr(l + 2) && k < C::hr(l + 1)))
                  put(C::offW(l + 2) + j * C::hr(l + 1) + k, acc.w32[l][jb][kb][e]);
              }
      } else {
#pragma unroll
      for (int l = 0; l < C::L - 1; ++l)
#pragma unroll
        for (int jb = 0; jb < C::NB; ++jb)
#pragma unroll
          for (int kb = 0; kb < C::NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 16 * jb + 4 * q + r, k = 16 * kb + p;
              if (!C::RAGGED || (j < C::hr(l + 2) && k < C::hr(l + 1)))
                put(C::offW(l + 2) + j * C::hr(l + 1) + k, acc.w[l][jb][kb][r]);
            }
      }
      if constexpr (C::NOUT == 1) {
        if (lane == 0) {
          put(C::offbout, bsum);
          if constexpr (C::SKIP != 0) {
#pragma unroll
            for (int a = 0; a < C::D; ++a) put(C::offS + a, ssum[a]);
          }
        }
      } else {
#pragma unroll
        for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * ob + 4 * q + r;
            if (u < C::NOUT) {
              if (p == 0) {
                put(C::offbout + u, acc.bo[ob][r]);
                if constexpr (C::SKIP != 0) {
#pragma unroll
                  for (int a = 0; a < C::D; ++a) put(C::offS + u * C::D + a, acc.so[a][ob][r]);
                }
              }
#pragma unroll
              for (int kb = 0; kb < C::NB; ++kb) {
                if (!C::RAGGED || 16 * kb + p < C::hr(C::L)) put(C::offWout + u * C::hr(C::L) + 16 * kb + p, acc.wo[ob][kb][r]);
              }
            }
          }
      }
      if constexpr (C::ACTP == 1) {           // activation parameters (formulas: Act<ACT_APTX>)
        if (lane == 0) {
#pragma unroll
          for (int l = 0; l < C::L; ++l) {
            const real dbeta = apsum[l][1] / act_pre<C>(prm, l + 1);
            if constexpr (C::AK == 1) {
              put(C::offA + l, dbeta);
            } else {
              put(C::offA + 3 * l, 0.5f * apsum[l][0]);
              put(C::offA + 3 * l + 1, dbeta);
              put(C::offA + 3 * l + 2, apsum[l][2] / prm[C::offA + 3 * l + 2]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  auto total = [&](int i) {
    real v = red0[i];
#pragma unroll
    for (int r = 1; r < R; ++r) v += red0[r * PP + i];
    return v;
  };
  if constexpr (C::ACTP == 0) {
    for (int i = tid; i < C::P; i += nt) out[i] = total(i);
  } else {
    // the tile loop ran on scaled weights (stage_weights: W' = f W): dW = f dW'
    auto segment = [&](int lo, int hi, real f) {
      for (int i = lo + tid; i < hi; i += nt) out[i] = total(i) * f;
    };
    segment(C::offW1, C::offb1 + C::hr(1), act_pre<C>(prm, 1));
    sfor<C::L - 1>([&](auto k_) {
      constexpr int l = decltype(k_)::value + 2;
      segment(C::offW(l), C::offb(l), act_pre<C>(prm, l) * act_post<C>(prm, l - 1));
      segment(C::offb(l), C::offb(l) + C::hr(l), act_pre<C>(prm, l));
    });
    segment(C::offWout, C::offbout, act_post<C>(prm, C::L));
    segment(C::offbout, C::P, 1.f);           // output bias, skip weights, (trainable) activation parameters
  }
}

template <class C>
__global__ __launch_bounds__(C::BWD_THREADS) void mlp_jet_bwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  constexpr int WAVES = C::BWD_THREADS / 64;
  const int ntiles = (a.n + 15) >> 4;
  // coordinates and (single-output networks) seeds of the first tile before the weights are staged, of every later tile one tile
  // ahead: loop-carried registers, so the loads cannot be sunk to their use (round 5; see mlp_jet_fwd_kernel)
  real xn[C::D], gn[C::NOUT == 1 ? C::NS : 1];
  {
    const int n0 = (blockIdx.x * WAVES + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn0];
    if constexpr (C::NOUT == 1) {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) gn[s] = a.gbar[(size_t)s * a.ldj + nn0];
    }
  }
  stage_weights<C, true>(lds, a.params);
  __syncthreads();
  real* stage = lds + C::ldsWeightsEnd(true) + wave * C::stageFloatsPerWave;
  GradAcc<C> acc;
  acc_init<C>(acc, lds, WAVES, wave, lane);
  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    const int n = tile * 16 + p;
    const bool valid = n < a.n;
    const int nn = valid ? n : a.n - 1;
    real x[C::D], gcur[C::NOUT == 1 ? C::NS : 1];
#pragma unroll
    for (int d = 0; d < C::D; ++d) x[d] = xn[d];
    if constexpr (C::NOUT == 1) {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) gcur[s] = gn[s];
    }
    {
      const int n1 = n + gridDim.x * WAVES * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn1];
      if constexpr (C::NOUT == 1) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) gn[s] = a.gbar[(size_t)s * a.ldj + nn1];
      }
    }
    const real* ldsw = lds + opaque_zero<C>();
    LayerState<C> st[C::L];
    real4 h[C::NS][C::NB];
    KeptPlanes<C> kp;
    tile_forward<C, true>(ldsw, lane, q, x, st, h, kp, stage);
    if constexpr (C::NOUT == 1) {
      real gout[C::NS];
#pragma unroll
      for (int s = 0; s < C::NS; ++s) gout[s] = valid ? gcur[s] : 0.f;
      tile_backward<C>(ldsw, stage, lane, p, q, x, gout, st, acc, kp);
    } else {
      real4 go[C::NS][C::NBO];
#pragma unroll
      for (int s = 0; s < C::NS; ++s)
#pragma unroll
        for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * ob + 4 * q + r;
            go[s][ob][r] = (valid && u < C::NOUT) ? a.gbar[((size_t)s * C::NOUT + u) * a.ldj + nn] : 0.f;
          }
      tile_backward_multi<C>(ldsw, stage, lane, p, q, x, go, st, acc, kp);
    }
  }
  block_reduce_store<C, WAVES>(lds, acc, wave, lane, p, q, a.partials + (size_t)blockIdx.x * C::P, a.params);
}

// ------------------------------------------------------------------------------------------------ fused train / eval kernel
// One kernel for the whole closure of a single-network system (solvers.py:369-395): forward streams -> generated
// pointwise stage PW (conditions + residuals + loss seeds, neurodiffeq_amd/codegen.py) -> reverse pass, with the
// layer states kept in registers in between: no forward recompute, no stream round trip through HBM, one launch.
// PW::apply(x, jets, seed, r, f, gj): per-point function; PW::loss(r): per-point loss term; PW::NEQ / PW::NF = number of
// residuals / function values.
struct FusedArgs {
  const real* coords;     // [D][ldc]
  const real* params;     // [P]
  real* partials;         // TRAIN: [gridDim.x][P]
  real* loss_partials;    // [gridDim.x] block sums of sum_e r_e^2
  real* funcs;            // optional [NF][ldj]
  real* resid;            // optional [NEQ][ldj]
  int n, ldc, ldj;
  real seed;              // adjoint seed scale 1 / (N_global * n_eq)
  const real* theta;      // PW::NT trainable scalars of the equations (inverse problems), else unused
  real* theta_partials;   // TRAIN: [gridDim.x][PW::NT] block sums of their per-point adjoints
};

// sum over the workgroup of NT per-lane values (non-zero in the lanes that carried a point), fixed order: lanes -> wave
// -> the waves in order; threads 0 .. NT-1 write the block's row.  `scratch`: (waves x NT) floats of LDS nobody else
// uses any more.
template <int NT>
__device__ __forceinline__ void theta_block_sum(real (&t)[NT > 0 ? NT : 1], real* scratch, int waves, real* out) {
  if constexpr (NT > 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NT; ++j) t[j] = point_sum(quad_sum(t[j]));
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) scratch[wave * NT + j] = t[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < NT && out) {
      real v = 0.f;
      for (int w = 0; w < waves; ++w) v += scratch[w * NT + threadIdx.x];
      out[threadIdx.x] = v;
    }
  }
}

// The body works on workgroup `blk` of `nblk` (the plain kernels pass blockIdx.x / gridDim.x; the train + validation
// launch below gives each half of its grid its own numbering, so that either half computes -- bit for bit -- what a
// launch of its own would).
// pull (fit(), small grids; csrc/ndq_tail.h): the launch first finishes the previous epoch -- sums, Adam, history -- and
// stages its weights from the parameters it has just computed (LDS vector behind the weight image); `writer`: this
// workgroup writes the global state.  fp32 builds without trainable activation parameters only (Cfg::ACTP == 0).
template <class C, class = void> struct no_pull_marker : std::false_type {};          // a Cfg may opt out: static constexpr bool NO_PULL
template <class C> struct no_pull_marker<C, std::void_t<decltype(C::NO_PULL)>> : std::true_type {};
template <class C> constexpr bool pull_supported() { return NDQ_F64 == 0 && C::ACTP == 0 && !no_pull_marker<C>::value; }
template <class C> constexpr int pull_floats() { return ((C::P + 3) & ~3) + 32; }

template <class C, class PW, bool TRAIN>
__device__ __forceinline__ void fused_closure_body(const FusedArgs& a, real* lds, const int blk, const int nblk,
                                                   const PullArgs& pull, bool writer) {
  static_assert(C::NOUT == 1, "the single-launch closure kernel needs a single-output network");
  NDQ_TS(0);
#ifdef NDQ_PHASE_TS
  if (threadIdx.x == 0 && blockIdx.x == 0) ndq_tile_iter = 0;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  constexpr int WAVES = C::BWD_THREADS / 64;
  const int ntiles = (a.n + 15) >> 4;
  // coordinates of a tile are fetched one tile ahead: the first tile's before the weights are staged (both global
  // latencies overlap), the next tile's while the current one is computed
  real xn[C::D];
  {
    const int n0 = (blk * WAVES + wave) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  const real* prm = a.params;
#if !NDQ_F64
  if constexpr (pull_supported<C>()) {
    if (pull.enabled) {
      float* pnew = lds + C::ldsWeightsEnd(TRAIN);
      NDQ_PT(0);
#ifdef NDQ_PHASE_TS
      if (threadIdx.x == 0 && blockIdx.x == 0) {      // when the previous launch's workgroups 0 / 1 ended
        ndq_pull_ts[4] = ndq_phase_ts[4 + 3];
        ndq_pull_ts[5] = ndq_phase_ts[8 + 4 + 3];
      }
#endif
      pull_prologue<C::BWD_THREADS, C::P, 1>(pull, pnew, 0, pnew + ((C::P + 3) & ~3), writer);
      NDQ_PT(2);
      __syncthreads();
      prm = pnew;
    }
  }
#endif
  stage_weights<C, TRAIN>(lds, prm);
  __syncthreads();
  NDQ_PT(3);
  NDQ_TS(1);
  real* stage = lds + C::ldsWeightsEnd(true) + wave * C::stageFloatsPerWave;
  GradAcc<C> acc;
  if constexpr (TRAIN) acc_init<C>(acc, lds, WAVES, wave, lane);
  real lsum = 0.f;
  // per-point data columns (rows D .. D + ND of the coordinate block) and trainable scalars of the equations
  constexpr int NXE = PW::ND + PW::NT;
  real xe[NXE > 0 ? NXE : 1], tsum[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
  for (int j = 0; j < PW::NT; ++j) { xe[PW::ND + j] = a.theta[j]; tsum[j] = 0.f; }
#if NDQ_STAGGER > 0
  // two waves per SIMD run the same phases (VALU-heavy activation math, MFMA-heavy GEMMs) in lockstep and then compete for
  // the same pipe; delaying the second wave of every SIMD by a fraction of a tile lets one's MFMAs run under the
  // other's VALU work
  if (WAVES > 4 && wave >= WAVES / 2) __builtin_amdgcn_s_sleep(NDQ_STAGGER);
#endif
  for (int tile = blk * WAVES + wave; tile < ntiles; tile += nblk * WAVES) {
    const int n = tile * 16 + p;
    const bool valid = n < a.n;
    real x[C::D];
#pragma unroll
    for (int d = 0; d < C::D; ++d) x[d] = xn[d];
    {
      const int n1 = n + nblk * WAVES * 16;
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
    const real* ldsw = lds + opaque_zero<C>();
    LayerState<C> st[C::L];
    real4 h[C::NS][C::NB];
    KeptPlanes<C> kp;
    NDQ_TT(0);
    tile_forward<C, TRAIN>(ldsw, lane, q, x, st, h, kp, stage);
    NDQ_TT(1);
    real jets[C::NS], gout[C::NS], r[PW::NR], f[PW::NF > 0 ? PW::NF : 1], gth[PW::NT > 0 ? PW::NT : 1];
    tile_output<C, TRAIN>(ldsw, q, x, h, jets);
    NDQ_TT(2);
    if constexpr (PW::ND > 0) {
      const int nn = valid ? n : a.n - 1;
#pragma unroll
      for (int j = 0; j < PW::ND; ++j) xe[j] = a.coords[(size_t)(C::D + j) * a.ldc + nn];
    }
    PW::apply(x, xe, jets, a.seed, TRAIN ? 1 : 0, r, f, gout, gth);
    NDQ_TT(3);
    if (valid && q == 0) {
      lsum += PW::loss(r);
      if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < PW::NT; ++j) tsum[j] += gth[j];
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
    if constexpr (TRAIN) {
#pragma unroll
      for (int s = 0; s < C::NS; ++s) gout[s] = valid ? gout[s] : 0.f;
      tile_backward<C>(ldsw, stage, lane, p, q, x, gout, st, acc, kp);
    }
    NDQ_TT(12);
#ifdef NDQ_PHASE_TS
    if (threadIdx.x == 0 && blockIdx.x == 0) ndq_tile_iter = 1;
#endif
  }
#ifdef NDQ_PHASE_TS
  __syncthreads();
  NDQ_TS(2);
#endif
  if constexpr (TRAIN) block_reduce_store<C, WAVES>(lds, acc, wave, lane, p, q, a.partials + (size_t)blk * C::P, a.params);
  // loss: lanes (only q == 0 lanes are non-zero) -> wave -> workgroup, fixed order
  lsum = point_sum(quad_sum(lsum));
  __syncthreads();
  real* wl = lds + C::ldsWeightsEnd(TRAIN);
  if (lane == 0) wl[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    real v = 0.f;
    for (int w = 0; w < WAVES; ++w) v += wl[w];
    a.loss_partials[blk] = v;
  }
  if constexpr (TRAIN) theta_block_sum<PW::NT>(tsum, wl + 16, WAVES, a.theta_partials ? a.theta_partials + (size_t)blk * PW::NT : nullptr);
  NDQ_TS(3);
}

template <class C, class PW, bool TRAIN>
__global__ __launch_bounds__(C::BWD_THREADS) void fused_closure_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const PullArgs none{};
  fused_closure_body<C, PW, TRAIN>(a, lds, blockIdx.x, gridDim.x, none, false);
}

// Training batch AND validation batch in ONE launch (fit(): solvers.py:443-497 runs a validation epoch after every
// training epoch; the validation loss of the parameters a training epoch starts from is evaluated by the spare
// workgroups of that epoch's closure launch): workgroups [0, train_blocks) run the training closure on `t`, the rest the
// forward-only closure on `v`.  Either count may be zero.
template <class C, class PW>
__global__ __launch_bounds__(C::BWD_THREADS) void fused_closure_tv_kernel(FusedArgs t, FusedArgs v, int train_blocks, PullArgs pull) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const bool writer = blockIdx.x == 0;
  if ((int)blockIdx.x < train_blocks) fused_closure_body<C, PW, true>(t, lds, blockIdx.x, train_blocks, pull, writer);
  else fused_closure_body<C, PW, false>(v, lds, (int)blockIdx.x - train_blocks, (int)gridDim.x - train_blocks, pull, writer);
}

// Loop mode (csrc/ndq_tail.h): one workgroup runs launches [e0, e1) of fit()'s pull-mode sequence back to back -- state
// in LDS behind everything the closure bodies use (loop_state_offset), the bodies themselves unchanged: they read their
// parameters from and leave their gradient row / loss partial in LDS through generic pointers.
template <class C> constexpr size_t fused_lds_bytes(bool train);
template <class C> constexpr int loop_state_offset() {
  const size_t a = fused_lds_bytes<C>(true), b = fused_lds_bytes<C>(false);
  return (int)((((a > b ? a : b) + 15) & ~(size_t)15) / sizeof(real));
}
template <class C> constexpr int loop_state_floats(int nets) { return nets * 4 * ((C::P + 3) & ~3) + 64; }
template <class C> constexpr size_t fused_loop_lds_bytes() { return sizeof(real) * (loop_state_offset<C>() + loop_state_floats<C>(1)); }

// state of K networks in LDS <- global at entry, -> global at exit (all threads; the caller synchronises)
template <int K>
__device__ __forceinline__ void loop_state_load(const LoopArgs& L, float* state, int pp, int len, float* misc) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float* s = state + (size_t)k * 4 * pp;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      s[i] = L.net[k].p_in[i];
      s[pp + i] = L.net[k].m_in[i];
      s[2 * pp + i] = L.net[k].v_in[i];
      s[3 * pp + i] = L.e0 >= 1 ? L.net[k].part_in[i] : 0.f;
    }
  }
  if (threadIdx.x == 0) {
    misc[0] = L.e0 >= 1 ? L.lp_in[0] : 0.f;
    misc[1] = (L.e0 >= 2 && L.has_valid) ? L.vp_in[0] : 0.f;
    misc[2] = L.best_loss[0];
    misc[3] = L.best_loss[1];
  }
}
template <int K>
__device__ __forceinline__ void loop_state_store(const LoopArgs& L, const float* state, int pp, int len, const float* misc) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float* s = state + (size_t)k * 4 * pp;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      L.net[k].p_out[i] = s[i];
      L.net[k].m_out[i] = s[pp + i];
      L.net[k].v_out[i] = s[2 * pp + i];
      L.net[k].part_out[i] = s[3 * pp + i];
    }
  }
  if (threadIdx.x == 0) {
    L.lp_out[0] = misc[0];
    if (L.has_valid) L.vp_out[0] = misc[1];
    L.best_loss[0] = misc[2];
    L.best_loss[1] = misc[3];
  }
}

#if !NDQ_F64
template <class C, class PW>
__global__ __launch_bounds__(C::BWD_THREADS) void fused_closure_loop_kernel(FusedArgs t, FusedArgs v, LoopArgs L) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  if constexpr (!pull_supported<C>()) return;         // (never launched: ndq_fused_loop_ok)
  constexpr int PP = (C::P + 3) & ~3;
  float* state = lds + loop_state_offset<C>();
  float* misc = state + 4 * PP;                       // [0] training loss partial [1] validation loss partial [2, 3] best loss
  const PullArgs none{};
  loop_state_load<1>(L, state, PP, C::P, misc);
  __syncthreads();
  for (int e = L.e0; e < L.e1; ++e) {
    if (e >= 1) {                                     // finish training epoch e - 1 (in place: every thread owns its columns)
      PullArgs pa{};
      loop_pull_args<1>(L, e, state, PP, C::P, misc, pa);
      pull_prologue<C::BWD_THREADS, C::P, 1>(pa, state, 0, misc + 8, true);
      __syncthreads();
    }
    if (L.has_valid && e >= 1) {                      // validation loss of the parameters epoch e starts from
      FusedArgs a = v;
      a.params = state; a.loss_partials = misc + 1;
      fused_closure_body<C, PW, false>(a, lds, 0, 1, none, false);
      __syncthreads();
    }
    if (e < L.n_epochs) {
      FusedArgs a = t;
      a.coords = t.coords + (size_t)e * (size_t)L.coord_stride;
      a.params = state; a.partials = state + 3 * PP; a.loss_partials = misc;
      fused_closure_body<C, PW, true>(a, lds, 0, 1, none, false);
      __syncthreads();
    }
  }
  loop_state_store<1>(L, state, PP, C::P, misc);
}
#endif

// ------------------------------------------------------------------------------------------------ multi-network closure
// The same single launch for K networks of ONE shape and stream set (systems of ODEs / PDEs with one network per
// unknown, the reference's default: solvers.py:136-140): all K weight images sit in LDS, each tile runs the K forward
// passes, the pointwise stage on all K stream sets, then -- network by network -- a second forward pass that keeps
// the layer states and the reverse pass (keeping K sets of states live at once would not fit the register file; the
// systems this serves are small and launch-bound, the extra per-point GEMMs are noise).  H = 32 class nets only.
constexpr int kMaxFusedNets = 4;
struct FusedMultiArgs {
  const real* coords;                    // [D][ldc]
  const real* params[kMaxFusedNets];     // K x [P]
  real* partials[kMaxFusedNets];         // TRAIN: K x [gridDim.x][P]
  real* loss_partials;                   // [gridDim.x]
  real* funcs;                           // optional [NF][ldj]
  real* resid;                           // optional [NEQ][ldj]
  int n, ldc, ldj;
  real seed;
  const real* theta;                     // see FusedArgs
  real* theta_partials;
};

// Workgroup shape (round 3): the K networks of a tile run CONCURRENTLY on K waves -- wave (k, g) carries network k for
// tile slot g of the round -- instead of one wave running K forward passes, the pointwise stage and then K times
// (forward again + reverse).  The K waves of a tile exchange their output streams through a small LDS block (double
// buffered: one workgroup barrier per round), every one of them evaluates the pointwise stage for the tile's points
// (a handful of operations for systems of ODEs) and keeps the adjoint seeds of its own network, whose layer states are
// still in its registers: no second forward pass, and the serial chain of a round is one network deep instead of K.
// Waves per workgroup: K x G with G = 2 tile slots for K = 2, one for K = 3, 4 -- never more than one wave per SIMD,
// so every wave has the whole register file (Cfg must be the 256-thread build: KEEP_H, no laundering).
template <int K> constexpr int multi_group() { return K == 2 ? NDQ_MULTI_G2 : 1; }
template <int K> constexpr int multi_threads() { return 64 * K * multi_group<K>(); }
// reduction regions of one network's G waves (they overlay those waves' transpose staging tiles)
template <class C, int K> constexpr int multi_regions() {
  const int pp = (C::P + 3) & ~3, g = multi_group<K>();
  int r = (g * C::stageFloatsPerWave) / pp;
  return r < 1 ? 1 : (r > g ? g : r);
}
template <class C, int K> constexpr int multi_group_floats() {
  const int pp = (C::P + 3) & ~3, st = multi_group<K>() * C::stageFloatsPerWave, red = multi_regions<C, K>() * pp;
  return ((st > red ? st : red) + 3) & ~3;
}
template <class C, int K> constexpr int multi_xchg_floats() { return 2 * multi_group<K>() * K * C::NS * 16; }

// loop mode (fused_multi_closure_loop_kernel): buffers in LDS that replace coords / params[k] / partials[k] /
// loss_partials of the argument struct; params_base == nullptr: unused
struct MultiLoopView {
  const real* coords; const real* params_base; real* partials_base; real* loss_partials; int stride;
};

template <class C, int K, class PW, bool TRAIN>
__device__ __forceinline__ void fused_multi_closure_body(const FusedMultiArgs& a, real* lds, const int blk, const int nblk,
                                                         const PullArgs& pull, bool writer, const MultiLoopView& lv) {
  static_assert(C::NOUT == 1 && !C::WIDE && K >= 2 && K <= kMaxFusedNets, "multi-network closure: n_out = 1, H <= 48, 2..4 nets");
  static_assert(C::BWD_THREADS == 256, "multi-network closure: one wave per SIMD (build without NDQ_BWD_THREADS)");
  constexpr int WS = C::ldsWeightsEnd(TRAIN);          // LDS floats per weight image
  constexpr int G = multi_group<K>(), WAVES = K * G;
  NDQ_TS(0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int k = wave / G, g = wave - k * G;            // this wave's network and its tile slot in a round
  // loop mode hands in its LDS-resident buffers through `lv` (the argument struct itself stays in the kernarg segment)
  const real* const coords = lv.params_base ? lv.coords : a.coords;
  const real* const prm_k = lv.params_base ? lv.params_base + k * lv.stride : a.params[k];
  real* const part_k = lv.params_base ? lv.partials_base + k * lv.stride : a.partials[k];
  real* const lpart = lv.params_base ? lv.loss_partials : a.loss_partials;
  const int ntiles = (a.n + 15) >> 4;
  // the first tile's coordinates are fetched before the weights are staged (both global latencies overlap)
  real xn[C::D];
  {
    const int n0 = (blk * G + g) * 16 + p;
    const int nn0 = n0 < a.n ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < C::D; ++d) xn[d] = coords[(size_t)d * a.ldc + nn0];
  }
  // every network's G waves stage that network's image: the K images are built at the same time (a staging pass is
  // bound by the latency of its parameter loads -- 2.97 us for two images one after the other at C1, measured)
  const real* prm = prm_k;
#if !NDQ_F64
  if constexpr (pull_supported<C>()) {
    if (pull.enabled) {
      // finish the previous epoch for all K networks (every thread of the workgroup on every network), results in LDS
      float* pn0 = lds + K * WS;
      pull_prologue<multi_threads<K>(), C::P, K>(pull, pn0, pull_floats<C>(), pn0 + K * pull_floats<C>() - 32, writer);
      __syncthreads();
      prm = pn0 + k * pull_floats<C>();
    }
  }
#endif
  stage_weights<C, TRAIN>(lds + k * WS, prm, g * 64 + lane, G * 64);
  __syncthreads();
  NDQ_TS(1);
  const real* ldsw = lds + k * WS;
  real* work = lds + K * WS;                           // per network: G staging tiles (later: its reduction regions)
  real* stage = work + k * multi_group_floats<C, K>() + g * C::stageFloatsPerWave;
  real* xchg = work + (TRAIN ? K * multi_group_floats<C, K>() : 0);
  GradAcc<C> acc;
  if constexpr (TRAIN) { acc_zero<C>(acc); acc.bias = nullptr; }
  real lsum = 0.f;
  constexpr int NXE = PW::ND + PW::NT;
  real xe[NXE > 0 ? NXE : 1], tsum[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
  for (int j = 0; j < PW::NT; ++j) { xe[PW::ND + j] = a.theta[j]; tsum[j] = 0.f; }
  int par = 0;
  // every wave of the workgroup runs the same number of rounds (one barrier each); a slot past the last tile works on
  // a copy of the last point with zero seeds
  for (int tile0 = blk * G; tile0 < ntiles; tile0 += nblk * G, par ^= 1) {
    const int n = (tile0 + g) * 16 + p;
    const bool valid = n < a.n;
    real x[C::D];
#pragma unroll
    for (int d = 0; d < C::D; ++d) x[d] = xn[d];
    {
      const int n1 = n + nblk * G * 16;                  // next round's tile, one round ahead
      const int nn1 = n1 < a.n ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < C::D; ++d) xn[d] = coords[(size_t)d * a.ldc + nn1];
    }
    LayerState<C> st[C::L];
    real4 h[C::NS][C::NB];
    KeptPlanes<C> kp;
    tile_forward<C, TRAIN>(ldsw, lane, q, x, st, h, kp, stage);
    real* xr = xchg + (par * G + g) * (K * C::NS * 16);        // [K][NS][16 points] of this tile
    {
      real mine[C::NS];
      tile_output<C, TRAIN>(ldsw, q, x, h, mine);
      if (q == 0) {
#pragma unroll
        for (int s = 0; s < C::NS; ++s) xr[(k * C::NS + s) * 16 + p] = mine[s];
      }
    }
    __syncthreads();
    real jets[K][C::NS], gout[K][C::NS], r[PW::NR], f[PW::NF > 0 ? PW::NF : 1], gth[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
#pragma unroll
      for (int s = 0; s < C::NS; ++s) jets[kk][s] = xr[(kk * C::NS + s) * 16 + p];
    if constexpr (PW::ND > 0) {
      const int nn = valid ? n : a.n - 1;
#pragma unroll
      for (int j = 0; j < PW::ND; ++j) xe[j] = coords[(size_t)(C::D + j) * a.ldc + nn];
    }
    PW::apply(x, xe, jets, a.seed, TRAIN ? 1 : 0, r, f, gout, gth);
    if (valid && q == 0 && k == 0) {
      lsum += PW::loss(r);
      if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < PW::NT; ++j) tsum[j] += gth[j];
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
    if constexpr (TRAIN) {
      real go[C::NS];
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        real v = gout[0][s];
#pragma unroll
        for (int kk = 1; kk < K; ++kk) v = (k == kk) ? gout[kk][s] : v;
        go[s] = valid ? v : 0.f;
      }
      tile_backward<C>(ldsw, stage, lane, p, q, x, go, st, acc, kp);
    }
  }
#ifdef NDQ_PHASE_TS
  __syncthreads();
  NDQ_TS(2);
#endif
  if constexpr (TRAIN) {
    // the G waves of a network add up among themselves (all networks at once: same barrier sequence), regions in that
    // network's own staging area; block_reduce_store addresses its regions behind "the" weight image of its base
    real* base = work + k * multi_group_floats<C, K>() - C::ldsWeightsEnd(true);
    block_reduce_store<C, G, multi_regions<C, K>()>(base, acc, g, lane, p, q, part_k + (size_t)blk * C::P, prm_k,
                                                     g * 64 + lane, G * 64);
  }
  lsum = point_sum(quad_sum(lsum));                      // non-zero in the waves of network 0 only
  __syncthreads();
  real* wl = lds + K * WS;
  if (lane == 0) wl[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    real v = 0.f;
    for (int w = 0; w < WAVES; ++w) v += wl[w];
    lpart[blk] = v;
  }
  if constexpr (TRAIN) theta_block_sum<PW::NT>(tsum, wl + 16, WAVES, a.theta_partials ? a.theta_partials + (size_t)blk * PW::NT : nullptr);
  NDQ_TS(3);
}

template <class C, int K, class PW, bool TRAIN>
__global__ __launch_bounds__(multi_threads<K>()) void fused_multi_closure_kernel(FusedMultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const PullArgs none{};
  fused_multi_closure_body<C, K, PW, TRAIN>(a, lds, blockIdx.x, gridDim.x, none, false, MultiLoopView{});
}

// training + validation batch in one launch, as fused_closure_tv_kernel
template <class C, int K, class PW>
__global__ __launch_bounds__(multi_threads<K>()) void fused_multi_closure_tv_kernel(FusedMultiArgs t, FusedMultiArgs v, int train_blocks,
                                                                                    PullArgs pull) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const bool writer = blockIdx.x == 0;
  if ((int)blockIdx.x < train_blocks) fused_multi_closure_body<C, K, PW, true>(t, lds, blockIdx.x, train_blocks, pull, writer, MultiLoopView{});
  else fused_multi_closure_body<C, K, PW, false>(v, lds, (int)blockIdx.x - train_blocks, (int)gridDim.x - train_blocks, pull, writer, MultiLoopView{});
}

template <class C, int K> constexpr size_t fused_multi_lds_bytes(bool train);
template <class C, int K> constexpr int multi_loop_state_offset() {
  const size_t a = fused_multi_lds_bytes<C, K>(true), b = fused_multi_lds_bytes<C, K>(false);
  return (int)((((a > b ? a : b) + 15) & ~(size_t)15) / sizeof(real));
}
template <class C, int K> constexpr size_t fused_multi_loop_lds_bytes() {
  return sizeof(real) * (multi_loop_state_offset<C, K>() + loop_state_floats<C>(K));
}
#if !NDQ_F64
// loop mode for K same-shape networks (see fused_closure_loop_kernel)
template <class C, int K, class PW>
__global__ __launch_bounds__(multi_threads<K>()) void fused_multi_closure_loop_kernel(FusedMultiArgs t, FusedMultiArgs v, LoopArgs L) {
  static_assert(K <= kLoopMaxNets, "loop mode: up to two networks");
  extern __shared__ __attribute__((aligned(16))) real lds[];
  if constexpr (!pull_supported<C>()) return;
  constexpr int PP = (C::P + 3) & ~3;
  float* state = lds + multi_loop_state_offset<C, K>();
  float* misc = state + K * 4 * PP;
  const PullArgs none{};
  loop_state_load<K>(L, state, PP, C::P, misc);
  __syncthreads();
  for (int e = L.e0; e < L.e1; ++e) {
    if (e >= 1) {
      PullArgs pa{};
      loop_pull_args<K>(L, e, state, PP, C::P, misc, pa);
      pull_prologue<multi_threads<K>(), C::P, K>(pa, state, 4 * PP, misc + 8, true);
      __syncthreads();
    }
    if (L.has_valid && e >= 1) {
      fused_multi_closure_body<C, K, PW, false>(v, lds, 0, 1, none, false, MultiLoopView{v.coords, state, state + 3 * PP, misc + 1, 4 * PP});
      __syncthreads();
    }
    if (e < L.n_epochs) {
      fused_multi_closure_body<C, K, PW, true>(t, lds, 0, 1, none, false,
                                               MultiLoopView{t.coords + (size_t)e * (size_t)L.coord_stride, state, state + 3 * PP, misc, 4 * PP});
      __syncthreads();
    }
  }
  loop_state_store<K>(L, state, PP, C::P, misc);
}
#endif

// ------------------------------------------------------------------------------------------------ grouped closure
// Single-launch closure for systems whose pointwise stage is too heavy to run 4x redundantly on the (p, q) lanes of a
// 16-point tile, or whose network has several outputs / reads only some of the batch coordinates (C4: FCNN(1 -> 25)
// coefficient network of a spherical-harmonics expansion, pde_spherical.py:253-254, conditions.py:1063-1096 -- 75
// stream values per point feed ~1 500 per-point operations).  A wave works on GROUPS of 64 points = 4 tiles:
//   phase 1  per tile: forward streams -> the tile's outputs go to this wave's LDS exchange tile X[64 points][XS]
//   phase 2  ONE point per lane: PW::apply reads its row of X (all streams of all outputs), leaves the adjoint seeds
//            in the same row                                    -- no redundancy, all 64 lanes carry distinct points
//   phase 3  per tile: forward again, this time keeping the layer states, seeds from X, reverse pass
// The second forward pass costs one extra forward (+1/3 of the per-point GEMMs); keeping 4 tiles of layer states in
// registers instead would not leave room for the per-point program.  Nothing crosses HBM except coordinates in and
// the workgroup's gradient partials out.  Row stride XS is odd: the per-lane row accesses of phase 2 (address =
// lane * XS + j) hit 64 distinct banks.
//   PW::NC          number of batch coordinates (rows of a.coords)
//   PW::dep(d)      batch coordinate fed to network input d
//   PW::apply(c, srow, seed, want_adj, r, f, grow): per-point function on the LDS row (srow == grow)
// Rows hold [NS][GW] values, GW = 1 for single-output networks, else the output count padded to whole 16-unit blocks
// (HO): every lane then stores / loads its 4 units of a block unconditionally (the padding entries are exact zeros:
// zero weight rows and zero bias on the way in, never written by the per-point stage, zero weight rows on the way back).
// tiles per group: 4 (64 points, every lane of phase 2 carries a point) with one wave per SIMD; the 8-wave build halves the
// exchange tile to fit the LDS (2 tiles = 32 points: phase 2 runs on half the lanes, but two waves per SIMD issue VALU
// instructions at about twice the rate of one)
template <class C> constexpr int group_tiles() { return C::BWD_THREADS == 256 ? 4 : 2; }
template <class C> constexpr int group_w() { return C::NOUT == 1 ? 1 : C::HO; }
template <class C> constexpr int group_xs() {
  const int w = C::NS * group_w<C>();
  return (w & 1) ? w : w + 1;
}

#ifndef NDQ_GROUP_U1
#define NDQ_GROUP_U1 2      // tiles of a group whose phase-1 / phase-3 bodies the compiler may interleave (unroll factor;
                            // C4 on MI355X, 131 072 points: U1 = 1 / 2 / 4: 60.2 / 59.0 / 59.8 us, U3 = 2: 59.9 us)
#endif
#ifndef NDQ_GROUP_U3
#define NDQ_GROUP_U3 1
#endif
#define NDQ_PRAGMA(x) _Pragma(#x)
#define NDQ_UNROLL(n) NDQ_PRAGMA(unroll n)

template <class C, class PW, bool TRAIN>
__device__ __forceinline__ void fused_group_closure_body(const FusedArgs& a, real* lds, const int blk, const int nblk,
                                                         const PullArgs& pull, bool writer) {
  static_assert(!C::ACC_LDS, "grouped closure: H <= 48");
  NDQ_TS(0);
  const real* prm = a.params;
#if !NDQ_F64
  if constexpr (pull_supported<C>()) {
    if (pull.enabled) {
      float* pnew = lds + C::ldsWeightsEnd(TRAIN);
      pull_prologue<C::BWD_THREADS, C::P, 1>(pull, pnew, 0, pnew + ((C::P + 3) & ~3), writer);
      __syncthreads();
      prm = pnew;
    }
  }
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  constexpr int WAVES = C::BWD_THREADS / 64;
  constexpr int XS = group_xs<C>();
  constexpr int G = group_tiles<C>(), GP = 16 * G;      // tiles / points per group
  const int ngroups = (a.n + GP - 1) / GP;
  // the first group's coordinates are requested before the weights are staged, every later group's one group ahead (round 5:
  // the load sat at the top of the group loop, its HBM latency exposed once per group -- the tile kernels always prefetched)
  real cn[PW::NC];
  {
    const int n0 = (blk * WAVES + wave) * GP + lane;
    const int nn0 = (n0 < a.n && lane < GP) ? n0 : a.n - 1;
#pragma unroll
    for (int d = 0; d < PW::NC; ++d) cn[d] = a.coords[(size_t)d * a.ldc + nn0];
  }
  stage_weights<C, TRAIN>(lds, prm);
  __syncthreads();
  NDQ_TS(1);
  real* stage = lds + C::ldsWeightsEnd(TRAIN) + wave * C::stageFloatsPerWave;
  real* X = lds + C::ldsWeightsEnd(TRAIN) + WAVES * C::stageFloatsPerWave + wave * (GP * XS);
  GradAcc<C> acc;
  if constexpr (TRAIN) { acc_zero<C>(acc); acc.bias = nullptr; }
  real lsum = 0.f;
  real th[PW::NT > 0 ? PW::NT : 1], tsum[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
  for (int j = 0; j < PW::NT; ++j) { th[j] = a.theta[j]; tsum[j] = 0.f; }
  for (int grp = blk * WAVES + wave; grp < ngroups; grp += nblk * WAVES) {
    const int n = grp * GP + lane;                       // this lane's point in phase 2 (lanes >= GP idle there)
    const bool valid = n < a.n && lane < GP;
    real c[PW::NC];
#pragma unroll
    for (int d = 0; d < PW::NC; ++d) c[d] = cn[d];
    {
      const int n1 = (grp + nblk * WAVES) * GP + lane;
      const int nn1 = (n1 < a.n && lane < GP) ? n1 : a.n - 1;
#pragma unroll
      for (int d = 0; d < PW::NC; ++d) cn[d] = a.coords[(size_t)d * a.ldc + nn1];
    }
#if !NDQ_GROUP_PREFETCH      // (A/B: the load at the top of the group loop, as in rounds 2 - 4)
    {
      const int nn = valid ? n : a.n - 1;
#pragma unroll
      for (int d = 0; d < PW::NC; ++d) c[d] = a.coords[(size_t)d * a.ldc + nn];
      asm volatile("" : "+v"(c[0]));
    }
#endif
    // ---- phase 1: forward streams of the 4 tiles -> X
    NDQ_TT(0);
    NDQ_UNROLL(NDQ_GROUP_U1)
    for (int t = 0; t < G; ++t) {
      real x[C::D];
      sfor<C::D>([&](auto d_) {
        constexpr int d = decltype(d_)::value;
        x[d] = __shfl(c[PW::dep(d)], 16 * t + p);
      });
      LayerState<C> st;
      first_layer<C, TRAIN>(lds, q, x, st);
      real4 h[C::NS][C::NB];
#pragma unroll
      for (int l = 2; l <= C::L; ++l) {
        act_forward<C>(st, h);
        if constexpr (C::BF16) {
          Planes<C> P;
          split_all<C>(h, P);
          hidden_layer_planes<C, TRAIN>(lds, l, lane, q, P, st);
        } else {
          hidden_layer<C, TRAIN>(lds, l, lane, q, h, st);
        }
      }
      act_forward<C>(st, h);
      real* row = X + (16 * t + p) * XS;
      if constexpr (C::NOUT == 1) {
        real out[C::NS];
        tile_output<C, TRAIN>(lds, q, x, h, out);
        if (q == 0) {
#pragma unroll
          for (int s = 0; s < C::NS; ++s) row[s] = out[s];
        }
      } else {
        real4 o[C::NS][C::NBO];
        output_layer_mfma<C, TRAIN>(lds, lane, q, h, o);
        output_skip_multi<C, TRAIN>(lds, q, x, o);
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
          for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) row[s * C::HO + 16 * ob + 4 * q + r] = o[s][ob][r];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase 2: the per-point program, one point per lane
    NDQ_TT(1);
    {
      real r[PW::NR], f[PW::NF > 0 ? PW::NF : 1], gth[PW::NT > 0 ? PW::NT : 1];
#pragma unroll
      for (int j = 0; j < PW::NT; ++j) gth[j] = 0.f;
      real* row = X + (lane < GP ? lane : 0) * XS;
      if (lane < GP) PW::apply(c, th, row, valid ? a.seed : 0.f, TRAIN ? 1 : 0, r, f, row, gth);
      if (valid) {
        lsum += PW::loss(r);
        if constexpr (TRAIN) {
#pragma unroll
          for (int j = 0; j < PW::NT; ++j) tsum[j] += gth[j];
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
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- phase 3: forward with kept states + reverse pass, tile by tile (seed = 0 for padding points: their rows
      // hold zero adjoints)
      NDQ_TT(2);
      NDQ_UNROLL(NDQ_GROUP_U3)
      for (int t = 0; t < G; ++t) {
        if ((grp * G + t) * 16 >= a.n) break;            // whole tile is padding (uniform over the wave)
        real x[C::D];
        sfor<C::D>([&](auto d_) {
          constexpr int d = decltype(d_)::value;
          x[d] = __shfl(c[PW::dep(d)], 16 * t + p);
        });
        LayerState<C> st[C::L];
        real4 h[C::NS][C::NB];
        KeptPlanes<C> kp;
        tile_forward<C, true>(lds, lane, q, x, st, h, kp);
        const real* row = X + (16 * t + p) * XS;
        if constexpr (C::NOUT == 1) {
          real gout[C::NS];
#pragma unroll
          for (int s = 0; s < C::NS; ++s) gout[s] = row[s];
          tile_backward<C>(lds, stage, lane, p, q, x, gout, st, acc, kp);
        } else {
          real4 go[C::NS][C::NBO];
#pragma unroll
          for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int ob = 0; ob < C::NBO; ++ob)
#pragma unroll
              for (int r = 0; r < 4; ++r) go[s][ob][r] = row[s * C::HO + 16 * ob + 4 * q + r];
          tile_backward_multi<C>(lds, stage, lane, p, q, x, go, st, acc, kp);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      NDQ_TT(3);
    }
  }
  NDQ_TS(2);
  if constexpr (TRAIN) block_reduce_store<C, WAVES>(lds, acc, wave, lane, p, q, a.partials + (size_t)blk * C::P, a.params);
  lsum = point_sum(quad_sum(lsum));                      // all 64 lanes carry a point here
  __syncthreads();
  real* wl = lds + C::ldsWeightsEnd(TRAIN);
  if (lane == 0) wl[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    real v = 0.f;
    for (int w = 0; w < WAVES; ++w) v += wl[w];
    a.loss_partials[blk] = v;
  }
  if constexpr (TRAIN) theta_block_sum<PW::NT>(tsum, wl + 16, WAVES, a.theta_partials ? a.theta_partials + (size_t)blk * PW::NT : nullptr);
  NDQ_TS(3);
}

template <class C, class PW, bool TRAIN>
__global__ __launch_bounds__(C::BWD_THREADS) void fused_group_closure_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const PullArgs none{};
  fused_group_closure_body<C, PW, TRAIN>(a, lds, blockIdx.x, gridDim.x, none, false);
}

// training + validation batch in one launch, as fused_closure_tv_kernel
template <class C, class PW>
__global__ __launch_bounds__(C::BWD_THREADS) void fused_group_closure_tv_kernel(FusedArgs t, FusedArgs v, int train_blocks, PullArgs pull) {
  extern __shared__ __attribute__((aligned(16))) real lds[];
  const bool writer = blockIdx.x == 0;
  if ((int)blockIdx.x < train_blocks) fused_group_closure_body<C, PW, true>(t, lds, blockIdx.x, train_blocks, pull, writer);
  else fused_group_closure_body<C, PW, false>(v, lds, (int)blockIdx.x - train_blocks, (int)gridDim.x - train_blocks, pull, writer);
}

// ------------------------------------------------------------------------------------------------ host-side sizes
template <class C> constexpr size_t group_lds_bytes(bool train) {
  const int waves = C::BWD_THREADS / 64;
  const int pp = (C::P + 3) & ~3;
  int work = waves * (C::stageFloatsPerWave + 16 * group_tiles<C>() * group_xs<C>());
  const int red = train ? bwd_regions<C>(waves) * pp : 0;          // overlays the staging + exchange tiles at the end
  if (red > work) work = red;
  if (pull_floats<C>() > work) work = pull_floats<C>();            // pull prologue: the updated parameter vector
  return sizeof(real) * (C::ldsWeightsEnd(train) + work + 16);
}
template <class C> constexpr size_t fwd_lds_bytes() { return sizeof(real) * C::ldsWeightsEnd(false); }
template <class C> constexpr size_t bwd_lds_bytes(int wavesPerBlock);
template <class C> constexpr size_t fused_lds_bytes(bool train) {
  const size_t pulled = sizeof(real) * (C::ldsWeightsEnd(train) + pull_floats<C>() + 16);   // pull prologue's vector
  const size_t plain = train ? bwd_lds_bytes<C>(C::BWD_THREADS / 64) : sizeof(real) * (C::ldsWeightsEnd(false) + 16);
  return plain > pulled ? plain : pulled;
}
template <class C, int K> constexpr size_t fused_multi_lds_bytes(bool train) {
  const size_t work = train ? (size_t)K * multi_group_floats<C, K>() : 0;
  const size_t plain = (size_t)K * C::ldsWeightsEnd(train) + work + multi_xchg_floats<C, K>() + 16;
  const size_t pulled = (size_t)K * C::ldsWeightsEnd(train) + (size_t)K * pull_floats<C>() + 16;   // pull prologue's K vectors
  return sizeof(real) * (plain > pulled ? plain : pulled);
}
template <class C> constexpr size_t bwd_lds_bytes(int wavesPerBlock) {
  const int pp = (C::P + 3) & ~3;
  const int stage = wavesPerBlock * C::stageFloatsPerWave;
  const int red = bwd_regions<C>(wavesPerBlock) * pp;
  return sizeof(real) * (C::ldsWeightsEnd(true) + (stage > red ? stage : red) + wavesPerBlock * C::biasFloats);
}
}  // namespace ndq
