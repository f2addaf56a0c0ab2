// Host-side launchers of the templated MLP kernels (ndq_mlp.h) packaged as an ndq_mlp_kernels record (include/ndq.h).
// Used twice: by ndq_api.hip for the kernels compiled into libndq.so, and by the small extension modules
// neurodiffeq_amd/codegen.py builds at run time (hipcc, one Cfg each) for network shapes / stream sets outside that
// table, which then join the same dispatch through ndq_mlp_register().
#pragma once
#ifndef NDQ_WG_TR
#define NDQ_WG_TR 1     // adjoint kernels of H = 32 networks: weight gradients from the bf16x3 planes (ndq_mlp.h Cfg::WG_TR)
#endif
#include "ndq_mlp.h"
#include "ndq_wide.h"     // one hidden layer of 65 .. 512 units (fp32; fp64: the forward-stream / adjoint kernels, round 5)
#if !NDQ_F64
#include "ndq_deep.h"     // 2 .. 8 hidden layers of 65 .. 512 units (fp32 only)
#endif
#include "../../include/ndq.h"

namespace ndq {

#ifndef NDQ_FWD_MAX_BLOCKS
#define NDQ_FWD_MAX_BLOCKS 768   // persistent-style grid: up to 3 workgroups per CU, waves loop over tiles
#endif
#ifndef NDQ_BWD_MAX_BLOCKS
#define NDQ_BWD_MAX_BLOCKS 256   // one workgroup per CU; every wave amortises its epilogue over several tiles
#endif

template <class C>
int launch_fwd(const MlpArgs& a, hipStream_t s) {
  static bool attr = false;
  const size_t lds = fwd_lds_bytes<C>();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_jet_fwd_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  constexpr int waves = C::FWD_THREADS / 64;
  const int tiles = (a.n + 15) / 16;
  int blocks = (tiles + waves - 1) / waves;
  if (blocks > NDQ_FWD_MAX_BLOCKS) blocks = NDQ_FWD_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(mlp_jet_fwd_kernel<C>, dim3(blocks), dim3(C::FWD_THREADS), lds, s, a);
  return (int)hipGetLastError();
}

template <class C>
int launch_bwd(const MlpArgs& a, int blocks, hipStream_t s) {
  static bool attr = false;
  const size_t lds = bwd_lds_bytes<C>(C::BWD_THREADS / 64);
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_jet_bwd_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(mlp_jet_bwd_kernel<C>, dim3(blocks), dim3(C::BWD_THREADS), lds, s, a);
  return (int)hipGetLastError();
}

template <class C>
int kernels_fwd(const real* coords, int ldc, int n, const real* params, real* jets, int ldj, void* stream) {
  MlpArgs a{};
  a.coords = coords; a.params = params; a.jets = jets; a.n = n; a.ldc = ldc; a.ldj = ldj;
  return launch_fwd<C>(a, static_cast<hipStream_t>(stream));
}

template <class C>
int kernels_bwd(const real* coords, int ldc, int n, const real* params, const real* gbar, int ldj, real* partials,
                int blocks, void* stream) {
  MlpArgs a{};
  a.coords = coords; a.params = params; a.gbar = gbar; a.partials = partials; a.n = n; a.ldc = ldc; a.ldj = ldj;
  return launch_bwd<C>(a, blocks, static_cast<hipStream_t>(stream));
}

#if NDQ_F64
typedef ndq64_mlp_kernels kernels_record;      // include/ndq.h: the fp64 record (double buffers)
#else
typedef ndq_mlp_kernels kernels_record;
#endif

template <class C>
kernels_record make_kernels() {
  kernels_record k{};
  k.desc = ndq_mlp_desc{C::D, C::SS::FIRST, (int)C::SS::M2, C::HR, C::L, C::ACT, C::NOUT, C::SS::LAP, C::SKIP,
                        (int)C::SS::M3, C::ACTP, (int)C::HRP, (int)C::MONO, (int)C::SS::M4};
  k.n_streams = C::NS;
  k.n_params = C::P;
  k.bwd_waves = C::BWD_THREADS / 64;
  const size_t f = fwd_lds_bytes<C>(), b = bwd_lds_bytes<C>(C::BWD_THREADS / 64);
  k.lds_bytes = (int)(f > b ? f : b);
  k.fwd = &kernels_fwd<C>;
  k.bwd = &kernels_bwd<C>;
  return k;
}

// ---- one hidden layer of 65 .. 512 units (ndq_wide.h): the same record, so ndq_mlp_register / ndq_mlp_jet_fwd / _bwd
// serve these shapes like any other
template <class C>
int wide_kernels_fwd(const real* coords, int ldc, int n, const real* params, real* jets, int ldj, void* stream) {
  MlpArgs a{};
  a.coords = coords; a.params = params; a.jets = jets; a.n = n; a.ldc = ldc; a.ldj = ldj;
  static bool attr = false;
  const size_t lds = wide_lds_bytes<C>();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_jet_fwd_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int tiles = (n + 15) / 16;
  int blocks = (tiles + C::WAVES - 1) / C::WAVES;
  if (blocks > NDQ_BWD_MAX_BLOCKS) blocks = NDQ_BWD_MAX_BLOCKS;      // every wave loads its units once: one workgroup per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(wide_jet_fwd_kernel<C>, dim3(blocks), dim3(C::THREADS), lds, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

template <class C>
int wide_kernels_bwd(const real* coords, int ldc, int n, const real* params, const real* gbar, int ldj, real* partials,
                     int blocks, void* stream) {
  MlpArgs a{};
  a.coords = coords; a.params = params; a.gbar = gbar; a.partials = partials; a.n = n; a.ldc = ldc; a.ldj = ldj;
  static bool attr = false;
  const size_t lds = wide_lds_bytes<C>();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_jet_bwd_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(wide_jet_bwd_kernel<C>, dim3(blocks), dim3(C::THREADS), lds, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

template <class C>
kernels_record make_wide_kernels() {
  kernels_record k{};
  k.desc = ndq_mlp_desc{C::D, C::SS::FIRST, (int)C::SS::M2, C::W, 1, C::ACT, C::NOUT, C::SS::LAP, 0, (int)C::SS::M3, 0, 0, 0};
  k.n_streams = C::NS;
  k.n_params = C::P;
  k.bwd_waves = C::WAVES;
  k.lds_bytes = (int)wide_lds_bytes<C>();
  k.fwd = &wide_kernels_fwd<C>;
  k.bwd = &wide_kernels_bwd<C>;
  return k;
}

#if !NDQ_F64
// ---- two or more hidden layers of 65 .. 512 units (ndq_deep.h): layer by layer through a workspace in HBM.  The module
// owns that workspace (hipMalloc on first use, grown when a larger batch arrives, never inside a timed steady state); the
// adjoint entry recomputes the forward layers like every ndq_mlp_jet_bwd and writes ONE row of "partials" -- the gradient
// itself, its own reductions being fixed-order already (bwd_waves is set so that ndq_mlp_bwd_blocks() == 1).
#ifndef NDQ_DEEP_BF16X3
#define NDQ_DEEP_BF16X3 1          // per-point GEMMs of the deep wide networks on the bf16 matrix core with 3-way split operands
#endif                             // (0: the exact-f32 MFMA kernels -- A/B runs, and the fallback the first version shipped with)
struct DeepPlan {
  int np, blocks_max;
  size_t X, z0, zb0, wp, wt, pw, pw_layer, pb, pb_layer, pw1, pwo, pbo, wpl, wtl, planes_layer, total;
};
template <class K>
int deep_set_lds(K kernel, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <class C>
DeepPlan deep_plan(int n) {
  DeepPlan q{};
  q.np = (n + 15) & ~15;
  q.blocks_max = 256;
  const size_t HP = C::HP, nwmax = 4 * (size_t)q.blocks_max * C::WAVES;      // most partial rows any kernel writes
  q.X = (size_t)C::NS * q.np * HP;
  size_t o = 0;
  q.z0 = o; o += (size_t)(C::L - 1) * q.X;            // Z_2 .. Z_L
  q.zb0 = o; o += 2 * q.X;                            // Zbar ping-pong
  q.wp = o; o += (size_t)(C::L - 1) * HP * HP;
  q.wt = o; o += (size_t)(C::L - 1) * HP * HP;
  q.pw_layer = (2 * (size_t)q.blocks_max / (C::NT * C::NT) + 1) * HP * HP;       // per hidden matrix
  q.pw = o; o += (size_t)(C::L - 1) * q.pw_layer;
  q.pb_layer = nwmax * HP;                                                          // per hidden layer
  q.pb = o; o += (size_t)C::L * q.pb_layer;
  q.pw1 = o; o += nwmax * HP * C::D;
  q.pwo = o; o += nwmax * C::NOUT * HP;
  q.pbo = o; o += nwmax * C::NOUT;
  q.planes_layer = (size_t)C::NB * ((C::HP + 31) / 32) * 3 * 64 * 4;             // floats per matrix: bf16x8 elements of 16 B
  q.wpl = o; o += (size_t)(C::L - 1) * q.planes_layer;
  q.wtl = o; o += (size_t)(C::L - 1) * q.planes_layer;
  q.total = o + 64;
  return q;
}
// One workspace per DEVICE and module (it used to be one per module, allocated on whichever device was current at
// first use -- a second FusedSystem on another GPU of the same process would have been handed that device's memory).  Still
// one STREAM at a time per device: growing it synchronises the device and frees the old block, and the forward-reuse hint
// below is per device too; concurrent streams of one process must not share a deep network's module.
constexpr int kDeepMaxDevices = 16;
inline int deep_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kDeepMaxDevices) d = 0; return d; }
inline real* deep_workspace(size_t floats) {
  static real* base[kDeepMaxDevices] = {};
  static size_t have[kDeepMaxDevices] = {};
  const int d = deep_device();
  if (floats > have[d]) {
    if (base[d]) { (void)hipDeviceSynchronize(); (void)hipFree(base[d]); base[d] = nullptr; have[d] = 0; }
    const size_t want = floats + floats / 8;
    if (hipMalloc(reinterpret_cast<void**>(&base[d]), want * sizeof(real)) != hipSuccess) { base[d] = nullptr; return nullptr; }
    have[d] = want;
  }
  return base[d];
}
// The adjoint entry recomputes the forward layers unless the caller vouches that the workspace still holds them: the engine
// raises this flag for the adjoint call that directly follows the forward call of the same training step (same parameter
// vector, same batch); a different coordinate / parameter pointer or point count recomputes regardless.
struct DeepLast { const void* coords; const void* params; int n, ldc; bool reuse; };
inline DeepLast& deep_last() { static DeepLast d[kDeepMaxDevices] = {}; return d[deep_device()]; }

inline int deep_blocks(long waves, int min_waves, int cap) {
  long w = waves > min_waves ? waves : min_waves;
  long b = (w + 3) / 4;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// forward layers 2 .. L into the workspace (shared by the two entry points)
// jets != nullptr (forward entry point): when one output chunk holds all units (NCH == 1: W <= 128) the LAST layer's GEMM
// also computes the output streams (deep_gemm_bf EPI 3) and the caller skips deep_head_fwd; returns 1 in *head_done then.
#ifndef NDQ_DEEP_HEAD_FWD_FUSED
#define NDQ_DEEP_HEAD_FWD_FUSED 1
#endif
template <class C>
int deep_forward_layers(const DeepPlan& q, real* ws, const real* coords, int ldc, int n, const real* params, hipStream_t st,
                        real* jets = nullptr, int ldj = 0, int* head_done = nullptr) {
  const int ntiles = q.np / 16;
  if (head_done) *head_done = 0;
#if NDQ_DEEP_BF16X3
  hipLaunchKernelGGL(deep_prep_planes<C>, dim3(64, C::L - 1), dim3(256), 0, st, params, reinterpret_cast<bf16x8*>(ws + q.wpl),
                     reinterpret_cast<bf16x8*>(ws + q.wtl));
  constexpr int NCHB0 = (C::NB + deep_bf_jb<C, 0>() - 1) / deep_bf_jb<C, 0>();
  static bool attr = false;
  if (!attr) {
    int e = deep_set_lds(&deep_gemm_bf<C, 0, 0>, deep_bf_lds_bytes<C, 0>());
    if (!e) e = deep_set_lds(&deep_gemm_bf<C, 1, 0>, deep_bf_lds_bytes<C, 0>());
    if constexpr (NCHB0 == 1 && NDQ_DEEP_HEAD_FWD_FUSED) {
      if (!e) e = deep_set_lds(&deep_gemm_bf<C, 0, 3>, deep_bf_lds_bytes<C, 0>());
      if (!e) e = deep_set_lds(&deep_gemm_bf<C, 1, 3>, deep_bf_lds_bytes<C, 0>());
    }
    if (e) return e;
    attr = true;
  }
  int stripes = (ntiles + kDeepBfWaves - 1) / kDeepBfWaves;                              // one workgroup per CU: its weight planes fill the LDS
  if (stripes > kDeepBfOcc * q.blocks_max / NCHB0) stripes = kDeepBfOcc * q.blocks_max / NCHB0;
  if (stripes < 1) stripes = 1;
#else
  hipLaunchKernelGGL(deep_prep<C>, dim3(64, C::L - 1), dim3(256), 0, st, params, ws + q.wp, ws + q.wt);
  const int blocks = deep_blocks((long)ntiles * C::NCH, C::NCH, 2 * q.blocks_max);      // two workgroups per CU
#endif
  for (int l = 2; l <= C::L; ++l) {
    DeepArgs a{};
    a.coords = coords; a.prm = params; a.n = n; a.np = q.np; a.ldc = ldc;
    a.wmat = ws + q.wp + (size_t)(l - 2) * C::HP * C::HP;
    a.wpl = ws + q.wpl + (size_t)(l - 2) * q.planes_layer;
    a.bias = params + C::offb(l);
    a.ow = C::wl(l);
    a.zin = l > 2 ? ws + q.z0 + (size_t)(l - 3) * q.X : nullptr;
    a.zout = ws + q.z0 + (size_t)(l - 2) * q.X;
#if NDQ_DEEP_BF16X3
    bool folded = false;
    if constexpr (NCHB0 == 1 && NDQ_DEEP_HEAD_FWD_FUSED) {
      if (jets && l == C::L) {
        a.jets = jets; a.ldj = ldj;
        if (l == 2) hipLaunchKernelGGL((deep_gemm_bf<C, 0, 3>), dim3(stripes), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 0>()), st, a);
        else hipLaunchKernelGGL((deep_gemm_bf<C, 1, 3>), dim3(stripes), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 0>()), st, a);
        folded = true;
        if (head_done) *head_done = 1;
      }
    }
    if (folded) continue;
    if (l == 2) hipLaunchKernelGGL((deep_gemm_bf<C, 0, 0>), dim3(stripes * NCHB0), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 0>()), st, a);
    else hipLaunchKernelGGL((deep_gemm_bf<C, 1, 0>), dim3(stripes * NCHB0), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 0>()), st, a);
#else
    if (l == 2) hipLaunchKernelGGL((deep_fwd_gemm<C, true>), dim3(blocks), dim3(C::THREADS), 0, st, a);
    else hipLaunchKernelGGL((deep_fwd_gemm<C, false>), dim3(blocks), dim3(C::THREADS), 0, st, a);
#endif
  }
  return (int)hipGetLastError();
}

template <class C>
int deep_kernels_fwd(const real* coords, int ldc, int n, const real* params, real* jets, int ldj, void* stream) {
  const DeepPlan q = deep_plan<C>(n);
  real* ws = deep_workspace(q.total);
  if (!ws) return (int)hipErrorOutOfMemory;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int head_done = 0;
  int rc = deep_forward_layers<C>(q, ws, coords, ldc, n, params, st, jets, ldj, &head_done);
  if (rc) return rc;
  DeepLast& last = deep_last();
  last.coords = coords; last.params = params; last.n = n; last.ldc = ldc;
  if (head_done) return (int)hipGetLastError();          // (the last layer's GEMM wrote the output streams itself)
  DeepHeadArgs h{};
  h.prm = params; h.z = ws + q.z0 + (size_t)(C::L - 2) * q.X; h.jets = jets; h.n = n; h.np = q.np; h.ldj = ldj;
  hipLaunchKernelGGL(deep_head_fwd<C>, dim3(deep_blocks(q.np / 16, 1, 4 * q.blocks_max)), dim3(C::THREADS), 0, st, h);
  return (int)hipGetLastError();
}

template <class C>
int deep_kernels_bwd(const real* coords, int ldc, int n, const real* params, const real* gbar, int ldj, real* grad,
                     int /*blocks == 1*/, void* stream) {
  const DeepPlan q = deep_plan<C>(n);
  real* ws = deep_workspace(q.total);
  if (!ws) return (int)hipErrorOutOfMemory;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DeepLast& last = deep_last();
  const bool have = last.reuse && last.coords == coords && last.params == params && last.n == n && last.ldc == ldc;
  last.coords = nullptr;                       // whatever happens next, the workspace is about to be overwritten
  if (!have) {
    int rc = deep_forward_layers<C>(q, ws, coords, ldc, n, params, st);
    if (rc) return rc;
  }
  // every kernel of the sweep leaves partial rows / tiles in its own region; ONE launch at the end adds them all up
  DeepReduceJobs J{};
  auto reduce = [&](const real* src, int nparts, int rows_p, int cols_p, int rows, int cols, real* dst) {
    DeepReduceJob& j = J.job[J.njobs++];
    j.src = src; j.dst = dst; j.nparts = nparts; j.rows_p = rows_p; j.cols_p = cols_p; j.rows = rows; j.cols = cols;
    j.block0 = J.nblocks;
    J.nblocks += (rows * cols + 63) / 64;
  };
  constexpr int UG = (C::HP + 63) / 64;
  if (!NDQ_DEEP_HEAD_FUSED || C::HP > 128) {
    DeepHeadArgs h{};
    h.prm = params; h.z = ws + q.z0 + (size_t)(C::L - 2) * q.X; h.gbar = gbar; h.zbar = ws + q.zb0;
    h.pwo = ws + q.pwo; h.pb = ws + q.pb + (size_t)(C::L - 1) * q.pb_layer; h.pbo = ws + q.pbo; h.n = n; h.np = q.np; h.ldj = ldj;
    const int blocks = deep_blocks((long)UG * q.np, UG, 4 * q.blocks_max);
    const int stripes = blocks * C::WAVES / UG;
    hipLaunchKernelGGL(deep_head_bwd<C>, dim3(blocks), dim3(C::THREADS), 0, st, h);
    reduce(h.pwo, stripes, C::NOUT, C::HP, C::NOUT, C::wl(C::L), grad + C::offWout);
    reduce(h.pb, stripes, 1, C::HP, 1, C::wl(C::L), grad + C::offb(C::L));
    reduce(h.pbo, stripes, 1, C::NOUT, 1, C::NOUT, grad + C::offbout);
  }
  int cur = 0;
  const int ntiles = q.np / 16;
  for (int l = C::L; l >= 2; --l) {
    DeepArgs a{};
    a.coords = coords; a.prm = params; a.n = n; a.np = q.np; a.ldc = ldc;
    a.zin = ws + q.zb0 + (size_t)cur * q.X;                               // Zbar_l
    a.zprev = l > 2 ? ws + q.z0 + (size_t)(l - 3) * q.X : nullptr;        // Z_{l-1}
#if NDQ_DEEP_HEAD_FUSED
    // layer L with the head folded in: no deep_head_bwd pass, no Zbar_L in HBM -- both consumers read Z_L and the seeds.
    // Only where every consumer forms a unit's Zbar_L ONCE (HP <= 128: one output chunk in the reverse GEMM, 2 x 2 tiles in the
    // weight-gradient GEMM); at W = 256 the same arithmetic would be repeated by 4 chunks and 4 tile columns -- measured
    // (profiles/r05b_deep_ab.md): 128 x 3 602 -> 585 us per step, 256 x 2 1112 -> 1171 us, so the wider shapes keep the pass
    const bool head = l == C::L && C::HP <= 128;
    if (head) {
      a.zin = ws + q.z0 + (size_t)(C::L - 2) * q.X;                       // Z_L
      a.gbar = gbar; a.ldj = ldj;
      a.pwo = ws + q.pwo; a.pbh = ws + q.pb + (size_t)(C::L - 1) * q.pb_layer; a.pbo = ws + q.pbo;
    }
#else
    constexpr bool head = false;
#endif
    {   // dW_l
      a.pw = ws + q.pw + (size_t)(l - 2) * q.pw_layer;
      // one workgroup per (tile, point slice)
      int KS = (q.np / 4 + C::WAVES - 1) / C::WAVES;
#if NDQ_DEEP_WGRAD_BF
      constexpr int kWgOcc = deep_wgbf_occ(C::NS);         // bf16x3 products (deep_wgrad_bf)
#define NDQ_WGRAD_KERNEL deep_wgrad_bf
#else
      constexpr int kWgOcc = 2;                            // exact-f32 products (deep_wgrad_gemm): two workgroups per CU
#define NDQ_WGRAD_KERNEL deep_wgrad_gemm
#endif
      if (KS > kWgOcc * q.blocks_max / (C::NT * C::NT)) KS = kWgOcc * q.blocks_max / (C::NT * C::NT);
      if (KS < 1) KS = 1;
      const int blocks = KS * C::NT * C::NT;
      if (head) {
        if (l == 2) hipLaunchKernelGGL((NDQ_WGRAD_KERNEL<C, true, true>), dim3(blocks), dim3(C::THREADS), 0, st, a);
        else hipLaunchKernelGGL((NDQ_WGRAD_KERNEL<C, false, true>), dim3(blocks), dim3(C::THREADS), 0, st, a);
        reduce(a.pwo, KS * C::WAVES, C::NOUT, C::HP, C::NOUT, C::wl(C::L), grad + C::offWout);
        reduce(a.pbh, KS * C::WAVES, 1, C::HP, 1, C::wl(C::L), grad + C::offb(C::L));
        reduce(a.pbo, KS * C::WAVES, 1, C::NOUT, 1, C::NOUT, grad + C::offbout);
      } else if (l == 2) hipLaunchKernelGGL((NDQ_WGRAD_KERNEL<C, true>), dim3(blocks), dim3(C::THREADS), 0, st, a);
      else hipLaunchKernelGGL((NDQ_WGRAD_KERNEL<C, false>), dim3(blocks), dim3(C::THREADS), 0, st, a);
#undef NDQ_WGRAD_KERNEL
      reduce(a.pw, KS, C::HP, C::HP, C::wl(l), C::wl(l - 1), grad + C::offW(l));
    }
    a.wmat = ws + q.wt + (size_t)(l - 2) * C::HP * C::HP;
    a.wpl = ws + q.wtl + (size_t)(l - 2) * q.planes_layer;               // planes of W_l^T
    a.pb = ws + q.pb + (size_t)(l - 2) * q.pb_layer;                      // db_{l-1}
#if NDQ_DEEP_BF16X3
    {
      static bool attr = false;
      if (!attr) {
        int e = deep_set_lds(&deep_gemm_bf<C, 2, 1>, deep_bf_lds_bytes<C, 1>());
        if (!e) e = deep_set_lds(&deep_gemm_bf<C, 2, 2>, deep_bf_lds_bytes<C, 2>());
#if NDQ_DEEP_HEAD_FUSED
        if (!e) e = deep_set_lds(&deep_gemm_bf<C, 3, 1>, deep_bf_lds_bytes<C, 1>());
        if (!e) e = deep_set_lds(&deep_gemm_bf<C, 3, 2>, deep_bf_lds_bytes<C, 2>());
#endif
        if (e) return e;
        attr = true;
      }
    }
    auto bf_stripes = [&](int nch) {
      int s = (ntiles + kDeepBfWaves - 1) / kDeepBfWaves;
      if (s > kDeepBfOcc * q.blocks_max / nch) s = kDeepBfOcc * q.blocks_max / nch;
      return s < 1 ? 1 : s;
    };
    if (l > 2) {
      constexpr int NCHB1 = (C::NB + deep_bf_jb<C, 1>() - 1) / deep_bf_jb<C, 1>();
      a.zout = ws + q.zb0 + (size_t)(cur ^ 1) * q.X;
      const int stripes = bf_stripes(NCHB1);
      if (head) hipLaunchKernelGGL((deep_gemm_bf<C, 3, 1>), dim3(stripes * NCHB1), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 1>()), st, a);
      else hipLaunchKernelGGL((deep_gemm_bf<C, 2, 1>), dim3(stripes * NCHB1), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 1>()), st, a);
      reduce(a.pb, stripes * kDeepBfWaves, 1, C::HP, 1, C::wl(l - 1), grad + C::offb(l - 1));
      cur ^= 1;
    } else {
      constexpr int NCHB2 = (C::NB + deep_bf_jb<C, 2>() - 1) / deep_bf_jb<C, 2>();
      a.pw1 = ws + q.pw1;
      const int stripes = bf_stripes(NCHB2);
      if (head) hipLaunchKernelGGL((deep_gemm_bf<C, 3, 2>), dim3(stripes * NCHB2), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 2>()), st, a);
      else hipLaunchKernelGGL((deep_gemm_bf<C, 2, 2>), dim3(stripes * NCHB2), dim3(kDeepBfThreads), (deep_bf_lds_bytes<C, 2>()), st, a);
      reduce(a.pb, stripes * kDeepBfWaves, 1, C::HP, 1, C::wl(1), grad + C::offb1);
      reduce(a.pw1, stripes * kDeepBfWaves, C::HP, C::D, C::wl(1), C::D, grad + C::offW1);
    }
#else
    if (l > 2) {
      a.zout = ws + q.zb0 + (size_t)(cur ^ 1) * q.X;
      const int blocks = deep_blocks((long)ntiles * C::NCHB, C::NCHB, 2 * q.blocks_max);
      hipLaunchKernelGGL((deep_bwd_gemm<C, false>), dim3(blocks), dim3(C::THREADS), 0, st, a);
      reduce(a.pb, blocks * C::WAVES / C::NCHB, 1, C::HP, 1, C::wl(l - 1), grad + C::offb(l - 1));
      cur ^= 1;
    } else {
      a.pw1 = ws + q.pw1;
      const int blocks = deep_blocks((long)ntiles * C::NCHF, C::NCHF, 2 * q.blocks_max);
      const int stripes = blocks * C::WAVES / C::NCHF;
      hipLaunchKernelGGL((deep_bwd_gemm<C, true>), dim3(blocks), dim3(C::THREADS), 0, st, a);
      reduce(a.pb, stripes, 1, C::HP, 1, C::wl(1), grad + C::offb1);
      reduce(a.pw1, stripes, C::HP, C::D, C::wl(1), C::D, grad + C::offW1);
    }
#endif
  }
  hipLaunchKernelGGL(deep_reduce_all, dim3(J.nblocks), dim3(64, 16), 0, st, J);
  return (int)hipGetLastError();
}

template <class C>
kernels_record make_deep_kernels() {
  kernels_record k{};
  k.desc = ndq_mlp_desc{C::D, C::SS::FIRST, (int)C::SS::M2, C::W, C::L, C::ACT, C::NOUT, C::SS::LAP, 0, (int)C::SS::M3, 0, (int)C::WP, 0};
  k.n_streams = C::NS;
  k.n_params = C::P;
  k.bwd_waves = 1 << 24;              // ndq_mlp_bwd_blocks() == 1: the adjoint entry writes the gradient row itself
  k.lds_bytes = 0;
  k.fwd = &deep_kernels_fwd<C>;
  k.bwd = &deep_kernels_bwd<C>;
  return k;
}
#endif

}  // namespace ndq
