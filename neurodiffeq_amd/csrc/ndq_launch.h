// Host-side launchers of the templated MLP kernels (ndq_mlp.h) packaged as an ndq_mlp_kernels record (include/ndq.h).
// Used twice: by ndq_api.hip for the kernels compiled into libndq.so, and by the small extension modules
// neurodiffeq_amd/codegen.py builds at run time (hipcc, one Cfg each) for network shapes / stream sets outside that
// table, which then join the same dispatch through ndq_mlp_register().
#pragma once
#ifndef NDQ_WG_TR
#define NDQ_WG_TR 1     // adjoint kernels of H = 32 networks: weight gradients from the bf16x3 planes (ndq_mlp.h Cfg::WG_TR)
#endif
#include "ndq_mlp.h"
#include "ndq_wide.h"
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
                        (int)C::SS::M3, C::ACTP, (int)C::HRP, (int)C::MONO};
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
#if !NDQ_F64
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
#endif

}  // namespace ndq
