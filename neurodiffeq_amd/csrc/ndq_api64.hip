// libndq64.so: the stream kernels of ndq_mlp.h compiled for fp64 (NDQ_F64) behind the ndq64_* entry points of
// include/ndq.h -- descriptor dispatch, forward streams, parameter-gradient adjoint, fixed-order second-stage sum.
// Serves fp64 networks (the reference's default precision) through the torch custom ops of autograd_ops.py.
#define NDQ_F64 1
#include <vector>
#include "ndq_launch.h"

namespace ndq {

// X(D, FIRST, MASK2, NB, L, ACT, NOUT): value-only and full second-order stream sets of the default network shapes;
// everything else is compiled on first use as an extension module (codegen.ensure_mlp_kernels(desc, f64=True))
#define NDQ64_CFG_TABLE(X)        \
  X(1, 0, 0, 2, 2, ACT_TANH, 1)   \
  X(1, 1, 1, 2, 2, ACT_TANH, 1)   \
  X(2, 0, 0, 2, 2, ACT_TANH, 1)   \
  X(2, 1, 7, 2, 2, ACT_TANH, 1)   \
  X(1, 0, 0, 2, 2, ACT_SIN, 1)    \
  X(1, 1, 1, 2, 2, ACT_SIN, 1)

#define NDQ64_ENTRY(D, F, M, NB, L, A, O) make_kernels<Cfg<D, F, M, NB, L, A, O, 0>>(),
static const ndq64_mlp_kernels kTable64[] = {NDQ64_CFG_TABLE(NDQ64_ENTRY)};
static std::vector<const ndq64_mlp_kernels*> g_registered64;

static bool same_desc(const ndq_mlp_desc& a, const ndq_mlp_desc& b) {
  return a.d == b.d && a.first == b.first && a.mask2 == b.mask2 && a.hidden == b.hidden && a.layers == b.layers &&
         a.act == b.act && a.n_out == b.n_out && a.lap == b.lap && a.skip == b.skip && a.mask3 == b.mask3 &&
         a.actp == b.actp && a.widths == b.widths && a.mono == b.mono;
}

static const ndq64_mlp_kernels* find64(const ndq_mlp_desc* d) {
  if (!d || d->hidden < 1 || d->hidden > 64) return nullptr;
  for (const ndq64_mlp_kernels& e : kTable64)
    if (same_desc(e.desc, *d)) return &e;
  for (const ndq64_mlp_kernels* e : g_registered64)
    if (same_desc(e->desc, *d)) return e;
  return nullptr;
}

static int bwd_blocks64(const ndq64_mlp_kernels* e, int n) {
  const int tiles = (n + 15) / 16;
  int blocks = (tiles + e->bwd_waves - 1) / e->bwd_waves;
  if (blocks > NDQ_BWD_MAX_BLOCKS) blocks = NDQ_BWD_MAX_BLOCKS;
  return blocks < 1 ? 1 : blocks;
}

// out[i] = (acc ? out[i] : 0) + scale * sum_r partials[r*len + i]; rows in fixed order, 4 independent chains per thread
__global__ __launch_bounds__(256) void reduce_partials64_kernel(const double* __restrict__ part, int nparts, int len,
                                                                double* __restrict__ out, int accumulate, double scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  int r = 0;
  for (; r + 3 < nparts; r += 4) {
    s0 += part[(size_t)r * len + i];
    s1 += part[(size_t)(r + 1) * len + i];
    s2 += part[(size_t)(r + 2) * len + i];
    s3 += part[(size_t)(r + 3) * len + i];
  }
  for (; r < nparts; ++r) s0 += part[(size_t)r * len + i];
  const double s = ((s0 + s1) + (s2 + s3)) * scale;
  out[i] = accumulate ? out[i] + s : s;
}

// torch.optim.Adam (amsgrad=False, maximize=False), the fp32 kernel of ndq_api.hip in double
__global__ __launch_bounds__(256) void adam64_kernel(double* __restrict__ p, const double* __restrict__ g,
                                                     double* __restrict__ m, double* __restrict__ v, int len, double lr,
                                                     double b1, double b2, double eps, double wd, double bc1, double bc2s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  double gi = g[i];
  const double pi = p[i];
  if (wd != 0.0) gi = fma(wd, pi, gi);
  const double mi = fma(b1, m[i], (1.0 - b1) * gi);
  const double vi = fma(b2, v[i], (1.0 - b2) * gi * gi);
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - (lr / bc1) * (mi / (sqrt(vi) / bc2s + eps));
}

}  // namespace ndq

using namespace ndq;

extern "C" {

int ndq64_mlp_supported(const ndq_mlp_desc* desc) { return find64(desc) ? 1 : 0; }

int ndq64_mlp_register(const ndq64_mlp_kernels* k) {
  if (!k || !k->fwd || !k->bwd || k->desc.hidden < 1 || k->desc.hidden > 64 || k->n_streams < 1 || k->n_params < 1 || k->bwd_waves < 1 ||
      k->lds_bytes > 160 * 1024)
    return NDQ_EINVAL;
  if (!find64(&k->desc)) g_registered64.push_back(k);
  return 0;
}

int ndq64_mlp_num_streams(const ndq_mlp_desc* desc) {
  const ndq64_mlp_kernels* e = find64(desc);
  return e ? e->n_streams : NDQ_EUNSUPPORTED;
}

int ndq64_mlp_num_params(const ndq_mlp_desc* desc) {
  const ndq64_mlp_kernels* e = find64(desc);
  return e ? e->n_params : NDQ_EUNSUPPORTED;
}

int ndq64_mlp_bwd_blocks(const ndq_mlp_desc* desc, int n) {
  const ndq64_mlp_kernels* e = find64(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (n <= 0) return NDQ_EINVAL;
  return bwd_blocks64(e, n);
}

int ndq64_mlp_jet_fwd(const ndq_mlp_desc* desc, const double* coords, int ldc, int n, const double* params, double* jets,
                      int ldj, void* stream) {
  const ndq64_mlp_kernels* e = find64(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (!coords || !params || !jets || n <= 0 || ldc < n || ldj < n) return NDQ_EINVAL;
  return e->fwd(coords, ldc, n, params, jets, ldj, stream);
}

int ndq64_mlp_jet_bwd(const ndq_mlp_desc* desc, const double* coords, int ldc, int n, const double* params,
                      const double* gbar, int ldj, double* partials, void* stream) {
  const ndq64_mlp_kernels* e = find64(desc);
  if (!e) return NDQ_EUNSUPPORTED;
  if (!coords || !params || !gbar || !partials || n <= 0 || ldc < n || ldj < n) return NDQ_EINVAL;
  return e->bwd(coords, ldc, n, params, gbar, ldj, partials, bwd_blocks64(e, n), stream);
}

int ndq64_reduce_partials(const double* partials, int nparts, int len, double* out, int accumulate, double scale,
                          void* stream) {
  if (!partials || !out || nparts <= 0 || len <= 0) return NDQ_EINVAL;
  hipLaunchKernelGGL(reduce_partials64_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     partials, nparts, len, out, accumulate, scale);
  return (int)hipGetLastError();
}

int ndq64_adam_step(double* params, const double* grad, double* exp_avg, double* exp_avg_sq, int len, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int step, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || len <= 0 || step <= 0) return NDQ_EINVAL;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2s = sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adam64_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), params, grad,
                     exp_avg, exp_avg_sq, len, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
  return (int)hipGetLastError();
}

}  // extern "C"
