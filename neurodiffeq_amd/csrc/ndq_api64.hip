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
         a.act == b.act && a.n_out == b.n_out && a.lap == b.lap && a.skip == b.skip && a.mask3 == b.mask3 && a.mask4 == b.mask4 &&
         a.actp == b.actp && a.widths == b.widths && a.mono == b.mono;
}

static const ndq64_mlp_kernels* find64(const ndq_mlp_desc* d) {
  // (hidden > 64: ONE hidden layer of up to 512 units -- csrc/ndq_wide.h compiled in double, registered extension modules only)
  if (!d || d->hidden < 1 || d->hidden > (d->layers == 1 ? 512 : 64)) return nullptr;
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

// out[i] = (acc ? out[i] : 0) + scale * sum_r partials[r*len + i], rows in fixed order.  A workgroup owns 64 columns; its
// 16 row groups each add up every 16th row in four independent chains, the 16 group sums are added in index order (the
// layout of ndq_api.hip's column_sum_1024).  An fp64 adjoint launch leaves up to 1 024 partial rows: with one thread per
// column walking all of them this sum took 15 us of a 130 us epoch, and the loss sum (len = 1) was a single thread.
__global__ __launch_bounds__(1024) void reduce_partials64_kernel(const double* __restrict__ part, int nparts, int len,
                                                                 double* __restrict__ out, int accumulate, double scale) {
  __shared__ double sm[16 * 64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  // Sixteen rows of a thread at a time as straight-line clamped loads, pinned by one empty asm statement, then added in
  // EXACTLY the order of the loops they replace:
  //     for (r = rg; r + 48 < nparts; r += 64) { s0 += row r; s1 += row r + 16; s2 += row r + 32; s3 += row r + 48; }
  //     for (; r < nparts; r += 16) s0 += row r;
  // which compile to "four loads, s_waitcnt vmcnt(0), four adds, branch" -- one memory round trip per 64 rows, sixteen of them
  // behind an fp64 adjoint launch's 1 024 partial rows (round 5; csrc/ndq_api.hip ColumnRows has the fp32 story).
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  const bool live = i < len;
  const int cc = live ? i : len - 1;
  for (int base = 0; base < nparts; base += 256) {
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = base + rg + 16 * j;
      x[j] = part[(size_t)(r < nparts ? r : nparts - 1) * len + cc];
    }
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                      "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r0 = base + rg + 64 * it;
      if (r0 + 48 < nparts) {
        s0 += x[4 * it]; s1 += x[4 * it + 1]; s2 += x[4 * it + 2]; s3 += x[4 * it + 3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (r0 + 16 * k < nparts) s0 += x[4 * it + k];
      }
    }
  }
  sm[rg * 64 + c] = live ? (s0 + s1) + (s2 + s3) : 0.;
  __syncthreads();
  if (rg == 0 && i < len) {
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k * 64 + c];
    s *= scale;
    out[i] = accumulate ? out[i] + s : s;
  }
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

// Device-side end of an fp64 epoch, the double twin of ndq_api.hip's epoch_tail_kernel: mean of the per-batch losses ->
// history ring, best-loss ping-pong + snapshot of the parameters the epoch was evaluated on, Adam (the arithmetic of
// adam64_kernel, operation for operation: an epoch through this kernel and one through ndq64_adam_step agree bit for bit).
struct Tail64Args {
  double* p; const double* g; double* m; double* v; int len;
  double lr, b1, b2, eps, wd, bc1, bc2s;
  const double* loss_slots; int nb; double* loss_hist; int hist_index; double* best_loss; int parity; double* best_flat;
  int write_scalars;
};
__global__ __launch_bounds__(256) void epoch_tail64_kernel(Tail64Args a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double loss = 0.;
  for (int k = 0; k < a.nb; ++k) loss += a.loss_slots[k];
  loss /= (double)a.nb;
  const double best = a.best_loss[a.parity];
  const bool better = (a.best_flat != nullptr) && (loss < best);   // false for NaN, like the reference's comparison
  if (i < a.len) {
    const double pi = a.p[i];
    if (better) a.best_flat[i] = pi;
    if (a.m != nullptr) {             // validation epochs pass no optimiser state: bookkeeping only
      double gi = a.g[i];
      if (a.wd != 0.0) gi = fma(a.wd, pi, gi);
      const double mi = fma(a.b1, a.m[i], (1.0 - a.b1) * gi);
      const double vi = fma(a.b2, a.v[i], (1.0 - a.b2) * gi * gi);
      a.m[i] = mi;
      a.v[i] = vi;
      a.p[i] = pi - (a.lr / a.bc1) * (mi / (sqrt(vi) / a.bc2s + a.eps));
    }
  }
  if (i == 0 && a.write_scalars) {
    a.loss_hist[a.hist_index] = loss;
    a.best_loss[a.parity ^ 1] = better ? loss : best;
  }
}

}  // namespace ndq

using namespace ndq;

extern "C" {

int ndq64_mlp_supported(const ndq_mlp_desc* desc) { return find64(desc) ? 1 : 0; }

int ndq64_mlp_register(const ndq64_mlp_kernels* k) {
  if (!k || !k->fwd || !k->bwd || k->desc.hidden < 1 || k->desc.hidden > (k->desc.layers == 1 ? 512 : 64) || k->n_streams < 1 || k->n_params < 1 || k->bwd_waves < 1 ||
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
  hipLaunchKernelGGL(reduce_partials64_kernel, dim3((len + 63) / 64), dim3(1024), 0, static_cast<hipStream_t>(stream),
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

int ndq64_epoch_tail(double* params, const double* grad, double* exp_avg, double* exp_avg_sq, int len, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int step, const double* loss_slots,
                     int n_batches, double* loss_hist, int hist_index, double* best_loss, int parity, double* best_flat,
                     int write_scalars, void* stream) {
  const bool adam = exp_avg != nullptr;
  if (!params || len <= 0 || !loss_slots || n_batches <= 0 || !loss_hist || !best_loss || hist_index < 0 ||
      (parity != 0 && parity != 1) || (adam && (!grad || !exp_avg_sq || step <= 0)))
    return NDQ_EINVAL;
  Tail64Args a{};
  a.p = params; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.len = len;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = adam ? 1.0 - pow(beta1, (double)step) : 1.0;
  a.bc2s = adam ? sqrt(1.0 - pow(beta2, (double)step)) : 1.0;
  a.loss_slots = loss_slots; a.nb = n_batches; a.loss_hist = loss_hist; a.hist_index = hist_index;
  a.best_loss = best_loss; a.parity = parity; a.best_flat = best_flat; a.write_scalars = write_scalars;
  hipLaunchKernelGGL(epoch_tail64_kernel, dim3((len + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

}  // extern "C"
