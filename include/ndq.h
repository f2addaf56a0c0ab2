/* ndq -- C-ABI of the MI355X-native PINN training core (libndq.so).
 *
 * The reference (NeuroDiffGym/neurodiffeq) has no FFI layer: its hot path is Python on top of torch autograd
 * (SURVEY.md 8b).  These entry points are what a binding for that path would call; each cites the reference code it
 * replaces.  All pointers are DEVICE pointers to fp32 unless stated; `stream` is a hipStream_t passed as void*.
 * No entry point allocates, synchronises or touches host memory.  Return value: 0 on success, a positive hipError_t,
 * or a negative NDQ_E* code.
 */
#ifndef NDQ_H
#define NDQ_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NDQ_EUNSUPPORTED (-1) /* no compiled kernel for this descriptor */
#define NDQ_EINVAL (-2)       /* bad argument */

#define NDQ_ACT_TANH 0 /* torch.nn.Tanh, default of FCNN (networks.py:27,52-53) */
#define NDQ_ACT_SIN 1  /* neurodiffeq.networks.SinActv (networks.py:142-152) */
#define NDQ_ACT_SIGMOID 2 /* torch.nn.Sigmoid passed as FCNN(actv=...) (networks.py:52-53) */
#define NDQ_ACT_SWISH 3   /* neurodiffeq.networks.Swish (networks.py:155-175); actp = 0: its default fixed beta = 1 */
#define NDQ_ACT_APTX 4    /* neurodiffeq.networks.APTx (networks.py:177-209); actp = 0: default fixed alpha = 1, beta = 1, gamma = 0.5 */
#define NDQ_ACT_ELU 5       /* torch.nn.ELU (alpha = 1) as FCNN(actv=nn.ELU): tests/test_pde.py:377 */
#define NDQ_ACT_SOFTPLUS 6  /* torch.nn.Softplus (beta = 1, threshold = 20): tests/test_pde.py:182 */
#define NDQ_ACT_GELU 7      /* torch.nn.GELU (approximate = 'none') */

/* Shape of one FCNN (networks.py:59-66: Linear(d,h) actv [Linear(h,h) actv]* Linear(h,n_out)) plus the set of
 * derivative streams of its raw output that the residual needs.  Stream order in every jets/gbar array:
 *   0: value | 1..d: d/dx_a (if first) | second-order pairs (a<=b) selected by mask2, enumerated
 *   (0,0),(0,1),..,(0,d-1),(1,1),..  (bit k of mask2 <-> k-th pair).
 *   With lap = 1 the second-order part is a single stream: sum over the diagonal pairs in mask2 of d2/dx_a^2.  */
#define NDQ_MAX_HIDDEN 512   /* widest hidden layer a descriptor may name: 1..64 csrc/ndq_mlp.h (fragment kernels, weights
                                resident in LDS), 65..512 csrc/ndq_wide.h (one hidden layer, units over lanes) */
typedef struct ndq_mlp_desc {
  int d;       /* number of input coordinates (1..3) */
  int first;   /* 1: first-order streams present */
  int mask2;   /* second-order pair mask */
  int hidden;  /* width of the hidden layers, 1..NDQ_MAX_HIDDEN -- the widest one if they differ (kernels lay every layer out
                  padded; the flat parameter vector holds the real widths) */
  int layers;  /* number of hidden layers */
  int act;     /* NDQ_ACT_* */
  int n_out;   /* output units */
  int lap;     /* 1: "Laplacian stream" -- the diagonal pairs of mask2 are carried as ONE stream holding their sum */
  int skip;    /* 1: Resnet (networks.py:73-106): out += S x with a trainable bias-free S (n_out x d, row-major) that
                  follows the output bias in the flat parameter vector */
  int mask3;   /* third-order triple mask: bit k <-> k-th triple a <= b <= c in lexicographic order; those streams follow
                  the second-order ones.  A triple needs its three pairs in mask2; lap must be 0.  (Sobolev losses of
                  second-order PDEs, losses.py:17-26; diff(u, t, order=3), neurodiffeq.py:21-34) */
  int actp;    /* 1: trainable activation parameters (Swish(trainable=True): beta; APTx(trainable=True): alpha, beta,
                  gamma -- one set per hidden layer, networks.py:166-169,196-203).  They sit at the END of the flat
                  parameter vector, layer after layer, and get gradient entries like every other parameter; beta and
                  gamma must be non-zero.
                  2: the same scalars as fixed non-default values (Swish(beta=2.0)): the layers x {1, 3} floats FOLLOW
                  the n_params trainable entries in the buffer `params` points to and have no gradient entries */
  int widths;  /* 0: every hidden layer is `hidden` wide.  Otherwise the widths of layers 1..layers, 8 bits each, layer 1
                  in the low byte (FCNN(hidden_units=(64, 32, 16)) -> 0x102040); `hidden` is their maximum.  Networks wider
                  than 64 units (hidden > 64): 10 bits each, two or three layers (hidden_units=(128, 64) -> 64 << 10 | 128) */
  int mono;    /* != 0: a MonomialNN (networks.py:109-139) in front of the first linear layer: bit k <-> degree k + 1
                  (ascending, 1..8).  The d coordinates become the d * n_degrees features x_a^deg, degree after degree,
                  and the first weight matrix is (hidden x d * n_degrees); streams up to second order, hidden <= 48, no skip */
  int mask4;   /* fourth-order quadruple mask (round 6): bit k <-> k-th quadruple a <= b <= c <= d in lexicographic order; those
                  streams follow the third-order ones.  A quadruple needs its six pairs in mask2 and its four triples in mask3;
                  lap, actp and mono must be 0; tanh / sin / sigmoid.  (diff(u, x, order=4), neurodiffeq.py:21-34: beam and
                  biharmonic equations) */
} ndq_mlp_desc;

/* One compiled kernel pair (forward streams / parameter-gradient adjoint) for ONE descriptor.  libndq.so carries a
 * table of them (csrc/ndq_api.hip); extension modules built at run time for other network shapes or stream sets
 * (neurodiffeq_amd/codegen.py: hipcc on csrc/ndq_launch.h with one Cfg) add theirs through ndq_mlp_register, after
 * which every entry point below serves that descriptor too. */
typedef struct ndq_mlp_kernels {
  ndq_mlp_desc desc;
  int n_streams, n_params;
  int bwd_waves;  /* waves per workgroup of the adjoint kernel (one 16-point tile per wave and iteration) */
  int lds_bytes;  /* dynamic LDS the larger of the two kernels asks for (a workgroup has 160 KiB) */
  int (*fwd)(const float* coords, int ldc, int n, const float* params, float* jets, int ldj, void* stream);
  int (*bwd)(const float* coords, int ldc, int n, const float* params, const float* gbar, int ldj, float* partials,
             int blocks, void* stream);
} ndq_mlp_kernels;
/* The record must stay valid for the life of the process.  Registering a descriptor twice is a no-op. */
int ndq_mlp_register(const ndq_mlp_kernels* kernels);

/* 1 if kernels for this descriptor are available (built in or registered). */
int ndq_mlp_supported(const ndq_mlp_desc* desc);
/* number of streams NS, of parameters P (flat torch order W1,b1,W2,b2,...,Wout,bout[,S]) */
int ndq_mlp_num_streams(const ndq_mlp_desc* desc);
int ndq_mlp_num_params(const ndq_mlp_desc* desc);
/* number of workgroups ndq_mlp_jet_bwd launches for n points == rows of `partials` it writes */
int ndq_mlp_bwd_blocks(const ndq_mlp_desc* desc, int n);

/* Fused FCNN forward with derivative streams.  Replaces FCNN.forward (networks.py:68-70) together with every
 * diff(net_out, x_a[, order=2]) sweep over it (neurodiffeq.py:21-34).
 *   coords [d][ldc], params [P], jets (out) [NS][n_out][ldj]  */
int ndq_mlp_jet_fwd(const ndq_mlp_desc* desc, const float* coords, int ldc, int n, const float* params, float* jets,
                    int ldj, void* stream);

/* Adjoint of ndq_mlp_jet_fwd w.r.t. the parameters: given gbar[s][o][n] = dLoss/d jets[s][o][n] it recomputes the
 * streams and writes per-workgroup partial sums of dLoss/dparams.  Replaces the part of loss.backward()
 * (solvers.py:393) that walks the differentiated network graph.
 *   partials (out) [ndq_mlp_bwd_blocks][P]  */
int ndq_mlp_jet_bwd(const ndq_mlp_desc* desc, const float* coords, int ldc, int n, const float* params,
                    const float* gbar, int ldj, float* partials, void* stream);

/* out[i] = (accumulate ? out[i] : 0) + scale * sum_{r<nparts} partials[r*len + i], fixed summation order.
 * Second stage of every reduction (parameter gradients: accumulate over n_batches like solvers.py:360-362;
 * loss: mean of squared residuals, solvers.py:218). */
int ndq_reduce_partials(const float* partials, int nparts, int len, float* out, int accumulate, float scale,
                        void* stream);

/* Fused Adam step on a flat parameter vector (torch.optim.Adam defaults, solvers.py:182); step is the 1-based
 * step count AFTER this update.  Optional fast path for the optimizer.step() of solvers.py:331-341. */
int ndq_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int len, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);

/* One fused launch for the two second-stage sums of a batch: out[i] = (accumulate ? out[i] : 0) + sum_r partials[r*len+i]
 * (parameter gradients) and *loss_out = loss_scale * sum_r loss_partials[r] (mean squared residual of the batch). */
int ndq_reduce_grad_loss(const float* partials, int nparts, int len, float* out, int accumulate,
                         const float* loss_partials, int n_loss_parts, float* loss_out, float loss_scale, void* stream);

/* Device-side end of a training epoch (solvers.py:407-419 without a host round trip): epoch loss = mean of the
 * n_batches loss slots -> loss_hist[hist_index]; best-network snapshot (solvers.py:434-441): if the epoch loss is
 * below best_loss[parity] the PRE-step parameters are copied to best_flat and best_loss[parity^1] is updated
 * (best_loss is a 2-slot ping-pong so no workgroup reads what another writes); then the fused Adam update.
 * write_scalars: only one network of a multi-network system records loss_hist / best_loss.
 * exp_avg == NULL (validation epoch): no Adam update, only the loss / best-snapshot bookkeeping. */
int ndq_epoch_tail(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int len, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, const float* loss_slots, int n_batches,
                   float* loss_hist, int hist_index, float* best_loss, int parity, float* best_flat, int write_scalars,
                   void* stream);

/* Device-side collocation-point sampler: the distributions of the reference's generators drawn with a counter-based
 * Philox4x32-10 stream instead of torch's host generator, so a fresh batch per epoch costs one small kernel and no
 * PCIe transfer.  (The reference samples on torch's default device with the global RNG, generators.py:150-152,
 * 253-266, 622-646; points drawn here follow the same distributions but are NOT the same numbers.) */
#define NDQ_SAMPLE_UNIFORM 0   /* Generator1D 'uniform' per coordinate: lo + (hi - lo) * U[0,1) */
#define NDQ_SAMPLE_GRID 1      /* Generator1D/2D/3D 'equally-spaced[-noisy]': ij-meshgrid of linspaces + N(0, std^2) */
#define NDQ_SAMPLE_SPHERICAL 2 /* GeneratorSpherical: direction as generators.py:622-635, radius per `radial` */
typedef struct ndq_sampler_desc {
  int kind;           /* NDQ_SAMPLE_* */
  int d;              /* coordinates per point (1..3; SPHERICAL: 3 = r, theta, phi) */
  int n[3];           /* GRID: points per axis (last axis fastest); otherwise n[0] = number of points */
  float lo[3], hi[3]; /* box per coordinate; SPHERICAL: lo[0] = r_min, hi[0] = r_max */
  float noise_std[3]; /* GRID: standard deviation of the jitter per axis (0: exact grid) */
  int radial;         /* SPHERICAL: 0: r^2 uniform ('equally-spaced-noisy'), 1: r uniform ('equally-radius-noisy') */
} ndq_sampler_desc;
/* coords (out) [d][ldc].  Point i uses Philox counter (i, draw, stream_id) under key `seed`: the same
 * (seed, draw, stream_id) always yields the same batch; ranks use distinct stream_id. */
int ndq_sample(const ndq_sampler_desc* desc, unsigned long long seed, unsigned long long draw, unsigned stream_id,
               float* coords, int ldc, void* stream);

/* Launcher exported by a generated single-network fused closure kernel (codegen.py: fused_source). */
typedef int (*ndq_fused_launch_fn)(const float* coords, int ldc, int n, const float* params, float* partials,
                                   float* loss_partials, float* funcs, float* resid, int ldj, float seed, int train,
                                   void* stream);

/* Sum-all-reduce of `count` fp32 values in place over the data-parallel ranks, enqueued on `stream`: the signature of
 * ncclAllReduce (RCCL) with dtype = ncclFloat32 (7) and op = ncclSum (0).  libndq.so does not link RCCL; the host
 * passes the function and its communicator in (parallel.py). */
typedef int (*ndq_allreduce_fn)(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm,
                                void* stream);

/* Everything one training epoch of a single-network system with n_batches_train = 1 needs, prepared once by the host:
 * closure kernel -> ndq_reduce_grad_loss [-> all-reduce of grad|loss] -> ndq_epoch_tail, issued back to back on
 * `stream` by ONE native call. */
typedef struct ndq_fused_step {
  ndq_fused_launch_fn launch;
  int n, ldc, ldj, blocks, n_params;
  float seed;                 /* 1 / (N_global * n_eq) */
  float* params;              /* [P] */
  float* partials;            /* [blocks][P] */
  float* loss_partials;       /* [blocks] */
  float* grad;                /* [P] */
  float* loss_slot;           /* [1] */
  float* adam_m;              /* [P]  (NULL: stop after the reductions) */
  float* adam_v;              /* [P] */
  float lr, beta1, beta2, eps, weight_decay;
  float* loss_hist;           /* ring of epoch losses */
  float* best_loss;           /* [2] */
  float* best_flat;           /* [P] */
  ndq_allreduce_fn allreduce; /* data parallel: sums grad[0..P) and the loss slot, which must be grad[P] (one message) */
  void* comm;                 /* ncclComm_t for `allreduce` */
  /* optional prefetch of the NEXT batch of a device generator: extra workgroups of the sums + tail kernel draw
   * (next_sampler, next_seed, next_draw, next_stream) into next_coords [d][next_ldc] -- the block this step's closure
   * kernel has just finished reading -- so the sampler launch leaves the step.  NULL: nothing drawn. */
  const struct ndq_sampler_desc* next_sampler;
  unsigned long long next_seed, next_draw;
  unsigned next_stream;
  float* next_coords;
  int next_ldc;
} ndq_fused_step;
int ndq_fused_step_run(const ndq_fused_step* s, const float* coords, int adam_step, int hist_index, int parity,
                       void* stream);

/* The same for a system of 2..4 networks of one shape served by ONE closure launch (generated launcher taking host
 * arrays of per-network device pointers): steps[k] describes network k (params, partials, grad, Adam state, best
 * snapshot; n / ld / blocks / seed / loss_partials / loss_hist / best_loss are read from steps[0]); after the closure
 * kernel every network gets its fused second-stage-sums + tail kernel.  No data-parallel hook on this entry point. */
typedef int (*ndq_fused_launch_multi_fn)(const float* coords, int ldc, int n, const float* const* params,
                                         float* const* partials, float* loss_partials, float* funcs, float* resid,
                                         int ldj, float seed, int train, void* stream);
int ndq_fused_multi_step_run(const ndq_fused_step* steps, int n_nets, ndq_fused_launch_multi_fn launch,
                             const float* coords, int adam_step, int hist_index, int parity, void* stream);

/* ---- fit(): several epochs of (training epoch, validation epoch) from ONE native call -------------------------------
 * The reference's fit loop (solvers.py:443-497) alternates run_train_epoch / run_valid_epoch and returns to Python after
 * each; at its default sizes (32 .. 1 024 points, solvers.py:1083-1086, 1501-1504) that host round trip costs several
 * times the kernels.  ndq_fused_fit_run enqueues n_epochs epochs back to back, two launches per epoch:
 *   1. closure launch: the training closure of epoch e on train_coords[e] AND -- spare workgroups of the same launch --
 *      the forward-only closure on the validation batch, both with the parameters epoch e starts from (= the parameters
 *      epoch e - 1's validation epoch is defined on, solvers.py:484-485);
 *   2. sums + tail launch: second-stage sums, training loss -> loss_hist[hist_index + e], validation loss of epoch e - 1
 *      -> valid_hist[valid_index + e - 1], best-network snapshot of the PRE-update parameters (solvers.py:434-441), Adam.
 * After the last epoch the validation batch is evaluated once more on its own (closure launch without training
 * workgroups + tail without Adam).  n_epochs = 0 with a validation batch is one stand-alone validation epoch,
 * n_epochs = 1 without one a stand-alone training epoch: the per-epoch entry points of the Solver run the SAME device
 * code, so a loss history does not depend on how its epochs were grouped into calls (bit for bit). */
typedef int (*ndq_fused_launch_tv_fn)(const float* coords, int ldc, int n, const float* const* params,
                                      float* const* partials, float* loss_partials, float seed,
                                      const float* valid_coords, int valid_ldc, int valid_n,
                                      float* valid_loss_partials, const void* pull, void* stream);
/* `pull` (NULL: none): an ndq::PullArgs of csrc/ndq_tail.h.  PULL MODE, used while training workgroups x networks <= 16: the
 * closure launch of epoch e first FINISHES epoch e - 1 itself -- every workgroup adds up the partial rows of the previous
 * launch (same summation order as the tail kernel), applies Adam to all parameters in registers and stages its weight
 * image from the result; workgroup 0 writes the new parameters / moments / history.  One launch per epoch instead of
 * two.  It needs a second set of buffers (a launch reads the state its predecessor wrote while writing the next one):
 * the alt_* members below; the call's last launch is an ordinary tail that leaves everything in the primary buffers. */
/* LOOP MODE (csrc/ndq_tail.h: ndq::LoopArgs), used when the training grid and the validation grid are ONE workgroup each
 * (the reference's default ODE solvers: 32 points) and there are at most two networks: one launch of one workgroup
 * stands for up to 64 launches of the pull-mode sequence -- prologue, validation closure, training closure, again -- with
 * parameters, moments, gradient row and loss partials in LDS.  Exported by the closure module as ndq_fused_launch_loop
 * (ndq_fused_loop_ok() != 0 when its kernel supports it).  Same device functions in the same order as the other two
 * routes: the three are interchangeable bit for bit. */
typedef int (*ndq_fused_launch_loop_fn)(const float* coords, int ldc, int n, float seed, const float* vcoords, int vldc,
                                        int vn, const void* loop, void* stream);
typedef struct ndq_fused_fit {
  ndq_fused_launch_tv_fn launch;   /* exported by the generated closure kernel module as ndq_fused_launch_tv */
  int n_nets;                      /* 1..4 networks behind the one closure launch */
  /* per network: params, partials, grad, loss_slot, n_params, Adam state and hyper-parameters, best_flat;
   * n / ldc / blocks / seed / loss_partials / loss_hist / best_loss are read from net[0]; launch / allreduce / comm /
   * next_* of the entries are ignored */
  ndq_fused_step net[4];
  const float* valid_coords;       /* [d][valid_ldc] resident validation batch; NULL: no validation epochs */
  int valid_n, valid_ldc, valid_blocks;
  float valid_scale;               /* 1 / (valid_n * n_eq): validation loss = valid_scale * sum of the block partials */
  float* valid_loss_partials;      /* [valid_blocks] */
  float* valid_hist;               /* ring of validation-epoch losses */
  int track_best;                  /* 0: no snapshot, 1: lowest TRAINING loss (n_batches_valid = 0, solvers.py:414-415),
                                      2: lowest VALIDATION loss */
  /* pull mode: pull_ok != 0 (the closure module supports it: ndq_fused_pull_ok) and every alt_* buffer present */
  int pull_ok;
  float* alt_params[4];            /* [P_k] second parameter vector of network k, likewise the Adam moments */
  float* alt_m[4];
  float* alt_v[4];
  float* alt_partials[4];          /* [blocks][P_k] */
  float* alt_loss_partials;        /* [blocks] */
  float* alt_valid_loss_partials;  /* [valid_blocks] */
  /* loop mode: both set by the caller when the module exports it (needs the alt_* buffers too) */
  ndq_fused_launch_loop_fn launch_loop;
  int loop_ok;
} ndq_fused_fit;
/* train_coords: HOST array of n_epochs device pointers, the [d][ldc] batch of every epoch.  adam_step: step count AFTER
 * the first update (epoch e uses adam_step + e).  parity: slot of best_loss[2] holding the current best; every tail
 * launch (n_epochs of them, + 1 with a validation batch) flips it. */
int ndq_fused_fit_run(const ndq_fused_fit* f, int n_epochs, const float* const* train_coords, int adam_step,
                      int hist_index, int valid_index, int parity, void* stream);

/* Signature of a generated pointwise kernel launcher (one per traced PDE system, built by
 * neurodiffeq_amd/codegen.py with hipcc).  It evaluates the condition re-parameterisation (conditions.py
 * `parameterize`), the user's residuals (`diff_eqs`, solvers.py:380), the squared-residual partial sums
 * (solvers.py:218) and -- when gbar != NULL -- the adjoint of the loss w.r.t. every network output stream.
 *   coords [d][ldc]; jets/gbar: arrays of n_nets device pointers, each [NS_k][n_out_k][ldj];
 *   funcs (opt, out) [n_funcs][ldj]; resid (opt, out) [n_eq][ldj];
 *   loss_partials (out) [ndq_pw_blocks(n)] block sums of r^2 over all equations;
 *   seed_scale: gbar = seed_scale * d(sum r^2)/d jets, i.e. 1/(N_global * n_eq) for the mean. */
typedef int (*ndq_pointwise_fn)(const float* coords, int ldc, int n, const float* const* jets,
                                float* const* gbar, int ldj, float* funcs, float* resid, float* loss_partials,
                                float seed_scale, void* stream);
typedef int (*ndq_pw_blocks_fn)(int n);
/* Inverse problems: the reference re-evaluates the user's diff_eqs under autograd every batch (solvers.py:380), so
 * nn.Parameter coefficients inside the equations receive gradients and (N, 1) data tensors are just operands.  In a
 * generated module the n_theta trainable scalars are a device vector every launch reads, their gradient the fixed-order
 * sum of per-point adjoints (block rows theta_partials [blocks][n_theta], second stage: ndq_reduce_partials), and
 * the n_data per-point columns are rows [d, d + n_data) of the coordinate block.  Every generated module (pointwise:
 * ndq_pw_bind_theta, closure: ndq_fused_bind_theta) exports the binder below; it is called before a launch whenever
 * n_theta > 0 (theta_partials may be NULL for forward-only launches). */
typedef void (*ndq_bind_theta_fn)(const float* theta, float* theta_partials);

/* ---- fp64 variants (libndq64.so) ------------------------------------------------------------------------------------
 * The reference's default precision is fp64 (neurodiffeq/__init__.py:22, utils.py:10-41).  libndq64.so carries the SAME
 * stream kernels compiled for double (csrc/ndq_mlp.h with NDQ_F64: per-point GEMMs on v_mfma_f64_16x16x4_f64, no bf16
 * splitting, libm transcendentals) behind entry points of the same meaning; every float buffer is a double buffer.  It
 * serves the torch custom-op seam (neurodiffeq_amd/autograd_ops.py) for fp64 networks: FCNN.forward `networks.py:68-70`
 * + the diff() sweeps `neurodiffeq.py:21-34` + the parameter part of loss.backward() `solvers.py:393`. */
typedef struct ndq64_mlp_kernels {
  ndq_mlp_desc desc;
  int n_streams, n_params;
  int bwd_waves;
  int lds_bytes;
  int (*fwd)(const double* coords, int ldc, int n, const double* params, double* jets, int ldj, void* stream);
  int (*bwd)(const double* coords, int ldc, int n, const double* params, const double* gbar, int ldj, double* partials,
             int blocks, void* stream);
} ndq64_mlp_kernels;
int ndq64_mlp_register(const ndq64_mlp_kernels* kernels);
int ndq64_mlp_supported(const ndq_mlp_desc* desc);
int ndq64_mlp_num_streams(const ndq_mlp_desc* desc);
int ndq64_mlp_num_params(const ndq_mlp_desc* desc);
int ndq64_mlp_bwd_blocks(const ndq_mlp_desc* desc, int n);
int ndq64_mlp_jet_fwd(const ndq_mlp_desc* desc, const double* coords, int ldc, int n, const double* params, double* jets,
                      int ldj, void* stream);
int ndq64_mlp_jet_bwd(const ndq_mlp_desc* desc, const double* coords, int ldc, int n, const double* params,
                      const double* gbar, int ldj, double* partials, void* stream);
int ndq64_reduce_partials(const double* partials, int nparts, int len, double* out, int accumulate, double scale,
                          void* stream);
/* ndq_adam_step in double (torch.optim.Adam, amsgrad = False, maximize = False) */
int ndq64_adam_step(double* params, const double* grad, double* exp_avg, double* exp_avg_sq, int len, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int step, void* stream);

/* ndq_epoch_tail in double: the device-side end of an epoch of an fp64 system (loss slots -> history ring, best snapshot,
 * Adam with ndq64_adam_step's arithmetic) -- with it an fp64 Solver epoch needs no host synchronisation either
 * (solvers.py:407-419, 434-441 of the reference). */
int ndq64_epoch_tail(double* params, const double* grad, double* exp_avg, double* exp_avg_sq, int len, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int step, const double* loss_slots,
                     int n_batches, double* loss_hist, int hist_index, double* best_loss, int parity, double* best_flat,
                     int write_scalars, void* stream);

/* ---- one-shot all-reduce of the small [gradient | loss] message (data-parallel training) ------------
 * Every rank writes its vector straight into every peer's inbox (fine-grained device memory shared through HIP IPC; on
 * MI355X one xGMI hop to each of the 7 peers) and adds up the world_size vectors it received, in rank order: ONE
 * kernel launch per call, bit-identical results on all ranks.  ndq_oneshot_allreduce has ncclAllReduce's signature
 * (fp32 sum only) so that ndq_fused_step.allreduce / .comm can point at it; replaces, for this message size, the RCCL
 * ring / tree behind solvers.py:393's gradient in a data-parallel run.
 *   create:  allocates this rank's inbox for vectors of up to max_len floats, returns its 64-byte IPC handle
 *   connect: handles = world_size x 64 bytes in rank order (every rank's own handle included), opens the peers' inboxes
 *   status:  number of peer-flag waits that hit their spin limit so far (synchronises); 0 = healthy */
int ndq_oneshot_create(int rank, int world_size, int max_len, void** ctx, unsigned char* handle64);
int ndq_oneshot_connect(void* ctx, const unsigned char* handles);
int ndq_oneshot_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* ctx, void* stream);
int ndq_oneshot_status(void* ctx);
int ndq_oneshot_destroy(void* ctx);

#ifdef __cplusplus
}
#endif
#endif
