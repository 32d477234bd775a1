/*
 * dcgp.h -- C-ABI of libdcgp.so: the MI355X (gfx950) conv-GP forward / ELBO hot path of DeepCGP.
 *
 * The reference (kekeblom/DeepCGP) has no FFI seam: the path is Python methods that build
 * TensorFlow graph nodes.  Every entry point below therefore names the reference *method* it
 * replaces (file:line under the reference tree); the Python classes in deepcgp_amd/ keep the
 * reference's names and call these through ctypes (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - every array is float64 ("double"), C-contiguous row-major, resident in DEVICE memory unless
 *     the parameter name ends in _host; labels are int32;
 *   - every function returns a status (DCGP_OK == 0); dcgp_last_error(ctx) gives the message;
 *   - one ctx <-> one device <-> one HIP stream; a ctx is not thread-safe; calls are stream-ordered
 *     and have completed (stream synchronised) on return unless stated otherwise;
 *   - the caller owns every buffer it allocates with dcgp_malloc; the library owns only internal
 *     workspaces hanging off the ctx / model;
 *   - "not positive definite" (the reference's tf.errors.InvalidArgumentError from tf.cholesky,
 *     conv_gp/experiment.py:45) is DCGP_ERR_NOT_PD with the 1-based failing column in *info.
 */
#ifndef DCGP_H
#define DCGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCGP_OK 0
#define DCGP_ERR_ARG (-1)      /* bad argument (NULL, non-positive size, unsupported shape)      */
#define DCGP_ERR_HIP (-2)      /* a HIP runtime call failed                                       */
#define DCGP_ERR_NOT_PD (-3)   /* Cholesky hit a non-positive pivot                               */
#define DCGP_ERR_RCCL (-4)     /* an RCCL call failed / communicator missing                      */
#define DCGP_ERR_ALLOC (-5)    /* device allocation failed                                        */

typedef struct dcgp_ctx dcgp_ctx;
typedef struct dcgp_model dcgp_model;

/* ---- context, memory, errors ------------------------------------------------------------- */
int dcgp_ctx_create(int device, dcgp_ctx** out);
int dcgp_ctx_destroy(dcgp_ctx* ctx);
const char* dcgp_last_error(dcgp_ctx* ctx);
int dcgp_device_count(int* count);
int dcgp_malloc(dcgp_ctx* ctx, size_t bytes, void** dptr);
int dcgp_free(dcgp_ctx* ctx, void* dptr);
int dcgp_h2d(dcgp_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int dcgp_d2h(dcgp_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int dcgp_memset(dcgp_ctx* ctx, void* dptr, int value, size_t bytes);
int dcgp_sync(dcgp_ctx* ctx);
/* Device memory the library itself holds for this ctx (the named, grow-only workspaces: factor scratch, K_uf / A1 slabs of the
 * sweep + GEMM route, partial sums) and the number of such workspaces.  SURVEY.md 8(b) planned a dcgp_workspace_query that SIZES
 * caller-provided workspaces; the library grows its own lazily instead (nothing on the reference side owns scratch memory either:
 * TensorFlow allocates graph temporaries itself), so the query reports what has been taken so far.                               */
int dcgp_workspace_query(dcgp_ctx* ctx, size_t* bytes_out, int* count_out);
/* A/B and debugging switches (csrc/common.h: DcgpOptions).  A ctx reads the environment variable DCGP_<NAME> once, in
 * dcgp_ctx_create, as the switch's initial value; afterwards only these calls change it -- no getenv on the step path.  Names are
 * lower case ("no_fused_layer", "kl_side", ...); an unknown name is DCGP_ERR_ARG.                                                 */
int dcgp_ctx_set_option(dcgp_ctx* ctx, const char* name, long value);
int dcgp_ctx_get_option(dcgp_ctx* ctx, const char* name, long* value_out);

/* ---- per-kernel HIP-event timing (bench.py's roofline leg) -------------------------------- */
/* When enabled, the launches of the named kernel families ("kuf", "gemm_cond", "potrf", "trtri",
 * "head_kzx", "head_kdiag", ...) are bracketed by hipEvents on the ctx stream.                 */
int dcgp_timing_enable(dcgp_ctx* ctx, int on);   /* 0 off, 1 every family, 2 only the roofline kernels ("conv_fused", "gemm_cond_s3", "kuf"), every 7th launch of them */
int dcgp_timing_reset(dcgp_ctx* ctx);
int dcgp_timing_query(dcgp_ctx* ctx, const char* name, int* launches, double* total_ms);
int dcgp_timing_names(dcgp_ctx* ctx, char* buf, size_t buflen);   /* ';'-separated list */

/* ---- patch view: FullView.extract_patches / extract_patches_PNL (conv_gp/views.py:32-54) --- */
/* X [N,H,W,C] -> out [N,P,L] (pnl == 0) or [P,N,L] (pnl != 0); VALID window, dilation 1;
 * p = oh*W' + ow, l = (kh*f + kw)*C + c.                                                         */
int dcgp_extract_patches(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                         double* out, int pnl);

/* ---- inducing-patch kernel matrices ------------------------------------------------------- */
/* MultiOutputConvKernel.Kuu (conv_gp/layers.py:18-21) == RBF.K(Z) + jitter*I; also the dispatch
 * Kuu(feature, kern, jitter) of conv_gp/kernels.py:172-174.  Z [M,L] -> out [M,M].               */
int dcgp_kuu_rbf(dcgp_ctx* ctx, const double* Z, int M, int L, double variance, double lengthscale,
                 double jitter, double* out_MM);
/* FullView.extract_patches_PNL + MultiOutputConvKernel.Kuf (conv_gp/views.py:40-44,
 * conv_gp/layers.py:23-32) fused: the patches are gathered from an LDS-staged image and never
 * materialised.  X [N,H,W,C], Z [M,L] -> out.  layout 0: [P,M,N] (the reference's Kuf layout);
 * layout 1: [M, N*P] with column n*P + p (the layout the fused conditional consumes).            */
int dcgp_kuf_patches_rbf(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                         const double* Z, int M, double variance, double lengthscale,
                         double* out, int layout);
/* The same two matrices for the ArcCosine(order = 0) base kernel that `--base-kernel acos` selects for the conv
 * layers (conv_gp/models.py:118-119; gpflow.kernels.ArcCosine: k = variance * (pi - theta) / pi,
 * theta = acos(1e-15 + (1 - 2e-15) cos), cos from <x,z> = weight_variance * x.z + bias_variance; Kdiag = variance). */
int dcgp_kuu_acos(dcgp_ctx* ctx, const double* Z, int M, int L, double variance, double weight_variance,
                  double bias_variance, double jitter, double* out_MM);
int dcgp_kuf_patches_acos(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                          const double* Z, int M, double variance, double weight_variance,
                          double bias_variance, double* out, int layout);

/* ---- dense M x M factorisations ------------------------------------------------------------ */
/* tf.cholesky (conv_gp/conditionals.py:29, layers.py:151,156): in place, lower factor, strict
 * upper triangle zeroed.  *info_host = 0, or the 1-based column of the first non-positive pivot
 * (return value DCGP_ERR_NOT_PD).                                                                */
int dcgp_potrf_lower(dcgp_ctx* ctx, double* A_MM, int M, int* info_host);
/* inverse of a lower-triangular factor (the form in which the triangular solves of
 * conv_gp/conditionals.py:31-33,44-47 are applied on the matrix cores).                          */
int dcgp_trtri_lower(dcgp_ctx* ctx, const double* L_MM, int M, double* Linv_MM);

/* ---- the conditional ---------------------------------------------------------------------- */
/* conditional(Kmn, Kmm, Knn, f, full_cov=False, q_sqrt, white) of conv_gp/conditionals.py:6-67.
 * Kmn [P,M,N], Kmm [M,M] (not overwritten), Knn [P,N], f [M,R], q_sqrt [R,M,M] lower-triangular
 * (its strict upper triangle is ignored, matrix_band_part at :55) or NULL.
 * out_mean [N,P,R], out_var [R,P,N].                                                            */
int dcgp_conditional(dcgp_ctx* ctx, const double* Kmn, const double* Kmm, const double* Knn,
                     const double* f, const double* q_sqrt, int white, int P, int M, int N, int R,
                     double* out_mean, double* out_var, int* info_host);

/* ConvLayer.conditional_ND (conv_gp/layers.py:96-135) for the Zero mean function, fused end to
 * end (patch gather -> Kuf -> Cholesky -> conditional), plus Layer.sample_from_conditional's
 * reparameterisation when z != NULL.  X [N, H*W*C]; out_* [N, P*R] (column p*R + r); any of the
 * three outputs may be NULL.  z [N, P*R].  identity_mean != 0 adds Conv2dMean
 * (conv_gp/mean_functions.py:28-41; odd filter sizes only).                                      */
int dcgp_conv_layer_forward(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                            const double* Z, int M, int R, double variance, double lengthscale,
                            const double* q_mu, const double* q_sqrt, int white, int identity_mean,
                            const double* z, double jitter,
                            double* out_sample, double* out_mean, double* out_var, int* info_host);

/* ---- classification-head kernels ----------------------------------------------------------- */
/* ConvKernel.Kzx (conv_gp/kernels.py:117-133) / AdditivePatchKernel.Kzx (:63-74; identical
 * arithmetic): out[m,n] = (1/P) sum_p w[p] k(Z[m], x[n,p]).  X [N,H,W,C], Z [M,L], w [P] -> [M,N] */
int dcgp_convkernel_kzx(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                        const double* Z, int M, double variance, double lengthscale, const double* w,
                        double* out_MN);
/* ConvKernel.Kdiag (conv_gp/kernels.py:106-115): out[n] = (1/P^2) sum_{p,p'} w[p] w[p'] k(x[n,p], x[n,p']). */
int dcgp_convkernel_kdiag(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                          double variance, double lengthscale, const double* w, double* out_N);
/* AdditivePatchKernel.Kdiag (conv_gp/kernels.py:53-61): out[n] = variance * mean_p w[p].        */
int dcgp_additive_kdiag(dcgp_ctx* ctx, int N, int P, double variance, const double* w, double* out_N);

/* doubly_stochastic_dgp SVGP_Layer.conditional_ND (call site conv_gp/models.py:192-198) given
 * Kuf [M,N], Ku [M,M] (jitter already added), Kdiag [N]; out_mean, out_var [N,R].               */
int dcgp_svgp_conditional(dcgp_ctx* ctx, const double* Kuf, const double* Ku, const double* Kdiag,
                          const double* q_mu, const double* q_sqrt, int white, int M, int N, int R,
                          double* out_mean, double* out_var, int* info_host);

/* ---- KL, likelihood, sampling --------------------------------------------------------------- */
/* gpflow.kullback_leiblers.gauss_kl(q_mu, q_sqrt, K) (call sites conv_gp/layers.py:145,147);
 * K == NULL is the whitened prior.  q_mu [M,R], q_sqrt [R,M,M], K [M,M].                         */
int dcgp_gauss_kl(dcgp_ctx* ctx, const double* q_mu, const double* q_sqrt, const double* K, int M, int R,
                  double* out_host, int* info_host);
/* gpflow MultiClass(K) + RobustMax(eps).variational_expectations, 20 Gauss-Hermite points (call
 * site conv_gp/models.py:67).  mu, var [n,K]; y [n] int32 -> out [n].                            */
int dcgp_robustmax_varexp(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n,
                          int K, double eps, double* out_n);
/* MultiClass.predict_mean_and_var: per-class probabilities [n,K].                                */
int dcgp_robustmax_predict(dcgp_ctx* ctx, const double* mu, const double* var, int n, int K, double eps,
                           double* out_p);
/* doubly_stochastic_dgp.utils.reparameterize: out = mean + z*sqrt(var + jitter), n elements.     */
int dcgp_reparam(dcgp_ctx* ctx, const double* mean, const double* var, const double* z, size_t n,
                 double jitter, double* out);

/* ---- model-level path: DGP_Base.propagate / _build_likelihood (doubly_stochastic_dgp, call sites
 *      conv_gp/models.py:65-70, conv_gp/utils/tensorboard.py:32) ------------------------------- */
int dcgp_model_create(dcgp_ctx* ctx, int num_samples, double jitter, dcgp_model** out);
int dcgp_model_destroy(dcgp_model* model);
/* ConvLayer (conv_gp/layers.py:52-94).  Parameter arrays are HOST pointers, copied to the device.
 * Z0 is the frozen initial Z of the KL prior (conv_gp/layers.py:149-152); NULL -> Z.             */
int dcgp_model_add_conv_layer(dcgp_model* model, int H, int W, int C, int f, int stride, int M, int R,
                              int white, int identity_mean, double variance, double lengthscale,
                              const double* Z_host, const double* Z0_host,
                              const double* q_mu_host, const double* q_sqrt_host);
/* SVGP_Layer(kern=ConvKernel|AdditivePatchKernel) (conv_gp/models.py:169-198).
 * kernel_type 0 = ConvKernel, 1 = AdditivePatchKernel.                                           */
int dcgp_model_set_head(dcgp_model* model, int H, int W, int C, int f, int stride, int M, int R,
                        int white, int kernel_type, double variance, double lengthscale,
                        const double* Z_host, const double* w_host,
                        const double* q_mu_host, const double* q_sqrt_host);
/* keep every layer's (sample, mean, var) of the next forward passes for dcgp_model_layer_output   */
int dcgp_model_set_keep_outputs(dcgp_model* model, int on);
/* Push a changed parameter: which = "Z", "Z0", "q_mu", "q_sqrt", "w", "variance", "lengthscale", or
 * "base_kernel" = {type, variance, p1, p2}: type 0 RBF (p1 = lengthscale), type 1 ArcCosine order 0 (p1 = weight
 * variance, p2 = bias variance; conv layers only, conv_gp/models.py:113-121), or "ard_lengthscales" = one lengthscale
 * per input dimension for a single-patch head (H = W = f = 1, C = D): gpflow RBF(D, ARD=True) on the flattened
 * features, the dense head of --last-kernel rbf (conv_gp/models.py:160-168).  "likelihood_epsilon" (one value in
 * (0, 1), `layer` ignored) is the RobustMax epsilon of the ELBO / predict_y entry points (default 1e-3).  */
int dcgp_model_set_param(dcgp_model* model, int layer, const char* which, const double* value_host,
                         size_t count);

/* One forward ELBO evaluation of a minibatch (compute_log_likelihood semantics):
 *   X [N, H*W*C], y [N] int32 (device).  z_per_layer_host: array of num_layers DEVICE pointers,
 *   each [S, N, D_l] standard-normal noise (entries may be NULL -> counter-based device RNG with
 *   `seed`); NULL -> RNG for all layers.  scale = num_data / global_batch.  dedup_layer0 != 0
 *   evaluates the first layer on the N distinct images only (propagate() tiles X S times, so the
 *   S copies are identical; results are bit-identical either way).
 *   If the ctx holds a communicator (dcgp_comm_init_rank) the data term is all-reduced (sum) over
 *   the ranks before scaling.  out_host[0] = ELBO, [1] = sum_n E_q log p(y_n) (global), [2] = sum KL. */
int dcgp_elbo_forward(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                      const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0,
                      double* out_host, int* info_host);
/* Throughput mode of dcgp_elbo_forward for loops that do not need step i's value before step i + 1 is queued (an
 * optimisation loop: session.run(train_op) at conv_gp/experiment.py:84-108 returns nothing; the logger reads the
 * objective every test_every steps only).  _enqueue queues the same launches and returns a ticket without waiting;
 * _collect waits for that step and hands back what dcgp_elbo_forward would have (same values, same error codes).
 * Tickets are collected in the order they were handed out, at most 4 may be outstanding; X, y and the noise buffers of
 * an enqueued step must stay untouched until it is collected.  The host then runs ahead of the device, which hides
 * the launch latency at the head of a step and the wake-up after it. */
int dcgp_elbo_forward_enqueue(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                              const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0,
                              uint64_t* ticket);
int dcgp_elbo_forward_collect(dcgp_model* model, uint64_t ticket, double* out_host, int* info_host);
/* The ELBO of dcgp_elbo_forward AND its gradient with respect to every trainable value (what TensorFlow autodiff
 * hands the optimiser at conv_gp/experiment.py:84-108): Z, q_mu, q_sqrt (lower triangle), the base-kernel
 * hyper-parameters of every layer (variance + lengthscale, ArcCosine: variance + weight / bias variances, dense head:
 * variance + one lengthscale per input dimension), patch_weights of the head -- constrained values, not gpflow's
 * unconstrained ones.  dedup_layer0 as in dcgp_elbo_forward: the first layer's conditional and its reverse pass run on
 * the N distinct images (the S gradients per image are added first) -- same values.
 * The gradients stay on the device; read them with dcgp_model_get_grad. */
int dcgp_elbo_grad(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                   const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, double* out_host,
                   int* info_host);
/* which = "Z" [M, L], "q_mu" [M, R], "q_sqrt" [R, M, M], "variance" [1], "lengthscale" [1] (RBF) or "weight_variances" [1] and
 * "bias_variance" [1] (ArcCosine conv layers), "w" [P] (head),
 * "ard_lengthscales" [D] (dense RBF(ARD) head; its scalar "lengthscale" gradient is 0). */
int dcgp_model_get_grad(dcgp_model* model, int layer, const char* which, double* out_host, size_t count);
/* Data parallelism of the gradient: every rank holds a shard of the batch and the full parameters.  The data part of
 * the gradient is a sum over shards, the KL part is replicated, so each rank computes scale * data_grad(shard) -
 * KL_grad / shards and the sum over ranks is the full gradient.  With a communicator on the ctx
 * (dcgp_comm_init_rank) dcgp_elbo_grad does that sum itself: one in-stream ncclAllReduce per layer over the layer's
 * contiguous gradient block.  Without one (host-side reduction, tests) set the shard count explicitly and reduce
 * the blocks yourself: block = [Z | q_mu | q_sqrt | w | variance, p1, p2 | ARD lengthscales (head)], device pointer
 * (p1 = lengthscale or ArcCosine weight variance, p2 = ArcCosine bias variance).  shards = 0
 * restores the default (ranks of the communicator, else 1). */
int dcgp_model_set_grad_shards(dcgp_model* model, int shards);
/* Multi-GPU: this rank holds images [first_image, first_image + N) of a minibatch of global_batch images.  With it
 * declared the device RNG of Layer.sample_from_conditional draws every element at its counter in the UN-sharded batch, so the ELBO
 * of a step does not depend on the number of ranks (global_batch = 0: not sharded, one RNG stream per rank). */
int dcgp_model_set_shard(dcgp_model* model, int first_image, int global_batch);
int dcgp_model_grad_block(dcgp_model* model, int layer, double** block_dev, size_t* count);
/* One optimiser step on the gradients dcgp_elbo_grad left on the device: tf.train.AdamOptimizer semantics
 * (lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); theta -= lr_t m / (sqrt(v) + eps); t = 1, 2, ...) ascending the
 * ELBO in gpflow's unconstrained space -- variance / lengthscale through transforms.positive (softplus + 1e-6),
 * q_sqrt on its lower triangle, Z / q_mu / patch_weights as they are (gpflow.train.AdamOptimizer at
 * conv_gp/experiment.py:104-107; the learning-rate schedule :71-73 stays with the caller).  t = 0: use the model's
 * own count of steps taken since its (zero-initialised) moment buffers were created -- a freshly built optimiser
 * restarts its beta powers whatever global_step a loaded checkpoint carries (experiment.py:84-89). */
int dcgp_model_adam_step(dcgp_model* model, double lr, double beta1, double beta2, double eps, int t);
/* One training step in one call -- dcgp_elbo_grad and dcgp_model_adam_step enqueued back to back with a single wait at the end: what
 * session.run(minimise_op) is to the reference (conv_gp/experiment.py:84-108).  Arguments: those of the two calls.  A step whose
 * factorisation fails returns DCGP_ERR_NOT_PD and leaves parameters, moments and step count as they were (the update reads the step's
 * status word on the device). */
int dcgp_model_train_step_adam(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                               const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, double lr, double beta1,
                               double beta2, double eps, int t, double* out_host, int* info_host);
/* Multi-rank training step (one process per GPU, dcgp_comm_init_rank): how a step's gradient is exchanged inside dcgp_model_train_step_adam.
 * 0 (default): ncclAllReduce of every layer's gradient block, every rank then updates every parameter.  1: ncclReduceScatter of the block
 * (each rank receives the sum of its shard only -- dcgp_shard_range), Adam on that shard of the layer's parameter block, ncclAllGather
 * of the updated parameters: half the bytes on the links of an all-reduce + nothing, and 1 / ranks of the optimiser arithmetic per rank
 * (the counterpart of nothing in the reference, which is single-device).  Frozen groups (dcgp_model_set_trainable)
 * pass through unchanged.  dcgp_elbo_grad on its own always all-reduces: its caller reads whole gradients. */
int dcgp_model_set_grad_exchange(dcgp_model* model, int mode);
/* Shards of a block of n values over nranks ranks: every shard ceil(n / nranks) long (the collectives want equal counts), rank r holds
 * [lo, hi) = [r * shard, min((r + 1) * shard, n)).  deepcgp_amd/dist.py grad_shard_range is the same arithmetic on the host. */
int dcgp_shard_range(long n, int nranks, int rank, long* lo, long* hi, long* shard /* may be NULL */);
/* Debugging aid: one Adam step taken the way `ranks` ranks take it in exchange mode 1, played on this one GPU from rank 0's point of view
 * (shard 0 in place, the others through the staging block and the unstage pass).  Needs the complete gradient (dcgp_elbo_grad).  Bit-identical
 * to dcgp_model_adam_step. */
int dcgp_model_debug_sharded_adam(dcgp_model* model, int ranks, double lr, double beta1, double beta2, double eps, int t);
/* Plain gradient ascent in the same unconstrained space (gpflow.train.GradientDescentOptimizer, the "SGD" branch
 * at conv_gp/experiment.py:100-103). */
int dcgp_model_sgd_step(dcgp_model* model, double lr);
/* One natural-gradient step of size gamma on every layer's (q_mu, q_sqrt) from the gradients of the last dcgp_elbo_grad:
 * gpflow.train.NatGradOptimizer(gamma) on var_list = [(l.q_mu, l.q_sqrt)] (conv_gp/experiment.py:90-99), natural-parameter
 * form theta <- theta + gamma dELBO/d eta.  Returns DCGP_ERR_NOT_PD (nothing written, *info = failing column) when a new
 * precision matrix is not positive definite: the reference's loop then scales gamma by 0.2 and retries (:36-49). */
int dcgp_model_natgrad_step(dcgp_model* model, double gamma, int* info_host);
/* param.set_trainable(False / True) (conv_gp/experiment.py:93-95, models.py:100): parameters switched off are left
 * alone by the Adam / SGD steps.  which = "Z", "q_mu", "q_sqrt", "w", or "hyper" (variance and lengthscale). */
int dcgp_model_set_trainable(dcgp_model* model, int layer, const char* which, int on);
/* current (constrained) value of a parameter, names as dcgp_model_get_grad; the inverse of dcgp_model_set_param -- what
 * sess.run(param.constrained_tensor) returns when the reference writes its checkpoint (conv_gp/experiment.py:56-64) */
int dcgp_model_get_param(dcgp_model* model, int layer, const char* which, double* out_host, size_t count);
/* DGP_Base.propagate(X, S) -> last layer's Fmean, Fvar [S*N, R] (device buffers owned by caller) */
int dcgp_model_propagate(dcgp_model* model, const double* X, int N, int S,
                         const double* const* z_per_layer_host, uint64_t seed,
                         double* out_fmean, double* out_fvar, int* info_host);
/* DGP_Base.predict_y(X, S) with the RobustMax likelihood, device end to end (doubly_stochastic_dgp
 * predict_y -> likelihood.predict_mean_and_var; caller at conv_gp/utils/log.py:62-66): out_p [S*N, K]
 * class probabilities per sample (the predictive variance is p - p^2), out_p_mean [N, K] their mean over
 * the S samples (what AccuracyLogger arg-maxes).  Either output may be NULL, not both.  Device buffers. */
int dcgp_model_predict_y(dcgp_model* model, const double* X, int N, int S,
                         const double* const* z_per_layer_host, uint64_t seed,
                         double* out_p, double* out_p_mean, int* info_host);
/* Parameter-only state across steps.  The reference's evaluation loops run hundreds of batches at ONE parameter state (AccuracyLogger /
 * LogLikelihoodLogger, conv_gp/utils/log.py:55-68; conv_gp/utils/tensorboard.py:22-42), and every session.run of them factors every Kuu again.
 * Here a step records the parameter version its chain (operand preparation, factorisations, inverses, G / alpha, KL pieces) ran at; every call that writes
 * a parameter -- dcgp_model_set_param, the Adam / SGD / natural-gradient steps -- starts a new version.  mode 0: never reuse; 1 (default):
 * dcgp_model_propagate and dcgp_model_predict_y skip the chain while the version stands; 2: the forward ELBO (dcgp_elbo_forward) as well -- for
 * evaluation sweeps; a TRAINING step's forward pass never reuses it, and bench.py's headline never runs in mode 2 (the reference's step recomputes).
 * Results are bit-identical to a step that runs the chain.  dcgp_model_chain_skips: steps that reused it so far. */
int dcgp_model_set_factor_reuse(dcgp_model* model, int mode);
int dcgp_model_chain_skips(dcgp_model* model, uint64_t* out);
/* Output of layer `layer` from the most recent forward: sample/mean/var [rows, D_l] device->device copy. */
int dcgp_model_layer_output(dcgp_model* model, int layer, double* out_sample, double* out_mean,
                            double* out_var, int* rows, int* width);

/* The strided fp64 GEMM of the training step, as an operator (every tf.matmul / tf.tensordot of the reverse pass, e.g.
 * the adjoints of conv_gp/conditionals.py:50,58, runs on it):
 *   C_b(i, j) (+)= alpha * colscale_b[j] * sum_k A_b(i, k) * kscale_b[k] * B_b(k, j),   b = 0 .. batch-1,
 *   A_b(i, k) = A[b a_bs + i a_rs + k a_cs],  B_b(k, j) = B[b b_bs + k b_rs + j b_cs],  C_b(i, j) = C[b c_bs + i c_rs + j];
 * colscale / kscale may be NULL; lower_only writes 0 above the diagonal of a square result.  Device pointers. */
int dcgp_gemm_strided(dcgp_ctx* ctx, const double* A, long a_rs, long a_cs, long a_bs, const double* B, long b_rs,
                      long b_cs, long b_bs, double* C, long c_rs, long c_bs, int M, int N, int K, int batch,
                      double alpha, int accumulate, const double* colscale, long cs_s, long cs_bs,
                      const double* kscale, long ks_s, long ks_bs, int lower_only);
/* The same product with the epilogues of the kernel adjoints (dZ = (E X - rowsum(E) o Z) / l^2 and its kin, Murray's Phi of the Cholesky
 * adjoint, symmetric products computed on the lower tiles only):
 *   C_b(i, j) (+)= alpha * (sum_k A_b(i, k) B_b(k, j) - sub_v[b sv_bs + i] * sub_x[b sx_bs + i sx_rs + j])      (sub_v, sub_x: both or neither)
 * flags bit 0: lower_only; bit 1 (with bit 0): mirror -- entries below the diagonal are also stored transposed, nothing else is written
 * above it; bit 2: phi -- the strictly lower part is kept, the diagonal halved, the rest written as zero. */
int dcgp_gemm_strided_ex(dcgp_ctx* ctx, const double* A, long a_rs, long a_cs, long a_bs, const double* B, long b_rs,
                         long b_cs, long b_bs, double* C, long c_rs, long c_bs, int M, int N, int K, int batch,
                         double alpha, int accumulate, const double* sub_v, long sv_bs, const double* sub_x, long sx_rs,
                         long sx_bs, int flags);

/* ---- initialisation ---------------------------------------------------------------------------------------------- */
/* Lloyd's k-means of n points [n, d] (device) into k centres [k, d] (device): the inducing-patch initialisation of
 * PatchInducingFeatures.from_images (conv_gp/kernels.py:147-164: sklearn KMeans(n_clusters=M, init='random')).
 * init_rows_host: k row indices, the 'random' initial centres (drawn by the caller); stops after max_iter iterations or
 * when the summed squared centre shift is <= tol.  Deterministic for given initial rows. */
int dcgp_kmeans(dcgp_ctx* ctx, const double* X, long n, int d, int k, const int32_t* init_rows_host, int max_iter,
                double tol, double* centers, int* iters_out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------ */
/* Side effect of the two set-up calls below: RCCL prints a version banner to stdout the first time; while the call runs the process's
 * file descriptor 1 points at stderr (so for that window other host threads' stdout lands there too).  The swap is serialised
 * process-wide. */
int dcgp_comm_unique_id(unsigned char* out_128bytes);
int dcgp_comm_init_rank(dcgp_ctx* ctx, int nranks, int rank, const unsigned char* id_128bytes);
int dcgp_comm_destroy(dcgp_ctx* ctx);
/* ranks RCCL itself reports for the ctx's communicator (ncclCommCount); 0 without one */
int dcgp_comm_count(dcgp_ctx* ctx, int* out_ranks);
int dcgp_allreduce_sum_f64(dcgp_ctx* ctx, double* buf_dev, int n);

/* ---- debugging aids: TEST-ONLY entry points (tests/, tools/); nothing of the reference's interface maps onto them and no product path calls them ---- */
/* With a communicator, a FORWARD step kept in flight (dcgp_elbo_forward_enqueue with no reverse pass behind it) hands the data term's all-reduce and
 * the ELBO assembly to a comm stream, so that the next kernels on the main stream do not queue behind the collective (a training step keeps its
 * collectives on the main stream: one communicator's collectives stay on one stream).  The gate lets a test
 * prove it: closed != 0 -- every such all-reduce enqueued from now on first waits (at most ~4 s) for the gate; 0 -- opens and removes it.
 * main_idle_out (may be NULL): bit 0 -- the ctx's main stream has drained (hipStreamQuery), bit 1 -- the comm stream has. */
int dcgp_debug_comm_gate(dcgp_ctx* ctx, int closed, int* main_idle_out);
/* Device buffer of 8 x 16 x 16 int64 into which the one-launch conv layer kernel (csrc/conv_fused.hip) stamps the shader
 * clock at its phase boundaries (8 sampled workgroups x 4 strips of a persistent one x 16 waves x 16 stamps = 8192 words); NULL switches it off (tools/fused_trace.py). */
int dcgp_debug_set_fused_trace(dcgp_ctx* ctx, long long* buf_dev);
/* The same for the patch sweeps (csrc/head_units.hip; tools/sweep_trace.py): [n_workgroups][waves per workgroup][8] int64 -- wall clock at entry,
 * shader clock at entry / image staged / set-up done / first unit done / last unit done, wall clock at exit, units run.            */
/* The ceilings bench.py prices kernels against, measured on this device (csrc/peaks.hip): the sustained fp64 MFMA rate (TFLOP/s, 4 waves
 * per SIMD, ~85 ms) and the rate of a pure store sweep writing an [M x N*P] matrix in the K_uf sweep's tile pattern (GB/s).            */
int dcgp_debug_mfma_f64_rate(dcgp_ctx* ctx, double* tflops_out);
int dcgp_debug_store_rate(dcgp_ctx* ctx, int N, int P, int M, double* gbs_out);
int dcgp_debug_set_sweep_trace(dcgp_ctx* ctx, long long* buf_dev, long n_workgroups, const char* family /* "kuf", "kuf_long", "head_sweep"; NULL: any */);

#ifdef __cplusplus
}
#endif
#endif /* DCGP_H */
