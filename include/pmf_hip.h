/*
 * pmf_hip.h -- C-ABI of libpmf_hip.so: the MI355X (gfx950) engine for the
 * latent-factor update loop of rogeliolopezcamara/prob-matrix-factorization.
 *
 * The reference has no FFI of its own: its hot path is the body of
 * `fit()` / `predict()` / `evaluate_rmse()` of the classes in src/models/ (pure NumPy).
 * This header is the boundary a replacement binds instead: every entry point
 * names the reference code it stands in for (paths relative to the reference
 * repository root).  The binding a maintainer adds on the reference side is a
 * ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types, caller-owned host buffers, 64-bit sizes, no torch types;
 *   - every function returns 0 on success and a negative PMF_E* code on
 *     failure; pmf_last_error() returns a thread-local message.  Nothing
 *     aborts or throws across the ABI;
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe;
 *     distinct contexts are independent;
 *   - host <-> device exchange of model state is always float64 (the
 *     reference's dtype); device storage is the
 *     context's dtype (PMF_F32 for throughput, PMF_F64 for parity runs).
 */
#ifndef PMF_HIP_H
#define PMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMF_ABI_VERSION 3

/* error codes */
#define PMF_OK 0
#define PMF_EINVAL (-1)   /* bad argument / state not set */
#define PMF_EHIP (-2)     /* a HIP runtime call failed (message has the hipError string) */
#define PMF_ENOMEM (-3)
#define PMF_ERANGE (-4)   /* id outside [0, n_rows) or unsupported n_factors */
#define PMF_ECOMM (-5)    /* a collective / communicator call failed (RCCL error string in the message) */

/* device storage / arithmetic type of a context */
#define PMF_F32 0
#define PMF_F64 1

/* which block of the bipartite model a call addresses */
#define PMF_SIDE_USER 0
#define PMF_SIDE_ITEM 1

/* Model-state arrays (second argument of pmf_set_array / pmf_get_array).
 * Reference attribute each one mirrors:
 *   FACTOR   E_theta / E_beta        (poisson_mf_cavi.py:41-42, hpf_cavi.py:54-55)
 *            m_theta / m_beta        (gaussian_mf_cavi_bias.py:33,35)        [rows x K]
 *   SHAPE    a_theta / a_beta, gamma_a_theta / gamma_a_beta                  [rows x K]
 *   RATE     b_theta / b_beta, gamma_b_theta / gamma_b_beta                  [rows x K]
 *   PRIOR_RATE  E_xi / E_eta         (hpf_cavi.py:56-57)                     [rows]
 *   HYPER_RATE  gamma_b_xi / gamma_b_eta (hpf_cavi.py:46,50)                 [rows]
 *   COV      V_theta / V_beta        (gaussian_mf_cavi_bias.py:34,36)        [rows x K x K] on the
 *            host side; packed lower-triangular K(K+1)/2 per row on the device
 *   BIAS     m_user_bias / m_item_bias (gaussian_mf_cavi_bias.py:39-40)      [rows]
 *   SCALE / SCALE_SHAPE / SCALE_RATE   E_phi, a_phi, b_phi / E_psi, a_psi, b_psi of the extended
 *            Poisson model (poisson_mf_extended_cavi.py:34-45)                [rows]
 */
#define PMF_ARR_FACTOR 0
#define PMF_ARR_SHAPE 1
#define PMF_ARR_RATE 2
#define PMF_ARR_PRIOR_RATE 3
#define PMF_ARR_HYPER_RATE 4
#define PMF_ARR_COV 5
#define PMF_ARR_BIAS 6
#define PMF_ARR_SCALE 7
#define PMF_ARR_SCALE_SHAPE 8
#define PMF_ARR_SCALE_RATE 9
#define PMF_ARR_COUNT 10

/* kernel classes for pmf_prof_get (live hipEvent timing on the context's stream) */
#define PMF_KERNEL_GAMMA_SWEEP 0   /* Poisson/HPF half-sweep accumulate(+finalise) */
#define PMF_KERNEL_GAMMA_FINAL 1   /* split-row / distributed finalise */
#define PMF_KERNEL_GAUSS_ACCUM 2   /* Gaussian normal-equation accumulate */
#define PMF_KERNEL_GAUSS_SOLVE 3   /* K x K SPD inverse + mean */
#define PMF_KERNEL_GAUSS_BIAS 4    /* Gaussian bias half-sweep */
#define PMF_KERNEL_EVAL 5          /* fused predict + error reduction */
#define PMF_KERNEL_PREDICT 6
#define PMF_KERNEL_TOPK 7
#define PMF_KERNEL_GAUSS_COMBINE 8 /* split-row partial sums -> row sums */
#define PMF_KERNEL_GAUSS_SGD 9     /* MAP gradient half-sweep (no reference counterpart) */
#define PMF_KERNEL_COMM_ALLREDUCE 10 /* item-statistics all-reduce, timed on the collective stream */
#define PMF_KERNEL_COMM_WAIT 11    /* compute stream idle until a chunk's all-reduce has landed = exposed communication */
#define PMF_KERNEL_COUNT 12

typedef struct pmf_ctx pmf_ctx;

/* ---- library ----------------------------------------------------------- */
int pmf_abi_version(void);
const char *pmf_last_error(void);
/* number of visible HIP devices (0 and PMF_EHIP when there is no usable GPU) */
int pmf_device_count(int *count);

/* ---- context ------------------------------------------------------------
 * One context = one model instance's device state on one GPU.
 * n_users / n_items follow `_infer_dimensions` (hpf_cavi.py:60-64): max id + 1
 * of the TRAINING ratings.  In a multi-GPU run n_users is the size of this
 * rank's user range (ids are local to it) and n_items is global. */
int pmf_ctx_create(int device, int64_t n_users, int64_t n_items, int n_factors, int dtype,
                   pmf_ctx **out);
int pmf_ctx_destroy(pmf_ctx *ctx);
/* run on an existing HIP stream (e.g. torch's current stream) instead of the
 * context's own; `hip_stream` is a hipStream_t.  NULL restores the own stream -- so the
 * null handle of the legacy default stream cannot be selected: order with torch / RCCL
 * through a non-default stream (pmf_hip/dist.py:StreamScope). */
int pmf_ctx_set_stream(pmf_ctx *ctx, void *hip_stream);
int pmf_ctx_sync(pmf_ctx *ctx);

/* Row chunks (no reference counterpart; multi-GPU pipelining).  `n_chunks` equal row ranges of
 * one side; the *_accumulate / *_finalize calls of that side then act on the chunk chosen with
 * pmf_ctx_select_chunk (-1 = all rows, the default) -- they read and write only rows
 * [row_begin, row_end) of the statistics buffer, at the same offsets as in an unchunked call,
 * so a caller can all-reduce the slice of chunk c while chunk c+1 is being accumulated.
 * The fused *_sweep calls always cover all rows.  Results do not depend on the chunking. */
int pmf_ctx_set_row_chunks(pmf_ctx *ctx, int side, int n_chunks);
int pmf_ctx_chunk_rows(pmf_ctx *ctx, int side, int chunk, int64_t *row_begin, int64_t *row_end);
int pmf_ctx_select_chunk(pmf_ctx *ctx, int side, int chunk);
/* bytes of device memory currently held by the context */
int pmf_ctx_device_bytes(pmf_ctx *ctx, int64_t *bytes);

/* Training ratings in their original order (COO).  Replaces
 * `_build_index_lists` (hpf_cavi.py:97-107, gaussian_mf_cavi_bias.py:69-86):
 * builds the by-user (CSR) and by-item (CSC) orderings with a stable counting
 * sort, so every row lists its ratings in ascending original position and
 * duplicate (u,i) pairs are kept.  Also builds the per-side work lists
 * (row chunks) the sweep kernels consume. */
int pmf_ctx_set_ratings(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids,
                        const int32_t *item_ids, const double *ratings);

/* Host float64 -> device (and back).  `host` holds rows x K (FACTOR, SHAPE,
 * RATE), rows (PRIOR_RATE, HYPER_RATE, BIAS) or rows x K x K (COV) doubles,
 * C-contiguous.  Setting an array allocates it. */
int pmf_set_array(pmf_ctx *ctx, int side, int array, const double *host);
int pmf_get_array(pmf_ctx *ctx, int side, int array, double *host);
/* Row subsets of the same arrays: `rows[n]` are row ids of `side` (any order, repeats allowed when
 * reading; when writing, a repeated row ends with one of its values); `host` holds n rows in the
 * layout above (n x K, n, or n x K x K doubles).  What the reference's row indexing does
 * (`V_beta[j_idx]`, `m_beta[j_idx]`, `V_theta[i] = ...`: gaussian_mf_cavi_bias.py:146-162) without
 * moving the whole stack: V_theta at 1M x 64 x 64 is 32.8 GB as host float64, 164 GB at the K = 128
 * shard of config C4.  A row id outside [0, rows) is PMF_ERANGE and nothing is copied. */
int pmf_get_array_rows(pmf_ctx *ctx, int side, int array, int64_t n, const int64_t *rows, double *host);
int pmf_set_array_rows(pmf_ctx *ctx, int side, int array, int64_t n, const int64_t *rows, const double *host);
/* COV shortcut for `_initialize_variational_params`
 * (gaussian_mf_cavi_bias.py:64-67): every row's covariance = scale * I. */
int pmf_set_cov_identity(pmf_ctx *ctx, int side, double scale);

/* ---- Poisson MF / HPF half-sweep ---------------------------------------
 * One block-Jacobi half-sweep over all rows of `side`
 * (poisson_mf_cavi.py:135-167 users, :173-197 items;
 *  hpf_cavi.py:126-153 users, :162-187 items):
 *     rate_j    = max(FACTOR_other[o_j] . FACTOR_side[r], 1e-10)
 *     SHAPE[r]  = shape_prior + sum_j x_j * FACTOR_other[o_j] * FACTOR_side[r] / rate_j
 *     RATE[r]   = rate_prior_r + sum_j FACTOR_other[o_j]
 *     FACTOR[r] = SHAPE[r] / RATE[r]
 * rate_prior_r is `rate_prior` when hierarchical == 0 (Poisson MF, b0) and
 * PRIOR_RATE[r] (E_xi / E_eta) when hierarchical != 0 (HPF).  In the
 * hierarchical case the epilogue also performs the xi / eta update of
 * hpf_cavi.py:155-159 / :189-193:
 *     HYPER_RATE[r] = hyper_rate_prior + sum_k FACTOR[r,k]
 *     PRIOR_RATE[r] = hyper_shape / HYPER_RATE[r]
 * Rows without ratings fall back to the priors (hpf_cavi.py:128-132). */
int pmf_gamma_sweep(pmf_ctx *ctx, int side, double shape_prior, double rate_prior,
                    int hierarchical, double hyper_shape, double hyper_rate_prior);

/* Extended Poisson MF half-sweep, x ~ Poisson(phi_u psi_i theta_u.beta_i)
 * (poisson_mf_extended_cavi.py:108-160 users, :163-215 items).  With s = SCALE of
 * the other side:
 *     SHAPE[r]  = a0 + sum_j x_j FACTOR_other[o_j] FACTOR_side[r] / (FACTOR_other[o_j].FACTOR_side[r])
 *     RATE[r]   = b0 + sum_j s[o_j] FACTOR_other[o_j];        FACTOR[r] = SHAPE[r] / RATE[r]
 *     SCALE_SHAPE[r] = a0 + sum_j x_j;
 *     SCALE_RATE[r]  = b0 + sum_j s[o_j] (FACTOR_other[o_j] . FACTOR[r])   -- with the NEW FACTOR[r]
 *     SCALE[r]       = SCALE_SHAPE[r] / SCALE_RATE[r]
 * The rate is not clamped (the reference divides by the raw dot product).  Rows
 * without ratings get the priors in SHAPE / RATE / SCALE_SHAPE / SCALE_RATE and keep
 * FACTOR and SCALE (the reference `continue`s before recomputing them). */
int pmf_gamma_ext_sweep(pmf_ctx *ctx, int side, double shape_prior, double rate_prior);

/* Multi-GPU form of the same half-sweep (ratings sharded by user range,
 * SURVEY.md section 8e).  `accumulate` writes this rank's raw sums
 * [rows x 2 x Kpad] (shape sums, then rate sums; Kpad from pmf_ctx_kpad) into
 * `stats_dev`, a DEVICE buffer the caller owns and all-reduces (RCCL) between
 * the two calls; `finalize` applies the priors and the epilogue above. */
int pmf_ctx_kpad(pmf_ctx *ctx, int *kpad);
int pmf_gamma_accumulate(pmf_ctx *ctx, int side, void *stats_dev);
int pmf_gamma_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double shape_prior,
                       double rate_prior, int hierarchical, double hyper_shape,
                       double hyper_rate_prior);

/* ---- Gaussian MF half-sweeps -------------------------------------------
 * Factor half-sweep (gaussian_mf_cavi_bias.py:132-165 users, :170-201 items;
 * bias-free twin gaussian_mf_cavi.py:121-147 / :152-178 when no BIAS array is
 * set): for every row with at least one rating
 *     S      = sum_j ( COV_other[o_j] + FACTOR_other[o_j] FACTOR_other[o_j]^T )
 *     COV[r] = inv( I/eta2 + S/sigma2 )
 *     FACTOR[r] = (1/sigma2) COV[r] . sum_j FACTOR_other[o_j] (x_j - BIAS_side[r] - BIAS_other[o_j])
 * Rows without ratings keep their mean and covariance.  n_factors <= 256 (PMF_ERANGE above; the reference
 * has no limit, its grids stop at 70); the MFMA kernels cover K <= 128 in fp32, K > 128 and fp64 run generic ones. */
int pmf_gauss_factor_sweep(pmf_ctx *ctx, int side, double sigma2, double eta2);
/* Bias half-sweep (gaussian_mf_cavi_bias.py:206-232 users, :237-263 items):
 *     BIAS[r] = var/sigma2 * sum_j (x_j - BIAS_other[o_j] - FACTOR_other[o_j].FACTOR_side[r]),
 *     var = 1 / (1/eta_bias2 + n_r/sigma2);   rows without ratings keep their value. */
int pmf_gauss_bias_sweep(pmf_ctx *ctx, int side, double sigma2, double eta_bias2);

/* Multi-GPU forms: raw per-row sums into / from a caller-owned DEVICE buffer.
 * Factor: [rows x (Kp + Kpad)] = packed lower triangle of S, then the
 * right-hand side (Kp = K(K+1)/2 rounded up to a multiple of 4, see
 * pmf_ctx_cov_stride).  Bias: [rows x 2] = residual sum, rating count. */
int pmf_ctx_cov_stride(pmf_ctx *ctx, int *stride);
int pmf_gauss_factor_accumulate(pmf_ctx *ctx, int side, void *stats_dev);
int pmf_gauss_factor_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double sigma2,
                              double eta2);
int pmf_gauss_bias_accumulate(pmf_ctx *ctx, int side, void *stats_dev);
int pmf_gauss_bias_finalize(pmf_ctx *ctx, int side, const void *stats_dev, double sigma2,
                            double eta_bias2);

/* ---- multi-GPU: RCCL inside the library (SURVEY.md section 8(b) `pmf_comm_init`, section 8(e)) ------
 * No reference counterpart (the reference is a single process).  One process per GPU; ratings are
 * sharded by USER RANGE (the context's n_users is this rank's range, ids local to it), the item block
 * is replicated.  Once a context has a communicator (of any size), every ITEM-side half-sweep
 * (pmf_gamma_sweep, pmf_gauss_factor_sweep, pmf_gauss_bias_sweep, pmf_gauss_sgd_sweep with
 * side = PMF_SIDE_ITEM) runs inside the library as
 *     accumulate raw per-item sums over this rank's ratings  ->  exchange (sum over ranks)  ->  finalize
 * on a library-owned statistics buffer, pipelined over the item row chunks of pmf_ctx_set_row_chunks:
 * the all-reduce of chunk c runs on a second, high-priority HIP stream and its finalisation on a third
 * (ordered against the compute stream by events, never by the host) while chunk c+1 is accumulated.  Per-row arithmetic is that of
 * the accumulate / finalize pair, so results do not depend on the chunking, and every rank ends the
 * half-sweep with bit-identical item state.  USER-side half-sweeps stay local.  The iteration order a
 * caller issues is unchanged (gaussian_mf_cavi_bias.py:129-263, hpf_cavi.py:121-193).
 * pmf_gamma_ext_sweep is not available with a communicator.
 *
 * pmf_comm_unique_id: 128 bytes from ncclGetUniqueId; rank 0 creates them and hands them to the other
 * ranks out of band (pmf_hip/dist.py: a file next to the launcher's rendezvous).  pmf_comm_init: RCCL
 * communicator over xGMI, collective (every rank must call it).  pmf_comm_attach lets a
 * second context of the same process and device share `owner`'s communicator; pmf_comm_destroy
 * detaches (the communicator goes with its last user; pmf_ctx_destroy detaches too).
 *
 * Exchange of a chunk's statistics (pmf_comm_set_exchange; the same mode on every rank):
 *   PMF_EXCHANGE_ALLREDUCE       ncclAllReduce, then every rank finalises every row;
 *   PMF_EXCHANGE_SCATTER_GATHER  ncclReduceScatter -> each rank finalises its 1/nranks of the chunk's rows ->
 *                                ncclAllGather of the finalised state (the reference solves each row once,
 *                                gaussian_mf_cavi_bias.py:170-201: so does the job as a whole, instead of
 *                                nranks times).  Same wire bytes for the Gaussian factor sweep, whose state
 *                                is as wide as its statistics; replicas stay bit-identical either way;
 *   PMF_EXCHANGE_AUTO (default)  SCATTER_GATHER for pmf_gauss_factor_sweep on more than one rank (finalize =
 *                                a K x K solve per row), ALLREDUCE for the sweeps whose finalize is
 *                                element-wise.  Environment PMF_COMM_EXCHANGE=allreduce|scatter_gather
 *                                sets the default of new contexts. */
#define PMF_UNIQUE_ID_BYTES 128
#define PMF_TRANSPORT_RCCL 0
#define PMF_OP_SUM 0
#define PMF_OP_MAX 1
#define PMF_EXCHANGE_AUTO 0
#define PMF_EXCHANGE_ALLREDUCE 1
#define PMF_EXCHANGE_SCATTER_GATHER 2
int pmf_comm_unique_id(void *id_out);
int pmf_comm_init(pmf_ctx *ctx, int nranks, int rank, const void *unique_id);
int pmf_comm_set_exchange(pmf_ctx *ctx, int mode);
#ifdef PMF_TEST_TRANSPORT
/* TEST BUILDS ONLY (libpmf_hip_test.so, -DPMF_TEST_TRANSPORT; the product library does not export it): the same
 * interface over POSIX shared memory for ranks that share ONE GPU -- a rehearsal transport for one-GPU
 * boxes (RCCL refuses two ranks per device). */
#define PMF_TRANSPORT_HOSTSHM 1
int pmf_comm_init_hostshm(pmf_ctx *ctx, int nranks, int rank, const void *unique_id);
#endif
int pmf_comm_attach(pmf_ctx *ctx, pmf_ctx *owner);
int pmf_comm_destroy(pmf_ctx *ctx);
int pmf_comm_info(pmf_ctx *ctx, int *nranks, int *rank, int *transport);
/* Waiting under a communicator: pmf_ctx_sync (of a context with a communicator), pmf_comm_barrier,
 * pmf_comm_allreduce_host and pmf_comm_gather_user_rows poll the stream, the communicator's asynchronous error
 * state and a deadline (environment PMF_COMM_TIMEOUT_S, seconds, default 1800, 0 = none).  When a peer died or
 * never reached its collective they abort the communicator and return PMF_ECOMM; every later collective call on
 * it returns PMF_ECOMM at once. */
/* all queued work of every rank (kernels and collectives) has finished when this returns */
int pmf_comm_barrier(pmf_ctx *ctx);
/* element-wise sum / max of n host doubles over the ranks, result on every rank (validation sums of
 * the sharded monitor -- hpf_cavi.py:196-211 --, timings) */
int pmf_comm_allreduce_host(pmf_ctx *ctx, double *values, int64_t n, int op);
/* the USER-side rows of `array` of every rank, concatenated in rank order, as host float64 on every
 * rank; bounds[nranks + 1] = the global user ranges (what `fit` needs to hand back full
 * E_theta / m_theta, train_poisson_full.py:68-76).  Collective. */
int pmf_comm_gather_user_rows(pmf_ctx *ctx, int array, const int64_t *bounds, double *host_full);

/* ---- Gaussian MF, MAP by stochastic gradient steps -------------
 * NO reference counterpart (the reference's Gaussian model is CAVI only): parity unpinned, the
 * oracle is this build's own restatement (oracle/cavi_oracle.py:gauss_sgd_half_sweep).  One call
 * walks every row of `side` through its ratings in input order with the other side fixed:
 *   e = x - b_r - b_o - f_r . f_o;  f_r += lr (e f_o / sigma2 - f_r / (eta2 n_r));
 *   b_r += lr (e / sigma2 - b_r / (eta_bias2 n_r))        (biases only if both BIAS arrays are set)
 * Rows longer than 256 ratings are cut into pieces that start from the row's old value; the row
 * moves by the rating-count-weighted mean of the pieces' displacements.  The accumulate / finalize
 * pair exposes that sum -- [rows x width] = sum len * d_f | sum len * d_b | sum len | 0 0, width from
 * pmf_ctx_sgd_stats_width -- so that ranks holding different ratings of an item can all-reduce it
 * (honours pmf_ctx_select_chunk like the other accumulate / finalize calls). */
int pmf_gauss_sgd_sweep(pmf_ctx *ctx, int side, double lr, double sigma2, double eta2, double eta_bias2);
int pmf_ctx_sgd_stats_width(pmf_ctx *ctx, int *width);
int pmf_gauss_sgd_accumulate(pmf_ctx *ctx, int side, void *stats_dev, double lr, double sigma2, double eta2,
                             double eta_bias2);
int pmf_gauss_sgd_finalize(pmf_ctx *ctx, int side, const void *stats_dev);

/* ---- predict / evaluate -------------------------------------------------
 * `predict` (hpf_cavi.py:215-231, poisson_mf_cavi.py:221-241,
 * gaussian_mf_cavi_bias.py:291-316): out[n] = FACTOR_user[u].FACTOR_item[i]
 * (+ BIAS_user[u] + BIAS_item[i] when bit 0 of `use_bias` is set; multiplied by
 * SCALE_user[u] SCALE_item[i] when bit 1 is set -- the extended Poisson model's
 * predict, poisson_mf_extended_cavi.py:239-259) for ids inside the trained
 * dimensions, 0 otherwise; `offset` (global_mean) is added to every row. */
#define PMF_PREDICT_BIAS 1
#define PMF_PREDICT_SCALE 2
int pmf_predict(pmf_ctx *ctx, int64_t n, const int32_t *user_ids, const int32_t *item_ids,
                int use_bias, double offset, double *out);

/* Validation set kept on the device for the per-iteration monitor of `fit`
 * (hpf_cavi.py:196-211, gaussian_mf_cavi_bias.py:268-284).  `label_index[n]`
 * maps each true rating to its position in np.unique(y_true) (n_labels <= 32)
 * for the per-label error sums of metrics.macro_mae (metrics.py:37-51). */
int pmf_eval_set(pmf_ctx *ctx, int64_t n, const int32_t *user_ids, const int32_t *item_ids,
                 const double *y_true, const int32_t *label_index, int n_labels);
/* Fused predict + reduction over the stored validation set: sum of squared
 * errors, per-label sum of |error| and per-label counts.  The host finishes
 * rmse = sqrt(sse / n) (metrics.py:6-10) and macro_mae = mean_l(abs_l / cnt_l). */
int pmf_eval_run(pmf_ctx *ctx, int use_bias, double offset, double *sum_sq_err,
                 double *abs_err_per_label, int64_t *count_per_label);

/* Top-k items per user from the dense reconstruction FACTOR_user . FACTOR_item^T
 * (the "identical top-k item rankings" check of the north star; the reference
 * has no ranking API -- scores follow `predict`: `use_bias` is predict's flag, 0, PMF_PREDICT_BIAS
 * (+ BIAS_user[u] + BIAS_item[i]) or PMF_PREDICT_SCALE (x SCALE_user[u] SCALE_item[i], the extended
 * Poisson model); ties are broken by the lower item id, NaN scores never rank).
 * out_items / out_scores hold n_query x k entries (-1 / 0 when fewer than k items rank).
 * fp32 contexts with Kpad <= 128 and k <= 64 run one fused kernel: score tiles on the matrix cores
 * (v_mfma_f32_32x32x2_f32), the running k best per user in LDS, no score matrix in HBM. */
int pmf_topk_items(pmf_ctx *ctx, int64_t n_query, const int32_t *user_ids, int k, int use_bias,
                   int32_t *out_items, double *out_scores);

/* ---- profiling ----------------------------------------------------------
 * When enabled every kernel launch is bracketed by hipEvents on the context's
 * stream; pmf_prof_get synchronises and returns the accumulated device time
 * and launch count of one kernel class. */
int pmf_prof_enable(pmf_ctx *ctx, int enable);
int pmf_prof_reset(pmf_ctx *ctx);
int pmf_prof_get(pmf_ctx *ctx, int kernel, double *total_ms, int64_t *launches);
/* Gather ceiling of the Poisson/HPF half-sweep of `side` on THIS context's ratings and tables: the
 * average device time of `repeats` launches of the sweep kernel's memory side alone (same tasks, same
 * index / rating streams, same 16-byte-per-lane row gathers; one add per loaded value, no row output).
 * The gathered table of C3 (25.6 MB of item rows, 256 MB of user rows) lives in L2 / Infinity Cache,
 * so the sweep is bound by what the caches deliver for this pattern, not by HBM: algorithmic bytes /
 * this time is the kernel's roofline (bench.py, `roofline.bound = "cache_gather"`). */
int pmf_prof_gather_ceiling(pmf_ctx *ctx, int side, int repeats, double *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* PMF_HIP_H */
