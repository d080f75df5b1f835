/*
 * causalimpact_amd.h -- C-ABI of the MI355X-native CausalImpact hot path.
 *
 * The reference (google/tfp-causalimpact) has no FFI: its hot path is the
 * Python call
 *     _train_causalimpact_sts(...)            causalimpact/causalimpact_lib.py:503-606
 *       -> _run_gibbs_sampler(...)            causalimpact/causalimpact_lib.py:345-395
 *            gibbs_sampler.fit_with_gibbs_sampling(...)          :365-388
 *            _get_posterior_means_and_trajectories(...)          :609-632
 * into TensorFlow-Probability.  This header is the boundary a maintainer would
 * bind instead (ctypes stub: INTEGRATION.md).  Plain pointers and sizes only;
 * the caller owns every host buffer; the library keeps no caller pointers after
 * a call returns.  All entry points return 0 on success, non-zero on error;
 * ci_last_error() then describes the failure (thread-local string).
 *
 * Shapes use the reference's names:  T = time steps handed to the sampler
 * (pre-period + everything after it, causalimpact_lib.py:548-562), P = design
 * columns incl. the trailing intercept (data.py:135), K = seasonal blocks,
 * W/S = warm-up / retained Gibbs iterations, C = chains, B = independent series.
 */
#ifndef CAUSALIMPACT_AMD_H_
#define CAUSALIMPACT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CI_ABI_VERSION 3
#define CI_MAX_BLOCKS 8

/* Per-series priors and initial Gibbs state.  One per series because every
 * constant scales with that series' outcome_sd.
 * Replaces the prior objects built in _build_default_gibbs_model
 * (causalimpact_lib.py:424-489) and the initial GibbsSamplerState
 * (causalimpact_lib.py:566-581, :370-383). */
typedef struct ci_series_params {
  double level_conc, level_scale, level_ub;   /* IG(sigma^2_level); clip on sigma   :424-432 */
  double slope_conc, slope_scale, slope_ub;   /* IG(sigma^2_slope) (LocalLinearTrend only) */
  double obs_conc, obs_scale, obs_ub;         /* IG(sigma^2_obs); upper bound        :434-443 */
  double drift_conc, drift_scale, drift_ub;   /* IG(sigma^2_drift), all blocks       :472-474 */
  double nonzero_prob;                        /* min(1, 3/P)                         :449-450 */
  double init_level_loc, init_level_scale;    /* N(y[0], sd)                         :467-469 */
  double init_slope_scale;                    /* N(0, .)  (LocalLinearTrend only) */
  double init_seasonal_scale;                 /* N(0, sd)                            :489     */
  double obs_scale0, level_scale0, slope_scale0;  /* initial state                   :566-572 */
  double drift_scale0[CI_MAX_BLOCKS];             /*                                 :573-574 */
  /* Multiplier of the weights-prior precision Omega = 0.01 (X'X/2 + diag(X'X)/2) / T (:451-453,
   * built on the device from X); 1 = the reference's prior.  The host package conditions a raw-
   * scale outcome as y -> (y - mu) / s before the float32 kernels see it and passes s^2 here:
   * Omega is then in the conditioned units and the chain is the reference's chain, mapped
   * (causalimpact_lib._internal_conditioning).  Must be positive. */
  double weights_prior_scale;
} ci_series_params;

/* The sampler's static configuration == the non-tensor arguments of
 * _run_gibbs_sampler (causalimpact_lib.py:346-353) plus the batch extensions
 * BASELINE.json asks for (chains, series). */
typedef struct ci_problem {
  int32_t abi_version;          /* CI_ABI_VERSION */
  int32_t T, P;
  int32_t has_slope;            /* 0 LocalLevel (reference default), 1 LocalLinearTrend */
  int32_t num_blocks;           /* K seasonal blocks */
  int32_t num_seasons[CI_MAX_BLOCKS];
  int32_t num_warmup;           /* W   InferenceOptions.num_warmup_steps :215-220 */
  int32_t num_results;          /* S   InferenceOptions.num_results */
  int32_t num_chains;           /* C chains run by THIS call (per series) */
  int32_t chain_offset;         /* global id of this call's first chain: results do not
                                   depend on how chains are split over devices */
  int32_t num_series;           /* B */
  uint32_t seed[2];             /* sanitized seed pair (int s -> (0,s)) :535-543 */
  int32_t device;               /* HIP device ordinal */
  int32_t flags;                /* CI_FLAG_* (0 = let the library choose the kernel) */
  int32_t series_offset;        /* global id of this call's first series: series b draws from the
                                   random streams of series id series_offset + b, so a batch split
                                   over devices gives the same draws as one launch, and the
                                   Monte-Carlo errors of different series are independent.  Series
                                   id 0 (any single-series fit) uses the plain chain streams. */
  int32_t reserved;             /* 0 */
} ci_problem;

/* Seasonal models (and any model with more than 52 design columns): use the one-wavefront-per-chain
 * sequential kernel even where a time-parallel kernel applies.  The route of a fit is otherwise a
 * function of the model and the series alone -- never of the launch size or the device -- so this
 * flag must be given to a batch and to the single-series fit it is compared with alike.  Higher
 * throughput for batches of hundreds of short multi-block series; test / diagnostic knob. */
#define CI_FLAG_SEQUENTIAL_SEASONAL 1
/* Batches: every series consumes the random streams of series 0 (series b of the batch then
 * reproduces a single-series fit of series b draw for draw; Monte-Carlo errors are perfectly
 * correlated across the batch).  Off by default. */
#define CI_FLAG_SHARED_SERIES_STREAMS 2
/* Use the four-wavefront Gibbs kernel where the eight-wavefront latency build (four time waves,
 * a regression wave that sweeps the next iteration's matrix during the Durbin-Koopman draw, three
 * randomness waves) would be chosen.  Same sampler, random stream and roundings: the two builds
 * give bit-identical draws, so this only changes the timing; test / diagnostic knob. */
#define CI_FLAG_FOUR_WAVES 4
/* Sequential seasonal kernel: keep its arrays over time in the per-chain HBM workspace even when
 * they would fit in LDS (the library does so by itself for long series / many covariates).
 * Test / diagnostic knob. */
#define CI_FLAG_SEASONAL_WORKSPACE 8
/* Time-parallel kernels: one workgroup per chain even where the library would spread a chain over a
 * cluster of 2, 4, 8 or 16 CUs (csrc/ci_wide.h: the Durbin-Koopman draw on up to eight of them, the
 * streaming phases on all; csrc/ci_seasonal_tp.h: up to 32) because the launch leaves CUs idle.
 * Every cluster size gives the same bits; test / diagnostic knob. */
#define CI_FLAG_NO_CLUSTER 16
/* Test knob: the last workgroup of every cluster exits at once, as if it had never been
 * scheduled -- the others must notice (time-out while assembling the cluster) and the chain's
 * main workgroup must carry on alone with unchanged results. */
#define CI_FLAG_TEST_DROP_HELPER 32
/* Seasonal models: take the wave-cooperative time-parallel kernel (csrc/ci_seasonal_tp.h: chunks of
 * the series on the wavefronts of a cluster of up to 32 workgroups of 4 wavefronts, any block list with a state of at
 * most 32 components) even for "trend + one block of 2-7 seasons" and for trend models with 53+
 * design columns, which by default run on the quad-split kernel of csrc/ci_wide.h (the faster one
 * since round 6: profiles/).  Test / diagnostic knob. */
#define CI_FLAG_CLUSTER_SEASONAL 64

/* Caller-allocated result buffers (float32, chain-major so per-device shards
 * are contiguous).  == GibbsSamplerState stack + (means, trajectories) returned
 * at causalimpact_lib.py:395.  Any pointer may be NULL to skip that output. */
typedef struct ci_outputs {
  float* observation_noise_scale;  /* [B,C,S]   */
  float* level_scale;              /* [B,C,S]   */
  float* slope_scale;              /* [B,C,S]   */
  float* seasonal_drift_scales;    /* [B,C,S,K] */
  float* weights;                  /* [B,C,S,P] */
  float* level;                    /* [B,C,S,T] */
  float* slope;                    /* [B,C,S,T] */
  float* seasonal_levels;          /* [B,C,S,T,K]  0-th latent of each block (:312-317) */
  float* posterior_means;          /* [B,C,T]   per-chain mean of the noise-free predictor (:627) */
  float* posterior_trajectories;   /* [B,C,S,T] (:629-631) */
} ci_outputs;

typedef struct ci_session ci_session;  /* device-resident fit: inputs + outputs live in HBM */

const char* ci_last_error(void);
int ci_abi_version(void);
int ci_device_count(int* count);
/* The Philox key of series `series_id` of a fit with `seed`: key = (seed[0] ^ h1(id), seed[1] ^
 * h2(id)), h1 / h2 bijective mixers that fix 0 -- series 0 (every single-series fit) keeps the
 * plain seeds, and batches under different seeds never share a stream.  A single-series fit with
 * seed = key reproduces series `series_id` of the batch (what the parity tests feed the oracle). */
void ci_series_stream_key(const uint32_t seed[2], int32_t series_id, uint32_t key[2]);
/* Waits for all work queued on `device` (bench.py brackets its timed region with it). */
int ci_device_synchronize(int device);
/* Device buffers of finished sessions are parked in a per-process pool (<= 32 GiB of the 288) for reuse by
 * the next fit, and so are their streams and events (creating and destroying a stream costs about
 * a millisecond on this runtime); this returns all of them to the driver. */
int ci_pool_trim(void);
/* Pinned (page-locked) host memory for result buffers; recycled through a pool like the device
 * buffers.  ci_host_free accepts only pointers returned by ci_host_alloc. */
int ci_host_alloc(void** ptr, size_t bytes);
int ci_host_free(void* ptr);

/* One-shot: upload, run all W+S Gibbs iterations for B*C chains, download.
 *   y     [B,T]    float32 outcome (standardised); value ignored where mask != 0
 *   mask  [B,T]    uint8, 1 = missing (NaN pre-period steps + the whole after-pre period)
 *   X     [B,T,P]  float32 row-major design, intercept last; NULL when P == 0
 *   season_change [K,T] uint8, 1 where step t is the last step of a season of block k
 *   params[B]
 * Replaces _run_gibbs_sampler (causalimpact_lib.py:345-395). */
int ci_fit_gibbs(const ci_problem* problem, const float* y, const uint8_t* mask, const float* X,
                 const uint8_t* season_change, const ci_series_params* params,
                 ci_outputs* outputs);

/* The same fit computed in FLOAT64 throughout (DataOptions.dtype = float64: the reference runs its
 * sampler in the requested dtype, causalimpact_lib.py:159; its numeric pin covers float32 and
 * float64, causalimpact_lib_test.py:655-662).  y, X and every result array are float64; any model
 * the float32 entry point takes (seasonal blocks, P up to 512, any T).  One wavefront per chain
 * (csrc/ci_gibbs64.h): trend models time-parallel over its 64 lanes, models with seasonal blocks
 * sequential in time -- the precision option, not the fast one.  Same random stream as
 * ci_fit_gibbs: the two agree draw for draw to float32 round-off. */
typedef struct ci_outputs_f64 {
  double* observation_noise_scale;  /* [B,C,S]   */
  double* level_scale;              /* [B,C,S]   */
  double* slope_scale;              /* [B,C,S]   */
  double* seasonal_drift_scales;    /* [B,C,S,K] */
  double* weights;                  /* [B,C,S,P] */
  double* level;                    /* [B,C,S,T] */
  double* slope;                    /* [B,C,S,T] */
  double* seasonal_levels;          /* [B,C,S,T,K] */
  double* posterior_means;          /* [B,C,T]   */
  double* posterior_trajectories;   /* [B,C,S,T] */
} ci_outputs_f64;
int ci_fit_gibbs_f64(const ci_problem* problem, const double* y, const uint8_t* mask,
                     const double* X, const uint8_t* season_change,
                     const ci_series_params* params, ci_outputs_f64* outputs);
/* Duration (HIP events) of the sampling kernel of the calling thread's last successful
 * ci_fit_gibbs_f64 -- the call itself also allocates, uploads and downloads. */
int ci_fit_gibbs_f64_kernel_ms(float* kernel_ms);

/* Device-resident variant (what bench.py times: inputs already in HBM). */
int ci_session_create(const ci_problem* problem, const float* y, const uint8_t* mask,
                      const float* X, const uint8_t* season_change,
                      const ci_series_params* params, ci_session** session);
/* Runs the fit on the session's stream and waits for it.  kernel_ms (optional)
 * receives the Gibbs kernel's duration measured with HIP events on that stream. */
int ci_session_run(ci_session* session, float* kernel_ms);
int ci_session_fetch(ci_session* session, ci_outputs* outputs);
/* ci_session_run + ci_session_fetch with the copies OVERLAPPED with the fit: the persistent
 * kernel publishes, every chunk_draws retained draws, how many rows of each chain are complete;
 * the host then queues the device-to-host copies of those rows ([chains, chunk, T] blocks of
 * level / slope / trajectories) on a second stream while sampling continues, and only the last
 * chunk and the small arrays remain after the kernel ends.  Give it buffers from ci_host_alloc
 * (pinned): the copies then run at PCIe rate; pageable buffers work but copy synchronously.
 * Models on the seasonal kernels are copied in the same chunks after the kernel has finished. */
int ci_session_run_streamed(ci_session* session, ci_outputs* outputs, int32_t chunk_draws,
                            float* kernel_ms);
/* Bytes the kernel must move per run (algorithmic bytes, DESIGN.md "Roofline"). */
int ci_session_algorithmic_bytes(const ci_session* session, double* bytes);
/* Name of the Gibbs kernel instantiation this session dispatches to, as a profiler shows it
 * (e.g. "ci::gibbs_kernel<2,4,1,false>"); NUL-terminated, truncated to buflen. */
int ci_session_kernel_name(const ci_session* session, char* buf, int32_t buflen);
/* Developer aid: when enabled, the next ci_session_run() accumulates shader-clock cycles of
 * the kernel's phases (block 0) into 32 counters; cycles16 (optional, 32 entries) receives the counters
 * of the previous run.  Slot meaning: DESIGN.md "Time budget". */
int ci_session_profile(ci_session* session, int enable, int64_t* cycles16);
int ci_session_destroy(ci_session* session);

/* On-device summarisation of the pooled posterior-predictive draws of a finished run, for all
 * B series of the session at once: replaces the T x (C*S) pandas/numpy work of _compute_impact
 * (causalimpact_lib.py:793-837 effect trajectories, posterior_processing.py:25-60 quantiles,
 * :966-1017 per-draw post-period totals).  All results are float64 on the data scale:
 *   value[b, n, t] = trajectory[b, n, t] * scale[b] + shift[b]   (standardize.py:60-64)
 *   point[b, n, t] = -(value[b, n, t] - observed[b, t])           NaN where observed is NaN
 *   cum[b, n, t]   = running sum of point from the treatment start (NaN steps skipped, reported NaN)
 * scale, shift [B]; observed [B,T] data-scale outcome, NaN = no observation; flags [B,T]: bit 0
 * = t >= treatment start, bit 1 = inside the post-period window.  ranks: num_ranks (<= 8)
 * 0-based order statistics over the N = C*S draws of a series (the caller interpolates
 * quantiles from them exactly as numpy does).  Outputs (host, caller-allocated, any may be NULL):
 *   value_order [B, num_ranks, T], cum_order [B, num_ranks, T],
 *   per_draw [B, 2, N]: sum over the window of value, and of point (NaN skipped),
 *   per_draw_order [B, 2, num_ranks]: the same order statistics of those two rows of totals
 *   (what the `summary` frame's bands interpolate: causalimpact_lib.py:1019-1075).
 * Any output pointer may be NULL. */
int ci_session_summarize(ci_session* session, const double* scale, const double* shift,
                         const double* observed, const uint8_t* flags, int32_t num_ranks,
                         const int32_t* ranks, double* value_order, double* cum_order,
                         double* per_draw, double* per_draw_order);
/* The same summary for draws that are on the host (pooled from several devices / processes, or
 * produced by the HMC path): trajectories [num_draws, T] float32 are uploaded to `device`,
 * summarised there and the (one-series) results returned as above. */
int ci_summarize_draws(int32_t device, int32_t num_draws, int32_t T, const float* trajectories,
                       double scale, double shift, const double* observed, const uint8_t* flags,
                       int32_t num_ranks, const int32_t* ranks, double* value_order,
                       double* cum_order, double* per_draw, double* per_draw_order);
/* ... for float64 trajectories (the draws of ci_fit_gibbs_f64: no rounding to float32 on the way
 * into the order statistics). */
int ci_summarize_draws_f64(int32_t device, int32_t num_draws, int32_t T, const double* trajectories,
                           double scale, double shift, const double* observed, const uint8_t* flags,
                           int32_t num_ranks, const int32_t* ranks, double* value_order,
                           double* cum_order, double* per_draw, double* per_draw_order);

/* Kalman-filter log-likelihood of the trend + regression model for num_evals parameter sets
 * (the objective an HMC / VI fit would use; the reference never
 * evaluates it directly -- upstream it is LinearGaussianStateSpaceModel.log_prob).
 *   theta [num_evals, 3 + P] float64: (sigma_obs, sigma_level, sigma_slope, weights[P])
 *   loglik [num_evals] float64.   One series (num_series must be 1), no seasonal blocks. */
int ci_kalman_loglik(const ci_problem* problem, const ci_series_params* params, const float* y,
                     const uint8_t* mask, const float* X, int32_t num_evals, const double* theta,
                     double* loglik);

/* Device-resident variant for samplers that evaluate l(theta) many times (HMC leapfrogs):
 * data stay in HBM; each eval moves only theta in and (loglik, grad) out.
 *   grad [num_evals, 3 + P] float64 = dl/d(sigma_obs, sigma_level, sigma_slope, weights);
 *   NULL skips the backward pass.
 * ci_ll_session_draw_latents draws, for each theta row, one latent path (Durbin-Koopman) and
 * one posterior-predictive trajectory (what one_step_predictive, causalimpact_lib.py:620-631,
 * does with GibbsSamplerState draws): level/slope/loc/traj are [num_draws, T] float32; RNG
 * stream (seed, chain = rng_chain, iteration = iter0 + draw). */
typedef struct ci_ll_session ci_ll_session;
int ci_ll_session_create(const ci_problem* problem, const ci_series_params* params, const float* y,
                         const uint8_t* mask, const float* X, int32_t max_evals,
                         ci_ll_session** session);
/* The same for ANY model and length (season_change [K,T] as in ci_fit_gibbs): trend + one block
 * of 2-7 seasons and trend-only series of more than 4096 steps are time-parallel
 * (csrc/ci_wide_score.h), other block lists -- or CI_FLAG_SEQUENTIAL_SEASONAL -- take the
 * sequential one-wavefront route (csrc/ci_score_seq.h).  theta / grad rows are then
 * [3 + K + P]: (sigma_obs, sigma_level, sigma_slope, sigma_drift[K], weights[P]). */
int ci_ll_session_create2(const ci_problem* problem, const ci_series_params* params, const float* y,
                          const uint8_t* mask, const float* X, const uint8_t* season_change,
                          int32_t max_evals, ci_ll_session** session);
int ci_ll_session_eval(ci_ll_session* session, int32_t num_evals, const double* theta,
                       double* loglik, double* grad);
int ci_ll_session_draw_latents(ci_ll_session* session, int32_t num_draws, const double* theta,
                               const uint32_t seed[2], uint32_t rng_chain, uint32_t iter0,
                               float* level, float* slope, float* loc, float* traj);
/* Hamiltonian Monte Carlo over the model's parameters entirely on the device: one workgroup per
 * chain runs num_warmup + num_results iterations of num_leapfrog steps (log-likelihood + score by
 * the same time-parallel scans as ci_ll_session_eval).  Warm-up is the three-stage windowed scheme
 * (step size by dual averaging -> doubling windows that re-estimate the diagonal mass -> step
 * size), csrc/ci_hmc.h.  Then ONE launch draws the latent path and the posterior-predictive
 * trajectory of every retained draw (what causalimpact_lib.py:609-632 does with the Gibbs
 * states), and the fit stays resident in HBM until ci_ll_session_hmc_fetch.
 * Target: l(theta) + the reference's inverse-gamma variance priors (causalimpact_lib.py:424-443)
 * + a continuous regression prior (the spike-and-slab prior has no density):
 *   CI_HMC_PRIOR_SLAB       the Gaussian slab of the reference's prior (:451-453);
 *                           theta = (beta[P], log sigma_obs, log sigma_level[, log sigma_slope])
 *   CI_HMC_PRIOR_HORSESHOE  the horseshoe of tfp.sts.SparseLinearRegression (what BASELINE.json's
 *                           north_star names): beta_j = z_j ln_j sqrt(lv_j) gn sqrt(gv) s0;
 *                           theta = (z[P], log ln[P], log lv[P], log gn, log gv, log scales)
 * EXTENSION (upstream analogues tfp.sts.fit_with_hmc /
 * tfp.experimental.mcmc.windowed_adaptive_hmc; the reference itself has no HMC path).  RNG stream
 * (seed, chain = chain_offset + c): results do not depend on how chains are split over devices.
 * init_theta (optional): [num_chains, dim] unconstrained starting points, e.g. draws of a fitted
 * surrogate posterior; NULL starts from the Gibbs sampler's initial state.
 * kernel_ms (optional, 2 floats): HIP-event durations on the session's stream of the HMC kernel
 * and of the latent/predictive pass. */
#define CI_HMC_PRIOR_SLAB 0
#define CI_HMC_PRIOR_HORSESHOE 1
typedef struct ci_hmc_options {
  int32_t num_chains, chain_offset;
  int32_t num_warmup, num_results, num_leapfrog;
  int32_t prior;                 /* CI_HMC_PRIOR_* */
  double target_accept;          /* dual-averaging target (0.75 upstream) */
  double initial_step_size;
  double horseshoe_scale;        /* s0 = weights_prior_scale (horseshoe only) */
  uint32_t seed[2];
} ci_hmc_options;
int ci_ll_session_hmc_run(ci_ll_session* session, const ci_hmc_options* options,
                          const double* init_theta, float* kernel_ms);
/* Copies the finished fit to the host; every pointer may be NULL.  draws [num_chains,
 * num_results, 3 + P] float64 rows (sigma_obs, sigma_level, sigma_slope, beta); accept_rate,
 * step_size [num_chains]; outputs: the sample container of ci_fit_gibbs with B = 1 (seasonal
 * fields ignored). */
int ci_ll_session_hmc_fetch(ci_ll_session* session, double* draws, double* accept_rate,
                            double* step_size, ci_outputs* outputs);
int ci_ll_session_kernel_name(const ci_ll_session* session, char* buf, int32_t buflen);
/* Algorithmic bytes of the last configured HMC fit (DESIGN.md "Roofline", cfg3). */
int ci_ll_session_algorithmic_bytes(const ci_ll_session* session, double* bytes);
int ci_ll_session_destroy(ci_ll_session* session);

/* ---- chain gather / diagnostics across the GPUs of one node ----
 * The reference runs one chain in one process and has no communication;
 * BASELINE.json shards independent chains over the GPUs of a node, "RCCL over xGMI only for chain
 * gather/diagnostics".  The fit itself never communicates.  One ci_comm per rank (one rank per
 * GPU); transports:
 *   CI_COMM_RCCL  librccl, loaded on first use: ncclAllGather / ncclAllReduce on device buffers.
 *   CI_COMM_HOST  a POSIX shared-memory segment on one node, for ranks that share a device (RCCL
 *                 refuses two ranks on one GPU) and GPU-less tests; same results, staged via host.
 * ci_comm_unique_id is called by ONE rank; the 128 bytes reach the others out of band (the host
 * package passes them through a file, causalimpact/_comm.py).  ci_comm_create is collective.
 * ranks_seen = ncclCommCount of the live communicator (host transport: attached ranks). */
#define CI_COMM_ID_BYTES 128
#define CI_COMM_RCCL 0
#define CI_COMM_HOST 1
#define CI_COMM_SUM 0
#define CI_COMM_MAX 1
typedef struct ci_comm ci_comm;
int ci_comm_unique_id(int32_t transport, uint8_t* id /* [CI_COMM_ID_BYTES] */);
int ci_comm_create(int32_t transport, const uint8_t* id, int32_t rank, int32_t world,
                   int32_t device, ci_comm** comm);
int ci_comm_info(const ci_comm* comm, int32_t* rank, int32_t* world, int32_t* ranks_seen);
/* Bound of ONE collective on this communicator, seconds (0: $CI_COMM_TIMEOUT_S, default 300).  A
 * collective that exceeds it returns an error; on the RCCL transport the communicator is aborted
 * (ncclCommAbort) so that its kernel leaves the GPU, and every later call on it fails at once. */
int ci_comm_set_timeout(ci_comm* comm, double seconds);
int ci_comm_barrier(ci_comm* comm);
/* values [n] float64 on the host, reduced in place over all ranks (every rank gets the same bits). */
int ci_comm_all_reduce(ci_comm* comm, double* values, int64_t n, int32_t op);
/* send [bytes] -> recv [world, bytes], host buffers, rank order. */
int ci_comm_all_gather(ci_comm* comm, const void* send, void* recv, int64_t bytes);
/* One result array of a finished fit, gathered STRAIGHT FROM HBM (no host round trip on the RCCL
 * transport): every rank's device-resident [B, C, ...] block -> recv [world, B, C, ...] on the
 * host of every rank.  All ranks must hold sessions of the same shape.  field = CI_FIELD_*, the
 * members of ci_outputs in order. */
#define CI_FIELD_OBSERVATION_NOISE_SCALE 0
#define CI_FIELD_LEVEL_SCALE 1
#define CI_FIELD_SLOPE_SCALE 2
#define CI_FIELD_SEASONAL_DRIFT_SCALES 3
#define CI_FIELD_WEIGHTS 4
#define CI_FIELD_LEVEL 5
#define CI_FIELD_SLOPE 6
#define CI_FIELD_SEASONAL_LEVELS 7
#define CI_FIELD_POSTERIOR_MEANS 8
#define CI_FIELD_POSTERIOR_TRAJECTORIES 9
int ci_comm_session_all_gather(ci_comm* comm, ci_session* session, int32_t field, float* recv);
int ci_comm_ll_session_all_gather(ci_comm* comm, ci_ll_session* session, int32_t field, float* recv);
int ci_comm_destroy(ci_comm* comm);

/* ---- component entry points used by the parity tests (tests/test_gpu_*.py) ---- */
/* normals/uniforms/gammas of the specified Philox stream, computed on device. */
int ci_test_rng(int device, const uint32_t seed[2], uint32_t chain, uint32_t iter, uint32_t site,
                uint32_t sub, int32_t n, float* uniforms, float* normals, double alpha,
                double* gamma_draw);
/* One Durbin-Koopman draw of the latent path given residuals and scales
 * (LinearGaussianStateSpaceModel.posterior_sample as reached from
 * gibbs_sampler._resample_latents).  out_latents is [T, d] float32. */
int ci_test_dk_draw(const ci_problem* problem, const ci_series_params* params,
                    const float* resid, const uint8_t* mask, const uint8_t* season_change,
                    double obs_scale, double level_scale, double slope_scale,
                    const double* drift_scales, uint32_t iter, float* out_latents);

#ifdef __cplusplus
}
#endif
#endif  /* CAUSALIMPACT_AMD_H_ */
