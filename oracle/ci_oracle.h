/*
 * ci_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C float64 restatement of the Gibbs hot path behind
 * causalimpact.fit_causalimpact():
 *   /root/reference/causalimpact/causalimpact_lib.py:345-395  (_run_gibbs_sampler)
 *   /root/reference/causalimpact/causalimpact_lib.py:398-500  (priors)
 *   /root/reference/causalimpact/causalimpact_lib.py:609-632  (predictive draws)
 * The arithmetic the reference delegates to lives in the un-vendored, un-pinned
 * third-party dependency `tensorflow-probability` (pyproject.toml:22):
 *   tfp.experimental.sts_gibbs.gibbs_sampler.{fit_with_gibbs_sampling,
 *   _resample_latents,_resample_scale,one_step_predictive},
 *   tfp.experimental.sts_gibbs.spike_and_slab.SpikeSlabSampler,
 *   tfd.LinearGaussianStateSpaceModel.posterior_sample (Durbin-Koopman),
 *   tfp.sts.{LocalLevel,LocalLinearTrend,Seasonal(constrained)}.
 * This file restates their published algorithms (Durbin & Koopman 2002;
 * Scott & Varian 2013 eq. 8; TFP source as recalled -- see DESIGN.md "Oracle").
 *
 * PARITY STATUS: TFP cannot be imported here, and the reference holds no golden
 * vectors for the sampler, so per-draw parity with TFP's RNG streams is
 * UNPINNED.  What pins this oracle:
 *   (1) TFP-independent exact maths (dense Gaussian posterior / log-lik,
 *       exhaustive 2^P spike-slab enumeration, inverse-gamma KS tests):
 *       tests/test_oracle_math.py;
 *   (2) the reference's own statistical tolerance tests
 *       (causalimpact_lib_test.py:242-271,319-338,361-379,504-535,655-773):
 *       tests/test_reference_stat_pins.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path never does.
 *
 * Randomness is a *specified* counter-based stream (Philox4x32-10) shared with
 * the HIP kernels so that kernel and oracle consume identical random numbers:
 *   key     = (seed0, seed1)
 *   counter = (call_index, site | (sub << 8), iteration, chain)
 * One Philox call yields 4 uniforms / 4 Box-Muller normals (components 0..3);
 * element `idx` of a site lives in call idx>>2, component idx&3.
 */
#ifndef CI_ORACLE_H_
#define CI_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CI_MAX_D 64        /* max state dimension handled by the oracle (the device: 64 full-effect lanes) */
#define CI_MAX_BLOCKS 8    /* max seasonal blocks */

/* RNG sites (shared with the HIP kernels: csrc/ci_rng.h). */
enum {
  CI_SITE_PERM = 1,        /* uniforms for the feature-visit permutation */
  CI_SITE_FLIP = 2,        /* uniforms for the inclusion flips */
  CI_SITE_OBSVAR = 3,      /* gamma attempts: sigma^2_obs (spike-slab branch) */
  CI_SITE_WEIGHTS = 4,     /* normals for the weights draw */
  CI_SITE_PRIOR_INIT = 5,  /* normals: DK prior initial state */
  CI_SITE_PRIOR_LEVEL = 6, /* normals: DK level disturbances */
  CI_SITE_PRIOR_SLOPE = 7, /* normals: DK slope disturbances */
  CI_SITE_PRIOR_OBS = 8,   /* normals: DK observation noise */
  CI_SITE_PRIOR_SEAS = 9,  /* normals: DK seasonal drift (sub = block) */
  CI_SITE_LEVEL_SCALE = 10,
  CI_SITE_SLOPE_SCALE = 11,
  CI_SITE_DRIFT_SCALE = 12, /* sub = block */
  CI_SITE_OBS_SCALE = 13,   /* gamma attempts: sigma_obs (no-regression branch) */
  CI_SITE_PRED = 14,        /* normals: posterior-predictive noise */
  /* HMC extension (csrc/ci_hmc.h) */
  CI_SITE_HMC_MOMENTUM = 15, CI_SITE_HMC_ACCEPT = 16, CI_SITE_HMC_INIT = 17
};

typedef struct {
  /* sizes */
  int32_t T;            /* time steps fed to the sampler (pre + after-pre) */
  int32_t P;            /* design columns incl. intercept; 0 = no regression */
  int32_t has_slope;    /* 0 = LocalLevel, 1 = LocalLinearTrend */
  int32_t num_blocks;   /* seasonal blocks K */
  int32_t num_seasons[CI_MAX_BLOCKS];
  int32_t num_warmup;   /* W */
  int32_t num_results;  /* S */
  uint32_t seed[2];
  int32_t chain;        /* global chain id (RNG counter word 3) */
  int32_t flags;        /* CI_ORACLE_FLAG_* (0 = the reference's behaviour) */
  /* data (caller owned) */
  const double* y;               /* [T]; ignored where mask != 0 */
  const uint8_t* mask;           /* [T]; 1 = missing */
  const double* X;               /* [T*P] row-major, intercept last */
  const uint8_t* season_change;  /* [K*T]; 1 if t -> t+1 changes season in block k */
  /* priors (causalimpact_lib.py:424-489) */
  double level_conc, level_scale, level_ub;     /* IG on sigma^2_level, clip on sigma */
  double slope_conc, slope_scale, slope_ub;     /* IG on sigma^2_slope (LLT only) */
  double obs_conc, obs_scale, obs_ub;           /* IG on sigma^2_obs */
  double drift_conc, drift_scale, drift_ub;     /* IG on sigma^2_drift (all blocks) */
  double nonzero_prob;                          /* pi = min(1, 3/P) */
  double init_level_loc, init_level_scale;      /* N(y[0], sd) */
  double init_slope_scale;                      /* N(0, .) (LLT only) */
  double init_seasonal_scale;                   /* N(0, sd) per effect */
  /* initial Gibbs state (causalimpact_lib.py:566-581) */
  double obs_scale0, level_scale0, slope_scale0, drift_scale0[CI_MAX_BLOCKS];
  /* optional starting point of the chain (NULL = the reference's zeros, :575-581); lets a test
   * apply ONE Gibbs transition to an arbitrary state (tests/test_geweke.py) */
  const double* weights0;  /* [P] */
  const double* latents0;  /* [T*d] */
  /* multiplier of the weights-prior precision Omega = 0.01 (X'X/2 + diag(X'X)/2) / T
   * (causalimpact_lib.py:451-453); 0 is read as 1.  The product's internal conditioning of a raw-
   * scale outcome, y -> (y - mu) / s, needs Omega in the conditioned units: s^2 Omega. */
  double weights_prior_scale;
} ci_oracle_problem;

/* Test-only: sample with the weights prior N(0, sigma^2 Omega^-1) exactly as the collapsed
 * spike-and-slab sampler is derived (Scott & Varian 2013), i.e. WITHOUT the reference's
 * experimental_use_weight_adjustment=True rescaling (causalimpact_lib.py:388), so that the
 * transition has a known invariant distribution (Geweke 2004 joint-distribution test). */
#define CI_ORACLE_FLAG_NO_WEIGHT_ADJUSTMENT 1
/* Test-only: exponent a_post instead of the reference's (a_post - 1) in the collapsed
 * spike-and-slab marginal (see ss_evaluate in ci_oracle.c). */
#define CI_ORACLE_FLAG_EXACT_MARGINAL 2

typedef struct {
  /* all caller-allocated; any pointer may be NULL to skip that output */
  double* obs_scale;     /* [S] */
  double* level_scale;   /* [S] */
  double* slope_scale;   /* [S] */
  double* drift_scales;  /* [S*K] */
  double* weights;       /* [S*P] */
  double* level;         /* [S*T] */
  double* slope;         /* [S*T] */
  double* seasonal;      /* [S*T*K]  0-th latent of each block */
  double* pred_mean;     /* [T]      mean over draws of the noise-free predictor */
  double* trajectories;  /* [S*T]    posterior-predictive draws */
  int32_t* nonzeros;     /* [S*P]    inclusion indicators (diagnostic) */
} ci_oracle_outputs;

/* Full Gibbs fit for one chain.  Returns 0 on success. */
int ci_oracle_fit_gibbs(const ci_oracle_problem* pb, ci_oracle_outputs* out);

/* cpu_baseline leg of bench.py only: n_chains whole fits (chain ids first_chain ...), each
 * producing every output array of ci_oracle_outputs into buffers of its own, OpenMP-parallel over
 * chains when built with -fopenmp (the `native` target: -O3 -march=native -fopenmp, BASELINE.md
 * section 2), serial otherwise.  Returns the number of chains that failed. */
int ci_oracle_fit_gibbs_chains(const ci_oracle_problem* pb, int first_chain, int n_chains);
/* Threads the call above uses (1 without OpenMP) / sets them. */
int ci_oracle_max_threads(void);
void ci_oracle_set_threads(int n);

/* ---- building blocks exposed for the unit tests ---- */
void ci_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double ci_oracle_uniform(const uint32_t seed[2], uint32_t chain, uint32_t iter,
                         uint32_t site, uint32_t sub, uint32_t idx);
double ci_oracle_normal(const uint32_t seed[2], uint32_t chain, uint32_t iter,
                        uint32_t site, uint32_t sub, uint32_t idx);
/* Gamma(alpha, 1) by Marsaglia-Tsang with indexed attempts. */
double ci_oracle_gamma(double alpha, const uint32_t seed[2], uint32_t chain,
                       uint32_t iter, uint32_t site, uint32_t sub);

/* State-space description used by the filter/smoother helpers. */
typedef struct {
  int32_t T, d, has_slope, num_blocks;
  int32_t num_seasons[CI_MAX_BLOCKS];
  const uint8_t* mask;
  const uint8_t* season_change;
  double obs_scale, level_scale, slope_scale, drift_scale[CI_MAX_BLOCKS];
  double init_level_loc, init_level_scale, init_slope_scale, init_seasonal_scale;
} ci_oracle_ssm;

int ci_oracle_state_dim(int has_slope, int num_blocks, const int32_t* num_seasons);

/* Kalman filter log-likelihood of data[T] (masked steps skipped). */
double ci_oracle_kalman_loglik(const ci_oracle_ssm* m, const double* data);

/* Smoothed state means E[x_t | data]; out is [T*d]. */
void ci_oracle_smoothed_mean(const ci_oracle_ssm* m, const double* data, double* out);

/* One Durbin-Koopman posterior draw of the latents given data[T];
 * normals come from the spec'd stream (seed, chain, iter). out is [T*d]. */
void ci_oracle_dk_draw(const ci_oracle_ssm* m, const double* data,
                       const uint32_t seed[2], uint32_t chain, uint32_t iter,
                       double* out);

/* Collapsed spike-and-slab log posterior of an inclusion pattern
 * (Scott & Varian 2013 eq. 8 as TFP states it). */
double ci_oracle_spike_slab_logp(int P, const double* xtx, const double* prior_prec,
                                 const double* xty, double yty, const uint8_t* nonzeros,
                                 double nonzero_prob, double post_conc, double prior_scale);

/* ---- row H (extension): score of the log-likelihood and the HMC sampler of csrc/ci_hmc.h ---- */

/* Log-likelihood of data[T] plus its score: e_out[T] = -dl/d data_t (0 where masked; so
 * dl/dbeta = X'e for data = y - X beta), g_scales = dl/d(sigma_obs, sigma_level, sigma_slope,
 * sigma_drift[K]).  Either output may be NULL. */
double ci_oracle_loglik_score(const ci_oracle_ssm* m, const double* data, double* e_out,
                              double* g_scales);

typedef struct {
  int32_t T, P, has_slope, num_blocks;
  int32_t num_seasons[CI_MAX_BLOCKS];
  int32_t num_warmup, num_results, num_leapfrog;
  int32_t prior_mode;            /* 0 Gaussian slab, 1 horseshoe */
  uint32_t seed[2];
  int32_t chain;
  int32_t reserved;
  const double* y;               /* [T] */
  const uint8_t* mask;           /* [T] */
  const double* X;               /* [T*P] row-major */
  const uint8_t* season_change;  /* [K*T] */
  const double* omega;           /* [P*P] slab precision (prior_mode 0) */
  const double* init;            /* [dim] unconstrained start, or NULL */
  /* scales in theta order: obs, level, [slope], drift[K] */
  double ig_a[3 + CI_MAX_BLOCKS], ig_b[3 + CI_MAX_BLOCKS], init_log[3 + CI_MAX_BLOCKS];
  double hs_scale0, target_accept, eps0;
  double init_level_loc, init_level_scale, init_slope_scale, init_seasonal_scale;
} ci_oracle_hmc_problem;

int ci_oracle_hmc_dim(const ci_oracle_hmc_problem* pb);
/* log posterior (up to a constant) and its gradient at the unconstrained point theta[dim]. */
double ci_oracle_hmc_logp(const ci_oracle_hmc_problem* pb, const double* theta, double* grad);
/* Warm-up schedule: step size only on [0, slow_begin) and [slow_end, W); mass windows tile
 * [slow_begin, slow_end), the first ending at first_end, each next one twice as long. */
void ci_oracle_hmc_windows(int W, int* slow_begin, int* slow_end, int* first_end, int* base);
/* One chain.  draws [S, 3 + K + P] rows (sigma_obs, sigma_level, sigma_slope, drift[K], beta[P]). */
int ci_oracle_fit_hmc(const ci_oracle_hmc_problem* pb, double* draws, double* accept_rate,
                      double* step_size);
/* Latent path and posterior-predictive trajectory of each of the S draws (rows as above), as
 * the device's latents pass: Durbin-Koopman draw with iteration = draw index.  Outputs [S*T]. */
int ci_oracle_hmc_latents(const ci_oracle_hmc_problem* pb, const double* draws, int S,
                          double* level, double* slope, double* loc, double* traj);

#ifdef __cplusplus
}
#endif
#endif  /* CI_ORACLE_H_ */
