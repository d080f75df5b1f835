/*
 * ci_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ci_oracle.h for scope, provenance and parity status ("per-draw parity
 * with TFP unpinned; pinned by exact maths + the reference's statistical
 * tests").  Float64 throughout, single thread, deliberately plain.
 */
#include "ci_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011, "Parallel random numbers: as    */
/* easy as 1, 2, 3").  Same stream as csrc/ci_rng.h on the device.     */
/* ------------------------------------------------------------------ */
void ci_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void site_call(const uint32_t seed[2], uint32_t chain, uint32_t iter,
                      uint32_t site, uint32_t sub, uint32_t call, uint32_t r[4]) {
  uint32_t ctr[4] = {call, site | (sub << 8), iter, chain};
  ci_oracle_philox(ctr, seed, r);
}

static double u01(uint32_t r) { return ((double)r + 0.5) * (1.0 / 4294967296.0); }

double ci_oracle_uniform(const uint32_t seed[2], uint32_t chain, uint32_t iter,
                         uint32_t site, uint32_t sub, uint32_t idx) {
  uint32_t r[4];
  site_call(seed, chain, iter, site, sub, idx >> 2, r);
  return u01(r[idx & 3]);
}

/* Box-Muller: components (0,1) from (r0,r1), (2,3) from (r2,r3). */
static void box_muller(uint32_t ra, uint32_t rb, double* z0, double* z1) {
  double u1 = u01(ra);
  double rev = (double)rb * (1.0 / 4294967296.0); /* angle in revolutions */
  double rad = sqrt(-2.0 * log(u1));
  *z0 = rad * cos(2.0 * M_PI * rev);
  *z1 = rad * sin(2.0 * M_PI * rev);
}

double ci_oracle_normal(const uint32_t seed[2], uint32_t chain, uint32_t iter,
                        uint32_t site, uint32_t sub, uint32_t idx) {
  uint32_t r[4];
  double z0, z1;
  site_call(seed, chain, iter, site, sub, idx >> 2, r);
  if ((idx & 3) < 2) box_muller(r[0], r[1], &z0, &z1);
  else box_muller(r[2], r[3], &z0, &z1);
  return (idx & 1) ? z1 : z0;
}

/* Marsaglia & Tsang (2000).  Attempt k uses Philox call k of the site:
 * normal from (r0,r1) (cos branch), accept-uniform from r2, boost-uniform r3.
 * At most 64 attempts (one wavefront evaluates them in parallel on device). */
double ci_oracle_gamma(double alpha, const uint32_t seed[2], uint32_t chain,
                       uint32_t iter, uint32_t site, uint32_t sub) {
  double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  double d = a - 1.0 / 3.0;
  double c = 1.0 / sqrt(9.0 * d);
  for (uint32_t k = 0; k < 64; ++k) {
    uint32_t r[4];
    double x, unused;
    site_call(seed, chain, iter, site, sub, k, r);
    box_muller(r[0], r[1], &x, &unused);
    double t = 1.0 + c * x;
    double v = t * t * t;
    if (v <= 0.0) continue;
    double u = u01(r[2]);
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) {
      double g = d * v;
      if (alpha < 1.0) g *= pow(u01(r[3]), 1.0 / alpha);
      return g;
    }
  }
  return d; /* unreachable in practice: P(64 rejections) < 1e-80 */
}

/* ------------------------------------------------------------------ */
/* small dense helpers                                                  */
/* ------------------------------------------------------------------ */
static int chol_lower(int n, double* a /* n*n row-major, in place */) {
  for (int j = 0; j < n; ++j) {
    double s = a[j * n + j];
    for (int k = 0; k < j; ++k) s -= a[j * n + k] * a[j * n + k];
    if (!(s > 0.0)) return -1;
    double l = sqrt(s);
    a[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double t = a[i * n + j];
      for (int k = 0; k < j; ++k) t -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = t / l;
    }
    for (int i = 0; i < j; ++i) a[i * n + j] = 0.0;
  }
  return 0;
}

static void chol_solve(int n, const double* l, const double* b, double* x) {
  /* solve (L L') x = b */
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= l[i * n + k] * x[k];
    x[i] = s / l[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= l[k * n + i] * x[k];
    x[i] = s / l[i * n + i];
  }
}

/* ------------------------------------------------------------------ */
/* state-space model pieces                      */
/* ------------------------------------------------------------------ */
int ci_oracle_state_dim(int has_slope, int num_blocks, const int32_t* num_seasons) {
  int d = 1 + (has_slope ? 1 : 0);
  for (int k = 0; k < num_blocks; ++k) d += num_seasons[k] - 1;
  return d;
}

static int block_offset(const ci_oracle_ssm* m, int k) {
  int o = 1 + (m->has_slope ? 1 : 0);
  for (int j = 0; j < k; ++j) o += m->num_seasons[j] - 1;
  return o;
}

/* x <- T_t x  (LocalLevel / LocalLinearTrend block + constrained seasonal
 * blocks: identity inside a season, companion rotation at a season change;
 * tfp.sts.Seasonal(constrain_mean_effect_to_zero=True)). */
static void apply_transition(const ci_oracle_ssm* m, int t, double* x) {
  if (m->has_slope) x[0] += x[1];
  for (int k = 0; k < m->num_blocks; ++k) {
    if (!m->season_change[(size_t)k * m->T + t]) continue;
    int o = block_offset(m, k), n1 = m->num_seasons[k] - 1;
    double s = 0.0;
    for (int i = 0; i < n1; ++i) s += x[o + i];
    for (int i = 0; i + 1 < n1; ++i) x[o + i] = x[o + i + 1];
    x[o + n1 - 1] = -s;
  }
}

/* P <- T_t P T_t' + Q_t  (dense, d x d row-major). */
static void propagate_cov(const ci_oracle_ssm* m, int t, double* P) {
  int d = m->d;
  double col[CI_MAX_D];
  /* columns: P <- T P  (apply T to each column) */
  for (int j = 0; j < d; ++j) {
    for (int i = 0; i < d; ++i) col[i] = P[i * d + j];
    apply_transition(m, t, col);
    for (int i = 0; i < d; ++i) P[i * d + j] = col[i];
  }
  /* rows: P <- P T' (apply T to each row) */
  for (int i = 0; i < d; ++i) apply_transition(m, t, &P[i * d]);
  P[0] += m->level_scale * m->level_scale;
  if (m->has_slope) P[1 * d + 1] += m->slope_scale * m->slope_scale;
  for (int k = 0; k < m->num_blocks; ++k) {
    if (!m->season_change[(size_t)k * m->T + t]) continue;
    int o = block_offset(m, k), n = m->num_seasons[k];
    double q = m->drift_scale[k] / n;
    q *= q;
    for (int i = 0; i < n - 1; ++i)
      for (int j = 0; j < n - 1; ++j) P[(o + i) * d + (o + j)] += q;
  }
}

static void initial_moments(const ci_oracle_ssm* m, double* a, double* P) {
  int d = m->d;
  memset(a, 0, sizeof(double) * d);
  memset(P, 0, sizeof(double) * d * d);
  a[0] = m->init_level_loc;
  P[0] = m->init_level_scale * m->init_level_scale;
  if (m->has_slope) P[1 * d + 1] = m->init_slope_scale * m->init_slope_scale;
  for (int k = 0; k < m->num_blocks; ++k) {
    int o = block_offset(m, k), n = m->num_seasons[k];
    double v = m->init_seasonal_scale * m->init_seasonal_scale;
    for (int i = 0; i < n - 1; ++i)
      for (int j = 0; j < n - 1; ++j)
        P[(o + i) * d + (o + j)] = v * ((i == j ? 1.0 : 0.0) - 1.0 / n);
  }
}

static double observe(const ci_oracle_ssm* m, const double* x) {
  double s = x[0];
  for (int k = 0; k < m->num_blocks; ++k) s += x[block_offset(m, k)];
  return s;
}

typedef struct {
  double* a;   /* [T*d] predicted means */
  double* P;   /* [T*d*d] predicted covariances */
  double* vf;  /* [T] v_t / F_t (0 where masked) */
  double* kf;  /* [T*d] filter gain P Z'/F (0 where masked) */
  double loglik;
} kf_store;

static void kalman_forward(const ci_oracle_ssm* m, const double* data, kf_store* st) {
  int d = m->d, T = m->T;
  double a[CI_MAX_D], P[CI_MAX_D * CI_MAX_D], pz[CI_MAX_D];
  double H = m->obs_scale * m->obs_scale;
  initial_moments(m, a, P);
  st->loglik = 0.0;
  for (int t = 0; t < T; ++t) {
    if (st->a) memcpy(&st->a[(size_t)t * d], a, sizeof(double) * d);
    if (st->P) memcpy(&st->P[(size_t)t * d * d], P, sizeof(double) * d * d);
    if (!m->mask[t]) {
      double v = data[t] - observe(m, a);
      for (int i = 0; i < d; ++i) pz[i] = observe(m, &P[i * d]); /* P Z' (P symmetric) */
      double F = observe(m, pz) + H;
      st->loglik += -0.5 * (log(2.0 * M_PI) + log(F) + v * v / F);
      for (int i = 0; i < d; ++i) {
        double k = pz[i] / F;
        if (st->kf) st->kf[(size_t)t * d + i] = k;
        a[i] += k * v;
      }
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) P[i * d + j] -= pz[i] * pz[j] / F;
      if (st->vf) st->vf[t] = v / F;
    } else {
      if (st->vf) st->vf[t] = 0.0;
      if (st->kf) memset(&st->kf[(size_t)t * d], 0, sizeof(double) * d);
    }
    if (t + 1 < T) {
      apply_transition(m, t, a);
      propagate_cov(m, t, P);
    }
  }
}

double ci_oracle_kalman_loglik(const ci_oracle_ssm* m, const double* data) {
  kf_store st = {0};
  kalman_forward(m, data, &st);
  return st.loglik;
}

/* x' <- T_t' x (transpose of apply_transition). */
static void apply_transition_T(const ci_oracle_ssm* m, int t, double* x) {
  if (m->has_slope) x[1] += x[0];
  for (int k = 0; k < m->num_blocks; ++k) {
    if (!m->season_change[(size_t)k * m->T + t]) continue;
    int o = block_offset(m, k), n1 = m->num_seasons[k] - 1;
    /* T = [[0 I],[-1 ... -1]]  =>  (T' x)_0 = -x_last ; (T' x)_j = x_{j-1} - x_last */
    double last = x[o + n1 - 1];
    for (int j = n1 - 1; j >= 1; --j) x[o + j] = x[o + j - 1] - last;
    x[o] = -last;
  }
}

void ci_oracle_smoothed_mean(const ci_oracle_ssm* m, const double* data, double* out) {
  int d = m->d, T = m->T;
  kf_store st;
  st.a = (double*)malloc(sizeof(double) * T * d);
  st.P = (double*)malloc(sizeof(double) * T * d * d);
  st.vf = (double*)malloc(sizeof(double) * T);
  st.kf = (double*)malloc(sizeof(double) * T * d);
  kalman_forward(m, data, &st);
  double r[CI_MAX_D];
  memset(r, 0, sizeof(r));
  for (int t = T - 1; t >= 0; --t) {
    /* r_{t-1} = Z' v/F + (I - Kf Z)' T_t' r_t   (observed)
     *         = T_t' r_t                         (masked)        */
    if (t + 1 < T) apply_transition_T(m, t, r); else memset(r, 0, sizeof(double) * d);
    if (!m->mask[t]) {
      double kr = 0.0;
      for (int i = 0; i < d; ++i) kr += st.kf[(size_t)t * d + i] * r[i];
      double add = st.vf[t] - kr;
      r[0] += add;
      for (int k = 0; k < m->num_blocks; ++k) r[block_offset(m, k)] += add;
    }
    for (int i = 0; i < d; ++i) {
      double s = st.a[(size_t)t * d + i];
      for (int j = 0; j < d; ++j) s += st.P[((size_t)t * d + i) * d + j] * r[j];
      out[(size_t)t * d + i] = s;
    }
  }
  free(st.a); free(st.P); free(st.vf); free(st.kf);
}

/* Durbin & Koopman (2002) simulation smoother, one-filter-pass form:
 * x+ ~ prior with ZERO initial mean, y+ = Z x+ + eps+, smooth (data - y+)
 * under the ORIGINAL prior mean, return smoothed + x+.
 * (tfd.LinearGaussianStateSpaceModel.posterior_sample, reached from
 *  gibbs_sampler._resample_latents.) */
void ci_oracle_dk_draw(const ci_oracle_ssm* m, const double* data,
                       const uint32_t seed[2], uint32_t chain, uint32_t iter,
                       double* out) {
  int d = m->d, T = m->T;
  double a0[CI_MAX_D], P0[CI_MAX_D * CI_MAX_D], x[CI_MAX_D], z[CI_MAX_D];
  double* xplus = (double*)malloc(sizeof(double) * T * d);
  double* ytil = (double*)malloc(sizeof(double) * T);
  initial_moments(m, a0, P0);
  /* x+_0 = chol(P_1) z */
  if (chol_lower(d, P0) != 0) {
    /* P_1 is PD for every model we build; keep going with a diagonal fallback */
    initial_moments(m, a0, P0);
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) P0[i * d + j] = (i == j) ? sqrt(fabs(P0[i * d + i])) : 0.0;
  }
  for (int i = 0; i < d; ++i) z[i] = ci_oracle_normal(seed, chain, iter, CI_SITE_PRIOR_INIT, 0, i);
  for (int i = 0; i < d; ++i) {
    double s = 0.0;
    for (int j = 0; j <= i; ++j) s += P0[i * d + j] * z[j];
    x[i] = s;
  }
  for (int t = 0; t < T; ++t) {
    memcpy(&xplus[(size_t)t * d], x, sizeof(double) * d);
    double yplus = observe(m, x) +
        m->obs_scale * ci_oracle_normal(seed, chain, iter, CI_SITE_PRIOR_OBS, 0, t);
    ytil[t] = m->mask[t] ? 0.0 : data[t] - yplus;
    if (t + 1 < T) {
      apply_transition(m, t, x);
      x[0] += m->level_scale * ci_oracle_normal(seed, chain, iter, CI_SITE_PRIOR_LEVEL, 0, t);
      if (m->has_slope)
        x[1] += m->slope_scale * ci_oracle_normal(seed, chain, iter, CI_SITE_PRIOR_SLOPE, 0, t);
      for (int k = 0; k < m->num_blocks; ++k) {
        if (!m->season_change[(size_t)k * T + t]) continue;
        int o = block_offset(m, k), n = m->num_seasons[k];
        double w = m->drift_scale[k] *
            ci_oracle_normal(seed, chain, iter, CI_SITE_PRIOR_SEAS, (uint32_t)k, t);
        for (int i = 0; i < n - 1; ++i) x[o + i] -= w / n;
      }
    }
  }
  ci_oracle_smoothed_mean(m, ytil, out);
  for (size_t i = 0; i < (size_t)T * d; ++i) out[i] += xplus[i];
  free(xplus); free(ytil);
}

/* ------------------------------------------------------------------ */
/* spike-and-slab (tfp.experimental.sts_gibbs.spike_and_slab)           */
/* ------------------------------------------------------------------ */
typedef struct {
  int na;
  int idx[512];
  double logp;
  double post_scale;       /* b0 + (yty - b'M^{-1}b)/2 */
  double* chol_post;       /* na*na */
  double* mean;            /* na */
} ss_eval;

/* logp(gamma) = 1/2 logdet(Omega_g) - 1/2 logdet(Omega_g + XtX_g)
 *             + sum_j log prior(gamma_j) - (a_post - 1) log(2 beta_post(g)) */
/* Exponent of the collapsed marginal.  Scott & Varian (2013) eq. 8 -- and TFP's
 * spike_and_slab as recalled, and bsts/BOOM -- write SS^-(N/2 - 1), i.e. (a_post - 1); integrating
 * sigma^2 out of the model exactly gives a_post.  The difference multiplies the inclusion odds by
 * SS_g'/SS_g = 1 + O(1/n): immaterial at the reference's sizes, visible at n = 4
 * (tests/test_geweke.py, which therefore runs the exact exponent through
 * CI_ORACLE_FLAG_EXACT_MARGINAL).  Default 1.0 = the reference's formula. */
static _Thread_local double g_ss_exponent_offset = 1.0;   /* per thread: fits are re-entrant */

static int ss_evaluate(int P, const double* xtx, const double* prior_prec, const double* xty,
                       double yty, const uint8_t* nz, double nonzero_prob, double post_conc,
                       double prior_scale, double* work /* >= 3*P*P + 2*P */, ss_eval* ev) {
  int na = 0;
  for (int j = 0; j < P; ++j) if (nz[j]) ev->idx[na++] = j;
  ev->na = na;
  double* lp = work;            /* prior chol */
  double* lm = work + P * P;    /* posterior chol */
  double* b = work + 2 * P * P;
  double* mu = b + P;
  for (int i = 0; i < na; ++i) {
    for (int j = 0; j < na; ++j) {
      double o = prior_prec[ev->idx[i] * P + ev->idx[j]];
      lp[i * na + j] = o;
      lm[i * na + j] = o + xtx[ev->idx[i] * P + ev->idx[j]];
    }
    b[i] = xty[ev->idx[i]];
  }
  double half_logdet_prior = 0.0, half_logdet_post = 0.0, quad = 0.0;
  if (na > 0) {
    if (chol_lower(na, lp) != 0 || chol_lower(na, lm) != 0) return -1;
    chol_solve(na, lm, b, mu);
    for (int i = 0; i < na; ++i) {
      half_logdet_prior += log(lp[i * na + i]);
      half_logdet_post += log(lm[i * na + i]);
      quad += mu[i] * b[i];
    }
  }
  double prior_term = 0.0;
  if (nonzero_prob < 1.0) {
    for (int j = 0; j < P; ++j) prior_term += nz[j] ? log(nonzero_prob) : log1p(-nonzero_prob);
  }
  ev->post_scale = prior_scale + 0.5 * (yty - quad);
  ev->logp = half_logdet_prior - half_logdet_post + prior_term -
             (post_conc - g_ss_exponent_offset) * log(2.0 * ev->post_scale);
  if (ev->chol_post) memcpy(ev->chol_post, lm, sizeof(double) * na * na);
  if (ev->mean) memcpy(ev->mean, mu, sizeof(double) * na);
  return 0;
}

double ci_oracle_spike_slab_logp(int P, const double* xtx, const double* prior_prec,
                                 const double* xty, double yty, const uint8_t* nonzeros,
                                 double nonzero_prob, double post_conc, double prior_scale) {
  double* work = (double*)malloc(sizeof(double) * (3 * P * P + 2 * P + 8));
  ss_eval ev;
  ev.chol_post = NULL; ev.mean = NULL;
  int rc = ss_evaluate(P, xtx, prior_prec, xty, yty, nonzeros, nonzero_prob, post_conc,
                       prior_scale, work, &ev);
  free(work);
  return rc == 0 ? ev.logp : NAN;
}

/* ------------------------------------------------------------------ */
/* the Gibbs sampler                                                    */
/* ------------------------------------------------------------------ */
static double draw_scale(double conc, double scale, double ub, double n, double ss,
                         const uint32_t seed[2], uint32_t chain, uint32_t iter, uint32_t site,
                         uint32_t sub) {
  /* gibbs_sampler._resample_scale: variance ~ IG(conc + n/2, scale + ss/2);
   * new_scale = min(sqrt(variance), prior.upper_bound). */
  double g = ci_oracle_gamma(conc + 0.5 * n, seed, chain, iter, site, sub);
  double s = sqrt((scale + 0.5 * ss) / g);
  return s < ub ? s : ub;
}

int ci_oracle_fit_gibbs(const ci_oracle_problem* pb, ci_oracle_outputs* out) {
  const int T = pb->T, P = pb->P, K = pb->num_blocks;
  const int W = pb->num_warmup, S = pb->num_results;
  if (T < 1 || P < 0 || P > 512 || K < 0 || K > CI_MAX_BLOCKS) return -1;
  ci_oracle_ssm m;
  memset(&m, 0, sizeof(m));
  m.T = T; m.has_slope = pb->has_slope; m.num_blocks = K;
  for (int k = 0; k < K; ++k) m.num_seasons[k] = pb->num_seasons[k];
  m.d = ci_oracle_state_dim(pb->has_slope, K, pb->num_seasons);
  if (m.d > CI_MAX_D) return -2;
  m.mask = pb->mask; m.season_change = pb->season_change;
  m.init_level_loc = pb->init_level_loc; m.init_level_scale = pb->init_level_scale;
  m.init_slope_scale = pb->init_slope_scale; m.init_seasonal_scale = pb->init_seasonal_scale;
  const int d = m.d;
  const uint32_t chain = (uint32_t)pb->chain;

  double n_obs = 0.0;
  for (int t = 0; t < T; ++t) if (!pb->mask[t]) n_obs += 1.0;

  /* state */
  double obs_scale = pb->obs_scale0, level_scale = pb->level_scale0;
  double slope_scale = pb->has_slope ? pb->slope_scale0 : 0.0;
  double drift[CI_MAX_BLOCKS];
  for (int k = 0; k < K; ++k) drift[k] = pb->drift_scale0[k];
  double* w = (double*)calloc(P > 0 ? P : 1, sizeof(double));
  double* lat = (double*)calloc((size_t)T * d, sizeof(double));
  if (pb->weights0) memcpy(w, pb->weights0, sizeof(double) * P);
  if (pb->latents0) memcpy(lat, pb->latents0, sizeof(double) * (size_t)T * d);
  double* resid = (double*)malloc(sizeof(double) * T);
  double* targets = (double*)malloc(sizeof(double) * T);
  double* pred_acc = (double*)calloc(T, sizeof(double));

  /* constants of the regression (spike_and_slab.SpikeSlabSampler.__init__;
   * causalimpact_lib.py:451-453 for the explicit prior precision). */
  double *xtx = NULL, *omega = NULL, *omega_eff = NULL, *xty = NULL, *work = NULL;
  double *chol_post = NULL, *mean = NULL, *zw = NULL;
  uint8_t* nz = NULL;
  double* perm_u = NULL; int* perm = NULL;
  if (P > 0) {
    xtx = (double*)calloc((size_t)P * P, sizeof(double));
    omega = (double*)calloc((size_t)P * P, sizeof(double));
    omega_eff = (double*)malloc(sizeof(double) * P * P);
    xty = (double*)malloc(sizeof(double) * P);
    work = (double*)malloc(sizeof(double) * (3 * P * P + 2 * P + 8));
    chol_post = (double*)malloc(sizeof(double) * P * P);
    mean = (double*)malloc(sizeof(double) * P);
    zw = (double*)malloc(sizeof(double) * P);
    nz = (uint8_t*)calloc(P, 1);
    perm_u = (double*)malloc(sizeof(double) * P);
    perm = (int*)malloc(sizeof(int) * P);
    for (int t = 0; t < T; ++t) {
      const double* xr = &pb->X[(size_t)t * P];
      for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) {
          double v = xr[i] * xr[j];
          omega[i * P + j] += v;                 /* all T rows (":458-459 cheats") */
          if (!pb->mask[t]) xtx[i * P + j] += v; /* masked rows are zeroed in the design */
        }
    }
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j) {
        double full = omega[i * P + j];
        omega[i * P + j] = 0.01 * (i == j ? full : 0.5 * full) / (double)T *
                           (pb->weights_prior_scale > 0.0 ? pb->weights_prior_scale : 1.0);
      }
  }
  const double post_conc = pb->obs_conc + 0.5 * n_obs;
  g_ss_exponent_offset = (pb->flags & CI_ORACLE_FLAG_EXACT_MARGINAL) ? 0.0 : 1.0;   /* test-only */

  for (int it = 0; it < W + S; ++it) {
    const uint32_t uit = (uint32_t)it;
    /* ---- (a) regression: spike-and-slab draw of (sigma^2_obs, weights) ---- */
    if (P > 0) {
      double yty = 0.0;
      for (int t = 0; t < T; ++t) {
        double v = 0.0;
        if (!pb->mask[t]) {
          v = pb->y[t] - lat[(size_t)t * d];
          for (int k = 0; k < K; ++k) v -= lat[(size_t)t * d + block_offset(&m, k)];
        }
        targets[t] = v;
        yty += v * v;
      }
      for (int j = 0; j < P; ++j) xty[j] = 0.0;
      for (int t = 0; t < T; ++t) {
        if (pb->mask[t]) continue;
        for (int j = 0; j < P; ++j) xty[j] += pb->X[(size_t)t * P + j] * targets[t];
      }
      /* experimental_use_weight_adjustment=True: prior precision rescaled by the
       * previous observation-noise variance (causalimpact_lib.py:388). */
      double prev_var = obs_scale * obs_scale;
      if (pb->flags & CI_ORACLE_FLAG_NO_WEIGHT_ADJUSTMENT) prev_var = 1.0;   /* test-only */
      for (int i = 0; i < P * P; ++i) omega_eff[i] = omega[i] * prev_var;
      ss_eval cur, prop;
      cur.chol_post = chol_post; cur.mean = mean;
      prop.chol_post = NULL; prop.mean = NULL;
      if (pb->nonzero_prob >= 1.0) {
        /* pi = 1 (P <= 3): every feature is always included
         * (causalimpact_lib_test.py:376-379 pins "no zero weights"). */
        for (int j = 0; j < P; ++j) nz[j] = 1;
      } else {
        for (int j = 0; j < P; ++j) nz[j] = (w[j] != 0.0);
        if (ss_evaluate(P, xtx, omega_eff, xty, yty, nz, pb->nonzero_prob, post_conc,
                        pb->obs_scale, work, &cur) != 0) return -3;
        /* random visiting order: argsort of P uniforms (stable) */
        for (int j = 0; j < P; ++j) {
          perm_u[j] = ci_oracle_uniform(pb->seed, chain, uit, CI_SITE_PERM, 0, j);
          perm[j] = j;
        }
        for (int i = 1; i < P; ++i) { /* insertion sort, stable */
          int pj = perm[i]; int q = i - 1;
          while (q >= 0 && perm_u[perm[q]] > perm_u[pj]) { perm[q + 1] = perm[q]; --q; }
          perm[q + 1] = pj;
        }
        for (int s = 0; s < P; ++s) {
          int j = perm[s];
          nz[j] ^= 1;
          if (ss_evaluate(P, xtx, omega_eff, xty, yty, nz, pb->nonzero_prob, post_conc,
                          pb->obs_scale, work, &prop) != 0) return -3;
          double u = ci_oracle_uniform(pb->seed, chain, uit, CI_SITE_FLIP, 0, s);
          double pflip = 1.0 / (1.0 + exp(-(prop.logp - cur.logp)));
          if (u < pflip) { cur.logp = prop.logp; }
          else nz[j] ^= 1;
        }
      }
      if (ss_evaluate(P, xtx, omega_eff, xty, yty, nz, pb->nonzero_prob, post_conc,
                      pb->obs_scale, work, &cur) != 0) return -3;
      /* sigma^2 ~ InverseGammaWithSampleUpperBound(a_post, beta_post, ub):
       * the VARIANCE draw is clipped at upper_bound (1.2*sd), as TFP does. */
      double g = ci_oracle_gamma(post_conc, pb->seed, chain, uit, CI_SITE_OBSVAR, 0);
      double var = cur.post_scale / g;
      if (var > pb->obs_ub) var = pb->obs_ub;
      obs_scale = sqrt(var);
      /* weights_g ~ N(mean, var * (Omega_g + XtX_g)^{-1}) ; solve L' u = z */
      int na = cur.na;
      for (int i = 0; i < na; ++i)
        zw[i] = ci_oracle_normal(pb->seed, chain, uit, CI_SITE_WEIGHTS, 0, (uint32_t)cur.idx[i]);
      for (int i = na - 1; i >= 0; --i) {
        double s = zw[i];
        for (int k2 = i + 1; k2 < na; ++k2) s -= chol_post[k2 * na + i] * zw[k2];
        zw[i] = s / chol_post[i * na + i];
      }
      for (int j = 0; j < P; ++j) w[j] = 0.0;
      for (int i = 0; i < na; ++i) w[cur.idx[i]] = mean[i] + obs_scale * zw[i];
      for (int t = 0; t < T; ++t) {
        double s = 0.0;
        for (int j = 0; j < P; ++j) s += pb->X[(size_t)t * P + j] * w[j];
        resid[t] = (pb->mask[t] ? 0.0 : pb->y[t]) - s;
      }
    } else {
      for (int t = 0; t < T; ++t) resid[t] = pb->mask[t] ? 0.0 : pb->y[t];
    }

    /* ---- (b) latent path: Durbin-Koopman draw with previous scales ---- */
    m.obs_scale = obs_scale; m.level_scale = level_scale; m.slope_scale = slope_scale;
    for (int k = 0; k < K; ++k) m.drift_scale[k] = drift[k];
    ci_oracle_dk_draw(&m, resid, pb->seed, chain, uit, lat);

    /* ---- (c) scales from the new path ---- */
    {
      double ss_level = 0.0, ss_slope = 0.0;
      for (int t = 0; t + 1 < T; ++t) {
        double dl = lat[(size_t)(t + 1) * d] - lat[(size_t)t * d];
        if (pb->has_slope) {
          dl -= lat[(size_t)t * d + 1];
          double dsl = lat[(size_t)(t + 1) * d + 1] - lat[(size_t)t * d + 1];
          ss_slope += dsl * dsl;
        }
        ss_level += dl * dl;
      }
      level_scale = draw_scale(pb->level_conc, pb->level_scale, pb->level_ub, (double)(T - 1),
                               ss_level, pb->seed, chain, uit, CI_SITE_LEVEL_SCALE, 0);
      if (pb->has_slope)
        slope_scale = draw_scale(pb->slope_conc, pb->slope_scale, pb->slope_ub, (double)(T - 1),
                                 ss_slope, pb->seed, chain, uit, CI_SITE_SLOPE_SCALE, 0);
      for (int k = 0; k < K; ++k) {
        int o = block_offset(&m, k), n = pb->num_seasons[k];
        double ss = 0.0, cnt = 0.0;
        for (int t = 0; t + 1 < T; ++t) {
          if (!pb->season_change[(size_t)k * T + t]) continue;
          double wdr;
          if (n >= 3) wdr = n * (lat[(size_t)t * d + o + 1] - lat[(size_t)(t + 1) * d + o]);
          else wdr = -2.0 * (lat[(size_t)(t + 1) * d + o] + lat[(size_t)t * d + o]);
          ss += wdr * wdr; cnt += 1.0;
        }
        drift[k] = draw_scale(pb->drift_conc, pb->drift_scale, pb->drift_ub, cnt, ss, pb->seed,
                              chain, uit, CI_SITE_DRIFT_SCALE, (uint32_t)k);
      }
      if (P == 0) {
        double ss = 0.0;
        for (int t = 0; t < T; ++t) {
          if (pb->mask[t]) continue;
          double v = pb->y[t] - lat[(size_t)t * d];
          for (int k = 0; k < K; ++k) v -= lat[(size_t)t * d + block_offset(&m, k)];
          ss += v * v;
        }
        obs_scale = draw_scale(pb->obs_conc, pb->obs_scale, pb->obs_ub, n_obs, ss, pb->seed, chain,
                               uit, CI_SITE_OBS_SCALE, 0);
      }
    }

    /* ---- (d) emit ---- */
    if (it >= W) {
      int s = it - W;
      if (out->obs_scale) out->obs_scale[s] = obs_scale;
      if (out->level_scale) out->level_scale[s] = level_scale;
      if (out->slope_scale) out->slope_scale[s] = slope_scale;
      for (int k = 0; k < K; ++k) if (out->drift_scales) out->drift_scales[(size_t)s * K + k] = drift[k];
      for (int j = 0; j < P; ++j) {
        if (out->weights) out->weights[(size_t)s * P + j] = w[j];
        if (out->nonzeros) out->nonzeros[(size_t)s * P + j] = nz[j];
      }
      for (int t = 0; t < T; ++t) {
        double lv = lat[(size_t)t * d];
        double loc = lv;
        if (out->level) out->level[(size_t)s * T + t] = lv;
        if (out->slope) out->slope[(size_t)s * T + t] = pb->has_slope ? lat[(size_t)t * d + 1] : 0.0;
        for (int k = 0; k < K; ++k) {
          double sv = lat[(size_t)t * d + block_offset(&m, k)];
          loc += sv;
          if (out->seasonal) out->seasonal[((size_t)s * T + t) * K + k] = sv;
        }
        for (int j = 0; j < P; ++j) loc += pb->X[(size_t)t * P + j] * w[j];
        pred_acc[t] += loc;
        /* one_step_predictive(use_zero_step_prediction=True): loc + sigma_obs * eps
         * (causalimpact_lib.py:620-631). */
        if (out->trajectories)
          out->trajectories[(size_t)s * T + t] =
              loc + obs_scale * ci_oracle_normal(pb->seed, chain, uit, CI_SITE_PRED, 0, (uint32_t)t);
      }
    }
  }
  if (out->pred_mean)
    for (int t = 0; t < T; ++t) out->pred_mean[t] = pred_acc[t] / (double)(S > 0 ? S : 1);

  free(w); free(lat); free(resid); free(targets); free(pred_acc);
  free(xtx); free(omega); free(omega_eff); free(xty); free(work);
  free(chol_post); free(mean); free(zw); free(nz); free(perm_u); free(perm);
  g_ss_exponent_offset = 1.0;
  return 0;
}

/* ------------------------------------------------------------------ */
/* Row H (extension): score of the Kalman          */
/* log-likelihood and the HMC sampler of csrc/ci_hmc.h, restated in     */
/* float64.  The reference has no HMC path (upstream analogues:          */
/* tfp.sts.fit_with_hmc, tfp.experimental.mcmc.windowed_adaptive_hmc);   */
/* parity with TFP unpinned.  The score is pinned by central differences */
/* of ci_oracle_kalman_loglik (itself pinned to the dense MVN density).  */
/* ------------------------------------------------------------------ */

/* Koopman & Shephard (1992), in the filter-gain form of SURVEY.md Appendix F:
 *   r_{t-1} = Z' v_t/F_t + (I - K_t Z)' T_t' r_t,      r_{T-1} = 0
 *   N_{t-1} = Z'Z/F_t + (I - K_t Z)' T_t' N_t T_t (I - K_t Z),  N_{T-1} = 0
 *   e_t = v_t/F_t - K_t' T_t' r_t = -dl/d data_t                 (observed t)
 *   dl/dH = 1/2 sum_obs (e_t^2 - D_t),  D_t = 1/F_t + K_t' T_t' N_t T_t K_t
 *   dl/dQ_t = 1/2 (r_t r_t' - N_t)  for the disturbance entering step t+1. */
double ci_oracle_loglik_score(const ci_oracle_ssm* m, const double* data, double* e_out,
                              double* g_scales) {
  int d = m->d, T = m->T, K = m->num_blocks;
  kf_store st;
  st.a = NULL; st.P = NULL;
  st.vf = (double*)malloc(sizeof(double) * T);
  st.kf = (double*)malloc(sizeof(double) * T * d);
  /* F_t is needed below: recover it from kf and a second pass would cost a filter; store it */
  double* Fv = (double*)malloc(sizeof(double) * T);
  {
    /* forward pass (kalman_forward with F_t recorded) */
    double a[CI_MAX_D], P[CI_MAX_D * CI_MAX_D], pz[CI_MAX_D];
    double H = m->obs_scale * m->obs_scale;
    initial_moments(m, a, P);
    st.loglik = 0.0;
    for (int t = 0; t < T; ++t) {
      Fv[t] = 0.0;
      if (!m->mask[t]) {
        double v = data[t] - observe(m, a);
        for (int i = 0; i < d; ++i) pz[i] = observe(m, &P[i * d]);
        double F = observe(m, pz) + H;
        Fv[t] = F;
        st.loglik += -0.5 * (log(2.0 * M_PI) + log(F) + v * v / F);
        for (int i = 0; i < d; ++i) {
          double k = pz[i] / F;
          st.kf[(size_t)t * d + i] = k;
          a[i] += k * v;
        }
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) P[i * d + j] -= pz[i] * pz[j] / F;
        st.vf[t] = v / F;
      } else {
        st.vf[t] = 0.0;
        memset(&st.kf[(size_t)t * d], 0, sizeof(double) * d);
      }
      if (t + 1 < T) {
        apply_transition(m, t, a);
        propagate_cov(m, t, P);
      }
    }
  }
  double r[CI_MAX_D], N[CI_MAX_D * CI_MAX_D], z[CI_MAX_D], mk[CI_MAX_D], col[CI_MAX_D];
  memset(r, 0, sizeof(r));
  memset(N, 0, sizeof(N));
  memset(z, 0, sizeof(z));
  z[0] = 1.0;
  for (int k = 0; k < K; ++k) z[block_offset(m, k)] = 1.0;
  double gH = 0.0, gQl = 0.0, gQs = 0.0, gQd[CI_MAX_BLOCKS];
  for (int k = 0; k < K; ++k) gQd[k] = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    if (t + 1 < T) {
      /* disturbance of the transition t -> t+1 */
      gQl += 0.5 * (r[0] * r[0] - N[0]);
      if (m->has_slope) gQs += 0.5 * (r[1] * r[1] - N[1 * d + 1]);
      for (int k = 0; k < K; ++k) {
        if (!m->season_change[(size_t)k * T + t]) continue;
        int o = block_offset(m, k), n = m->num_seasons[k];
        double sr = 0.0, sn = 0.0;
        for (int i = 0; i < n - 1; ++i) {
          sr += r[o + i];
          for (int j = 0; j < n - 1; ++j) sn += N[(o + i) * d + (o + j)];
        }
        gQd[k] += 0.5 * (sr * sr - sn) / ((double)n * n);   /* Q = (sigma_d/n)^2 11' */
      }
      /* r <- T' r ;  N <- T' N T */
      apply_transition_T(m, t, r);
      for (int j = 0; j < d; ++j) {
        for (int i = 0; i < d; ++i) col[i] = N[i * d + j];
        apply_transition_T(m, t, col);
        for (int i = 0; i < d; ++i) N[i * d + j] = col[i];
      }
      for (int i = 0; i < d; ++i) apply_transition_T(m, t, &N[i * d]);
    }
    if (e_out) e_out[t] = 0.0;
    if (!m->mask[t]) {
      const double* kf = &st.kf[(size_t)t * d];
      double F = Fv[t], kr = 0.0, kmk = 0.0;
      for (int i = 0; i < d; ++i) kr += kf[i] * r[i];
      for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += N[i * d + j] * kf[j];
        mk[i] = s;
      }
      for (int i = 0; i < d; ++i) kmk += kf[i] * mk[i];
      double e = st.vf[t] - kr;
      if (e_out) e_out[t] = e;
      gH += 0.5 * (e * e - (1.0 / F + kmk));
      /* r_{t-1} = (I - K Z)' r + Z' v/F  = r + Z' (v/F - K'r) */
      for (int i = 0; i < d; ++i) r[i] += z[i] * e;
      /* N_{t-1} = (I - K Z)' N (I - K Z) + Z'Z/F */
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
          N[i * d + j] += -z[i] * mk[j] - mk[i] * z[j] + z[i] * z[j] * (kmk + 1.0 / F);
    }
  }
  if (g_scales) {
    g_scales[0] = 2.0 * m->obs_scale * gH;
    g_scales[1] = 2.0 * m->level_scale * gQl;
    g_scales[2] = m->has_slope ? 2.0 * m->slope_scale * gQs : 0.0;
    for (int k = 0; k < K; ++k) g_scales[3 + k] = 2.0 * m->drift_scale[k] * gQd[k];
  }
  double ll = st.loglik;
  free(st.vf); free(st.kf); free(Fv);
  return ll;
}

static double clamp30(double v) { return v < -30.0 ? -30.0 : (v > 30.0 ? 30.0 : v); }

void ci_oracle_hmc_windows(int W, int* slow_begin, int* slow_end, int* first_end, int* base) {
  if (W < 20) { *slow_begin = *slow_end = *first_end = W; *base = 0; return; }
  int ib = 75, tb = 50, bw = 25;
  if (ib + tb + bw > W) { ib = (int)(0.15 * W); tb = (int)(0.10 * W); bw = W - ib - tb; }
  *slow_begin = ib; *slow_end = W - tb; *base = bw;
  int e = ib + bw;
  if (e + 2 * bw > *slow_end) e = *slow_end;
  *first_end = e;
}

/* log posterior and gradient at the unconstrained point th (csrc/ci_hmc.h "Target"). */
static double hmc_target(const ci_oracle_hmc_problem* pb, ci_oracle_ssm* m, const double* th,
                         double* g, double* dev /* 3+K+P */, double* resid, double* e) {
  const int T = pb->T, P = pb->P, K = pb->num_blocks;
  const int hs = pb->prior_mode == 1;
  const int off_sc = hs ? 3 * P + 2 : P;
  const int nsc = 2 + (pb->has_slope ? 1 : 0) + K;
  const int dim = off_sc + nsc;
  double hsc[256];
  double* beta = dev + 3 + K;
  for (int j = 0; j < P; ++j) {
    if (hs) {
      hsc[j] = exp(clamp30(th[P + j]) + 0.5 * clamp30(th[2 * P + j]) + clamp30(th[3 * P]) +
                   0.5 * clamp30(th[3 * P + 1])) * pb->hs_scale0;
      beta[j] = th[j] * hsc[j];
    } else {
      beta[j] = th[j];
    }
  }
  /* scale order in theta: obs, level, [slope], drift[K];  dev: obs, level, slope, drift[K] */
  int q = off_sc;
  m->obs_scale = exp(clamp30(th[q++]));
  m->level_scale = exp(clamp30(th[q++]));
  m->slope_scale = pb->has_slope ? exp(clamp30(th[q++])) : 0.0;
  for (int k = 0; k < K; ++k) m->drift_scale[k] = exp(clamp30(th[q++]));
  for (int t = 0; t < T; ++t) {
    double r = 0.0;
    if (!pb->mask[t]) {
      r = pb->y[t];
      for (int j = 0; j < P; ++j) r -= pb->X[(size_t)t * P + j] * beta[j];
    }
    resid[t] = r;
  }
  double gs[3 + CI_MAX_BLOCKS];
  double lp = ci_oracle_loglik_score(m, resid, e, gs);
  double gbeta[256];
  for (int j = 0; j < P; ++j) {
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += pb->X[(size_t)t * P + j] * e[t];
    gbeta[j] = s;
  }
  /* scales: IG(a, b) on sigma^2 + Jacobian of lam = log sigma */
  {
    double sig[3 + CI_MAX_BLOCKS], gsig[3 + CI_MAX_BLOCKS];
    int n = 0;
    sig[n] = m->obs_scale; gsig[n++] = gs[0];
    sig[n] = m->level_scale; gsig[n++] = gs[1];
    if (pb->has_slope) { sig[n] = m->slope_scale; gsig[n++] = gs[2]; }
    for (int k = 0; k < K; ++k) { sig[n] = m->drift_scale[k]; gsig[n++] = gs[3 + k]; }
    for (int k = 0; k < nsc; ++k) {
      double lam = clamp30(th[off_sc + k]), e2 = exp(-2.0 * lam);
      lp += -2.0 * pb->ig_a[k] * lam - pb->ig_b[k] * e2;
      g[off_sc + k] = sig[k] * gsig[k] - 2.0 * pb->ig_a[k] + 2.0 * pb->ig_b[k] * e2;
    }
  }
  if (!hs) {
    for (int i = 0; i < P; ++i) {
      double ob = 0.0;
      for (int k = 0; k < P; ++k) ob += th[k] * pb->omega[(size_t)k * P + i];
      lp += -0.5 * th[i] * ob;
      g[i] = gbeta[i] - ob;
    }
  } else {
    double sgb = 0.0;
    for (int j = 0; j < P; ++j) sgb += gbeta[j] * beta[j];
    for (int j = 0; j < P; ++j) {
      double zj = th[j];
      lp += -0.5 * zj * zj;
      g[j] = gbeta[j] * hsc[j] - zj;
      double u = clamp30(th[P + j]), ee = exp(2.0 * u);
      lp += -0.5 * ee + u;
      g[P + j] = gbeta[j] * beta[j] - ee + 1.0;
      u = clamp30(th[2 * P + j]); ee = exp(-u);
      lp += -0.5 * u - 0.5 * ee;
      g[2 * P + j] = 0.5 * gbeta[j] * beta[j] - 0.5 + 0.5 * ee;
    }
    double u = clamp30(th[3 * P]), ee = exp(2.0 * u);
    lp += -0.5 * ee + u;
    g[3 * P] = sgb - ee + 1.0;
    u = clamp30(th[3 * P + 1]); ee = exp(-u);
    lp += -0.5 * u - 0.5 * ee;
    g[3 * P + 1] = 0.5 * sgb - 0.5 + 0.5 * ee;
  }
  if (!(lp == lp) || lp > 1e300 || lp < -1e300) {
    lp = -INFINITY;
    for (int i = 0; i < dim; ++i) g[i] = 0.0;
  }
  return lp;
}

int ci_oracle_hmc_dim(const ci_oracle_hmc_problem* pb) {
  return (pb->prior_mode == 1 ? 3 * pb->P + 2 : pb->P) + 2 + (pb->has_slope ? 1 : 0) + pb->num_blocks;
}

double ci_oracle_hmc_logp(const ci_oracle_hmc_problem* pb, const double* theta, double* grad) {
  ci_oracle_ssm m;
  memset(&m, 0, sizeof(m));
  m.T = pb->T; m.has_slope = pb->has_slope; m.num_blocks = pb->num_blocks;
  for (int k = 0; k < pb->num_blocks; ++k) m.num_seasons[k] = pb->num_seasons[k];
  m.d = ci_oracle_state_dim(pb->has_slope, pb->num_blocks, pb->num_seasons);
  m.mask = pb->mask; m.season_change = pb->season_change;
  m.init_level_loc = pb->init_level_loc; m.init_level_scale = pb->init_level_scale;
  m.init_slope_scale = pb->init_slope_scale; m.init_seasonal_scale = pb->init_seasonal_scale;
  double dev[3 + CI_MAX_BLOCKS + 256], g[1024];
  double* resid = (double*)malloc(sizeof(double) * pb->T);
  double* e = (double*)malloc(sizeof(double) * pb->T);
  double lp = hmc_target(pb, &m, theta, g, dev, resid, e);
  if (grad) memcpy(grad, g, sizeof(double) * ci_oracle_hmc_dim(pb));
  free(resid); free(e);
  return lp;
}

int ci_oracle_fit_hmc(const ci_oracle_hmc_problem* pb, double* draws, double* accept_rate,
                      double* step_size) {
  const int T = pb->T, P = pb->P, K = pb->num_blocks, W = pb->num_warmup, S = pb->num_results;
  if (T < 1 || P < 0 || P > 255 || K < 0 || K > CI_MAX_BLOCKS) return -1;
  const int hs = pb->prior_mode == 1;
  const int off_sc = hs ? 3 * P + 2 : P;
  const int nsc = 2 + (pb->has_slope ? 1 : 0) + K;
  const int dim = off_sc + nsc;
  if (dim > 1024) return -2;
  ci_oracle_ssm m;
  memset(&m, 0, sizeof(m));
  m.T = T; m.has_slope = pb->has_slope; m.num_blocks = K;
  for (int k = 0; k < K; ++k) m.num_seasons[k] = pb->num_seasons[k];
  m.d = ci_oracle_state_dim(pb->has_slope, K, pb->num_seasons);
  if (m.d > CI_MAX_D) return -3;
  m.mask = pb->mask; m.season_change = pb->season_change;
  m.init_level_loc = pb->init_level_loc; m.init_level_scale = pb->init_level_scale;
  m.init_slope_scale = pb->init_slope_scale; m.init_seasonal_scale = pb->init_seasonal_scale;
  const uint32_t chain = (uint32_t)pb->chain;
  double theta[1024], grad[1024], th[1024], g[1024], mom[1024], imass[1024];
  double wmean[1024], wm2[1024], dev[3 + CI_MAX_BLOCKS + 256];
  double* resid = (double*)malloc(sizeof(double) * T);
  double* e = (double*)malloc(sizeof(double) * T);
  for (int i = 0; i < dim; ++i) {
    double v = i >= off_sc ? pb->init_log[i - off_sc] : 0.0;
    th[i] = pb->init ? pb->init[i]
                     : v + 0.01 * ci_oracle_normal(pb->seed, chain, 0u, CI_SITE_HMC_INIT, 0, (uint32_t)i);
    imass[i] = 1.0;
    wmean[i] = 0.0; wm2[i] = 0.0;
  }
  double lp_cur = hmc_target(pb, &m, th, g, dev, resid, e);
  memcpy(theta, th, sizeof(double) * dim);
  memcpy(grad, g, sizeof(double) * dim);
  double eps = pb->eps0, mu = log(10.0 * pb->eps0), hbar = 0.0, log_eps_bar = 0.0, t_da = 0.0;
  const double gamma_da = 0.05, t0_da = 10.0, kappa_da = 0.75;
  int slow_begin, slow_end, win_end, win_size;
  ci_oracle_hmc_windows(W, &slow_begin, &slow_end, &win_end, &win_size);
  double wn = 0.0, accepted = 0.0;
  for (int it = 0; it < W + S; ++it) {
    double kin = 0.0;
    for (int i = 0; i < dim; ++i) {
      double z = ci_oracle_normal(pb->seed, chain, (uint32_t)it, CI_SITE_HMC_MOMENTUM, 0, (uint32_t)i);
      mom[i] = z / sqrt(imass[i]);
      th[i] = theta[i];
      g[i] = grad[i];
      kin += 0.5 * mom[i] * mom[i] * imass[i];
    }
    double h0 = -lp_cur + kin, lp_new = lp_cur;
    for (int l = 0; l < pb->num_leapfrog; ++l) {
      for (int i = 0; i < dim; ++i) {
        mom[i] += 0.5 * eps * g[i];
        th[i] += eps * imass[i] * mom[i];
      }
      lp_new = hmc_target(pb, &m, th, g, dev, resid, e);
      for (int i = 0; i < dim; ++i) mom[i] += 0.5 * eps * g[i];
    }
    double k1 = 0.0;
    for (int i = 0; i < dim; ++i) k1 += 0.5 * mom[i] * mom[i] * imass[i];
    double h1 = -lp_new + k1;
    int fin = (h1 == h1) && h1 < 1e300 && h1 > -1e300;
    double log_acc = fin ? h0 - h1 : -INFINITY;
    double acc_prob = fin ? exp(log_acc < 0.0 ? log_acc : 0.0) : 0.0;
    double u = ci_oracle_uniform(pb->seed, chain, (uint32_t)it, CI_SITE_HMC_ACCEPT, 0, 0);
    int take = log(u) < log_acc;
    if (take) {
      memcpy(theta, th, sizeof(double) * dim);
      memcpy(grad, g, sizeof(double) * dim);
      lp_cur = lp_new;
    }
    if (it < W) {
      t_da += 1.0;
      hbar = (1.0 - 1.0 / (t_da + t0_da)) * hbar + (pb->target_accept - acc_prob) / (t_da + t0_da);
      double log_eps = mu - sqrt(t_da) / gamma_da * hbar;
      double eta = pow(t_da, -kappa_da);
      log_eps_bar = eta * log_eps + (1.0 - eta) * log_eps_bar;
      eps = exp(log_eps);
      if (it >= slow_begin && it < slow_end) {
        wn += 1.0;
        for (int i = 0; i < dim; ++i) {
          double x = theta[i], d0 = x - wmean[i];
          wmean[i] += d0 / wn;
          wm2[i] += d0 * (x - wmean[i]);
        }
        if (it + 1 == win_end) {
          for (int i = 0; i < dim; ++i) {
            if (wn >= 2.0) {
              double var = wm2[i] / (wn - 1.0);
              double v = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
              if (v == v && v < 1e300 && v > 0.0) imass[i] = v;
            }
            wmean[i] = 0.0; wm2[i] = 0.0;
          }
          wn = 0.0;
          eps = exp(log_eps_bar);
          mu = log(10.0 * eps); hbar = 0.0; log_eps_bar = 0.0; t_da = 0.0;
          if (win_end < slow_end) {
            win_size *= 2;
            int en = win_end + win_size;
            if (en + 2 * win_size > slow_end) en = slow_end;
            win_end = en;
          }
        }
      }
      if (it == W - 1 && t_da > 0.0) eps = exp(log_eps_bar);
    } else {
      accepted += take ? 1.0 : 0.0;
      /* rows (sigma_obs, sigma_level, sigma_slope, drift[K], beta[P]) */
      double* o = draws + (size_t)(it - W) * (3 + K + P);
      int q = off_sc;
      o[0] = exp(clamp30(theta[q++]));
      o[1] = exp(clamp30(theta[q++]));
      o[2] = pb->has_slope ? exp(clamp30(theta[q++])) : 0.0;
      for (int k = 0; k < K; ++k) o[3 + k] = exp(clamp30(theta[q++]));
      for (int j = 0; j < P; ++j) {
        double b = theta[j];
        if (hs)
          b *= exp(clamp30(theta[P + j]) + 0.5 * clamp30(theta[2 * P + j]) + clamp30(theta[3 * P]) +
                   0.5 * clamp30(theta[3 * P + 1])) * pb->hs_scale0;
        o[3 + K + j] = b;
      }
    }
  }
  if (accept_rate) *accept_rate = accepted / (double)(S > 0 ? S : 1);
  if (step_size) *step_size = eps;
  free(resid); free(e);
  return 0;
}

/* Latent path + posterior-predictive trajectory of every retained HMC draw, as the device does
 * after the chain (latents_kernel): Durbin-Koopman draw with iteration = draw index, then
 * loc = Z x + X beta, trajectory = loc + sigma_obs * eps (causalimpact_lib.py:620-631).
 * draws rows as ci_oracle_fit_hmc writes them; outputs [S*T] (any may be NULL). */
int ci_oracle_hmc_latents(const ci_oracle_hmc_problem* pb, const double* draws, int S,
                          double* level, double* slope, double* loc_out, double* traj) {
  const int T = pb->T, P = pb->P, K = pb->num_blocks;
  ci_oracle_ssm m;
  memset(&m, 0, sizeof(m));
  m.T = T; m.has_slope = pb->has_slope; m.num_blocks = K;
  for (int k = 0; k < K; ++k) m.num_seasons[k] = pb->num_seasons[k];
  m.d = ci_oracle_state_dim(pb->has_slope, K, pb->num_seasons);
  if (m.d > CI_MAX_D) return -3;
  m.mask = pb->mask; m.season_change = pb->season_change;
  m.init_level_loc = pb->init_level_loc; m.init_level_scale = pb->init_level_scale;
  m.init_slope_scale = pb->init_slope_scale; m.init_seasonal_scale = pb->init_seasonal_scale;
  const int d = m.d;
  double* resid = (double*)malloc(sizeof(double) * T);
  double* xw = (double*)malloc(sizeof(double) * T);
  double* lat = (double*)malloc(sizeof(double) * T * d);
  for (int s = 0; s < S; ++s) {
    const double* o = draws + (size_t)s * (3 + K + P);
    m.obs_scale = o[0]; m.level_scale = o[1]; m.slope_scale = o[2];
    for (int k = 0; k < K; ++k) m.drift_scale[k] = o[3 + k];
    for (int t = 0; t < T; ++t) {
      double v = 0.0;
      for (int j = 0; j < P; ++j) v += pb->X[(size_t)t * P + j] * o[3 + K + j];
      xw[t] = v;
      resid[t] = pb->mask[t] ? 0.0 : pb->y[t] - v;
    }
    ci_oracle_dk_draw(&m, resid, pb->seed, (uint32_t)pb->chain, (uint32_t)s, lat);
    for (int t = 0; t < T; ++t) {
      double lc = observe(&m, &lat[(size_t)t * d]) + xw[t];
      if (level) level[(size_t)s * T + t] = lat[(size_t)t * d];
      if (slope) slope[(size_t)s * T + t] = pb->has_slope ? lat[(size_t)t * d + 1] : 0.0;
      if (loc_out) loc_out[(size_t)s * T + t] = lc;
      if (traj)
        traj[(size_t)s * T + t] = lc + o[0] * ci_oracle_normal(pb->seed, (uint32_t)pb->chain,
                                                               (uint32_t)s, CI_SITE_PRED, 0, (uint32_t)t);
    }
  }
  free(resid); free(xw); free(lat);
  return 0;
}


/* ---- cpu_baseline: whole chains, OpenMP over chains ---- */
#ifdef _OPENMP
#include <omp.h>
#endif
int ci_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ci_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int ci_oracle_fit_gibbs_chains(const ci_oracle_problem* pb, int first_chain, int n_chains) {
  int failed = 0;
  const size_t T = (size_t)pb->T, S = (size_t)pb->num_results;
  const size_t P = (size_t)(pb->P > 0 ? pb->P : 1), K = (size_t)(pb->num_blocks > 0 ? pb->num_blocks : 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed)
#endif
  for (int c = 0; c < n_chains; ++c) {
    ci_oracle_problem q = *pb;
    q.chain = first_chain + c;
    ci_oracle_outputs o;
    memset(&o, 0, sizeof(o));
    o.obs_scale = (double*)malloc(sizeof(double) * S);
    o.level_scale = (double*)malloc(sizeof(double) * S);
    o.slope_scale = (double*)malloc(sizeof(double) * S);
    o.drift_scales = (double*)malloc(sizeof(double) * S * K);
    o.weights = (double*)malloc(sizeof(double) * S * P);
    o.level = (double*)malloc(sizeof(double) * S * T);
    o.slope = (double*)malloc(sizeof(double) * S * T);
    o.seasonal = pb->num_blocks > 0 ? (double*)malloc(sizeof(double) * S * T * K) : NULL;
    o.pred_mean = (double*)malloc(sizeof(double) * T);
    o.trajectories = (double*)malloc(sizeof(double) * S * T);
    if (ci_oracle_fit_gibbs(&q, &o) != 0) failed += 1;
    free(o.obs_scale); free(o.level_scale); free(o.slope_scale); free(o.drift_scales);
    free(o.weights); free(o.level); free(o.slope); free(o.seasonal); free(o.pred_mean);
    free(o.trajectories);
  }
  return failed;
}
