// ci_hmc.h -- Hamiltonian Monte Carlo over the model's parameters, entirely on the device.
// EXTENSION (BASELINE config "64 HMC chains sharded across 8 GPUs"):
// the reference is Gibbs-only; upstream analogues tfp.sts.fit_with_hmc and
// tfp.experimental.mcmc.windowed_adaptive_hmc.  Parity with TFP: unpinned; the float64 CPU
// restatement of exactly this sampler is oracle/ci_oracle.c::ci_oracle_fit_hmc (same random stream, so the two
// agree draw for draw until float32 round-off in the score separates the trajectories).
//
// One 256-thread workgroup per chain runs ALL warm-up and sampling iterations: every leapfrog
// step evaluates the Kalman-filter log-likelihood and its score with the time-parallel scans of
// loglik_grad_block (ci_kernels.h); the momentum / position updates, the prior terms, the
// Metropolis test and the adaptation are a few float64 lanes.
//
// Target.  Scales enter as lam_k = log sigma_k with the reference's inverse-gamma variance priors
// (causalimpact_lib.py:424-443) and the Jacobian:  -2 a_k lam_k - b_k exp(-2 lam_k).
// Regression prior (prior_mode):
//   0  Gaussian slab of the reference's spike-and-slab prior (:451-453): theta = (beta[P], lam),
//      log p = l(sigma, beta) - 1/2 beta' Omega beta + ...
//   1  horseshoe, the parameterisation of tfp.sts.SparseLinearRegression [UPSTREAM-RECALL]:
//      beta_j = z_j * ln_j * sqrt(lv_j) * gn * sqrt(gv) * s0,   z_j ~ N(0,1),
//      ln_j, gn ~ HalfNormal(1),  lv_j, gv ~ InverseGamma(1/2, 1/2)  (half-Cauchy local and
//      global scales as scale mixtures), s0 = weights_prior_scale;
//      theta = (z[P], log ln[P], log lv[P], log gn, log gv, lam).
//
// Adaptation ("windowed", the three-stage scheme of Stan / windowed_adaptive_hmc): dual averaging
// of the step size on every warm-up iteration (Nesterov 2009; Hoffman & Gelman 2014); an initial
// fast buffer (step size only), a sequence of doubling slow windows at whose ends the diagonal
// inverse mass is set to the regularised per-coordinate variance of that window's positions and
// the dual averaging restarts, and a terminal fast buffer (step size only).  hmc_windows() below
// is the schedule; the oracle restates it.
#pragma once
#include "ci_kernels.h"

// hmc_kernel's lambdas are inlined by force (a call moves every captured variable and the kernel
// arguments to scratch: 432 bytes per lane at L = 4) -- except in the L = 16 build, which sits at
// the register limit and is better off with the call.
#if defined(CI_L) && CI_L >= 16
#define CI_HMC_INLINE
#else
#define CI_HMC_INLINE __attribute__((always_inline))
#endif

namespace ci {

constexpr int HMC_MAXDIM = 3 * HMC_MAXP + 5;

struct HmcWindows { int slow_begin, slow_end, first_end, base; };
// Warm-up iterations [0, slow_begin) and [slow_end, W) adapt the step size only; mass windows tile
// [slow_begin, slow_end): the first ends at first_end, each following one is twice as long, and a
// window that would leave less than its own doubled length before slow_end absorbs the rest.
__host__ __device__ inline HmcWindows hmc_windows(int W) {
  HmcWindows w;
  if (W < 20) { w.slow_begin = w.slow_end = w.first_end = W; w.base = 0; return w; }
  int ib = 75, tb = 50, bw = 25;
  if (ib + tb + bw > W) { ib = (int)(0.15 * W); tb = (int)(0.10 * W); bw = W - ib - tb; }
  w.slow_begin = ib;
  w.slow_end = W - tb;
  w.base = bw;
  int e = ib + bw;
  if (e + 2 * bw > w.slow_end) e = w.slow_end;
  w.first_end = e;
  return w;
}

struct HmcArgs {
  int T, P, C, W, S, n_leap, chain_offset, x_in_lds, prior_mode;
  uint32_t seed0, seed1;
  const float* y;
  const uint8_t* mask;
  const float* Xt;
  const double* omega;      // [P, P]
  double ig_a[3], ig_b[3];  // inverse-gamma (concentration, scale) of sigma^2: obs, level, slope
  double init_log[3];       // log of the initial scales (causalimpact_lib.py:566-572)
  double hs_scale0;         // horseshoe: weights_prior_scale
  float a1, p10, p11;
  double target_accept, eps0;
  const double* init;       // optional [C, dim] unconstrained starting points (e.g. draws of a
                            // fitted surrogate posterior); NULL = the Gibbs sampler's initial state
  int legacy_driver;        // tests: 1 = the round 2-4 driver (five barriers around every score), whatever dim
  long long* prof;          // tools/exp_hmc_phases.py: 32 cycle counters of chain 0's thread 0, or NULL
  double* draws;            // [C, S, 3 + P]  (sigma_obs, sigma_level, sigma_slope, beta)
  double* accept_rate;      // [C]
  double* step_size;        // [C]
};

__host__ __device__ inline int hmc_dim(int P, int D, int prior_mode) {
  return (prior_mode == 1 ? 3 * P + 2 : P) + (D == 2 ? 3 : 2);
}

// float64 wave total on the DPP crossbar (row_shr 1, 2, 4, 8, then row_bcast:15 / :31): lane 63
// holds the sum, broadcast through v_readlane.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_add_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v = dpp_add_d<0x111, 0xF>(v);
  v = dpp_add_d<0x112, 0xF>(v);
  v = dpp_add_d<0x114, 0xF>(v);
  v = dpp_add_d<0x118, 0xF>(v);
  v = dpp_add_d<0x142, 0xA>(v);
  v = dpp_add_d<0x143, 0xC>(v);
  return readlane_d(v, 63);
}

__device__ __forceinline__ double clamp30(double v) { return v < -30.0 ? -30.0 : (v > 30.0 ? 30.0 : v); }

// WIDE: the fit has more than MAXP design columns -- its arrays are sized for HMC_MAXP.  Two builds
// because the sizes are compile-time constants (addresses as immediates): sized for 128 columns the
// arrays push cfg3's design matrix beyond the 64 KB an LDS instruction reaches with an immediate
// offset (7.7 -> 9.2 us per leapfrog), sized at run time they cost ten more live SGPRs (7.8).
template <int D, int L, bool WIDE>
__global__ __launch_bounds__(NT) void hmc_kernel(HmcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = a.P, T = a.T;
  constexpr int NSC = (D == 2) ? 3 : 2;          // number of scales in the parameter vector
  const bool hs = a.prior_mode == 1;
  const int off_sc = hs ? 3 * P + 2 : P;         // where the log scales start
  const int dim = off_sc + NSC;
  float* slots = (float*)smem_h;                 // 3 * NW * 16
  float* part = slots + 3 * NW * 16;             // NW * (P + 4)
  double* dbl = (double*)(smem_h + (((3 * NW * 16 + NW * (P + 4)) * sizeof(float) + 15) & ~(size_t)15));
  constexpr int MP = WIDE ? HMC_MAXP : MAXP;
  constexpr int dimp = 3 * MP + 5 + ((3 * MP + 5) & 1), devp = MP + 3 + ((MP + 3) & 1), hscp = MP + (MP & 1);
  double* theta = dbl;                 // current position (unconstrained)                        [dimp]
  double* grad = theta + dimp;         // its gradient
  double* th = grad + dimp;            // trajectory position
  double* g = th + dimp;               // trajectory gradient
  double* mom = g + dimp;              // trajectory momentum
  double* imass = mom + dimp;          // inverse mass (diagonal)
  double* dev = imass + dimp;          // device layout of th: (s_obs, s_level, s_slope, beta)   [devp]
  double* gdev = dev + devp;           // score in device layout                                 [devp]
  double* hsc = gdev + devp;           // horseshoe: d beta_j / d z_j                            [hscp]
  double* sc = hsc + hscp;             // scalars: [0] ll, [1] lp of the trajectory, [2] lp current
  // feature-major design matrix, zero padded to NT * L columns, resident in LDS for the whole fit
  // (every leapfrog step reads it twice: residual and d l / d beta)
  double* pre = sc + 8;                // theta-only pieces of the prior terms                 [dimp]
  constexpr int TPAD = NT * L;
  float* Xs = (float*)(pre + dimp);
  const bool x_in_lds = a.x_in_lds != 0;
  if (x_in_lds) {
    for (int j = 0; j < P; ++j)
      for (int t = tid; t < TPAD; t += NT) Xs[j * TPAD + t] = t < T ? a.Xt[(size_t)j * T + t] : 0.f;
  }
  const int chain = blockIdx.x;
  Rng rng{a.seed0, a.seed1, (uint32_t)(a.chain_offset + chain)};

  // d beta_j / d z_j of the horseshoe at the unconstrained point v
  auto hs_scale = [&](const double* v, int j) CI_HMC_INLINE {
    return exp(clamp30(v[P + j]) + 0.5 * clamp30(v[2 * P + j]) + clamp30(v[3 * P]) +
               0.5 * clamp30(v[3 * P + 1])) * a.hs_scale0;
  };

  // The per-coordinate work of a leapfrog step -- momentum / position update, the device layout of
  // the new position, prior terms, second momentum half step -- is a few float64 lanes.  When the
  // whole parameter vector fits one wavefront (dim <= 64: cfg3 has 14 coordinates, its horseshoe
  // form 38) lane i of wave 0 owns coordinate i, and what the prior terms need of the POSITION alone
  // (Omega th, exp(-2 lam) = a float64 division, the horseshoe's exponentials: 1.7k of the 2.4k cycles
  // they cost after the score) is computed by waves 1 and 2 while wave 0 sums the waves' shares of
  // the score in float64 -- beside the score's tail instead of after it.  The slab prior's Omega column stays in registers for
  // P <= 16 (the loop was 11 dependent LDS + L2 round trips), the observations too (a round trip to
  // L2 per leapfrog step).  Same expressions in the same order as the general driver below: the two
  // give the same bits (tests/test_gpu_hmc.py); phase cycles: tools/exp_hmc_phases.py.
  const bool fused = dim <= 64 && a.legacy_driver == 0;
  const bool om_regs = !hs && P <= 16 && L <= 4;      // (the L = 8, 16 builds have no registers to spare)
  double om[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) om[k] = (om_regs && lane < P && k < P) ? a.omega[k * P + lane] : 0.0;
  double my_iga = 0.0, my_igb = 0.0;
  // (a select chain: a run-time index into the by-value arguments would move them to scratch)
  const double iga0 = a.ig_a[0], iga1 = a.ig_a[1], iga2 = a.ig_a[2];
  const double igb0 = a.ig_b[0], igb1 = a.ig_b[1], igb2 = a.ig_b[2];
  const double il0 = a.init_log[0], il1 = a.init_log[1], il2 = a.init_log[2];
  auto pick3 = [](double v0, double v1, double v2, int k) { return k == 0 ? v0 : (k == 1 ? v1 : v2); };
  if (lane >= off_sc && lane < dim && dim <= 64) {
    my_iga = pick3(iga0, iga1, iga2, lane - off_sc);
    my_igb = pick3(igb0, igb1, igb2, lane - off_sc);
  }

  // device layout of th (threads 0 .. P-1, 64 .. 66, 128; or, fused, the lanes of wave 0)
  auto prep_dev = [&](bool on_wave0) CI_HMC_INLINE {
    const int jb = on_wave0 ? lane : tid;
    if (jb < P && (!on_wave0 || wave == 0)) {
      if (hs) {
        const double s = hs_scale(th, jb);
        hsc[jb] = s;
        dev[3 + jb] = th[jb] * s;
      } else {
        dev[3 + jb] = th[jb];
      }
    }
    if (on_wave0) {
      if (lane >= off_sc && lane < dim) dev[lane - off_sc] = exp(clamp30(th[lane]));
      if (D == 1 && lane == 0) dev[2] = 0.0;
    } else {
      if (tid >= 64 && tid < 64 + NSC) dev[tid - 64] = exp(clamp30(th[off_sc + tid - 64]));
      if (D == 1 && tid == 128) dev[2] = 0.0;
    }
  };
  Prof hp;
  hp.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && tid == 0);
  // the thread's own observations stay in registers for the whole fit (the L = 16 build, at the
  // register limit already, reloads them per evaluation)
  float yv[L];
  uint32_t maskbits;
  if constexpr (L <= 8) loglik_load_obs<L>(T, a.y, a.mask, tid, yv, maskbits);
  auto score = [&]() CI_HMC_INLINE {
    if constexpr (L > 8) loglik_load_obs<L>(T, a.y, a.mask, tid, yv, maskbits);
    if (x_in_lds)
      loglik_grad_block_obs<D, L>(T, P, yv, maskbits, Xs, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                                  gdev, tid, lane, wave, TPAD, &hp);
    else
      loglik_grad_block_obs<D, L>(T, P, yv, maskbits, a.Xt, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                                  gdev, tid, lane, wave, 0, &hp);
  };
  // Omega th for coordinate i (slab prior): the column from registers (P <= 16, i == lane) or L2
  auto omega_dot = [&](int i) CI_HMC_INLINE {
    double ob = 0.0;
    if (om_regs) {
      double tk[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) tk[k] = th[k < P ? k : 0];
#pragma unroll
      for (int k = 0; k < 16; ++k) ob = k < P ? fma(tk[k], om[k], ob) : ob;
    } else {
      for (int k = 0; k < P; ++k) ob = fma(th[k], a.omega[k * P + i], ob);
    }
    return ob;
  };
  // fused driver: the expensive pieces of the prior terms that need the position only -- Omega th,
  // exp(-2 lam) (a float64 division), the horseshoe's exponentials -- by waves 1 and 2 while wave 0
  // finishes the score (its float64 block sums); prior_terms(true) reads them back.  The same expressions
  // as prior_terms(false) evaluates in place.
  auto prior_pre = [&]() CI_HMC_INLINE {
    const int i = lane;
    if (wave == 1 && i < off_sc) {
      double v = 0.0;
      if (!hs) v = omega_dot(i);
      else if (i < P) v = 0.0;
      else if (i < 2 * P) v = exp(2.0 * clamp30(th[i]));
      else if (i < 3 * P) v = exp(-clamp30(th[i]));
      else if (i == 3 * P) v = exp(2.0 * clamp30(th[i]));
      else v = exp(-clamp30(th[i]));
      pre[i] = v;
    }
    if (wave == 2 && i >= off_sc && i < dim) {
      const double d = dev[i - off_sc];                    // exp(lam_k), laid out before the score
      pre[i] = 1.0 / (d * d);
    }
  };
  // prior terms + Jacobians: g, sc[1] (wave 0)
  auto prior_terms = [&](const bool use_pre, const bool ob_pre) CI_HMC_INLINE {
    double sgb = 0.0;                    // horseshoe: sum_j beta_j dl/dbeta_j
    if (hs) {
      for (int j = lane; j < P; j += 64) sgb = fma(gdev[3 + j], dev[3 + j], sgb);
      sgb = wave_sum_d(sgb);
    }
    hp.tick(15);
    double contrib = 0.0;
    for (int i = lane; i < dim; i += 64) {
      double ci, gi;
      if (i >= off_sc) {
        const int k = i - off_sc;
        const double lam = clamp30(th[i]);
        const double e2 = use_pre ? pre[i] : 1.0 / (dev[k] * dev[k]);   // exp(-2 lam): dev[k] = exp(lam)
        const double iga = dim <= 64 ? my_iga : pick3(iga0, iga1, iga2, k), igb = dim <= 64 ? my_igb : pick3(igb0, igb1, igb2, k);
        ci = -2.0 * iga * lam - igb * e2;
        gi = dev[k] * gdev[k] - 2.0 * iga + 2.0 * igb * e2;
      } else if (!hs) {
        const double ob = (use_pre || ob_pre) ? pre[i] : omega_dot(i);
        ci = -0.5 * th[i] * ob;
        gi = gdev[3 + i] - ob;
      } else if (i < P) {                  // z_j ~ N(0, 1)
        const double z = th[i];
        ci = -0.5 * z * z;
        gi = gdev[3 + i] * hsc[i] - z;
      } else if (i < 2 * P) {              // log of a HalfNormal(1) local scale
        const int j = i - P;
        const double e = use_pre ? pre[i] : exp(2.0 * clamp30(th[i]));
        ci = -0.5 * e + clamp30(th[i]);
        gi = gdev[3 + j] * dev[3 + j] - e + 1.0;
      } else if (i < 3 * P) {              // log of an InverseGamma(1/2, 1/2) local variance
        const int j = i - 2 * P;
        const double u = clamp30(th[i]);
        const double e = use_pre ? pre[i] : exp(-u);
        ci = -0.5 * u - 0.5 * e;
        gi = 0.5 * gdev[3 + j] * dev[3 + j] - 0.5 + 0.5 * e;
      } else if (i == 3 * P) {             // log of the HalfNormal(1) global scale
        const double e = use_pre ? pre[i] : exp(2.0 * clamp30(th[i]));
        ci = -0.5 * e + clamp30(th[i]);
        gi = sgb - e + 1.0;
      } else {                             // log of the InverseGamma(1/2, 1/2) global variance
        const double u = clamp30(th[i]);
        const double e = use_pre ? pre[i] : exp(-u);
        ci = -0.5 * u - 0.5 * e;
        gi = 0.5 * sgb - 0.5 + 0.5 * e;
      }
      contrib += ci;
      g[i] = gi;
    }
    hp.tick(16);
    double lp = sc[0] + wave_sum_d(contrib);
    hp.tick(17);
    const bool bad = !(lp == lp) || lp > 1e300 || lp < -1e300;
    if (bad) {
      lp = -INFINITY;
      for (int i = lane; i < dim; i += 64) g[i] = 0.0;
    }
    if (lane == 0) sc[1] = lp;
  };
  // log posterior and gradient at th -> sc[1], g   (all threads; contains barriers)
  // General driver with many columns: Omega th by one thread per coordinate BEFORE the score (wave 0
  // alone would walk its 64 + 64 + ... coordinates through P dependent L2 loads each after it);
  // the same fma chain in the same order as omega_dot.
  const bool ob_all = !hs && P > 16;
  auto target = [&]() CI_HMC_INLINE {
    prep_dev(false);
    if (ob_all && tid < P) pre[tid] = omega_dot(tid);
    __syncthreads();
    score();
    __syncthreads();
    if (wave == 0) prior_terms(false, ob_all);
    __syncthreads();
  };

  // ---- initial state: the Gibbs sampler's initial scales, zero weights, a little jitter
  // (horseshoe: all auxiliary log scales 0, z = 0)
  if (tid < dim) {
    double v = 0.0;
    if (tid >= off_sc) v = pick3(il0, il1, il2, tid - off_sc);
    if (a.init) th[tid] = a.init[(size_t)chain * dim + tid];
    else th[tid] = v + 0.01 * normal_d(rng, 0u, SITE_HMC_INIT, 0, (uint32_t)tid);
    imass[tid] = 1.0;
  }
  __syncthreads();
  target();
  if (tid < dim) { theta[tid] = th[tid]; grad[tid] = g[tid]; }
  if (tid == 0) sc[2] = sc[1];
  __syncthreads();

  // adaptation state (every thread carries the same scalars; thread i owns coordinate i's Welford)
  double eps = a.eps0, mu = log(10.0 * a.eps0), hbar = 0.0, log_eps_bar = 0.0, t_da = 0.0;
  const double gamma_da = 0.05, t0_da = 10.0, kappa_da = 0.75;
  const HmcWindows wnd = hmc_windows(a.W);
  int win_end = wnd.first_end, win_size = wnd.base;
  double wn = 0.0, wmean = 0.0, wm2 = 0.0;
  double accepted = 0.0;
  const int n_iter = a.W + a.S;
  for (int it = 0; it < n_iter; ++it) {
    // momentum ~ N(0, M), Hamiltonian at the start
    if (tid < dim) {
      const double z = normal_d(rng, (uint32_t)it, SITE_HMC_MOMENTUM, 0, (uint32_t)tid);
      mom[tid] = z / sqrt(imass[tid]);
      th[tid] = theta[tid];
      g[tid] = grad[tid];
    }
    __syncthreads();
    double h0 = 0.0;
    if (wave == 0) {
      double kin = 0.0;
      for (int i = lane; i < dim; i += 64) kin = fma(0.5 * mom[i] * mom[i], imass[i], kin);
      h0 = -sc[2] + wave_sum_d(kin);
    }
    hp.tick(7);
    if (fused) {
      for (int l = 0; l < a.n_leap; ++l) {
        if (wave == 0) {
          if (lane < dim) {
            const double ph = mom[lane] + 0.5 * eps * g[lane];
            mom[lane] = ph;
            th[lane] += eps * imass[lane] * ph;
          }
          wave_sync();                     // the new position, before the cross-lane reads below
          prep_dev(true);
        }
        hp.tick(0);
        __syncthreads();                   // position in device layout
        hp.tick(1);
        score();                           // (ends with wave 0 summing the waves' shares in float64 ...
        if (wave != 0) prior_pre();        //  ... while waves 1 and 2 prepare the prior terms)
        __syncthreads();                   // ll, score, prior pieces in LDS
        hp.tick(2);
        if (wave == 0) {
          prior_terms(true, false);
          hp.tick(3);
          wave_sync();                     // g, sc[1]
          if (lane < dim) mom[lane] += 0.5 * eps * g[lane];
        }
        hp.tick(4);
      }
      if (wave == 0) wave_sync();
    } else {
      for (int l = 0; l < a.n_leap; ++l) {
        if (tid < dim) {
          const double ph = mom[tid] + 0.5 * eps * g[tid];
          mom[tid] = ph;
          th[tid] += eps * imass[tid] * ph;
        }
        __syncthreads();
        target();
        if (tid < dim) mom[tid] += 0.5 * eps * g[tid];
        __syncthreads();
      }
    }
    // Metropolis test (wave 0), broadcast through LDS
    if (wave == 0) {
      double k1 = 0.0;
      for (int i = lane; i < dim; i += 64) k1 = fma(0.5 * mom[i] * mom[i], imass[i], k1);
      const double h1 = -sc[1] + wave_sum_d(k1);
      const bool fin = (h1 == h1) && h1 < 1e300 && h1 > -1e300;
      const double log_acc = fin ? h0 - h1 : -INFINITY;
      const double acc_prob = fin ? exp(log_acc < 0.0 ? log_acc : 0.0) : 0.0;
      const double u = uniform_d(rng, (uint32_t)it, SITE_HMC_ACCEPT, 0, 0);
      const bool take = log(u) < log_acc;
      if (lane == 0) { sc[3] = take ? 1.0 : 0.0; sc[4] = acc_prob; }
    }
    __syncthreads();
    const bool take = sc[3] != 0.0;
    const double acc_prob = sc[4];
    if (take) {
      if (tid < dim) { theta[tid] = th[tid]; grad[tid] = g[tid]; }
      if (tid == 0) sc[2] = sc[1];
    }
    __syncthreads();
    if (it < a.W) {
      // dual averaging of the log step size
      t_da += 1.0;
      hbar = (1.0 - 1.0 / (t_da + t0_da)) * hbar + (a.target_accept - acc_prob) / (t_da + t0_da);
      const double log_eps = mu - sqrt(t_da) / gamma_da * hbar;
      const double eta = pow(t_da, -kappa_da);
      log_eps_bar = eta * log_eps + (1.0 - eta) * log_eps_bar;
      eps = exp(log_eps);
      if (it >= wnd.slow_begin && it < wnd.slow_end) {
        if (tid < dim) {
          wn += 1.0;
          const double x = theta[tid], d0 = x - wmean;
          wmean += d0 / wn;
          wm2 += d0 * (x - wmean);
        }
        if (it + 1 == win_end) {
          // inverse mass = regularised variance of this window (Stan's shrinkage towards 1e-3)
          if (tid < dim && wn >= 2.0) {
            const double var = wm2 / (wn - 1.0);
            const double v = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
            if (v == v && v < 1e300 && v > 0.0) imass[tid] = v;
          }
          wn = 0.0; wmean = 0.0; wm2 = 0.0;
          eps = exp(log_eps_bar);
          mu = log(10.0 * eps); hbar = 0.0; log_eps_bar = 0.0; t_da = 0.0;
          if (win_end < wnd.slow_end) {
            win_size *= 2;
            int e = win_end + win_size;
            if (e + 2 * win_size > wnd.slow_end) e = wnd.slow_end;
            win_end = e;
          }
        }
      }
      if (it == a.W - 1 && t_da > 0.0) eps = exp(log_eps_bar);
    } else {
      accepted += take ? 1.0 : 0.0;
      double* o = a.draws + ((size_t)chain * a.S + (it - a.W)) * (3 + P);
      if (tid < P) o[3 + tid] = hs ? theta[tid] * hs_scale(theta, tid) : theta[tid];
      if (tid >= 64 && tid < 64 + NSC) o[tid - 64] = exp(clamp30(theta[off_sc + tid - 64]));
      if (D == 1 && tid == 128) o[2] = 0.0;
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.accept_rate[chain] = accepted / (double)(a.S > 0 ? a.S : 1);
    a.step_size[chain] = eps;
  }
}

__host__ __device__ inline size_t hmc_lds_bytes(int P, int tpad_if_x_in_lds) {
  const size_t f = (((size_t)(3 * NW * 16 + NW * (P + 4)) * sizeof(float)) + 15) & ~(size_t)15;
  const int MP = P > MAXP ? HMC_MAXP : MAXP;
  const size_t dimp = (size_t)(3 * MP + 5 + ((3 * MP + 5) & 1)), devp = (size_t)(MP + 3 + ((MP + 3) & 1)),
               hscp = (size_t)(MP + (MP & 1));
  return f + sizeof(double) * (7 * dimp + 2 * devp + hscp + 8) +
         sizeof(float) * (size_t)P * tpad_if_x_in_lds;
}

}  // namespace ci
