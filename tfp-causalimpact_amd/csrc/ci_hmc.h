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

namespace ci {

constexpr int HMC_MAXDIM = 3 * MAXP + 5;

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

template <int D, int L>
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
  double* theta = dbl;                 // current position (unconstrained)
  double* grad = theta + HMC_MAXDIM;   // its gradient
  double* th = grad + HMC_MAXDIM;      // trajectory position
  double* g = th + HMC_MAXDIM;         // trajectory gradient
  double* mom = g + HMC_MAXDIM;        // trajectory momentum
  double* imass = mom + HMC_MAXDIM;    // inverse mass (diagonal)
  double* dev = imass + HMC_MAXDIM;    // device layout of th: (s_obs, s_level, s_slope, beta)   [3 + MAXP]
  double* gdev = dev + (MAXP + 3);     // score in device layout                                 [3 + MAXP]
  double* hsc = gdev + (MAXP + 3);     // horseshoe: d beta_j / d z_j                            [MAXP]
  double* sc = hsc + MAXP;             // scalars: [0] ll, [1] lp of the trajectory, [2] lp current
  // feature-major design matrix, zero padded to NT * L columns, resident in LDS for the whole fit
  // (every leapfrog step reads it twice: residual and d l / d beta)
  constexpr int TPAD = NT * L;
  float* Xs = (float*)(sc + 8);
  const bool x_in_lds = a.x_in_lds != 0;
  if (x_in_lds) {
    for (int j = 0; j < P; ++j)
      for (int t = tid; t < TPAD; t += NT) Xs[j * TPAD + t] = t < T ? a.Xt[(size_t)j * T + t] : 0.f;
  }
  const int chain = blockIdx.x;
  Rng rng{a.seed0, a.seed1, (uint32_t)(a.chain_offset + chain)};

  // d beta_j / d z_j of the horseshoe at the unconstrained point v
  auto hs_scale = [&](const double* v, int j) {
    return exp(clamp30(v[P + j]) + 0.5 * clamp30(v[2 * P + j]) + clamp30(v[3 * P]) +
               0.5 * clamp30(v[3 * P + 1])) * a.hs_scale0;
  };

  // log posterior and gradient at th -> sc[1], g   (all threads; contains barriers)
  auto target = [&]() {
    if (tid < P) {
      if (hs) {
        const double s = hs_scale(th, tid);
        hsc[tid] = s;
        dev[3 + tid] = th[tid] * s;
      } else {
        dev[3 + tid] = th[tid];
      }
    }
    if (tid >= 64 && tid < 64 + NSC) dev[tid - 64] = exp(clamp30(th[off_sc + tid - 64]));
    if (D == 1 && tid == 128) dev[2] = 0.0;
    __syncthreads();
    if (x_in_lds)
      loglik_grad_block<D, L>(T, P, a.y, a.mask, Xs, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                              gdev, tid, lane, wave, TPAD);
    else
      loglik_grad_block<D, L>(T, P, a.y, a.mask, a.Xt, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                              gdev, tid, lane, wave);
    __syncthreads();
    if (wave == 0) {
      double sgb = 0.0;                    // horseshoe: sum_j beta_j dl/dbeta_j
      if (hs) {
        for (int j = lane; j < P; j += 64) sgb = fma(gdev[3 + j], dev[3 + j], sgb);
        sgb = wave_sum_d(sgb);
      }
      double contrib = 0.0;
      for (int i = lane; i < dim; i += 64) {
        double ci, gi;
        if (i >= off_sc) {
          const int k = i - off_sc;
          const double lam = clamp30(th[i]);
          const double e2 = 1.0 / (dev[k] * dev[k]);       // exp(-2 lam): dev[k] = exp(lam)
          ci = -2.0 * a.ig_a[k] * lam - a.ig_b[k] * e2;
          gi = dev[k] * gdev[k] - 2.0 * a.ig_a[k] + 2.0 * a.ig_b[k] * e2;
        } else if (!hs) {
          double ob = 0.0;
          for (int k = 0; k < P; ++k) ob = fma(th[k], a.omega[k * P + i], ob);
          ci = -0.5 * th[i] * ob;
          gi = gdev[3 + i] - ob;
        } else if (i < P) {                  // z_j ~ N(0, 1)
          const double z = th[i];
          ci = -0.5 * z * z;
          gi = gdev[3 + i] * hsc[i] - z;
        } else if (i < 2 * P) {              // log of a HalfNormal(1) local scale
          const int j = i - P;
          const double e = exp(2.0 * clamp30(th[i]));
          ci = -0.5 * e + clamp30(th[i]);
          gi = gdev[3 + j] * dev[3 + j] - e + 1.0;
        } else if (i < 3 * P) {              // log of an InverseGamma(1/2, 1/2) local variance
          const int j = i - 2 * P;
          const double u = clamp30(th[i]);
          const double e = exp(-u);
          ci = -0.5 * u - 0.5 * e;
          gi = 0.5 * gdev[3 + j] * dev[3 + j] - 0.5 + 0.5 * e;
        } else if (i == 3 * P) {             // log of the HalfNormal(1) global scale
          const double e = exp(2.0 * clamp30(th[i]));
          ci = -0.5 * e + clamp30(th[i]);
          gi = sgb - e + 1.0;
        } else {                             // log of the InverseGamma(1/2, 1/2) global variance
          const double u = clamp30(th[i]);
          const double e = exp(-u);
          ci = -0.5 * u - 0.5 * e;
          gi = 0.5 * sgb - 0.5 + 0.5 * e;
        }
        contrib += ci;
        g[i] = gi;
      }
      double lp = sc[0] + wave_sum_d(contrib);
      const bool bad = !(lp == lp) || lp > 1e300 || lp < -1e300;
      if (bad) {
        lp = -INFINITY;
        for (int i = lane; i < dim; i += 64) g[i] = 0.0;
      }
      if (lane == 0) sc[1] = lp;
    }
    __syncthreads();
  };

  // ---- initial state: the Gibbs sampler's initial scales, zero weights, a little jitter
  // (horseshoe: all auxiliary log scales 0, z = 0)
  if (tid < dim) {
    double v = 0.0;
    if (tid >= off_sc) v = a.init_log[tid - off_sc];
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
  return f + sizeof(double) * (6 * HMC_MAXDIM + 2 * (MAXP + 3) + MAXP + 8) +
         sizeof(float) * (size_t)P * tpad_if_x_in_lds;
}

}  // namespace ci
