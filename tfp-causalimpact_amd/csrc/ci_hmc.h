// ci_hmc.h -- Hamiltonian Monte Carlo over the model's parameters, entirely on the device.
// EXTENSION (BASELINE config "64 HMC chains sharded across 8 GPUs"):
// the reference is Gibbs-only; upstream analogue tfp.sts.fit_with_hmc.  Parity with TFP: unpinned.
//
// One 256-thread workgroup per chain runs ALL warm-up and sampling iterations: every leapfrog
// step evaluates the Kalman-filter log-likelihood and its score with the time-parallel scans of
// loglik_grad_block (ci_kernels.h), the momentum / position updates, the Metropolis test,
// dual-averaging step-size adaptation (Nesterov 2009; Hoffman & Gelman 2014) and the diagonal
// mass estimate (per chain, Welford over the middle half of warm-up) are a few lanes of wave 0.
// The host-driven version this replaces (causalimpact/_hmc.py::fit_hmc_host, kept as the
// statistical reference) paid one launch + one PCIe round trip per leapfrog step.
//
// Target (same as _hmc.py): theta = (beta[P], log sigma_obs, log sigma_level[, log sigma_slope]);
//   log p = l(sigma, beta) - 1/2 beta' Omega beta
//           + sum_k [ -2 a_k lam_k - b_k exp(-2 lam_k) ]      (IG(a, b) on sigma^2 + Jacobian)
#pragma once
#include "ci_kernels.h"

namespace ci {

constexpr int HMC_MAXDIM = MAXP + 3;

struct HmcArgs {
  int T, P, C, W, S, n_leap, chain_offset, x_in_lds;
  uint32_t seed0, seed1;
  const float* y;
  const uint8_t* mask;
  const float* Xt;
  const double* omega;      // [P, P]
  double ig_a[3], ig_b[3];  // inverse-gamma (concentration, scale) of sigma^2: obs, level, slope
  double init_log[3];       // log of the initial scales (causalimpact_lib.py:566-572)
  float a1, p10, p11;
  double target_accept, eps0;
  const double* init;       // optional [C, P + 2 or 3] unconstrained starting points (e.g. draws of a
                            // fitted surrogate posterior); NULL = the Gibbs sampler's initial state
  double* draws;            // [C, S, 3 + P]  (sigma_obs, sigma_level, sigma_slope, beta)
  double* accept_rate;      // [C]
  double* step_size;        // [C]
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int D, int L>
__global__ __launch_bounds__(NT) void hmc_kernel(HmcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = a.P, T = a.T;
  constexpr int NSC = (D == 2) ? 3 : 2;          // number of scales in the parameter vector
  const int dim = P + NSC;
  float* slots = (float*)smem_h;                 // 3 * NW * 16
  float* part = slots + 3 * NW * 16;             // NW * (P + 4)
  double* dbl = (double*)(smem_h + (((3 * NW * 16 + NW * (P + 4)) * sizeof(float) + 15) & ~(size_t)15));
  double* theta = dbl;                 // current position (unconstrained)
  double* grad = theta + HMC_MAXDIM;   // its gradient
  double* th = grad + HMC_MAXDIM;      // trajectory position
  double* g = th + HMC_MAXDIM;         // trajectory gradient
  double* mom = g + HMC_MAXDIM;        // trajectory momentum
  double* imass = mom + HMC_MAXDIM;    // inverse mass (diagonal)
  double* dev = imass + HMC_MAXDIM;    // device layout of th: (s_obs, s_level, s_slope, beta)
  double* gdev = dev + HMC_MAXDIM;     // score in device layout
  double* sc = gdev + HMC_MAXDIM;      // scalars: [0] ll, [1] lp of the trajectory, [2] lp current
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

  // log posterior and gradient at th -> sc[1], g   (all threads; contains barriers)
  auto target = [&]() {
    if (tid < dim) {
      if (tid < P) dev[3 + tid] = th[tid];
      else {
        double lam = th[tid];
        lam = lam < -30.0 ? -30.0 : (lam > 30.0 ? 30.0 : lam);
        dev[tid - P] = exp(lam);
      }
    }
    if (D == 1 && tid == 0) dev[2] = 0.0;
    __syncthreads();
    if (x_in_lds)
      loglik_grad_block<D, L>(T, P, a.y, a.mask, Xs, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                              gdev, tid, lane, wave, TPAD);
    else
      loglik_grad_block<D, L>(T, P, a.y, a.mask, a.Xt, dev, a.a1, a.p10, a.p11, slots, part, &sc[0],
                              gdev, tid, lane, wave);
    __syncthreads();
    if (wave == 0) {
      double contrib = 0.0, gi = 0.0;
      if (lane < P) {
        double ob = 0.0;
        for (int k = 0; k < P; ++k) ob = fma(th[k], a.omega[k * P + lane], ob);
        contrib = -0.5 * th[lane] * ob;
        gi = gdev[3 + lane] - ob;
      } else if (lane < dim) {
        const int k = lane - P;
        double lam = th[lane];
        lam = lam < -30.0 ? -30.0 : (lam > 30.0 ? 30.0 : lam);
        const double e2 = exp(-2.0 * lam);
        contrib = -2.0 * a.ig_a[k] * lam - a.ig_b[k] * e2;
        gi = dev[k] * gdev[k] - 2.0 * a.ig_a[k] + 2.0 * a.ig_b[k] * e2;
      }
      double lp = sc[0] + wave_sum_d(contrib);
      const bool bad = !(lp == lp) || lp > 1e300 || lp < -1e300;
      if (bad) { lp = -INFINITY; gi = 0.0; }
      if (lane < dim) g[lane] = gi;
      if (lane == 0) sc[1] = lp;
    }
    __syncthreads();
  };

  // ---- initial state: the Gibbs sampler's initial scales, zero weights, a little jitter
  if (tid < dim) {
    double v = 0.0;
    if (tid >= P) {
      const int k = tid - P;
      v = a.init_log[k];
    }
    if (a.init) th[tid] = a.init[(size_t)chain * dim + tid];
    else th[tid] = v + 0.01 * normal_d(rng, 0u, SITE_HMC_INIT, 0, (uint32_t)tid);
    imass[tid] = 1.0;
  }
  __syncthreads();
  target();
  if (tid < dim) { theta[tid] = th[tid]; grad[tid] = g[tid]; }
  if (tid == 0) sc[2] = sc[1];
  __syncthreads();

  // adaptation state (lane 0 of wave 0; every thread carries a copy of eps)
  double eps = a.eps0, mu = log(10.0 * a.eps0), hbar = 0.0, log_eps_bar = 0.0, t_da = 0.0;
  const double gamma_da = 0.05, t0_da = 10.0, kappa_da = 0.75;
  const int win_lo = (int)(0.25 * a.W), win_hi = (int)(0.75 * a.W);
  double wn = 0.0, wmean = 0.0, wm2 = 0.0;       // Welford over the window (lane i: coordinate i)
  double accepted = 0.0;
  const int n_iter = a.W + a.S;
  for (int it = 0; it < n_iter; ++it) {
    // momentum ~ N(0, M), Hamiltonian at the start
    double kin = 0.0;
    if (tid < dim) {
      const double z = normal_d(rng, (uint32_t)it, SITE_HMC_MOMENTUM, 0, (uint32_t)tid);
      const double p0 = z / sqrt(imass[tid]);
      mom[tid] = p0;
      th[tid] = theta[tid];
      g[tid] = grad[tid];
      kin = 0.5 * p0 * p0 * imass[tid];
    }
    double h0 = 0.0;
    if (wave == 0) h0 = -sc[2] + wave_sum_d(kin);
    __syncthreads();
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
      if (lane < dim) k1 = 0.5 * mom[lane] * mom[lane] * imass[lane];
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
    if (it < a.W) {
      // dual averaging of the log step size (every thread keeps the same scalars)
      t_da += 1.0;
      hbar = (1.0 - 1.0 / (t_da + t0_da)) * hbar + (a.target_accept - acc_prob) / (t_da + t0_da);
      const double log_eps = mu - sqrt(t_da) / gamma_da * hbar;
      const double eta = pow(t_da, -kappa_da);
      log_eps_bar = eta * log_eps + (1.0 - eta) * log_eps_bar;
      eps = exp(log_eps);
      __syncthreads();     // theta updated above
      if (it >= win_lo && it < win_hi && tid < dim) {
        wn += 1.0;
        const double x = theta[tid], d0 = x - wmean;
        wmean += d0 / wn;
        wm2 += d0 * (x - wmean);
      }
      if (it == win_hi - 1 && win_hi - win_lo >= 10) {
        // inverse mass = per-coordinate variance, normalised to mean 1
        double var = (tid < dim) ? wm2 / wn + 1e-8 : 0.0;
        if (tid < dim) dev[tid] = var;       // dev is free between target() calls
        __syncthreads();
        double mean = 0.0;
        bool ok = true;
        for (int i = 0; i < dim; ++i) { mean += dev[i]; ok = ok && (dev[i] == dev[i]) && dev[i] < 1e300; }
        mean /= (double)dim;
        if (ok && tid < dim) imass[tid] = var / mean;
        eps = exp(log_eps_bar);
        mu = log(10.0 * eps); hbar = 0.0; log_eps_bar = 0.0; t_da = 0.0;
      }
      if (it == a.W - 1) eps = exp(log_eps_bar);
    } else {
      accepted += take ? 1.0 : 0.0;
      __syncthreads();
      double* o = a.draws + ((size_t)chain * a.S + (it - a.W)) * (3 + P);
      if (tid < P) o[3 + tid] = theta[tid];
      else if (tid < dim) {
        double lam = theta[tid];
        lam = lam < -30.0 ? -30.0 : (lam > 30.0 ? 30.0 : lam);
        o[tid - P] = exp(lam);
      }
      if (D == 1 && tid == 0) o[2] = 0.0;
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
  return f + sizeof(double) * (8 * HMC_MAXDIM + 8) + sizeof(float) * (size_t)P * tpad_if_x_in_lds;
}

}  // namespace ci
