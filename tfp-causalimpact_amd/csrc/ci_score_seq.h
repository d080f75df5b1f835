// ci_score_seq.h -- Kalman log-likelihood and its score for ANY model of the package (trend,
// optional slope, any list of seasonal blocks, regression) and ANY length: one wavefront per
// evaluation, sequential in time.  SURVEY.md section 8 row H (extension: the reference has no
// log-likelihood objective; oracle: ci_oracle_loglik_score / hmc_target in oracle/ci_oracle.c).
//
// It complements the time-parallel loglik_grad_block of ci_kernels.h, which takes trend +
// regression models with T <= 4096 at ~8 us per evaluation; this route is the capability one:
// seasonal blocks (constrained tfp.sts.Seasonal, causalimpact_lib.py:471-489) and T > 4096.
//
// Layout = the sequential Gibbs kernel's (ci_seasonal.h): lane i holds component i of every
// state-sized vector, each block in its FULL n-effect form (cyclic shift + rank-one drift noise
// sigma_d^2 g g', g = e_last - 1/n), covariance P and the smoother's N as D x D float matrices in
// LDS updated through per-entry source tables; arrays over time (residual, K_t, v_t/F_t, 1/F_t,
// e_t) in a per-evaluation HBM workspace.  Sums are float64.
//   forward : v_t, F_t, K~_t = P_t Z'/F_t;  l = -1/2 sum_obs (log 2 pi + log F_t + v_t^2/F_t)
//   backward: r~ = T' r_t, M = T' N_t T;  e_t = v_t/F_t - K~' r~;  D_t = 1/F_t + K~' M K~
//             r_{t-1} = r~ + Z' e_t;   N_{t-1} = M - Z'(M K~)' - (M K~) Z + Z'Z D_t
//   score   : dl/d beta = sum_t x_t e_t;  dl/d sigma_obs = sigma_obs sum_obs (e_t^2 - D_t);
//             dl/d sigma_level = sigma_level sum_t (r_t[0]^2 - N_t[0][0])  (slope: index 1);
//             dl/d sigma_d,k = sigma_d,k sum_{t: block k changes} ((g'r_t)^2 - g'N_t g)
// (Koopman & Shephard 1992; the same formulas as loglik_grad_block.)
#pragma once
#include "ci_seasonal.h"
#include "ci_hmc.h"      // wave_sum_d

namespace ci {

struct SeqScoreArgs {
  int T, P, K, has_slope, E;         // E evaluations (blocks)
  int nseas[SMAXK];
  const float* y;                    // [T] 0 where masked
  const uint8_t* mask;               // [T]
  const float* Xt;                   // [P, T]
  const uint8_t* season_change;      // [K, T]
  const double* theta;               // [E, 3 + K + P]: sigma_obs, sigma_level, sigma_slope, drift[K], beta[P]
  float a1, p10, p11, p1e;           // prior mean of the level, prior variances: level, slope, seasonal effects
  double* out_ll;                    // [E]
  double* out_grad;                  // [E, 3 + K + P] (NULL: log-likelihood only)
  float* ws;                         // [E, seq_score_ws_floats(T, D)]
};

__host__ __device__ inline size_t seq_score_ws_floats(int T, int D) {
  const size_t TS = (size_t)((T + 3) & ~3);
  return 4 * TS + (size_t)T * D;      // resid, vf, rF, e, kf
}
__host__ __device__ inline size_t seq_score_lds_bytes(int D, int K) {
  return sizeof(float) * ((size_t)4 * D * D + 4 * D + SMAXK) + sizeof(uint32_t) * 2 * (size_t)D * D + 64;
}

// Geometry of the lane-distributed state (shared by the score and the HMC kernel).
struct SeqGeom {
  int D, trend;
  int off[SMAXK], nsz[SMAXK];
  int blk, pos, nb, boff;
  bool comp, isz;
  int fwd_src, bwd_src;
  float gpos;
};
__device__ __forceinline__ SeqGeom seq_geometry(int K, int has_slope, const int* nseas, int lane) {
  SeqGeom q;
  q.trend = has_slope ? 2 : 1;
  q.D = q.trend;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) {
    q.off[k] = q.D; q.nsz[k] = (k < K) ? nseas[k] : 0;
    if (k < K) q.D += q.nsz[k];
  }
  q.blk = -1; q.pos = 0; q.nb = 1; q.boff = 0;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k)
    if (k < K && lane >= q.off[k] && lane < q.off[k] + q.nsz[k]) {
      q.blk = k; q.pos = lane - q.off[k]; q.nb = q.nsz[k]; q.boff = q.off[k];
    }
  q.comp = lane < q.D;
  q.isz = q.comp && (lane == 0 || (q.blk >= 0 && q.pos == 0));
  q.fwd_src = q.blk >= 0 ? q.boff + (q.pos + 1 == q.nb ? 0 : q.pos + 1) : lane;
  q.bwd_src = q.blk >= 0 ? q.boff + (q.pos == 0 ? q.nb - 1 : q.pos - 1) : lane;
  q.gpos = q.blk >= 0 ? ((q.pos == q.nb - 1) ? 1.f - 1.f / (float)q.nb : -1.f / (float)q.nb) : 0.f;
  return q;
}

// LDS of one evaluation
struct SeqLds {
  float *Pa, *Pb, *Na, *Nb, *pzv, *kfv, *uv, *gv, *d2;
  uint32_t *fmeta, *bmeta;
};
__device__ __forceinline__ SeqLds seq_lds(unsigned char* smem, int D) {
  SeqLds s;
  float* f = reinterpret_cast<float*>(smem);
  s.Pa = f; f += D * D; s.Pb = f; f += D * D; s.Na = f; f += D * D; s.Nb = f; f += D * D;
  s.pzv = f; f += D; s.kfv = f; f += D; s.uv = f; f += D; s.gv = f; f += D;
  s.d2 = f; f += SMAXK;
  s.fmeta = reinterpret_cast<uint32_t*>(f); s.bmeta = s.fmeta + D * D;
  return s;
}

// One evaluation by one wavefront.  dev: sigma_obs, sigma_level, sigma_slope, drift[K], beta[P]
// (float64, any address space); gdev (may be NULL): the score in the same layout; returns l.
// `Xt` has row stride `xstride`.
__device__ __forceinline__ double seq_loglik_score(const SeqScoreArgs& a, const SeqGeom& q,
                                                   const double* dev, double* gdev, float* ws,
                                                   unsigned char* smem, int lane) {
  const int T = a.T, P = a.P, K = a.K, D = q.D;
  const int TS = (T + 3) & ~3;
  float* resid = ws; float* vf = resid + TS; float* rFv = vf + TS; float* ev = rFv + TS;
  float* kf = ev + TS;
  SeqLds L = seq_lds(smem, D);
  const int DD = D * D;
  // ---- per-entry source tables: forward (P <- T P T') and backward (N <- T' N T)
  for (int e = lane; e < DD; e += 64) {
    const int i = e / D, j = e - i * D;
    int bi = 15, bj = 15, fi = i, fj = j, ri = i, rj = j;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        if (i >= q.off[k] && i < q.off[k] + q.nsz[k]) {
          bi = k; const int p = i - q.off[k];
          fi = q.off[k] + (p + 1 == q.nsz[k] ? 0 : p + 1);
          ri = q.off[k] + (p == 0 ? q.nsz[k] - 1 : p - 1);
        }
        if (j >= q.off[k] && j < q.off[k] + q.nsz[k]) {
          bj = k; const int p = j - q.off[k];
          fj = q.off[k] + (p + 1 == q.nsz[k] ? 0 : p + 1);
          rj = q.off[k] + (p == 0 ? q.nsz[k] - 1 : p - 1);
        }
      }
    L.fmeta[e] = (uint32_t)fi | ((uint32_t)fj << 8) | ((uint32_t)bi << 16) | ((uint32_t)bj << 20);
    L.bmeta[e] = (uint32_t)ri | ((uint32_t)rj << 8) | ((uint32_t)bi << 16) | ((uint32_t)bj << 20);
  }
  if (q.comp) L.gv[lane] = q.gpos;
  const float so = (float)dev[0], sl = (float)dev[1], ssc = (float)dev[2];
  const float H = so * so, ql = sl * sl, qs = ssc * ssc;
  float myd2 = 0.f;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k)
    if (k < K) {
      const float dk = (float)dev[3 + k];
      if (lane == 0) L.d2[k] = dk * dk;
      if (q.blk == k) myd2 = dk * dk;
    }
  // residual y - X beta (0 where masked)
  for (int t = lane; t < TS; t += 64) {
    float s = 0.f;
    if (t < T && !a.mask[t]) {
      for (int j = 0; j < P; ++j) s = fmaf(a.Xt[(size_t)j * T + t], (float)dev[3 + K + j], s);
      s = a.y[t] - s;
    }
    resid[t] = s;
  }
  // prior covariance of x_0 (full-effect form: sd^2 (I - 11'/n) per block)
  for (int e = lane; e < DD; e += 64) {
    const int i = e / D, j = e - i * D;
    const uint32_t mt = L.fmeta[e];
    const unsigned bi = (mt >> 16) & 15u, bj = (mt >> 20) & 15u;
    float v = 0.f;
    if (e == 0) v = a.p10;
    else if (a.has_slope && i == 1 && j == 1) v = a.p11;
    else if (bi != 15u && bi == bj) {
      int nn = 1;
#pragma unroll
      for (int k = 0; k < SMAXK; ++k) if (k < K && bi == (unsigned)k) nn = q.nsz[k];
      v = a.p1e * ((i == j ? 1.f : 0.f) - 1.f / (float)nn);
    }
    L.Pa[e] = v;
    L.Na[e] = 0.f;
  }
  wave_sync();
  auto zsum = [&](float x) -> float {
    float s = readlane_f(x, 0);
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) s += readlane_f(x, q.off[k]);
    return s;
  };
  auto chg = [&](int t) -> unsigned {
    unsigned cb = 0u;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K && a.season_change[(size_t)k * T + t]) cb |= 1u << k;
    return cb;
  };
  // ---- forward: Kalman filter
  double ll = 0.0;
  {
    float* Pcur = L.Pa; float* Pnxt = L.Pb;
    float am = (lane == 0) ? a.a1 : 0.f;
    for (int t = 0; t < T; ++t) {
      const bool obs = a.mask[t] == 0;
      const unsigned cb = (t + 1 < T) ? chg(t) : 0u;
      float kfi = 0.f, rF = 0.f, vfv = 0.f;
      if (obs) {
        float pz = 0.f;
        if (q.comp) {
          pz = Pcur[lane * D];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K) pz += Pcur[lane * D + q.off[k]];
          L.pzv[lane] = pz;
        }
        const float F = zsum(pz) + H;
        rF = 1.0f / F;
        const float v = resid[t] - zsum(am);
        kfi = pz * rF;
        vfv = v * rF;
        am = fmaf(kfi, v, am);
        if (lane == 0) ll -= 0.5 * (1.8378770664093453 + (double)__logf(F) + (double)v * (double)vfv);
      } else if (q.comp) {
        L.pzv[lane] = 0.f;
      }
      if (q.comp) kf[(size_t)t * D + lane] = kfi;
      if (lane == 0) { vf[t] = vfv; rFv[t] = rF; }
      if (t + 1 == T) break;
      {   // a <- T a
        const float sh = __shfl(am, q.fwd_src, 64);
        float r = (q.blk >= 0 && ((cb >> q.blk) & 1u)) ? sh : am;
        if (a.has_slope) { const float s1 = readlane_f(am, 1); if (lane == 0) r += s1; }
        am = r;
      }
      wave_sync();
      for (int e = lane; e < DD; e += 64) {
        const uint32_t mt = L.fmeta[e];
        const int i = e / D, j = e - i * D;
        const unsigned bi = (mt >> 16) & 15u, bj = (mt >> 20) & 15u;
        const bool ci = bi != 15u && ((cb >> bi) & 1u), cj = bj != 15u && ((cb >> bj) & 1u);
        const int si = ci ? (int)(mt & 255u) : i, sj = cj ? (int)((mt >> 8) & 255u) : j;
        float v = Pcur[si * D + sj] - L.pzv[si] * L.pzv[sj] * rF;
        if (a.has_slope) {
          if (i == 0) v += Pcur[D + sj] - L.pzv[1] * L.pzv[sj] * rF;
          if (j == 0) v += Pcur[si * D + 1] - L.pzv[si] * L.pzv[1] * rF;
          if (i == 0 && j == 0) v += Pcur[D + 1] - L.pzv[1] * L.pzv[1] * rF;
          if (i == 1 && j == 1) v += qs;
        }
        if (e == 0) v += ql;
        if (ci && bi == bj) v = fmaf(L.d2[bi], L.gv[i] * L.gv[j], v);
        Pnxt[e] = v;
      }
      wave_sync();
      float* tmp = Pcur; Pcur = Pnxt; Pnxt = tmp;
    }
  }
  ll = readlane_d(ll, 0);
  if (!gdev) return ll;
  // ---- backward: r_t, N_t and the score sums
  double gH = 0.0, gl = 0.0, gs = 0.0, gd = 0.0;     // gd: lane k < K accumulates block k
  {
    float* Ncur = L.Na; float* Nnxt = L.Nb;
    float r = 0.f;
    for (int t = T - 1; t >= 0; --t) {
      const bool obs = a.mask[t] == 0;
      if (t + 1 < T) {
        const unsigned cb = chg(t);
        // disturbance of the transition t -> t+1
        const float r0 = readlane_f(r, 0);
        if (lane == 0) gl += (double)(r0 * r0 - Ncur[0]);
        if (a.has_slope) {
          const float r1 = readlane_f(r, 1);
          if (lane == 0) gs += (double)(r1 * r1 - Ncur[D + 1]);
        }
        if (cb != 0u) {
          // g'r and g'N g of every changing block: lane p of the block forms g_p r_p and
          // g_p sum_q g_q N[p][q]; block sums through the wave (blocks are contiguous lanes)
          float gr = 0.f, gng = 0.f;
          const bool mine = q.blk >= 0 && ((cb >> q.blk) & 1u);
          if (mine) {
            gr = q.gpos * r;
            float s = 0.f;
            for (int qq = 0; qq < q.nb; ++qq) s = fmaf(L.gv[q.boff + qq], Ncur[lane * D + q.boff + qq], s);
            gng = q.gpos * s;
          }
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K && ((cb >> k) & 1u)) {
              const float sr = wave_sum_dpp(q.blk == k ? gr : 0.f);
              const float sn = wave_sum_dpp(q.blk == k ? gng : 0.f);
              if (lane == k) gd += (double)(sr * sr - sn);
            }
        }
        // r <- T' r ;  N <- T' N T
        {
          const float sh = __shfl(r, q.bwd_src, 64);
          float rr = (q.blk >= 0 && ((cb >> q.blk) & 1u)) ? sh : r;
          if (a.has_slope) { const float x0 = readlane_f(r, 0); if (lane == 1) rr += x0; }
          r = rr;
        }
        for (int e = lane; e < DD; e += 64) {
          const uint32_t mt = L.bmeta[e];
          const int i = e / D, j = e - i * D;
          const unsigned bi = (mt >> 16) & 15u, bj = (mt >> 20) & 15u;
          const bool ci = bi != 15u && ((cb >> bi) & 1u), cj = bj != 15u && ((cb >> bj) & 1u);
          const int si = ci ? (int)(mt & 255u) : i, sj = cj ? (int)((mt >> 8) & 255u) : j;
          float v = Ncur[si * D + sj];
          if (a.has_slope) {       // T = I + E_01: M = N + [j==1] N[.][0] + [i==1] (N[0][.] + [j==1] N[0][0])
            if (j == 1) v += Ncur[si * D];
            if (i == 1) v += Ncur[sj] + (j == 1 ? Ncur[0] : 0.f);
          }
          Nnxt[e] = v;
        }
        wave_sync();
        float* tmp = Ncur; Ncur = Nnxt; Nnxt = tmp;
      }
      float et = 0.f;
      if (obs) {
        const float kfi = q.comp ? kf[(size_t)t * D + lane] : 0.f;
        if (q.comp) L.kfv[lane] = kfi;
        wave_sync();
        float u = 0.f;
        if (q.comp)
          for (int j = 0; j < D; ++j) u = fmaf(Ncur[lane * D + j], L.kfv[j], u);
        if (q.comp) L.uv[lane] = u;
        const float kr = wave_sum_dpp(kfi * r);
        const float kmk = wave_sum_dpp(kfi * u);
        const float rF = rFv[t];
        et = vf[t] - kr;
        const float Dt = rF + kmk;
        if (lane == 0) gH += (double)(et * et - Dt);
        if (q.isz) r += et;
        wave_sync();
        // N += -z u' - u z' + z z' D_t   (z = the rows of Z: lane 0 and the first lane of every block)
        for (int e = lane; e < DD; e += 64) {
          const int i = e / D, j = e - i * D;
          const uint32_t mt = L.bmeta[e];
          const unsigned bi = (mt >> 16) & 15u, bj = (mt >> 20) & 15u;
          bool zi = i == 0, zj = j == 0;
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K) { zi = zi || (bi == (unsigned)k && i == q.off[k]); zj = zj || (bj == (unsigned)k && j == q.off[k]); }
          if (zi || zj) {
            float v = Ncur[e];
            if (zi) v -= L.uv[j];
            if (zj) v -= L.uv[i];
            if (zi && zj) v += Dt;
            Ncur[e] = v;
          }
        }
        wave_sync();
      }
      if (lane == 0) ev[t] = et;
    }
  }
  wave_sync();
  // ---- assemble the score
  const double g_obs = (double)so * readlane_d(gH, 0);
  const double g_lev = (double)sl * readlane_d(gl, 0);
  const double g_slp = (double)ssc * readlane_d(gs, 0);
  if (lane == 0) { gdev[0] = g_obs; gdev[1] = g_lev; gdev[2] = a.has_slope ? g_slp : 0.0; }
  if (lane < K) gdev[3 + lane] = dev[3 + lane] * gd;
  for (int j = 0; j < P; ++j) {
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s = fmaf(a.Xt[(size_t)j * T + t], ev[t], s);
    const double tot = wave_sum_d((double)s);
    if (lane == 0) gdev[3 + K + j] = tot;
  }
  wave_sync();
  return ll;
}

#ifndef CI_SEASONAL_DECL_ONLY
// E evaluations, one wavefront each.
__global__ __launch_bounds__(64) void seq_score_kernel(SeqScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  const int lane = threadIdx.x, ev = blockIdx.x;
  const SeqGeom q = seq_geometry(a.K, a.has_slope, a.nseas, lane);
  const int dim = 3 + a.K + a.P;
  float* ws = a.ws + (size_t)ev * seq_score_ws_floats(a.T, q.D);
  const double ll = seq_loglik_score(a, q, a.theta + (size_t)ev * dim,
                                     a.out_grad ? a.out_grad + (size_t)ev * dim : nullptr, ws, smem_q,
                                     lane);
  if (lane == 0) a.out_ll[ev] = ll;
}
#endif

}  // namespace ci

// ------------------------------------------------------------------------------------
// Hamiltonian Monte Carlo for ANY model / length: the driver of hmc_kernel (ci_hmc.h -- same target,
// same windowed adaptation, same random stream, statement for statement) over the sequential
// score above.  One 256-thread workgroup per chain: wave 0 evaluates the score, the float64
// bookkeeping is spread over the threads as in hmc_kernel.  theta = (regression block, log sigma_obs,
// log sigma_level, [log sigma_slope], log sigma_drift[K]); oracle: ci_oracle_fit_hmc.
// ------------------------------------------------------------------------------------
namespace ci {

constexpr int HMC_SEQ_MAXDIM = 3 * MAXP + 5 + SMAXK;

struct HmcSeqArgs {
  SeqScoreArgs q;            // data and geometry (theta / out_* / E unused; ws = [C, seq_score_ws_floats])
  int C, W, S, n_leap, chain_offset, prior_mode;
  uint32_t seed0, seed1;
  const double* omega;       // [P, P]
  double ig_a[3 + SMAXK], ig_b[3 + SMAXK], init_log[3 + SMAXK];   // in the order of theta's scales
  double hs_scale0, target_accept, eps0;
  const double* init;        // optional [C, dim]
  double* draws;             // [C, S, 3 + K + P]  (sigma_obs, sigma_level, sigma_slope, drift[K], beta)
  double* accept_rate;       // [C]
  double* step_size;         // [C]
};

__host__ __device__ inline size_t hmc_seq_dbl_count() {
  return 6 * (size_t)HMC_SEQ_MAXDIM + 2 * (size_t)(MAXP + 3 + SMAXK) + MAXP + 8;
}
__host__ __device__ inline size_t hmc_seq_lds_bytes(int D, int K) {
  return sizeof(double) * hmc_seq_dbl_count() + ((seq_score_lds_bytes(D, K) + 15) & ~(size_t)15);
}

// The driver.  `score(dev, gdev, ll_out)` is called by ALL threads of the workgroup with uniform
// control flow; it evaluates l and its gradient at the device-layout parameters `dev` into
// *ll_out / `gdev` (same layout) and ends with a barrier.  smem_s: hmc_seq_dbl_count() doubles.
template <class Score>
__device__ __forceinline__ void hmc_drive(const HmcSeqArgs& a, unsigned char* smem_s, Score& score) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = a.q.P, K = a.q.K;
  const int ntr = a.q.has_slope ? 3 : 2;          // trend scales in theta
  const int nsc = ntr + K;
  const bool hs = a.prior_mode == 1;
  const int off_sc = hs ? 3 * P + 2 : P;
  const int dim = off_sc + nsc;
  const int ob = 3 + K;                           // where beta starts in the device layout
  double* dbl = (double*)smem_s;
  double* theta = dbl;
  double* grad = theta + HMC_SEQ_MAXDIM;
  double* th = grad + HMC_SEQ_MAXDIM;
  double* g = th + HMC_SEQ_MAXDIM;
  double* mom = g + HMC_SEQ_MAXDIM;
  double* imass = mom + HMC_SEQ_MAXDIM;
  double* dev = imass + HMC_SEQ_MAXDIM;           // (s_obs, s_level, s_slope, drift[K], beta)
  double* gdev = dev + (MAXP + 3 + SMAXK);
  double* hsc = gdev + (MAXP + 3 + SMAXK);
  double* sc = hsc + MAXP;
  const int chain = blockIdx.x;
  Rng rng{a.seed0, a.seed1, (uint32_t)(a.chain_offset + chain)};
  // theta's scale k -> slot of the device layout
  auto slot = [&](int k) { return k < ntr ? k : 3 + (k - ntr); };

  auto hs_scale = [&](const double* v, int j) {
    return exp(clamp30(v[P + j]) + 0.5 * clamp30(v[2 * P + j]) + clamp30(v[3 * P]) +
               0.5 * clamp30(v[3 * P + 1])) * a.hs_scale0;
  };

  auto target = [&]() {
    if (tid < P) {
      if (hs) {
        const double s = hs_scale(th, tid);
        hsc[tid] = s;
        dev[ob + tid] = th[tid] * s;
      } else {
        dev[ob + tid] = th[tid];
      }
    }
    if (tid >= 64 && tid < 64 + nsc) dev[slot(tid - 64)] = exp(clamp30(th[off_sc + tid - 64]));
    if (!a.q.has_slope && tid == 128) dev[2] = 0.0;
    __syncthreads();
    score(dev, gdev, &sc[0]);
    if (wave == 0) {
      double sgb = 0.0;
      if (hs) {
        for (int j = lane; j < P; j += 64) sgb = fma(gdev[ob + j], dev[ob + j], sgb);
        sgb = wave_sum_d(sgb);
      }
      double contrib = 0.0;
      for (int i = lane; i < dim; i += 64) {
        double ci, gi;
        if (i >= off_sc) {
          const int k = i - off_sc, sk = slot(k);
          const double lam = clamp30(th[i]);
          const double e2 = 1.0 / (dev[sk] * dev[sk]);
          ci = -2.0 * a.ig_a[k] * lam - a.ig_b[k] * e2;
          gi = dev[sk] * gdev[sk] - 2.0 * a.ig_a[k] + 2.0 * a.ig_b[k] * e2;
        } else if (!hs) {
          double obv = 0.0;
          for (int k = 0; k < P; ++k) obv = fma(th[k], a.omega[k * P + i], obv);
          ci = -0.5 * th[i] * obv;
          gi = gdev[ob + i] - obv;
        } else if (i < P) {
          const double z = th[i];
          ci = -0.5 * z * z;
          gi = gdev[ob + i] * hsc[i] - z;
        } else if (i < 2 * P) {
          const int j = i - P;
          const double e = exp(2.0 * clamp30(th[i]));
          ci = -0.5 * e + clamp30(th[i]);
          gi = gdev[ob + j] * dev[ob + j] - e + 1.0;
        } else if (i < 3 * P) {
          const int j = i - 2 * P;
          const double u = clamp30(th[i]);
          const double e = exp(-u);
          ci = -0.5 * u - 0.5 * e;
          gi = 0.5 * gdev[ob + j] * dev[ob + j] - 0.5 + 0.5 * e;
        } else if (i == 3 * P) {
          const double e = exp(2.0 * clamp30(th[i]));
          ci = -0.5 * e + clamp30(th[i]);
          gi = sgb - e + 1.0;
        } else {
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

  double eps = a.eps0, mu = log(10.0 * a.eps0), hbar = 0.0, log_eps_bar = 0.0, t_da = 0.0;
  const double gamma_da = 0.05, t0_da = 10.0, kappa_da = 0.75;
  const HmcWindows wnd = hmc_windows(a.W);
  int win_end = wnd.first_end, win_size = wnd.base;
  double wn = 0.0, wmean = 0.0, wm2 = 0.0;
  double accepted = 0.0;
  const int n_iter = a.W + a.S;
  for (int it = 0; it < n_iter; ++it) {
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
      double* o = a.draws + ((size_t)chain * a.S + (it - a.W)) * (3 + K + P);
      if (tid < P) o[ob + tid] = hs ? theta[tid] * hs_scale(theta, tid) : theta[tid];
      if (tid >= 64 && tid < 64 + nsc) o[slot(tid - 64)] = exp(clamp30(theta[off_sc + tid - 64]));
      if (!a.q.has_slope && tid == 128) o[2] = 0.0;
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.accept_rate[chain] = accepted / (double)(a.S > 0 ? a.S : 1);
    a.step_size[chain] = eps;
  }
}

#ifndef CI_SEASONAL_DECL_ONLY
// The sequential route: wave 0 evaluates the score (seq_loglik_score), the others wait.
struct SeqScoreFn {
  const SeqScoreArgs* q;
  SeqGeom geo;
  float* ws;
  unsigned char* smem;
  int lane, wave;
  __device__ __forceinline__ void operator()(const double* dev, double* gdev, double* ll_out) {
    if (wave == 0) {
      const double ll = seq_loglik_score(*q, geo, dev, gdev, ws, smem, lane);
      if (lane == 0) *ll_out = ll;
    }
    __syncthreads();
  }
};

__global__ __launch_bounds__(NT) void hmc_seq_kernel(HmcSeqArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63;
  SeqScoreFn fn;
  fn.q = &a.q;
  fn.geo = seq_geometry(a.q.K, a.q.has_slope, a.q.nseas, lane);
  fn.ws = a.q.ws + (size_t)blockIdx.x * seq_score_ws_floats(a.q.T, fn.geo.D);
  fn.smem = smem_s + sizeof(double) * hmc_seq_dbl_count();
  fn.lane = lane;
  fn.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  hmc_drive(a, smem_s, fn);
}
#endif

}  // namespace ci
