// ci_seasonal.h -- Gibbs kernel for models with seasonal blocks (tfp.sts.Seasonal,
// allow_drift=True, constrain_mean_effect_to_zero=True; reference call site
// causalimpact/causalimpact_lib.py:471-489).
//
// State dimension is 1 (+1 slope) + sum_k num_seasons_k, too wide for the register-resident
// scan elements of ci_kernels.h, so this first device version is ONE WAVEFRONT PER CHAIN and
// sequential in time (DESIGN.md "Seasonal models"):
//   * lane i holds component i of every state-sized vector; the covariance lives in LDS;
//   * every seasonal block is carried in its FULL n-effect form (the n-th effect is minus the
//     sum of the others).  The constrained dynamics then are a pure cyclic shift of the block's
//     lanes plus rank-1 noise, so no per-step block sums are needed; the covariance is
//     singular but F = Z P Z' + H > 0.  This is the same Gaussian as the oracle's
//     (n-1)-dimensional form, so draws agree per random number;
//   * de Jong / Koopman fast state smoother: forward filter (store K_t, v_t/F_t), backward
//     r-recursion (store r_t), forward reconstruction x^_{t+1} = T x^_t + Q_t r_t -- no
//     per-step covariance storage;
//   * regression block, gamma draws and the random stream are shared with ci_kernels.h.
#pragma once
#include "ci_kernels.h"

namespace ci {

constexpr int SMAXK = 8;

// v_readlane on a float (the builtin is typed int: passing a float would convert by value)
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

struct DevSeasonalParams {
  double drift_conc, drift_scale, drift_ub, init_seasonal_scale;
  double drift_scale0[SMAXK];
};

struct SArgs {
  KArgs k;                           // common fields (x_in_lds unused)
  int K, has_slope, dred;            // blocks, trend type, reduced state dim (oracle's d)
  int nseas[SMAXK];
  const uint8_t* season_change;      // [K,T]
  const DevSeasonalParams* ssp;      // [B]
  const float* p1_chol;              // [B,dred,dred] lower Cholesky factor of the prior cov of x_0
  float* out_drift;                  // [B,C,S,K]
  float* out_seasonal;               // [B,C,S,T,K]
};

struct SLayout {
  size_t yv, lev, slp, xw, ytil, vf, zl, zs, zo, seas, zk, kf, rs, Pa, Pb, pzv, zi, x0r, mask, chg,
      ei, ej, xtx, omega, bvec, w, total;
};

__host__ __device__ inline SLayout make_slayout(int T, int P, int K, int D, int dred,
                                                int has_slope) {
  SLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const size_t Tf = sizeof(float) * (size_t)((T + 3) & ~3);
  const int Pp = P > 0 ? P : 1;
  l.xtx = take(sizeof(double) * Pp * Pp);
  l.omega = take(sizeof(double) * Pp * Pp);
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.yv = take(Tf); l.lev = take(Tf); l.slp = take(has_slope ? Tf : 16); l.xw = take(Tf);
  l.ytil = take(Tf); l.vf = take(Tf); l.zl = take(Tf); l.zs = take(has_slope ? Tf : 16);
  l.zo = take(Tf);
  l.seas = take(Tf * K); l.zk = take(Tf * K);
  l.kf = take(sizeof(float) * (size_t)T * D); l.rs = take(sizeof(float) * (size_t)T * D);
  l.Pa = take(sizeof(float) * D * D); l.Pb = take(sizeof(float) * D * D);
  l.pzv = take(sizeof(float) * D); l.zi = take(sizeof(float) * (dred + 1));
  l.x0r = take(sizeof(float) * (dred + 1));
  l.w = take(sizeof(float) * Pp);
  l.mask = take((size_t)T); l.chg = take((size_t)T * K);
  l.ei = take((size_t)D * D); l.ej = take((size_t)D * D);
  l.total = o;
  return l;
}

#ifndef CI_SEASONAL_DECL_ONLY
__global__ __launch_bounds__(64) void gibbs_seasonal_kernel(SArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const KArgs& g = a.k;
  const int T = g.T, P = g.P, K = a.K;
  const int series = blockIdx.x / g.C, chain = blockIdx.x % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  const int trend = a.has_slope ? 2 : 1;
  int off[SMAXK], nsz[SMAXK], roff[SMAXK];
  int D = trend;
  {
    int rr = trend;
    for (int k = 0; k < K; ++k) { off[k] = D; nsz[k] = a.nseas[k]; roff[k] = rr; D += nsz[k]; rr += nsz[k] - 1; }
  }
  const SLayout L = make_slayout(T, P, K, D, a.dred, a.has_slope);
  float* yv = (float*)(smem + L.yv); float* lev = (float*)(smem + L.lev);
  float* slp = (float*)(smem + L.slp); float* xw = (float*)(smem + L.xw);
  float* ytil = (float*)(smem + L.ytil); float* vf = (float*)(smem + L.vf);
  float* zl = (float*)(smem + L.zl); float* zs = (float*)(smem + L.zs);
  float* zo = (float*)(smem + L.zo); float* seas = (float*)(smem + L.seas);
  float* zk = (float*)(smem + L.zk); float* kf = (float*)(smem + L.kf);
  float* rs = (float*)(smem + L.rs); float* Pcur = (float*)(smem + L.Pa);
  float* Pnxt = (float*)(smem + L.Pb); float* pzv = (float*)(smem + L.pzv);
  float* zi = (float*)(smem + L.zi); float* x0r = (float*)(smem + L.x0r);
  uint8_t* msk = smem + L.mask; uint8_t* chg = smem + L.chg;
  uint8_t* ei = smem + L.ei; uint8_t* ej = smem + L.ej;
  const int TS = (T + 3) & ~3;       // row stride of the [K][T] float arrays
  RegLds R;
  R.xtx = (double*)(smem + L.xtx); R.omega = (double*)(smem + L.omega);
  R.bvec = (double*)(smem + L.bvec); R.w = (float*)(smem + L.w);
  R.aug[0] = R.aug[1] = R.pri[0] = R.pri[1] = R.chol = R.zv = R.uperm = nullptr;
  R.nz = R.perm = R.idx = nullptr;

  const DevSeriesParams sp = g.sp[series];
  const DevSeasonalParams ss = a.ssp[series];
  Rng rng{g.seed0, g.seed1, (uint32_t)(g.chain_offset + chain)};
  const float* Xg = g.Xt + (size_t)series * P * T;
  const float* chol1 = a.p1_chol + (size_t)series * a.dred * a.dred;

  // ---- lane roles: component `lane` of the state, block membership
  int blk = -1, pos = 0, nb = 1, boff = 0;
  for (int k = 0; k < K; ++k)
    if (lane >= off[k] && lane < off[k] + nsz[k]) { blk = k; pos = lane - off[k]; nb = nsz[k]; boff = off[k]; }
  const bool comp = lane < D;
  const bool isz = comp && (lane == 0 || (blk >= 0 && pos == 0));   // rows of Z

  // ---- stage constants
  for (int t = lane; t < T; t += 64) {
    const bool m = g.mask[(size_t)series * T + t] != 0;
    msk[t] = m ? 1 : 0;
    yv[t] = m ? 0.f : g.y[(size_t)series * T + t];
    lev[t] = 0.f; xw[t] = 0.f;
    if (a.has_slope) slp[t] = 0.f;
    for (int k = 0; k < K; ++k) { seas[k * TS + t] = 0.f; chg[k * T + t] = a.season_change[(size_t)k * T + t]; }
  }
  for (int e = lane; e < P * P; e += 64) {
    R.xtx[e] = g.xtx[(size_t)series * P * P + e];
    R.omega[e] = g.omega[(size_t)series * P * P + e];
  }
  for (int e = lane; e < D * D; e += 64) { ei[e] = (uint8_t)(e / D); ej[e] = (uint8_t)(e % D); }
  if (lane < P) R.w[lane] = 0.f;
  wave_sync();
  double n_changes[SMAXK];
  for (int k = 0; k < K; ++k) {
    float c = 0.f;
    for (int t = lane; t + 1 < T; t += 64) c += chg[k * T + t] ? 1.f : 0.f;
    n_changes[k] = (double)wave_sum(c);
  }

  double obs_scale = sp.obs_scale0, level_scale = sp.level_scale0, slope_scale = sp.slope_scale0;
  double drift[SMAXK];
  for (int k = 0; k < K; ++k) drift[k] = ss.drift_scale0[k];
  float ssl = 0.f, sss = 0.f, ssd = 0.f;   // lane 0 / lane 1 / lane off[k] accumulate
  const float p1l = (float)(sp.init_level_scale * sp.init_level_scale);
  const float p1s = (float)(sp.init_slope_scale * sp.init_slope_scale);
  const float p1e = (float)(ss.init_seasonal_scale * ss.init_seasonal_scale);
  Prof prof;
  prof.start(nullptr, false);
  PriorCarry pc;
  pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;

  auto zsum = [&](float x) -> float {   // Z x for a lane-distributed vector
    float s = readlane_f(x, 0);
    for (int k = 0; k < K; ++k) s += readlane_f(x, off[k]);
    return s;
  };
  // x <- T_t x : cyclic shift of the blocks that change season at t (+ level += slope)
  auto transition = [&](float x, int t) -> float {
    const float nxt = __shfl_down(x, 1, 64);
    float r = x;
    for (int k = 0; k < K; ++k) {
      if (chg[k * T + t]) {
        const float first = readlane_f(x, off[k]);
        if (blk == k) r = (pos == nb - 1) ? first : nxt;
      }
    }
    if (a.has_slope) {
      const float s1 = readlane_f(x, 1);
      if (lane == 0) r += s1;
    }
    return r;
  };
  // x <- T_t' x
  auto transition_T = [&](float x, int t) -> float {
    const float prv = __shfl_up(x, 1, 64);
    float r = x;
    for (int k = 0; k < K; ++k) {
      if (chg[k * T + t]) {
        const float last = readlane_f(x, off[k] + nsz[k] - 1);
        if (blk == k) r = (pos == 0) ? last : prv;
      }
    }
    if (a.has_slope) {
      const float r0 = readlane_f(x, 0);
      if (lane == 1) r += r0;
    }
    return r;
  };
  // disturbance of the prior simulation entering at transition t (after the shift)
  auto sim_noise = [&](float x, int t, float sl, float ssl_, const double* dr) -> float {
    float r = x;
    if (lane == 0) r = fmaf(sl, zl[t], r);
    if (a.has_slope && lane == 1) r = fmaf(ssl_, zs[t], r);
    for (int k = 0; k < K; ++k) {
      if (chg[k * T + t] && blk == k) {
        const float w = (float)dr[k] * zk[k * TS + t];
        r += (pos == nb - 1) ? w - w / (float)nb : -w / (float)nb;
      }
    }
    return r;
  };

  const int n_iter = g.W + g.S;
  float pm_acc_dummy = 0.f;
  (void)pm_acc_dummy;
  for (int it = 0; it <= n_iter; ++it) {
    // ---- (1) X~'targets, y'y from the current latents
    {
      float yty = 0.f;
      for (int t = lane; t < T; t += 64) {
        float tg = 0.f;
        if (!msk[t]) {
          tg = yv[t] - lev[t];
          for (int k = 0; k < K; ++k) tg -= seas[k * TS + t];
        }
        ytil[t] = tg;            // reused as the targets buffer here
        yty = fmaf(tg, tg, yty);
      }
      wave_sync();
      for (int j = 0; j < P; ++j) {
        float pj = 0.f;
        for (int t = lane; t < T; t += 64) pj = fmaf(Xg[(size_t)j * T + t], ytil[t], pj);
        const float s = wave_sum(pj);
        if (lane == 0) R.bvec[j] = (double)s;
      }
      const float s0 = wave_sum(yty);
      if (lane == 0) R.bvec[P] = (double)s0;
      wave_sync();
    }
    // ---- (2) scale draws of iteration it-1, regression draw of iteration it
    double emit_obs = obs_scale;
    if (it > 0) {
      const uint32_t pit = (uint32_t)(it - 1);
      const double v_l = (double)readlane_f(ssl, 0);
      level_scale = scale_draw(sp.level_conc, sp.level_scale, sp.level_ub, (double)(T - 1), v_l,
                               rng, pit, SITE_LEVEL_SCALE, lane);
      if (a.has_slope) {
        const double v_s = (double)readlane_f(sss, 1);
        slope_scale = scale_draw(sp.slope_conc, sp.slope_scale, sp.slope_ub, (double)(T - 1), v_s,
                                 rng, pit, SITE_SLOPE_SCALE, lane);
      }
      for (int k = 0; k < K; ++k) {
        const double v_d = (double)readlane_f(ssd, off[k]);
        const double gk = gamma_wave(ss.drift_conc + 0.5 * n_changes[k], rng, pit, SITE_DRIFT_SCALE,
                                     (uint32_t)k, lane);
        const double sd = (double)__fsqrt_rn((float)((ss.drift_scale + 0.5 * v_d) * fast_rcp(gk)));
        drift[k] = sd < ss.drift_ub ? sd : ss.drift_ub;
      }
      if (P == 0)
        obs_scale = scale_draw(sp.obs_conc, sp.obs_scale, sp.obs_ub, sp.n_obs, R.bvec[P], rng, pit,
                               SITE_OBS_SCALE, lane);
      emit_obs = obs_scale;
      const int s = it - 1 - g.W;
      if (s >= 0) {
        const size_t o = chain_lin * g.S + s;
        if (lane == 0) {
          if (g.out_obs) g.out_obs[o] = (float)obs_scale;
          if (g.out_level_scale) g.out_level_scale[o] = (float)level_scale;
          if (g.out_slope_scale) g.out_slope_scale[o] = (float)(a.has_slope ? slope_scale : 0.0);
        }
        if (a.out_drift && lane < K) a.out_drift[o * K + lane] = (float)drift[lane < K ? lane : 0];
        if (g.out_weights && lane < P) g.out_weights[o * P + lane] = R.w[lane];
        // level / seasonal contributions / posterior-predictive trajectory of iteration it-1
        const float so = (float)emit_obs;
        const size_t row = o * T;
        for (int c = lane; c < (T + 3) / 4; c += 64) {
          float zp[4];
          normals4(site_call(rng, pit, SITE_PRED, 0, (uint32_t)c), zp);
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * c + q;
            if (t < T) {
              float loc = lev[t] + xw[t];
              for (int k = 0; k < K; ++k) {
                const float sv = seas[k * TS + t];
                loc += sv;
                if (a.out_seasonal) a.out_seasonal[(row + t) * K + k] = sv;
              }
              if (g.out_level) g.out_level[row + t] = lev[t];
              if (g.out_slope && a.has_slope) g.out_slope[row + t] = slp[t];
              if (g.out_traj) g.out_traj[row + t] = fmaf(so, zp[q], loc);
              if (g.out_pred_mean) {
                float* pm = g.out_pred_mean + chain_lin * T + t;   // running sum, scaled at the end
                *pm = (s == 0 ? 0.f : *pm) + loc;
              }
            }
          }
        }
      }
    }
    if (it == n_iter) break;
    if (P > 0) {
      const double g_obs = gamma_wave(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
      obs_scale = spike_slab_draw_regs(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, pc);
    }
    wave_sync();

    // ---- (3) residual, normals of this iteration
    for (int t = lane; t < T; t += 64) {
      float s = 0.f;
      for (int j = 0; j < P; ++j) s = fmaf(Xg[(size_t)j * T + t], R.w[j], s);
      xw[t] = s;
    }
    for (int c = lane; c < (T + 3) / 4; c += 64) {
      float z4[4];
      normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_LEVEL, 0, (uint32_t)c), z4);
      for (int q = 0; q < 4; ++q) if (4 * c + q < T) zl[4 * c + q] = z4[q];
      normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_OBS, 0, (uint32_t)c), z4);
      for (int q = 0; q < 4; ++q) if (4 * c + q < T) zo[4 * c + q] = z4[q];
      if (a.has_slope) {
        normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_SLOPE, 0, (uint32_t)c), z4);
        for (int q = 0; q < 4; ++q) if (4 * c + q < T) zs[4 * c + q] = z4[q];
      }
      for (int k = 0; k < K; ++k) {
        normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_SEAS, (uint32_t)k, (uint32_t)c), z4);
        for (int q = 0; q < 4; ++q) if (4 * c + q < T) zk[k * TS + 4 * c + q] = z4[q];
      }
    }
    if (lane < a.dred) {
      float z1[1];
      fill_normals<1>(rng, (uint32_t)it, SITE_PRIOR_INIT, 0, (uint32_t)lane, z1);
      zi[lane] = z1[0];
    }
    wave_sync();
    // x+_0 = chol(P_1) z in the oracle's reduced coordinates, folded into the prior mean
    if (lane < a.dred) {
      float s = 0.f;
      for (int j = 0; j <= lane; ++j) s = fmaf(chol1[lane * a.dred + j], zi[j], s);
      x0r[lane] = s;
    }
    wave_sync();
    float a1e = 0.f;
    if (lane == 0) a1e = (float)sp.init_level_loc + x0r[0];
    if (a.has_slope && lane == 1) a1e = x0r[1];
    if (blk >= 0) {
      if (pos < nb - 1) a1e = x0r[roff[blk] + pos];
      else { float s = 0.f; for (int q = 0; q < nb - 1; ++q) s += x0r[roff[blk] + q]; a1e = -s; }
    }
    const float so = (float)obs_scale, sl = (float)level_scale, ssc = (float)slope_scale;
    const float H = so * so;

    // ---- (4) pass 0: simulate x+ (zero initial state) and form y~ = resid - y+
    {
      float xp = 0.f;
      for (int t = 0; t < T; ++t) {
        const float zx = zsum(xp);
        if (lane == 0) ytil[t] = (yv[t] - xw[t]) - (zx + so * zo[t]);
        if (t + 1 < T) xp = sim_noise(transition(xp, t), t, sl, ssc, drift);
      }
    }
    // prior covariance of x_0 in full-effect form: sd^2 (I - 11'/n) per block
    for (int e = lane; e < D * D; e += 64) {
      const int i = ei[e], j = ej[e];
      float v = 0.f;
      if (i == j && i == 0) v = p1l;
      else if (a.has_slope && i == j && i == 1) v = p1s;
      else if (i >= trend && j >= trend) {
        int bi = -1, bj = -2, nn = 1;
        for (int k = 0; k < K; ++k) {
          if (i >= off[k] && i < off[k] + nsz[k]) { bi = k; nn = nsz[k]; }
          if (j >= off[k] && j < off[k] + nsz[k]) bj = k;
        }
        if (bi == bj) v = p1e * ((i == j ? 1.f : 0.f) - 1.f / (float)nn);
      }
      Pcur[e] = v;
    }
    wave_sync();

    // ---- (5) pass 1: Kalman filter, storing K_t and v_t / F_t
    {
      float am = a1e;
      for (int t = 0; t < T; ++t) {
        const bool obs = msk[t] == 0;
        float kfi = 0.f;
        if (obs) {
          float pz = 0.f;
          if (comp) {
            pz = Pcur[lane * D + 0];
            for (int k = 0; k < K; ++k) pz += Pcur[lane * D + off[k]];
            pzv[lane] = pz;
          }
          const float F = zsum(pz) + H;
          const float rF = 1.0f / F;
          const float v = ytil[t] - zsum(am);
          kfi = pz * rF;
          if (lane == 0) vf[t] = v * rF;
          am = fmaf(kfi, v, am);
          wave_sync();
          for (int e = lane; e < D * D; e += 64) Pcur[e] -= pzv[ei[e]] * pzv[ej[e]] * rF;
          wave_sync();
        } else if (lane == 0) {
          vf[t] = 0.f;
        }
        if (comp) kf[(size_t)t * D + lane] = kfi;
        if (t + 1 < T) {
          am = transition(am, t);
          bool moved = a.has_slope != 0;
          for (int k = 0; k < K; ++k) moved = moved || chg[k * T + t] != 0;
          if (!moved) {
            if (lane == 0) Pcur[0] += sl * sl;
            wave_sync();
          } else {
            for (int e = lane; e < D * D; e += 64) {
              const int i = ei[e], j = ej[e];
              // source index of row/column under the block shifts
              int si = i, sj = j, bi = -1, bj = -2, nn = 1;
              for (int k = 0; k < K; ++k) {
                const bool ci = i >= off[k] && i < off[k] + nsz[k];
                const bool cj = j >= off[k] && j < off[k] + nsz[k];
                if (ci) { bi = k; nn = nsz[k]; }
                if (cj) bj = k;
                if (chg[k * T + t]) {
                  if (ci) si = off[k] + (i - off[k] + 1) % nsz[k];
                  if (cj) sj = off[k] + (j - off[k] + 1) % nsz[k];
                }
              }
              float v = Pcur[si * D + sj];
              if (a.has_slope) {       // level <- level + slope
                if (i == 0) v += Pcur[1 * D + sj];
                if (j == 0) v += Pcur[si * D + 1];
                if (i == 0 && j == 0) v += Pcur[1 * D + 1];
              }
              if (i == 0 && j == 0) v += sl * sl;
              if (a.has_slope && i == 1 && j == 1) v += ssc * ssc;
              if (bi == bj && bi >= 0 && chg[bi * T + t]) {
                const float dk = (float)drift[bi], inv = 1.0f / (float)nn;
                const float gi = (i - off[bi] == nn - 1) ? 1.f - inv : -inv;
                const float gj = (j - off[bi] == nn - 1) ? 1.f - inv : -inv;
                v += dk * dk * gi * gj;
              }
              Pnxt[e] = v;
            }
            wave_sync();
            float* tmp = Pcur; Pcur = Pnxt; Pnxt = tmp;
          }
        }
      }
    }
    wave_sync();
    // ---- (6) pass 2: backward recursion, rs[t] = r_{t-1}
    {
      float r = 0.f;
      for (int t = T - 1; t >= 0; --t) {
        r = (t + 1 < T) ? transition_T(r, t) : 0.f;
        if (msk[t] == 0) {
          const float kfi = comp ? kf[(size_t)t * D + lane] : 0.f;
          const float kr = wave_sum(kfi * r);
          if (isz) r += vf[t] - kr;
        }
        if (comp) rs[(size_t)t * D + lane] = r;
      }
    }
    wave_sync();
    // ---- (7) pass 3: reconstruct x^ forward, re-simulate x+, write the draw, gather statistics
    {
      float xh = a1e;
      {   // x^_0 = a_1 + P_1 r_{-1}
        const float r0 = comp ? rs[lane] : 0.f;
        if (lane == 0) xh += p1l * r0;
        if (a.has_slope && lane == 1) xh += p1s * r0;
        if (blk >= 0) {
          float sb = 0.f;
          for (int q = 0; q < nb; ++q) sb += rs[boff + q];
          xh += p1e * (r0 - sb / (float)nb);
        }
      }
      float xp = 0.f, prev = 0.f, prev_next = 0.f;
      ssl = 0.f; sss = 0.f; ssd = 0.f;
      for (int t = 0; t < T; ++t) {
        const float xt = xh + xp;
        if (t > 0) {
          if (lane == 0) {
            float dl = xt - prev;
            if (a.has_slope) dl -= prev_next;       // slope_{t-1} is lane 1 = "next" of lane 0
            ssl = fmaf(dl, dl, ssl);
          }
          if (a.has_slope && lane == 1) { const float ds = xt - prev; sss = fmaf(ds, ds, sss); }
          if (blk >= 0 && pos == 0 && chg[blk * T + t - 1]) {
            const float w = (float)nb * (prev_next - xt);   // n (e_{t-1,1} - e_{t,0})
            ssd = fmaf(w, w, ssd);
          }
        }
        if (lane == 0) lev[t] = xt;
        if (a.has_slope && lane == 1) slp[t] = xt;
        if (blk >= 0 && pos == 0) seas[blk * TS + t] = xt;
        prev = xt;
        prev_next = __shfl_down(xt, 1, 64);
        if (t + 1 < T) {
          xh = transition(xh, t);
          const float rn = comp ? rs[(size_t)(t + 1) * D + lane] : 0.f;
          if (lane == 0) xh = fmaf(sl * sl, rn, xh);
          if (a.has_slope && lane == 1) xh = fmaf(ssc * ssc, rn, xh);
          if (blk >= 0 && chg[blk * T + t]) {
            float sb = 0.f;
            for (int q = 0; q < nb; ++q) sb += rs[(size_t)(t + 1) * D + boff + q];
            const float inv = 1.0f / (float)nb;
            const float gdot = rs[(size_t)(t + 1) * D + boff + nb - 1] - sb * inv;
            const float gi = (pos == nb - 1) ? 1.f - inv : -inv;
            const float dk = (float)drift[blk];
            xh = fmaf(dk * dk * gi, gdot, xh);
          }
          xp = sim_noise(transition(xp, t), t, sl, ssc, drift);
        }
      }
    }
    wave_sync();
  }
  if (g.out_pred_mean) {
    const float inv = 1.0f / (float)(g.S > 0 ? g.S : 1);
    for (int t = lane; t < T; t += 64) g.out_pred_mean[chain_lin * T + t] *= inv;
  }
}
#endif  // CI_SEASONAL_DECL_ONLY

}  // namespace ci
