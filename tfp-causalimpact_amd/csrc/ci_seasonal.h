// ci_seasonal.h -- Gibbs kernel for models with seasonal blocks (tfp.sts.Seasonal,
// allow_drift=True, constrain_mean_effect_to_zero=True; reference call site
// causalimpact/causalimpact_lib.py:471-489).
//
// State dimension is 1 (+1 slope) + sum_k num_seasons_k, too wide for the register-resident
// scan elements of ci_kernels.h, so this kernel is ONE WAVEFRONT PER CHAIN and sequential in time
// (DESIGN.md 3.3):
//   * lane i holds component i of every state-sized vector and ROW i of the covariance (LDS);
//   * every seasonal block is carried in its FULL n-effect form (the n-th effect is minus the
//     sum of the others), in SLOT coordinates: the effect of season s stays in lane off[k] + s and
//     the index c_k(t) of the observed slot moves (+1 mod n at each season change).  The
//     constrained dynamics then are the identity on the block plus rank-1 noise
//     sigma eta (e_{slot just observed} - 1/n), the observation row is e_0 + sum_k e_{off[k]+c_k(t)};
//     the covariance is singular but F = Z P Z' + H > 0.  Same Gaussian as the oracle's
//     (n-1)-dimensional form in permuted coordinates, so draws agree per random number.
//     (ci_gibbs64.h and ci_score_seq.h keep the rotated form of rounds 1-2: observed effect first,
//     cyclic lane shift at each change);
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
  float* ws;                         // time-parallel kernel (ci_wide.h): per-chain HBM workspace
  int Lc;                            //   and steps per thread
  // time-parallel kernel, several workgroups per chain (ci_wide.h "clusters"):
  int cluster;                       //   workgroups per chain (ci_wide.h: 1, 2, 4, 8 or 16; ci_seasonal_tp.h: 1 .. 32)
  int cluster_drop;                  //   test knob: this role exits before checking in (0 = none)
  int dk_lds;                        //   clusters of 16: the DK workers keep the draw's per-step rows in LDS
  int* csync;                        //   [B*C][32] handshake counters, zeroed before the launch
  float* cpart;                      //   [B*C][segments][4][RS] partial sums of X~'targets, y'y
  float* cw;                         //   [B*C][64] weights + emission scale of the iteration
  double* cv;                        //   [B*C][presweep_doubles(P)] regression matrix swept ahead + its pivot rows (presweep_export)
  size_t ws_stride;                  // sequential kernel: bytes of `ws` per chain (arrays over time when
                                     //   they are not in LDS, then the P > MAXP regression block)
  // Sequential kernel, LATENTS-ONLY mode (the pass after an HMC fit, ci_ll_session_hmc_run): block n
  // takes its parameters from lat_theta[n] = (sigma_obs, sigma_level, sigma_slope, drift[K],
  // beta[P]) instead of sampling them, draws ONE latent path (Durbin-Koopman, random stream of
  // chain chain_offset + n / lat_S, iteration n % lat_S) and emits it with its posterior-
  // predictive trajectory: what one_step_predictive does with each retained draw
  // (causalimpact_lib.py:609-632).  Launch with W = 0, S = 1, C = number of draws.
  const double* lat_theta;           //   NULL = the Gibbs sampler
  int lat_S;
};

struct SLayout {
  // arrays over time (LDS when they fit, else the per-chain HBM workspace)
  size_t yv, lev, slp, xw, ytil, vf, zl, zs, zo, seas, zk, gd, kf, rs, mask, cbits, cidx, t_total;
  // always in LDS
  size_t Pa, pzv, zi, x0r, d2, xtx, omega, bvec, w,
      aug0, pri0, chol, zv, uperm, nz, perm, idx, total;
};

// global_ws: the arrays over time live in a per-chain HBM workspace (offsets from its base,
// t_total bytes) instead of LDS, which removes the LDS bound on the series length; the
// LDS-resident one-wavefront regression block of P > 16 needs its buffers in LDS as well.
__host__ __device__ inline SLayout make_slayout(int T, int P, int K, int D, int dred,
                                                int has_slope, int global_ws = 0) {
  SLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const size_t TS = (size_t)((T + 3) & ~3);
  const size_t Tf = sizeof(float) * TS;
  const int Pp = P > 0 ? P : 1, Kp = K > 0 ? K : 1;
  const bool bigp = P > MAXP;          // every O(P^2) array lives in the HBM workspace (spike_slab_draw_big)
  const bool big = P > 16 && !bigp;
  l.xtx = take(bigp ? 16 : sizeof(double) * Pp * Pp);
  l.omega = take(bigp ? 16 : sizeof(double) * Pp * Pp);
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.aug0 = take(big ? sizeof(double) * sweep_padded((size_t)(Pp + 1) * (Pp + 1)) : 16);
  l.pri0 = take(big ? sizeof(double) * sweep_padded((size_t)Pp * Pp) : 16);
  l.chol = take(big ? sizeof(double) * Pp * Pp : 16);
  l.zv = take(big ? sizeof(double) * Pp : 16);
  l.uperm = take(big ? sizeof(double) * Pp : 16);
  l.nz = take(big ? sizeof(int) * Pp : 16);
  l.perm = take(big ? sizeof(int) * Pp : 16);
  l.idx = take(big ? sizeof(int) * Pp : 16);
  // the covariance (mirror of the lanes' register rows): D rows of stride ((D + 7) & ~7) + 4 floats
  l.Pa = take(sizeof(float) * D * ((((size_t)D + 7) & ~(size_t)7) + 4));
  // P z [72], the blocks' shock vectors [SMAXK][72], the observed columns [16]
  l.pzv = take(sizeof(float) * (72 + SMAXK * 72 + 16)); l.zi = take(sizeof(float) * (dred + 1));
  l.x0r = take(sizeof(float) * (dred + 1));
  l.d2 = take(sizeof(float) * SMAXK);
  l.w = take(sizeof(float) * (Pp > 16 ? Pp : 16));
  const size_t lds_fixed = o;
  if (global_ws) o = 0;
  l.yv = take(Tf); l.lev = take(Tf); l.slp = take(has_slope ? Tf : 16); l.xw = take(Tf);
  l.ytil = take(Tf); l.vf = take(Tf); l.zl = take(Tf); l.zs = take(has_slope ? Tf : 16);
  l.zo = take(Tf);
  l.seas = take(Tf * Kp); l.zk = take(Tf * Kp); l.gd = take(Tf * Kp);
  l.kf = take(sizeof(float) * (size_t)T * D); l.rs = take(sizeof(float) * (size_t)T * D);
  l.mask = take(TS); l.cbits = take(TS); l.cidx = take(TS * Kp);
  if (global_ws) { l.t_total = o; l.total = lds_fixed; }
  else { l.t_total = 0; l.total = o; }
  return l;
}

#ifndef CI_SEASONAL_DECL_ONLY
// Pass 1 of the sequential kernel (the Kalman filter in slot coordinates), its own function so that
// the per-step loop gets its own register allocation: inlined, the kernel's ~60 live scalars spill
// and every step reloads dozens of them through v_readlane.
struct SeasFilterArgs {
  int T, D, DS, lane, blk, pos, nb, boff, has_slope;
  float a1e, H, ql, qs, myd2, rnb;
  float* Pm;            // [D][DS] covariance rows
  float* pzv;           // [72] P z  | gvk [SMAXK][72] shock vector of each block | zcol [16]
  float* kf;            // [T][D]
  float* vf;            // [T]
  const float* ytil;
  const uint8_t* cbv;
  const uint8_t* msk;
  const uint8_t* cidb;  // c_k(t) of this lane's block
};
// Pointers arrive as generic ones (a struct in private memory); they are cast back to their address
// spaces here -- LDS for the covariance and the step's vectors, LDS or (GWS) global for the arrays
// over time -- so that the loop is ds_read / global_load instead of flat_load.
#define CI_LDS __attribute__((address_space(3)))
#define CI_GLB __attribute__((address_space(1)))
typedef float ci_f4v __attribute__((ext_vector_type(4)));
typedef int ci_i4v __attribute__((ext_vector_type(4)));
// NCH: the own covariance row lives in REGISTERS (8 NCH floats, D <= 8 NCH); LDS holds a mirror for
// the accesses with a run-time column (P z) and for lane 0's view of row 1.
template <bool GWS, int NCH>
static __device__ __noinline__ void seasonal_filter_pass(const SeasFilterArgs& p) {
  const int T = p.T, D = p.D, DS = p.DS, lane = p.lane, blk = p.blk, pos = p.pos, nb = p.nb;
  const bool slope = p.has_slope != 0;
  const bool comp = lane < D;
  const float H = p.H, ql = p.ql, qs = p.qs, myd2 = p.myd2, rnb = p.rnb;
  CI_LDS float* Pm = (CI_LDS float*)p.Pm;
  CI_LDS float* Prow = Pm + (comp ? lane : 0) * DS;
  CI_LDS float* pzv = (CI_LDS float*)p.pzv;
  CI_LDS float* gvk = pzv + 72;
  CI_LDS int* zcol = (CI_LDS int*)(gvk + SMAXK * 72);
  const int blk0 = blk >= 0 ? blk : 0;
  CI_LDS const float* gmine = gvk + blk0 * 72;   // the shock vector of the own block (zero elsewhere)
  CI_LDS float* gslot = gvk + blk0 * 72 + lane;  // where lane j publishes g_j
  using TF = typename std::conditional<GWS, CI_GLB float, CI_LDS float>::type;
  using TB = typename std::conditional<GWS, CI_GLB const uint8_t, CI_LDS const uint8_t>::type;
  using TF4 = typename std::conditional<GWS, CI_GLB ci_f4v, CI_LDS ci_f4v>::type;
  using TU = typename std::conditional<GWS, CI_GLB const uint32_t, CI_LDS const uint32_t>::type;
  TF* kfw = (TF*)p.kf + lane;
  TF* vfp = (TF*)p.vf;
  TF* ytil = (TF*)p.ytil;
  TB* cbv = (TB*)p.cbv;
  TB* msk = (TB*)p.msk;
  TB* cidb = (TB*)p.cidb;
  auto ld4 = [](CI_LDS const float* q) -> ci_f4v { return *(CI_LDS const ci_f4v*)q; };
  auto ldt4 = [](TF* q) -> ci_f4v { return *(TF4*)q; };
  auto ldb4 = [](TB* q) { return *(TU*)q; };
  auto at4 = [](const ci_f4v& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };
  auto lds_sync = []() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  // observed seasonal columns: zcol[k] = off[k] + c_k(t); unused entries point at a zero column
  if (lane < 8) zcol[lane] = D;
  for (int e = lane; e < 72 + SMAXK * 72; e += 64) pzv[e] = 0.f;
  lds_sync();
  if (blk >= 0 && pos == 0) zcol[blk] = p.boff;
  lds_sync();
  float prow[8 * NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ci_f4v a0 = ci_f4v{0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (8 * c < D) { a0 = ld4(Prow + 8 * c); a1 = ld4(Prow + 8 * c + 4); }
    prow[8 * c] = a0.x; prow[8 * c + 1] = a0.y; prow[8 * c + 2] = a0.z; prow[8 * c + 3] = a0.w;
    prow[8 * c + 4] = a1.x; prow[8 * c + 5] = a1.y; prow[8 * c + 6] = a1.z; prow[8 * c + 7] = a1.w;
  }
  float am = p.a1e;
  for (int t4 = 0; t4 < T; t4 += 4) {
    const ci_f4v yt4 = ldt4(ytil + t4);
    const uint32_t cb4 = ldb4(cbv + t4), mk4 = ldb4(msk + t4), cw4 = ldb4(cidb + t4);
    float vfq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t4 + q;
      vfq[q] = 0.f;
      if (t >= T) continue;
      const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
      const unsigned cb = (t + 1 < T) ? ((cb4 >> (8 * q)) & 0xFFu) : 0u;
      const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
      const bool isz = lane == 0 || (blk >= 0 && pos == mycur);
      const bool mych = blk >= 0 && ((cb >> blk) & 1u) != 0u;
      float kfi = 0.f, rF = 0.f, pz = 0.f;
      if (obs) {
        if (comp) {
          const ci_i4v z0 = *(CI_LDS const ci_i4v*)zcol, z1 = *(CI_LDS const ci_i4v*)(zcol + 4);
          pz = prow[0];
          pz += Prow[z0.x]; pz += Prow[z0.y]; pz += Prow[z0.z]; pz += Prow[z0.w];
          pz += Prow[z1.x]; pz += Prow[z1.y]; pz += Prow[z1.z]; pz += Prow[z1.w];
        }
        const float F = wave_sum_dpp(isz ? pz : 0.f) + H;
        rF = __builtin_amdgcn_rcpf(F);
        rF = fmaf(fmaf(-F, rF, 1.0f), rF, rF);
        const float v = at4(yt4, q) - wave_sum_dpp(isz ? am : 0.f);
        kfi = pz * rF;
        vfq[q] = v * rF;
        am = fmaf(kfi, v, am);
      }
      const float gi = mych ? ((pos == mycur ? 1.f : 0.f) - rnb) : 0.f;   // g_i of this step's shock
      if (comp) {
        *kfw = kfi;
        pzv[lane] = pz;
        if (blk >= 0) *gslot = gi;
      }
      kfw += D;
      if (t + 1 == T) continue;
      if (slope) {
        const float m1 = readlane_f(am, 1);
        if (lane == 0) am += m1;
      }
      if (!obs && cb == 0u && !slope) {
        if (lane == 0) { prow[0] += ql; Prow[0] = prow[0]; }
        continue;
      }
      lds_sync();
      if (comp) {
        // One sweep of the own row (registers), every load of the step's vectors issued first:
        //   P'[i][j] = P[i][j] - (Pz)_i (Pz)_j / F  + sigma_k^2 g_i g_j   (g: own block's shock)
        // and, with a slope, level <- level + slope on rows (lane 0 adds row 1) and columns.
        const float pz1 = slope ? pzv[1] : 0.f;
        ci_f4v pj4[2 * NCH], gj4[2 * NCH], p14[2 * NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (8 * c < D) {
            pj4[2 * c] = ld4(pzv + 8 * c); pj4[2 * c + 1] = ld4(pzv + 8 * c + 4);
            gj4[2 * c] = ld4(gmine + 8 * c); gj4[2 * c + 1] = ld4(gmine + 8 * c + 4);
            if (slope) { p14[2 * c] = ld4(Pm + DS + 8 * c); p14[2 * c + 1] = ld4(Pm + DS + 8 * c + 4); }
          }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (8 * c < D) {
            const float pj[8] = {pj4[2 * c].x, pj4[2 * c].y, pj4[2 * c].z, pj4[2 * c].w,
                                 pj4[2 * c + 1].x, pj4[2 * c + 1].y, pj4[2 * c + 1].z, pj4[2 * c + 1].w};
            const float gj[8] = {gj4[2 * c].x, gj4[2 * c].y, gj4[2 * c].z, gj4[2 * c].w,
                                 gj4[2 * c + 1].x, gj4[2 * c + 1].y, gj4[2 * c + 1].z, gj4[2 * c + 1].w};
            if (slope) {
              const float p1[8] = {p14[2 * c].x, p14[2 * c].y, p14[2 * c].z, p14[2 * c].w,
                                   p14[2 * c + 1].x, p14[2 * c + 1].y, p14[2 * c + 1].z, p14[2 * c + 1].w};
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                float r = fmaf(-(pz * pj[u]), rF, prow[8 * c + u]);
                if (lane == 0) r += fmaf(-(pz1 * pj[u]), rF, p1[u]);
                prow[8 * c + u] = fmaf(myd2, gi * gj[u], r);
              }
              if (c == 0) {
                prow[0] += prow[1];
                if (lane == 1) prow[1] += qs;
              }
            } else {
#pragma unroll
              for (int u = 0; u < 8; ++u)
                prow[8 * c + u] = fmaf(myd2, gi * gj[u], fmaf(-(pz * pj[u]), rF, prow[8 * c + u]));
            }
            if (c == 0 && lane == 0) prow[0] += ql;
            *(CI_LDS ci_f4v*)(Prow + 8 * c) = ci_f4v{prow[8 * c], prow[8 * c + 1], prow[8 * c + 2], prow[8 * c + 3]};
            *(CI_LDS ci_f4v*)(Prow + 8 * c + 4) = ci_f4v{prow[8 * c + 4], prow[8 * c + 5], prow[8 * c + 6], prow[8 * c + 7]};
          }
      }
      // the changing blocks observe their next slot from t + 1 on
      if (mych && pos == 0) zcol[blk] = p.boff + ((mycur + 1 == nb) ? 0 : mycur + 1);
      lds_sync();
    }
    if (lane == 0) *(TF4*)(vfp + t4) = ci_f4v{vfq[0], vfq[1], vfq[2], vfq[3]};
  }
}

// BIGP: the P > MAXP build (regression block in the HBM workspace, spike_slab_draw_big); its own
// instantiation so that the call does not cost the P <= MAXP builds a stack frame.
template <bool GWS, bool BIGP = false>
__global__ __launch_bounds__(64) void gibbs_seasonal_kernel(SArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const KArgs& g = a.k;
  const int T = g.T, P = g.P, K = a.K;
  const int series = blockIdx.x / g.C, chain = blockIdx.x % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  const int trend = a.has_slope ? 2 : 1;
  // block geometry in registers: every loop over blocks is fully unrolled with static indices
  int off[SMAXK], nsz[SMAXK], roff[SMAXK];
  int D = trend;
  {
    int rr = trend;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) {
      off[k] = D; roff[k] = rr; nsz[k] = (k < K) ? a.nseas[k] : 0;
      if (k < K) { D += nsz[k]; rr += nsz[k] - 1; }
    }
  }
  const SLayout L = make_slayout(T, P, K, D, a.dred, a.has_slope, GWS ? 1 : 0);
  // the arrays over time: LDS, or (GWS) this chain's slice of the HBM workspace -- every access
  // below is either one 16-byte row per 4 steps or lane-contiguous, and a chain only ever reads
  // what it wrote, so the slice stays in this XCD's L2
  unsigned char* tb_ = smem;
  unsigned char* wsc = reinterpret_cast<unsigned char*>(a.ws) + chain_lin * a.ws_stride;   // this chain's slice
  if constexpr (GWS) tb_ = wsc;
  float* yv = (float*)(tb_ + L.yv); float* lev = (float*)(tb_ + L.lev);
  float* slp = (float*)(tb_ + L.slp); float* xw = (float*)(tb_ + L.xw);
  float* ytil = (float*)(tb_ + L.ytil); float* vf = (float*)(tb_ + L.vf);
  float* zl = (float*)(tb_ + L.zl); float* zs = (float*)(tb_ + L.zs);
  float* zo = (float*)(tb_ + L.zo); float* seas = (float*)(tb_ + L.seas);
  float* zk = (float*)(tb_ + L.zk); float* gd = (float*)(tb_ + L.gd);
  float* kf = (float*)(tb_ + L.kf);
  float* rs = (float*)(tb_ + L.rs); float* Pm = (float*)(smem + L.Pa);   // covariance rows, stride DS
  float* pzv = (float*)(smem + L.pzv);
  float* zi = (float*)(smem + L.zi); float* x0r = (float*)(smem + L.x0r);
  float* d2 = (float*)(smem + L.d2);
  uint8_t* msk = tb_ + L.mask; uint8_t* cbv = tb_ + L.cbits;
  uint8_t* cidx = tb_ + L.cidx;       // [K][TS]: the slot of block k that step t observes
  const int TS = (T + 3) & ~3;       // padded length of every T-array (4-step blocks)
  RegLds R;
  R.xtx = (double*)(smem + L.xtx); R.omega = (double*)(smem + L.omega);
  R.bvec = (double*)(smem + L.bvec); R.w = (float*)(smem + L.w);
  // P > 16: the LDS-resident one-wavefront regression block (spike_slab_draw) and its buffers
  R.aug[0] = (double*)(smem + L.aug0); R.aug[1] = R.aug[0];
  R.pri[0] = (double*)(smem + L.pri0); R.pri[1] = R.pri[0];
  R.chol = (double*)(smem + L.chol); R.zv = (double*)(smem + L.zv);
  R.uperm = (double*)(smem + L.uperm);
  R.nz = (int*)(smem + L.nz); R.perm = (int*)(smem + L.perm); R.idx = (int*)(smem + L.idx);
  constexpr bool bigp = BIGP;
  if constexpr (BIGP) {
    // any number of covariates: the O(P^2) arrays in this chain's workspace, X'X and Omega read
    // where the setup kernel left them
    R.xtx = const_cast<double*>(g.xtx) + (size_t)series * P * P;
    R.omega = const_cast<double*>(g.omega) + (size_t)series * P * P;
    bigp_point(R, wsc + (GWS ? ((L.t_total + 255) & ~(size_t)255) : 0), P);
  }

  const DevSeriesParams sp = g.sp[series];
  const DevSeasonalParams ss = a.ssp[series];
  Rng rng{stream_key0(g.seed0, g.series_stream_base, series), stream_key1(g.seed1, g.series_stream_base, series), (uint32_t)(g.chain_offset + chain)};
  const bool lat = a.lat_theta != nullptr;
  uint32_t itb = 0u;                          // iteration offset of the random stream
  const double* lth = nullptr;
  if (lat) {
    rng.chain = (uint32_t)(g.chain_offset + chain / a.lat_S);
    itb = (uint32_t)(chain % a.lat_S);
    lth = a.lat_theta + (size_t)chain * (3 + K + P);
  }
  const float* Xg = g.Xt + (size_t)series * P * T;
  const float* chol1 = a.p1_chol + (size_t)series * a.dred * a.dred;

  // ---- lane roles: component `lane` of the state, block membership, shift partners
  int blk = -1, pos = 0, nb = 1, boff = 0, rbase = 0;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k)
    if (k < K && lane >= off[k] && lane < off[k] + nsz[k]) {
      blk = k; pos = lane - off[k]; nb = nsz[k]; boff = off[k]; rbase = roff[k];
    }
  const bool comp = lane < D;
  const int blk0 = blk >= 0 ? blk : 0;

  // ---- stage constants
  for (int t = lane; t < TS; t += 64) {
    const bool in = t < T;
    const bool m = in ? g.mask[(size_t)series * T + t] != 0 : true;
    msk[t] = m ? 1 : 0;
    yv[t] = m ? 0.f : g.y[(size_t)series * T + t];
    lev[t] = 0.f; xw[t] = 0.f; ytil[t] = 0.f; vf[t] = 0.f; zl[t] = 0.f; zo[t] = 0.f;
    if (a.has_slope) { slp[t] = 0.f; zs[t] = 0.f; }
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        seas[k * TS + t] = 0.f; zk[k * TS + t] = 0.f; gd[k * TS + t] = 0.f;
        if (in && a.season_change[(size_t)k * T + t]) bits |= 1u << k;
      }
    cbv[t] = (uint8_t)bits;
  }
  if (!bigp && !lat)
    for (int e = lane; e < P * P; e += 64) {
      R.xtx[e] = g.xtx[(size_t)series * P * P + e];
      R.omega[e] = g.omega[(size_t)series * P * P + e];
    }
  for (int j = lane; j < (P > 16 ? P : 16); j += 64) R.w[j] = 0.f;
  wave_sync();
  double n_changes[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) {
    n_changes[k] = 0.0;
    if (k < K) {
      float c = 0.f;
      for (int t = lane; t + 1 < T; t += 64) c += ((cbv[t] >> k) & 1) ? 1.f : 0.f;
      n_changes[k] = (double)wave_sum_dpp(c);
    }
  }

  double obs_scale = sp.obs_scale0, level_scale = sp.level_scale0, slope_scale = sp.slope_scale0;
  double drift[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) drift[k] = (k < K) ? ss.drift_scale0[k] : 0.0;
  float ssl = 0.f, sss = 0.f, ssd = 0.f;   // lane 0 / lane 1 / lane off[k] accumulate
  const float p1l = (float)(sp.init_level_scale * sp.init_level_scale);
  const float p1s = (float)(sp.init_slope_scale * sp.init_slope_scale);
  const float p1e = (float)(ss.init_seasonal_scale * ss.init_seasonal_scale);
  Prof prof;
  prof.start(g.prof, g.prof != nullptr && blockIdx.x == 0 && lane == 0);
  PriorCarry pc;
  pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;

  auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  auto ldb4 = [](const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); };
  auto at4 = [](const float4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };

  const int n_iter = g.W + g.S;
  if (lat) {
    obs_scale = lth[0]; level_scale = lth[1]; slope_scale = lth[2];
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) if (k < K) drift[k] = lth[3 + k];
    for (int j = lane; j < P; j += 64) R.w[j] = (float)lth[3 + K + j];
    wave_sync();
  }
  for (int it = 0; it <= n_iter; ++it) {
    // ---- (1) X~'targets, y'y from the current latents
    if (!lat) {
      float yty = 0.f;
      for (int t = lane; t < T; t += 64) {
        float tg = 0.f;
        if (!msk[t]) {
          tg = yv[t] - lev[t];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K) tg -= seas[k * TS + t];
        }
        ytil[t] = tg;            // reused as the targets buffer here
        yty = fmaf(tg, tg, yty);
      }
      wave_sync();
      for (int j = 0; j < P; ++j) {
        float pj = 0.f;
        for (int t = lane; t < T; t += 64) pj = fmaf(Xg[(size_t)j * T + t], ytil[t], pj);
        const float s = wave_sum_dpp(pj);
        if (lane == 0) R.bvec[j] = (double)s;
      }
      const float s0 = wave_sum_dpp(yty);
      if (lane == 0) R.bvec[P] = (double)s0;
      wave_sync();
    }
    prof.tick(20);
    // ---- (2) scale draws of iteration it-1, regression draw of iteration it
    double emit_obs = obs_scale;
    if (it > 0) {
      const uint32_t pit = (uint32_t)(it - 1) + itb;
      if (!lat) {
      const double v_l = (double)readlane_f(ssl, 0);
      level_scale = scale_draw(sp.level_conc, sp.level_scale, sp.level_ub, (double)(T - 1), v_l,
                               rng, pit, SITE_LEVEL_SCALE, lane);
      if (a.has_slope) {
        const double v_s = (double)readlane_f(sss, 1);
        slope_scale = scale_draw(sp.slope_conc, sp.slope_scale, sp.slope_ub, (double)(T - 1), v_s,
                                 rng, pit, SITE_SLOPE_SCALE, lane);
      }
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          const double v_d = (double)readlane_f(ssd, off[k]);
          const double gk = gamma_wave(ss.drift_conc + 0.5 * n_changes[k], rng, pit,
                                       SITE_DRIFT_SCALE, (uint32_t)k, lane);
          const double sd = (double)__fsqrt_rn((float)((ss.drift_scale + 0.5 * v_d) * fast_rcp(gk)));
          drift[k] = sd < ss.drift_ub ? sd : ss.drift_ub;
        }
      if (P == 0)
        obs_scale = scale_draw(sp.obs_conc, sp.obs_scale, sp.obs_ub, sp.n_obs, R.bvec[P], rng, pit,
                               SITE_OBS_SCALE, lane);
      }
      emit_obs = obs_scale;
      const int s = it - 1 - g.W;
      if (s >= 0) {
        const size_t o = chain_lin * g.S + s;
        if (lane == 0) {
          if (g.out_obs) g.out_obs[o] = (float)obs_scale;
          if (g.out_level_scale) g.out_level_scale[o] = (float)level_scale;
          if (g.out_slope_scale) g.out_slope_scale[o] = (float)(a.has_slope ? slope_scale : 0.0);
        }
        if (a.out_drift) {
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K && lane == k) a.out_drift[o * K + k] = (float)drift[k];
        }
        if (g.out_weights)
          for (int j = lane; j < P; j += 64) g.out_weights[o * P + j] = R.w[j];
        // level / seasonal contributions / posterior-predictive trajectory of iteration it-1
        const float so = (float)emit_obs;
        const size_t row = o * T;
        for (int c = lane; c < (T + 3) / 4; c += 64) {
          float zp[4];
          normals4(site_call(rng, pit, SITE_PRED, 0, (uint32_t)c), zp);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * c + q;
            if (t < T) {
              float loc = lev[t] + xw[t];
#pragma unroll
              for (int k = 0; k < SMAXK; ++k)
                if (k < K) {
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
    if (P > 0 && !lat) {
      const double g_obs = gamma_wave(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
      if (P <= 16)
        obs_scale = spike_slab_draw_regs(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, pc);
      else if constexpr (!BIGP)
        obs_scale = spike_slab_draw(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, it == 0);
      else
        obs_scale = spike_slab_draw_big(R, R.w, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, it == 0);
    }
    wave_sync();
    prof.tick(21);

    // ---- (3) residual, normals of this iteration
    for (int t = lane; t < T; t += 64) {
      float s = 0.f;
      for (int j = 0; j < P; ++j) s = fmaf(Xg[(size_t)j * T + t], R.w[j], s);
      xw[t] = s;
    }
    for (int c = lane; c < (T + 3) / 4; c += 64) {
      float z4[4];
      normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_LEVEL, 0, (uint32_t)c), z4);
      *reinterpret_cast<float4*>(zl + 4 * c) = make_float4(z4[0], z4[1], z4[2], z4[3]);
      normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_OBS, 0, (uint32_t)c), z4);
      *reinterpret_cast<float4*>(zo + 4 * c) = make_float4(z4[0], z4[1], z4[2], z4[3]);
      if (a.has_slope) {
        normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_SLOPE, 0, (uint32_t)c), z4);
        *reinterpret_cast<float4*>(zs + 4 * c) = make_float4(z4[0], z4[1], z4[2], z4[3]);
      }
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_SEAS, (uint32_t)k, (uint32_t)c), z4);
          *reinterpret_cast<float4*>(zk + k * TS + 4 * c) = make_float4(z4[0], z4[1], z4[2], z4[3]);
        }
    }
    if (lane < a.dred) {
      float z1[1];
      fill_normals<1>(rng, (uint32_t)it + itb, SITE_PRIOR_INIT, 0, (uint32_t)lane, z1);
      zi[lane] = z1[0];
    }
    float mydrift = 0.f;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        if (blk == k) mydrift = (float)drift[k];
        if (lane == 0) d2[k] = (float)(drift[k] * drift[k]);
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
      if (pos < nb - 1) a1e = x0r[rbase + pos];
      else { float s = 0.f; for (int q = 0; q < nb - 1; ++q) s += x0r[rbase + q]; a1e = -s; }
    }
    const float so = (float)obs_scale, sl = (float)level_scale, ssc = (float)slope_scale;
    const float H = so * so, ql = sl * sl, qs = ssc * ssc;
    const float* zkb = zk + blk0 * TS;
    prof.tick(22);

    // ---- The four passes over time run in SLOT coordinates: a seasonal block keeps the effect of
    // season s in lane off[k] + s for the whole series, and what moves is the index c_k(t) of the
    // slot that step t observes (cidx[k][t]; +1 mod n at every season change).  The explicit form
    // (effects rotated so that the observed one sits first) is the same Gaussian in permuted
    // coordinates -- the position of slot s at time t is (s - c_k(t)) mod n -- but there the
    // transition is a cross-lane rotation of every state-sized vector and of the covariance's rows
    // AND columns at each change; here the transition is the identity on the block (level += slope
    // aside), the observation row is e_0 + sum_k e_{off[k] + c_k(t)}, and the drift shock of a
    // change is sigma_k eta (e_{slot that was observed} - 1/n): rank one.  Lane i owns ROW i of the
    // covariance in LDS (odd row stride: column reads are conflict-free), so the measurement and
    // time updates are one in-place sweep of the own row -- no tables, no second buffer.
    const int DS = ((D + 7) & ~7) + 4;           // 16-byte rows, 8-column batches, stride = 4 mod 8
    float* Prow = Pm + (comp ? lane : 0) * DS;
    const float rnb = 1.0f / (float)nb;
    auto lds_sync = []() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    // sum over the observed components of a lane-distributed vector (Z x)
    auto zsum_slot = [&](float x, int mycur) -> float {
      const bool isz = lane == 0 || (blk >= 0 && pos == mycur);
      return wave_sum_dpp(isz ? x : 0.f);
    };
    // ---- (4) pass 0: simulate x+ (zero initial state), form y~ = resid - y+, record c_k(t).
    // Every pass walks time in blocks of 4 steps so that the per-step scalars arrive as one
    // batch of 16-byte loads instead of one exposed round trip each.
    {
      float xp = 0.f;
      int mycur = 0;
      for (int t4 = 0; t4 < T; t4 += 4) {
        const float4 zo4 = ld4(zo + t4), zl4 = ld4(zl + t4), zk4 = ld4(zkb + t4);
        const float4 yv4 = ld4(yv + t4), xw4 = ld4(xw + t4);
        float4 zs4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.has_slope) zs4 = ld4(zs + t4);
        const uint32_t cb4 = ldb4(cbv + t4);
        float yt[4];
        uint32_t curw = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          curw |= (uint32_t)mycur << (8 * q);
          const float zx = zsum_slot(xp, mycur);
          yt[q] = (at4(yv4, q) - at4(xw4, q)) - (zx + so * at4(zo4, q));
          if (t + 1 < T) {
            const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
            float r = xp;
            if (a.has_slope) {
              const float s1 = readlane_f(xp, 1);
              if (lane == 0) r += s1;
            }
            if (lane == 0) r = fmaf(sl, at4(zl4, q), r);
            if (a.has_slope && lane == 1) r = fmaf(ssc, at4(zs4, q), r);
            if (blk >= 0 && ((cb >> blk) & 1u)) {
              const float gi = (pos == mycur ? 1.f : 0.f) - rnb;
              r = fmaf(mydrift * gi, at4(zk4, q), r);
              mycur = (mycur + 1 == nb) ? 0 : mycur + 1;
            }
            xp = r;
          }
        }
        if (lane == 0) *reinterpret_cast<float4*>(ytil + t4) = make_float4(yt[0], yt[1], yt[2], yt[3]);
        if (blk >= 0 && pos == 0) *reinterpret_cast<uint32_t*>(cidx + blk * TS + t4) = curw;
      }
    }
    prof.tick(23);
    // prior covariance of x_0 (c_k(0) = 0: slots are positions): sd^2 (I - 11'/n) per block
    if (comp) {
      for (int j = D; j < DS; ++j) Prow[j] = 0.f;
      Prow[0] = lane == 0 ? p1l : 0.f;
      if (a.has_slope) Prow[1] = lane == 1 ? p1s : 0.f;
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K)
          for (int q = 0; q < nsz[k]; ++q)
            Prow[off[k] + q] = (blk == k) ? p1e * ((pos == q ? 1.f : 0.f) - rnb) : 0.f;
    }
    wave_sync();
    prof.tick(24);
    const uint8_t* cidb = cidx + blk0 * TS;      // c of this lane's block

    // ---- (5) pass 1: Kalman filter, storing K_t and v_t / F_t (seasonal_filter_pass above).  Per
    // step: P z from the K + 1 observed entries of the own row, then one sweep of the own row
    //   P'[i][j] = P[i][j] - (Pz)_i (Pz)_j / F   (+ the trend's T . T' and Q_t on the way).
    {
      SeasFilterArgs fa;
      fa.T = T; fa.D = D; fa.DS = DS; fa.lane = lane; fa.blk = blk; fa.pos = pos; fa.nb = nb; fa.boff = boff;
      fa.has_slope = a.has_slope;
      fa.a1e = a1e; fa.H = H; fa.ql = ql; fa.qs = qs; fa.myd2 = d2[blk0]; fa.rnb = rnb;
      fa.Pm = Pm; fa.pzv = pzv; fa.kf = kf; fa.vf = vf; fa.ytil = ytil; fa.cbv = cbv; fa.msk = msk; fa.cidb = cidb;
      if (D <= 16) seasonal_filter_pass<GWS, 2>(fa);
      else if (D <= 24) seasonal_filter_pass<GWS, 3>(fa);
      else if (D <= 32) seasonal_filter_pass<GWS, 4>(fa);
      else seasonal_filter_pass<GWS, 8>(fa);
    }
    wave_sync();
    prof.tick(25);
    // ---- (6) pass 2: backward recursion, rs[t] = r_{t-1}; then gd[k][t] = g . r_{t-1} per block
    // (the projection the forward reconstruction needs at season changes)
    {
      float r = 0.f;
      for (int t4 = ((T - 1) & ~3); t4 >= 0; t4 -= 4) {
        const float4 vf4 = ld4(vf + t4);
        const uint32_t mk4 = ldb4(msk + t4), cw4 = ldb4(cidb + t4);
        float kfq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kfq[q] = (comp && t4 + q < T) ? kf[(size_t)(t4 + q) * D + lane] : 0.f;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
          const int t = t4 + q;
          if (t >= T) continue;
          if (t + 1 < T) {
            if (a.has_slope) {                       // r <- T' r
              const float r0 = readlane_f(r, 0);
              if (lane == 1) r += r0;
            }
          } else {
            r = 0.f;
          }
          if (((mk4 >> (8 * q)) & 0xFFu) == 0u) {
            const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
            const float kr = wave_sum_dpp(kfq[q] * r);
            if (lane == 0 || (blk >= 0 && pos == mycur)) r += at4(vf4, q) - kr;
          }
          if (comp) rs[(size_t)t * D + lane] = r;
        }
      }
    }
    wave_sync();
    // g . r_{t-1} per block, time-parallel: g = e_{slot observed at t-1} - 1/n
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        const float rn = 1.0f / (float)nsz[k];
        for (int t = lane; t < T; t += 64) {
          const float* rr = rs + (size_t)t * D + off[k];
          float sb = 0.f;
          for (int q = 0; q < nsz[k]; ++q) sb += rr[q];
          const int cprev = t > 0 ? (int)cidx[k * TS + t - 1] : 0;
          gd[k * TS + t] = rr[cprev] - sb * rn;
        }
      }
    wave_sync();
    prof.tick(26);
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
      float xp = 0.f, prev = 0.f;
      ssl = 0.f; sss = 0.f; ssd = 0.f;
      bool ch_prev = false;
      const float* gdb = gd + blk0 * TS;
      for (int t4 = 0; t4 < T; t4 += 4) {
        const float4 zl4 = ld4(zl + t4), zk4 = ld4(zkb + t4);
        float4 zs4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.has_slope) zs4 = ld4(zs + t4);
        const uint32_t cb4 = ldb4(cbv + t4), cw4 = ldb4(cidb + t4);
        float rnq[4], gdq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // r_t = rs[t + 1] and g . r_t of this lane's block
          const int t1 = t4 + q + 1;
          rnq[q] = (comp && t1 < T) ? rs[(size_t)t1 * D + lane] : 0.f;
          gdq[q] = (t1 < T) ? gdb[t1] : 0.f;
        }
        float xo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          xo[q] = 0.f;
          if (t >= T) continue;
          const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
          const float xt = xh + xp;
          xo[q] = xt;
          if (t > 0) {
            const float pslope = readlane_f(prev, 1);     // slope_{t-1}
            if (lane == 0) {
              float dl = xt - prev;
              if (a.has_slope) dl -= pslope;
              ssl = fmaf(dl, dl, ssl);
            }
            if (a.has_slope && lane == 1) { const float ds = xt - prev; sss = fmaf(ds, ds, sss); }
            if (ch_prev && pos == mycur) {
              // the slot observed now received -eta/n at the change: eta = n (before - after)
              const float w = (float)nb * (prev - xt);
              ssd = fmaf(w, w, ssd);
            }
          }
          if (blk >= 0 && pos == mycur) seas[blk * TS + t] = xt;
          prev = xt;
          if (t + 1 < T) {
            const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
            const bool mych = blk >= 0 && ((cb >> blk) & 1u);
            float h = xh, r = xp;
            if (a.has_slope) {
              const float h1 = readlane_f(xh, 1), s1 = readlane_f(xp, 1);
              if (lane == 0) { h += h1; r += s1; }
            }
            if (lane == 0) { h = fmaf(ql, rnq[q], h); r = fmaf(sl, at4(zl4, q), r); }
            if (a.has_slope && lane == 1) { h = fmaf(qs, rnq[q], h); r = fmaf(ssc, at4(zs4, q), r); }
            if (mych) {
              const float dgi = mydrift * ((pos == mycur ? 1.f : 0.f) - rnb);   // sigma_d g_i
              h = fmaf(mydrift * dgi, gdq[q], h);
              r = fmaf(dgi, at4(zk4, q), r);
            }
            xh = h; xp = r;
            ch_prev = mych;
          }
        }
        // the trend components of the draw, 4 steps at a time
        if (lane == 0) *reinterpret_cast<float4*>(lev + t4) = make_float4(xo[0], xo[1], xo[2], xo[3]);
        if (a.has_slope && lane == 1)
          *reinterpret_cast<float4*>(slp + t4) = make_float4(xo[0], xo[1], xo[2], xo[3]);
      }
      // the drift statistic of a block: every slot collected its own changes
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          const float tot = wave_sum_dpp(blk == k ? ssd : 0.f);
          if (lane == off[k]) ssd = tot;
        }
    }
    wave_sync();
    prof.tick(27);
  }
  if (g.out_pred_mean) {
    const float inv = 1.0f / (float)(g.S > 0 ? g.S : 1);
    for (int t = lane; t < T; t += 64) g.out_pred_mean[chain_lin * T + t] *= inv;
  }
}
#endif  // CI_SEASONAL_DECL_ONLY

}  // namespace ci
