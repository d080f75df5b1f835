// ci_kernels5.h -- the latency build of the register-resident Gibbs kernel: FIVE wavefronts per
// chain.  Same sampler, same random stream, the same arithmetic in the same order as
// gibbs_kernel<D, L, 1> (ci_kernels.h), and -- the library is built with -ffp-contract=on, under
// which a multiply-add is fused iff it is ONE source expression, whatever it is inlined into --
// the same roundings: every draw of the two kernels is bit-identical
// (tests/test_gpu_gibbs.py::test_five_wave_latency_kernel_equals_the_four_wave_kernel), so the
// dispatch by launch size in ci_api.hip never changes a result.  What changes is WHO computes
// WHAT WHEN.
//
// In gibbs_kernel the serial regression section of wave 0 is half of every iteration
// (profiles/r02_a_phase_cycles.txt: 13.6k of 27k cycles), and most of it is a chain of
// dependent float64 sweeps of the posterior block A = sigma^2_prev Omega + X'X:
//   * sweeping the included features in        (4.7k cycles for 5 features),
//   * un-sweeping them again for the weights   (3.9k cycles).
// Neither depends on the data of the iteration: A is fixed by sigma^2_prev and the included set,
// both known when the PREVIOUS serial section ends; only the right-hand side X'targets and the
// normals are new.  So a fifth wavefront -- the regression wave -- owns the serial section and,
// while waves 0-3 run the Durbin-Koopman draw, already sweeps the NEXT iteration's matrix,
// recording for every sweep the multipliers (t_j, 1/pivot) and for every un-sweep (t_j, V_aa)
// in LDS.  The next serial section then only REPLAYS them on the new right-hand side
// (two fused multiply-adds per sweep instead of a full sweep) -- the same numbers the full sweeps
// would have produced, because a sweep's multipliers never depended on the right-hand side.
// An accepted inclusion flip invalidates the recorded un-sweeps for that iteration only: it falls
// back to the on-the-fly route of gibbs_kernel.
//
// Waves 0-3 are now all free during the serial section: each emits its own part of the
// previous draw and generates its own Durbin-Koopman normals (no hand-off buffer for wave 0).
// Barriers are workgroup-wide, so the regression wave executes the two barriers of dk_draw
// between its precompute steps (each step is shorter than the phase of the time waves it
// overlaps, so it is never the last to arrive).
#pragma once
#include "ci_kernels.h"

namespace ci {

constexpr int NT5 = NT + 64;          // 4 time waves + the regression wave
constexpr int PRE_MAXS = 16;          // recorded sweeps / un-sweeps (P <= 16)

// LDS tables written by the regression wave's precompute, read by its next serial section.
struct PreTables {
  double* tsw;      // [PRE_MAXS][64]  sweep s: t_j of lane j (1 at the pivot)
  double* tun;      // [PRE_MAXS][64]  un-sweep s: t_j
  double* rdsw;     // [PRE_MAXS]      sweep s: 1 / pivot
  double* vun;      // [PRE_MAXS]      un-sweep s: V_aa
  int* ksw;         // [PRE_MAXS]      sweep s: pivot feature
  int* kun;         // [PRE_MAXS]      un-sweep s: feature
};
__host__ __device__ inline size_t pre_tables_bytes() {
  return sizeof(double) * (2 * PRE_MAXS * 64 + 2 * PRE_MAXS) + sizeof(int) * 2 * PRE_MAXS + 16;
}

// What the precompute leaves in the regression wave's registers for the next serial section.
struct PreState {
  double c[4];      // swept posterior block (quadrant layout of QCols)
  double diag;
  unsigned long long S;   // the set it is swept on
  int n_sw, n_un;
  int valid;
};

// sweep_q_kr<KR, false> that also returns its multipliers (t of this lane, 1 / pivot).
template <int KR>
__device__ __forceinline__ void sweep_rec_kr(QCols& m, int k, int lane, double& t_out, double& rd_out) {
  const int kq = k >> 2, j = lane & 15, q = lane >> 4;
  const double ckr = m.c[KR];
  const double rd = fast_rcp(readlane_d(ckr, k + 16 * kq));
  const double rowk = bperm_d(ckr, j + 16 * kq);      // A[k][j]
  const bool isk = j == k;
  const double t = isk ? 1.0 : rowk * rd;
  double colk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) colk[r] = bperm_d(m.c[r], k + 16 * q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) m.c[r] = isk ? colk[r] * rd : m.c[r] - colk[r] * t;
  m.c[KR] = (q == kq) ? (isk ? -rd : t) : m.c[KR];
  m.diag = isk ? -rd : m.diag - rowk * rowk * rd;
  t_out = t;
  rd_out = rd;
}
__device__ __forceinline__ void sweep_rec(QCols& m, int k, int lane, double& t_out, double& rd_out) {
  switch (k & 3) {
    case 0: sweep_rec_kr<0>(m, k, lane, t_out, rd_out); break;
    case 1: sweep_rec_kr<1>(m, k, lane, t_out, rd_out); break;
    case 2: sweep_rec_kr<2>(m, k, lane, t_out, rd_out); break;
    default: sweep_rec_kr<3>(m, k, lane, t_out, rd_out); break;
  }
}

// The matrix work of the NEXT serial section: posterior block for sigma^2 = var_next swept on S,
// multipliers recorded; then (on a copy) the descending un-sweeps of the weights draw, recorded.
// `sync` is called after every step; the caller uses it to place the workgroup barriers of the
// phase the time waves are in.
template <class Sync>
__device__ __forceinline__ void regression_precompute(const RegLds& R, int P, double var_next,
                                                      unsigned long long S, int lane,
                                                      const PreTables& tb, PreState& ps, Sync sync) {
  const int j = lane & 15, q = lane >> 4;
  const bool live = j < P;
  const int col = live ? j : 0;
  QCols m;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * q + r;
    double om = 0.0, xx = 0.0;
    if (live && i < P) {
      om = R.omega[i * P + col];
      xx = R.xtx[i * P + col];
    }
    m.c[r] = om * var_next + xx;
    m.p[r] = 0.0;
  }
  m.diag = live ? R.omega[col * P + col] * var_next + R.xtx[col * P + col] : 1.0;
  m.cb = 0.0; m.corner = 0.0; m.pdiag = 0.0;
  int n = 0;
  for (unsigned long long pending = S; pending != 0ull; pending &= pending - 1ull) {
    const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
    double t, rd;
    sweep_rec(m, k, lane, t, rd);
    tb.tsw[n * 64 + lane] = t;
    if (lane == 0) { tb.rdsw[n] = rd; tb.ksw[n] = k; }
    ++n;
    sync();
  }
  ps.n_sw = n;
#pragma unroll
  for (int r = 0; r < 4; ++r) ps.c[r] = m.c[r];
  ps.diag = m.diag;
  ps.S = S;
  int nu = 0;
  for (unsigned long long mm = S; mm != 0ull;) {
    const int aidx = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)mm));   // descending
    mm &= ~(1ull << aidx);
    const double vaa = -readlane_d(m.diag, aidx);
    const double t = unsweep_q(m, aidx, lane);
    tb.tun[nu * 64 + lane] = t;
    if (lane == 0) { tb.vun[nu] = vaa; tb.kun[nu] = aidx; }
    ++nu;
    sync();
  }
  ps.n_un = nu;
  ps.valid = 1;
}

// spike_slab_draw_regs with the matrix sweeps replayed from the precompute (same results).
template <class PF>
__device__ __forceinline__ double spike_slab_draw_pre(const RegLds& R, int P,
                                                      const DevSeriesParams& sp,
                                                      double prev_obs_scale, double g_obs,
                                                      const Rng& rng, uint32_t iter, int lane,
                                                      PriorCarry& pc, const double* pre,
                                                      const PreTables& tb, const PreState& ps,
                                                      PF& prof) {
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  const int j = lane & 15, q = lane >> 4;
  const bool live = j < P;
  const int col = live ? j : 0;
  QCols m;
#pragma unroll
  for (int r = 0; r < 4; ++r) { m.c[r] = ps.c[r]; m.p[r] = pc.p[r]; }
  m.diag = ps.diag;
  m.pdiag = pc.pdiag;
  m.cb = live ? R.bvec[col] : 0.0;
  m.corner = R.bvec[P];
  unsigned long long S = ps.S;
  // ---- right-hand side: with V the matrix swept on S (precompute) and b = X~'targets,
  //   b~_j = (j in S ? 0 : b_j) - sum_{k in S} V_kj b_k ,   corner = y'y - sum_{k in S} b_k b~_k
  // (what carrying b through the recorded sweeps gives, as one matrix-vector product: lane
  // (j, q) holds V_{4q+r, j}, r < 4; the four quadrants are added across the rows of 16 lanes)
  {
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * q + r;
      const double bk = k < P ? R.bvec[k] : 0.0;
      if ((S >> k) & 1ull) part = fma(m.c[r], bk, part);
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    const bool inj = ((S >> j) & 1ull) != 0ull;
    const double bj = m.cb;
    m.cb = live ? (inj ? 0.0 : bj) - part : 0.0;
    double term = (live && inj) ? bj * m.cb : 0.0;      // (the same in every quadrant)
    term += __shfl_xor(term, 1, 64); term += __shfl_xor(term, 2, 64);
    term += __shfl_xor(term, 4, 64); term += __shfl_xor(term, 8, 64);
    m.corner -= term;
  }
  prof.tick(21);
  bool dirty = false;
  if (!all_in) {
    const int rank = reinterpret_cast<const int*>(pre + 16)[j];
    const double uflip = pre[j];
    const double logit_pi =
        (double)(__logf((float)sp.nonzero_prob) - __logf((float)(1.0 - sp.nonzero_prob)));
    int s_cur = 0;
    const double inv_prev_var = fast_rcp(prev_var);
    for (;;) {
      const bool in = ((S >> j) & 1ull) != 0ull;
      const double sg = in ? -1.0 : 1.0;
      const double rap = fast_rcp(sg * m.diag);
      const double beta_old = sp.obs_scale + 0.5 * m.corner;
      const double x = -0.5 * sg * m.cb * m.cb * rap * fast_rcp(beta_old);
      const double pscale = in ? inv_prev_var : prev_var;
      const double delta = 0.5 * (double)__logf((float)(sg * m.pdiag * pscale * rap)) +
                           sg * logit_pi - (a_post - 1.0) * fast_log1p(x);
      const float prob = 1.0f / (1.0f + __expf(-(float)delta));
      const bool acc = live && q == 0 && rank >= s_cur && uflip < (double)prob;
      unsigned long long cand = __ballot(acc);
      if (cand == 0ull) break;
      int best = -1, best_rank = 1 << 20;
      for (; cand != 0ull; cand &= cand - 1ull) {
        const int jj = __ffsll((long long)cand) - 1;
        const int rj = __builtin_amdgcn_readlane(rank, jj);
        if (rj < best_rank) { best_rank = rj; best = jj; }
      }
      best = __builtin_amdgcn_readfirstlane(best);
      sweep_q<true>(m, best, ((S >> best) & 1ull) != 0ull, lane);
      S ^= 1ull << best;
      s_cur = best_rank + 1;
      dirty = true;
    }
  }
  prof.tick(22);
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = m.p[r];
  pc.pdiag = m.pdiag;
  pc.S = S;
  pc.valid = 1;
  const double beta_post = sp.obs_scale + 0.5 * m.corner;
  double var = beta_post * fast_rcp(g_obs);
  if (var > sp.obs_ub) var = sp.obs_ub;
  const double new_scale = (double)__fsqrt_rn((float)var);
  const float zf = reinterpret_cast<const float*>(pre + 24)[col];
  const double mean = m.cb;
  double mu = 0.0, umine = 0.0;
  if (!dirty) {
    // recorded un-sweeps (descending feature order): feature a ~ N(mu_a, V_aa), the rest
    // conditioned on it.  The increments sqrt(V_aa) z_a do not depend on the running means, so
    // the deviation of feature j is a plain sum over the steps -- its own increment plus
    // t_s[j] times those of the features drawn before it -- with no chain from step to step.
#pragma unroll 4
    for (int s = 0; s < ps.n_un; ++s) {
      const int aidx = __builtin_amdgcn_readfirstlane(tb.kun[s]);
      const double t = tb.tun[s * 64 + lane];
      const double za = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(zf), aidx));
      const double cs = (double)__fsqrt_rn((float)tb.vun[s]) * za;
      umine += aidx > j ? t * cs : (aidx == j ? cs : 0.0);
    }
  } else {
    for (unsigned long long mm = S; mm != 0ull;) {
      const int aidx = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)mm));
      mm &= ~(1ull << aidx);
      const double vaa = -readlane_d(m.diag, aidx);
      const double mua = readlane_d(mu, aidx);
      const double za = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(zf), aidx));
      const double ua = mua + (double)__fsqrt_rn((float)vaa) * za;
      const double t = unsweep_q(m, aidx, lane);
      if (j == aidx) umine = ua; else mu += t * (ua - mua);
    }
  }
  if (q == 0 && live) R.w[j] = ((S >> j) & 1ull) ? (float)(mean + new_scale * umine) : 0.f;
  wave_sync();
  prof.tick(23);
  return new_scale;
}

// Serial section of iteration `it` on the regression wave (serial_section<1> of ci_kernels.h with
// the replayed draw).  It starts right after (B1) -- X~'targets and y'y are complete then, and the
// regression draw needs nothing else -- and passes the workgroup's (B2) itself (`b2`) right after
// gathering them, about when the time waves arrive there.  The level / slope scale draws need
// the increments the time waves sum between (B1) and (B2), and nobody needs those scales before
// (B3): they come last.  Returns through cx / scal / R.w.
template <class PF, class SyncFn>
static __device__ __forceinline__ void serial_section5(SerialCtx* cx, const RegLds& R,
                                                       const float* red, float* scal, int it,
                                                       int lane, PriorCarry& pc, const double* gam,
                                                       const double* pre, const PreTables& tb,
                                                       const PreState& ps, PF& prof, SyncFn b2) {
  const int P = cx->P;
  constexpr int RS = 16 + 4;
  for (int j = lane; j < P + 1; j += 64) {
    const int src = j < P ? j : RS - 4;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += (double)red[w * RS + src];
    R.bvec[j] = s;
  }
  const float wprev = lane < P ? R.w[lane] : 0.f;      // the previous draw's weights (stored below)
  wave_sync();
  b2();                                                 // (B2)
  double obs_scale = cx->obs_scale, level_scale = cx->level_scale, slope_scale = cx->slope_scale;
  const double emit_obs = obs_scale;
  const double g_level = gam[0], g_slope = gam[1], g_obs = gam[2];
  prof.tick(20);
  if (it < cx->n_iter) {
    NoProf np;
    if (ps.valid && ps.S == pc.S)
      obs_scale = spike_slab_draw_pre(R, P, cx->sp, obs_scale, g_obs, cx->rng, (uint32_t)it, lane, pc,
                                      pre, tb, ps, prof);
    else
      obs_scale = spike_slab_draw_regs(R, P, cx->sp, obs_scale, g_obs, cx->rng, (uint32_t)it, lane, np,
                                       pc, pre);
  }
  auto clipped_scale = [](double scale, double ss, double g, double ub) {
    const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
    return s < ub ? s : ub;
  };
  if (it > 0) {
    double ssl = 0.0, sss = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      ssl += (double)red[w * RS + RS - 3];
      sss += (double)red[w * RS + RS - 2];
    }
    level_scale = clipped_scale(cx->sp.level_scale, ssl, g_level, cx->sp.level_ub);
    if (cx->D == 2) slope_scale = clipped_scale(cx->sp.slope_scale, sss, g_slope, cx->sp.slope_ub);
    const int s = it - 1 - cx->W;
    if (s >= 0) {
      const size_t o = cx->chain_lin * cx->S + s;
      if (lane == 0) {
        if (cx->out_obs) cx->out_obs[o] = (float)emit_obs;
        if (cx->out_level_scale) cx->out_level_scale[o] = (float)level_scale;
        if (cx->out_slope_scale) cx->out_slope_scale[o] = (float)(cx->D == 2 ? slope_scale : 0.0);
      }
      if (cx->out_weights && lane < P) cx->out_weights[o * P + lane] = wprev;
    }
  }
  if (lane == 0) {
    cx->obs_scale = obs_scale;
    cx->level_scale = level_scale;
    cx->slope_scale = slope_scale;
    scal[SC_OBS_DK] = (float)obs_scale;
    scal[SC_OBS_EMIT] = (float)emit_obs;
    scal[SC_LEVEL] = (float)level_scale;
    scal[SC_SLOPE] = (float)slope_scale;
  }
  wave_sync();
}

struct LdsLayout5 {
  LdsLayout base;
  size_t off_pre, total;
};
__host__ __device__ inline LdsLayout5 make_layout5(int P, int D, int tpad) {
  LdsLayout5 l;
  l.base = make_layout(P, D, tpad, 1);
  l.off_pre = (l.base.total + 15) & ~(size_t)15;
  l.total = l.off_pre + pre_tables_bytes();
  return l;
}

// 0 < P <= 16, X resident in LDS (the dispatch conditions of gibbs_kernel<D, L, 1>).
template <int D, int L, bool PROF = false>
__global__ __launch_bounds__(NT5) void gibbs_kernel5(KArgs a) {
  using PF = typename std::conditional<PROF, Prof, NoProf>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool reg_wave = wave == NW;
  const int series = blockIdx.x / a.C, chain = blockIdx.x % a.C;
  const int T = a.T, P = a.P;
  constexpr int TPAD = NT * L;
  const LdsLayout5 lay5 = make_layout5(P, D, TPAD);
  const LdsLayout& lay = lay5.base;
  SerialCtx* cx = (SerialCtx*)(smem + lay.off_ctx);
  float* scal = (float*)(smem + lay.off_scal);
  float* red = (float*)(smem + lay.off_red);
  float* slots = (float*)(smem + lay.off_slots);
  float* xlast = (float*)(smem + lay.off_xlast);
  float* wls = (float*)(smem + lay.off_w);
  const float* Xs = (const float*)(smem + lay.off_x);
  const size_t chain_lin = (size_t)series * a.C + chain;
  constexpr int RS = 16 + 4;

  Rng rng;
  rng.k0 = a.seed0;
  rng.k1 = a.seed1;
  rng.chain = stream_id(a.chain_offset + chain, a.series_stream_base, series);

  const float* yg = a.y + (size_t)series * T;
  const uint8_t* mg = a.mask + (size_t)series * T;
  const float* Xg = a.Xt + (size_t)series * P * T;
  const int t0 = tid * L;                    // (time threads only)
  RegLds R;
  R.xtx = (double*)(smem + lay.off_xtx);
  R.omega = (double*)(smem + lay.off_omega);
  R.aug[0] = (double*)(smem + lay.off_aug0);
  R.aug[1] = (double*)(smem + lay.off_aug1);
  R.pri[0] = (double*)(smem + lay.off_pri0);
  R.pri[1] = (double*)(smem + lay.off_pri1);
  R.chol = (double*)(smem + lay.off_chol);
  R.bvec = (double*)(smem + lay.off_bvec);
  R.zv = (double*)(smem + lay.off_zv);
  R.uperm = (double*)(smem + lay.off_uperm);
  R.nz = (int*)(smem + lay.off_nz);
  R.perm = (int*)(smem + lay.off_perm);
  R.idx = (int*)(smem + lay.off_idx);
  R.w = wls;
  PreTables tb;
  {
    double* d = (double*)(smem + lay5.off_pre);
    tb.tsw = d; d += PRE_MAXS * 64;
    tb.tun = d; d += PRE_MAXS * 64;
    tb.rdsw = d; d += PRE_MAXS;
    tb.vun = d; d += PRE_MAXS;
    tb.ksw = (int*)d;
    tb.kun = tb.ksw + PRE_MAXS;
  }
  if (tid == 0) {
    cx->sp = a.sp[series];
    cx->obs_scale = cx->sp.obs_scale0;           // causalimpact_lib.py:566-572
    cx->level_scale = cx->sp.level_scale0;
    cx->slope_scale = cx->sp.slope_scale0;
    cx->R = R;
    cx->scal = scal;
    cx->red = red;
    cx->out_obs = a.out_obs;
    cx->out_level_scale = a.out_level_scale;
    cx->out_slope_scale = a.out_slope_scale;
    cx->out_weights = a.out_weights;
    cx->chain_lin = chain_lin;
    cx->rng = rng;
    cx->P = P; cx->T = T; cx->D = D; cx->W = a.W; cx->S = a.S; cx->n_iter = a.W + a.S;
    cx->prof = nullptr;
    scal[8] = (float)cx->sp.init_level_loc;
    scal[9] = (float)(cx->sp.init_level_scale * cx->sp.init_level_scale);
    scal[10] = (float)(cx->sp.init_slope_scale * cx->sp.init_slope_scale);
  }
  float yv[L];
  uint32_t maskbits = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    const bool in = !reg_wave && t < T;
    yv[l] = in ? yg[t] : 0.f;
    const bool m = in ? (mg[t] != 0) : true;
    if (m) { maskbits |= (1u << l); yv[l] = 0.f; }
  }
  {
    double* lx = (double*)(smem + lay.off_xtx);
    double* lo = (double*)(smem + lay.off_omega);
    for (int e = tid; e < P * P; e += NT5) {
      lx[e] = a.xtx[(size_t)series * P * P + e];
      lo[e] = a.omega[(size_t)series * P * P + e];
    }
    float* xw_ = (float*)(smem + lay.off_x);
    for (int j = 0; j < P; ++j)
      for (int t = tid; t < TPAD; t += NT5) xw_[j * TPAD + t] = (t < T) ? Xg[(size_t)j * T + t] : 0.f;
    if (tid < 16) wls[tid] = 0.f;                   // weights = 0            :575-578
  }
  __syncthreads();
  double* gam = (double*)(smem + lay.off_gam);
  if (wave == 1) serial_gammas<1>(cx, 0, lane, gam);
  if (wave == 2) spike_slab_randoms(rng, 0u, P, lane, gam + 8);
  const float init_loc = scal[8], init_var = scal[9], init_svar = scal[10];
  const int n_iter = a.W + a.S;

  if (reg_wave) {
    // ================================ regression wave ==========================================
    PriorCarry pc;
    pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;
    PreState ps;
    ps.valid = 0; ps.S = 0ull; ps.n_sw = 0; ps.n_un = 0; ps.diag = 1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) ps.c[r] = 0.0;
    PF rprof;     // slots 16.. : the regression wave's own budget (lane 0 of block 0)
    rprof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && lane == 0);
    for (int it = 0; it <= n_iter; ++it) {
      __syncthreads();      // (B1) X~'targets, y'y partials complete (and the boundary exchange)
      rprof.tick(16);
      // the serial section is the critical path of the iteration and shares its SIMD with one of
      // the time waves, which has slack until (B3): win the issue arbitration while it lasts
      __builtin_amdgcn_s_setprio(3);
      serial_section5(cx, R, red, scal, it, lane, pc, gam + 4 * (it & 1), gam + 8 + 32 * (it & 1), tb, ps,
                      rprof, []() { __syncthreads(); });      // (B2) inside, after the gather
      rprof.tick(17);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();      // (B3) scalars and weights of iteration `it` published
      rprof.tick(18);
      if (it == n_iter) break;
      // the time waves now run dk_draw (two barriers: the scan of its prior simulation was split
      // around (B3)); meanwhile: next iteration's matrix work
      int done = 0;
      const int n_steps = 2 * __popcll(pc.S);
      int step = 0;
      // barriers after half of the steps and after the last but one
      const int mark1 = n_steps / 2 > 1 ? n_steps / 2 : 2, mark2 = n_steps - 1 > 2 ? n_steps - 1 : 3;
      auto sync = [&]() {
        ++step;
        for (;;) {
          const int mark = done == 0 ? mark1 : mark2;
          if (done >= 2 || step < mark) break;
          __syncthreads();
          ++done;
        }
      };
      const double var_next = cx->obs_scale * cx->obs_scale;
      regression_precompute(R, P, var_next, pc.S, lane, tb, ps, sync);
      while (done < 2) { __syncthreads(); ++done; }
      rprof.tick(19);
    }
    return;
  }

  // ==================================== time waves =============================================
  // above the regression wave's precompute (priority 0), below its serial section (3): the wave
  // that shares a SIMD with it must not be held up during the Durbin-Koopman draw
  __builtin_amdgcn_s_setprio(1);
  float lev[L], slp[L], xw[L], pm_acc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { lev[l] = 0.f; slp[l] = 0.f; xw[l] = 0.f; pm_acc[l] = 0.f; }  // :580-581
  float* o_level = a.out_level ? a.out_level + chain_lin * a.S * T : nullptr;
  float* o_slope = a.out_slope ? a.out_slope + chain_lin * a.S * T : nullptr;
  float* o_traj = a.out_traj ? a.out_traj + chain_lin * a.S * T : nullptr;
  PF prof;
  prof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && tid == 0);
  for (int it = 0; it <= n_iter; ++it) {
    // ---- partial sums over the owned steps (targets use the CURRENT level)
    {
      float tg[L];
      float yty = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const bool obs = ((maskbits >> l) & 1u) == 0u;
        tg[l] = obs ? (yv[l] - lev[l]) : 0.f;
        yty = fmaf(tg[l], tg[l], yty);
      }
      float pj[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jj = j < P ? j : P - 1;
        float xr[L];
        lds_row_load<L>(Xs + jj * TPAD + t0, xr);
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) s = fmaf(xr[l], tg[l], s);
        pj[j] = s;
      }
      const float tot = wave_reduce_scatter16(pj, lane);
      if (lane < 16) red[wave * RS + lane] = tot;
      {
        const float s0 = wave_prefix_dpp(yty);           // y'y goes with X~'targets: before (B1)
        if (lane == 63) red[wave * RS + RS - 4] = s0;
      }
      xlast[tid * D] = lev[L - 1];
      if constexpr (D == 2) xlast[tid * D + 1] = slp[L - 1];
      __syncthreads();                                     // (B1)
      float ssl = 0.f, sss = 0.f;
      float pl = (tid > 0) ? xlast[(tid - 1) * D] : 0.f;
      float ps_ = 0.f;
      if constexpr (D == 2) ps_ = (tid > 0) ? xlast[(tid - 1) * D + 1] : 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int t = t0 + l;
        if (t >= 1 && t < T) {
          float dl = lev[l] - pl;
          if constexpr (D == 2) {
            dl -= ps_;
            const float ds = slp[l] - ps_;
            sss = fmaf(ds, ds, sss);
          }
          ssl = fmaf(dl, dl, ssl);
        }
        pl = lev[l];
        if constexpr (D == 2) ps_ = slp[l];
      }
      const float s1 = wave_prefix_dpp(ssl), s2 = wave_prefix_dpp(sss);
      if (lane == 63) {
        red[wave * RS + RS - 3] = s1;
        red[wave * RS + RS - 2] = s2;
      }
    }
    const float so_prev = scal[SC_OBS_DK];
    __syncthreads();                                       // (B2)
    prof.tick(0);

    // ---- while the regression wave is in its serial section: next iteration's gamma variates and
    // regression randomness, emission of iteration it-1, this iteration's Durbin-Koopman normals
    float zl[L], zs[L], zo[L];
#pragma unroll
    for (int l = 0; l < L; ++l) zs[l] = 0.f;
    if (wave == 1 && it < n_iter) serial_gammas<1>(cx, it + 1, lane, gam + 4 * ((it + 1) & 1));
    if (wave == 2 && it + 1 < n_iter)
      spike_slab_randoms(rng, (uint32_t)(it + 1), P, lane, gam + 8 + 32 * ((it + 1) & 1));
    if (it > a.W) {
      const int s = it - 1 - a.W;
      float zp[L];
      fill_normals<L>(rng, (uint32_t)(it - 1), SITE_PRED, 0, (uint32_t)t0, zp);
      float tr[L];
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float loc = lev[l] + xw[l];
        pm_acc[l] += loc;
        tr[l] = fmaf(so_prev, zp[l], loc);
      }
      const size_t row = (size_t)s * T;
      bool vec_done = false;
      if constexpr (L % 4 == 0) {
        if ((T & 3) == 0) {
          vec_done = true;
#pragma unroll
          for (int q = 0; q < L / 4; ++q) {
            const int t = t0 + 4 * q;
            if (t < T) {
              if (o_level) *(float4*)(o_level + row + t) = make_float4(lev[4 * q], lev[4 * q + 1], lev[4 * q + 2], lev[4 * q + 3]);
              if (o_slope) *(float4*)(o_slope + row + t) = make_float4(slp[4 * q], slp[4 * q + 1], slp[4 * q + 2], slp[4 * q + 3]);
              if (o_traj) *(float4*)(o_traj + row + t) = make_float4(tr[4 * q], tr[4 * q + 1], tr[4 * q + 2], tr[4 * q + 3]);
            }
          }
        }
      }
      if (!vec_done) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const int t = t0 + l;
          if (t < T) {
            if (o_level) o_level[row + t] = lev[l];
            if (o_slope) o_slope[row + t] = slp[l];
            if (o_traj) o_traj[row + t] = tr[l];
          }
        }
      }
    }
    if (it < n_iter) dk_normals<D, L>(rng, (uint32_t)it, tid, zl, zs, zo);
    if (wave == 3 && lane < D && it < n_iter) {
      // the normals of the simulated initial state (thread 0 needs them first thing in the draw:
      // two Philox calls on its critical path otherwise), drawn here by a wave with slack
      float zi[1];
      fill_normals<1>(rng, (uint32_t)it, SITE_PRIOR_INIT, 0, (uint32_t)lane, zi);
      scal[12 + lane] = zi[0];
    }
    // The level / slope disturbance scales of this iteration -- the draw the regression wave makes
    // in its serial section too (same expression, same inputs: the increments' sums in `red`,
    // complete since (B2), and wave 1's gamma variates) -- computed by every time wave here so
    // that the scan of the prior simulation (step (1) of dk_draw) runs before (B3) instead of
    // after it: its first half now, the second half behind (B3), which doubles as its barrier.
    Vec<D> sig;
    PElem<D> pincl = pelem_identity<D>();
    {
      auto clipped_scale = [](double scale, double ss, double g, double ub) {
        const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
        return s < ub ? s : ub;
      };
      double level_scale = cx->sp.level_scale0, slope_scale = cx->sp.slope_scale0;
      if (it > 0) {
        const double* gm = gam + 4 * (it & 1);
        double ssl = 0.0, sss = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          ssl += (double)red[w * RS + RS - 3];
          sss += (double)red[w * RS + RS - 2];
        }
        level_scale = clipped_scale(cx->sp.level_scale, ssl, gm[0], cx->sp.level_ub);
        if constexpr (D == 2) slope_scale = clipped_scale(cx->sp.slope_scale, sss, gm[1], cx->sp.slope_ub);
      }
      sig.v[0] = (float)level_scale;
      if constexpr (D == 2) sig.v[1] = (float)slope_scale;
      if (it < n_iter) pincl = dk_prior_begin<D, L>(sig, zl, zs, slots, lane, wave);
    }
    const bool publish = a.progress != nullptr && it > a.W &&
                         ((it - a.W) % a.progress_every == 0 || it - a.W == a.S);
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows are in L2
    prof.tick(2);
    __syncthreads();                                       // (B3)
    prof.tick(1);
    if (publish) {
      const int done = it - a.W;
      {
        // the time waves' stores of the first `done` rows were issued before (B3); make them
        // visible to the copy engines and publish (the scalars / weights of these draws are
        // copied after the kernel has ended)
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(a.progress + chain_lin, (unsigned int)done, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (it == n_iter) break;

    // ---- residual and the latent-path draw of iteration it
    float resid[L];
#pragma unroll
    for (int l = 0; l < L; ++l) xw[l] = 0.f;
    {
      float wv[16];
      lds_row_load<16>(wls, wv);           // the weights vector is padded to 16 floats
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jj = j < P ? j : P - 1;
        const float wj = j < P ? wv[j] : 0.f;
        float xr[L];
        lds_row_load<L>(Xs + jj * TPAD + t0, xr);
#pragma unroll
        for (int l = 0; l < L; ++l) xw[l] = fmaf(xr[l], wj, xw[l]);
      }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) resid[l] = yv[l] - xw[l];
    DkModel<D> md;
    {
      const float so = scal[SC_OBS_DK];
      md.H = so * so;
      md.sig = sig;                        // == scal[SC_LEVEL], scal[SC_SLOPE]
      md.a1 = vzero<D>();
      md.a1.v[0] = init_loc;
      md.p1.v[0] = init_var;
      if constexpr (D == 2) md.p1.v[1] = init_svar;
    }
    const PElem<D> ppre = dk_prior_finish<D>(pincl, slots, lane, wave);
    Vec<D> x[L];
    prof.tick(3);
    dk_draw<D, L>(md, resid, maskbits, rng, (uint32_t)it, tid, lane, wave, slots, x, prof, zl, zs, zo,
                  scal + 12, &ppre);                       // (B4) (B5)
#pragma unroll
    for (int l = 0; l < L; ++l) {
      lev[l] = x[l].v[0];
      if constexpr (D == 2) slp[l] = x[l].v[1];
    }
  }

  if (a.out_pred_mean) {
    const float inv = 1.0f / (float)(a.S > 0 ? a.S : 1);
    float* pm = a.out_pred_mean + chain_lin * T;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int t = t0 + l;
      if (t < T) pm[t] = pm_acc[l] * inv;
    }
  }
}

}  // namespace ci
