// ci_kernels8.h -- the latency build of the register-resident Gibbs kernel: EIGHT wavefronts per
// chain, one chain per compute unit (dispatched when every chain has a CU to itself, 0 < P <= 16,
// X resident in LDS).  Same sampler, same random stream, the same arithmetic in the same order as
// gibbs_kernel<D, L, 1> (ci_kernels.h) -- both call the same functions for everything that
// rounds, and the library is built with -ffp-contract=on -- so every draw of the two kernels is
// bit-identical (tests/test_gpu_gibbs.py) and the dispatch by launch size in ci_api.hip never
// changes a result.  What changes is WHO computes WHAT WHEN.
//
//   waves 0-3  TIME waves: thread i owns the L consecutive steps [iL, (i+1)L) -- targets, the
//              Durbin-Koopman draw, emission of the finished draw;
//   wave 4     REGRESSION wave: owns the serial section (sigma^2_obs, weights) and, during the
//              draw, already sweeps the NEXT iteration's posterior block (record / replay, as in
//              rounds 2-3);
//   waves 5-7  RANDOMNESS waves: the 3 L normals per time thread of the Durbin-Koopman draw, the
//              gamma variates and the regression block's permutation / uniforms, one iteration
//              ahead, through LDS (the L predictive normals stay with the time waves: they have
//              idle time while sigma^2_obs is drawn).  Philox + Box-Muller were 40 % of what the
//              time waves executed between two barriers.
//
// One Gibbs iteration, barriers (B1) ... (B5) + (Bs) shared by all eight waves:
//
//   time waves                         regression wave                 randomness waves
//   targets -> LDS                     .                               .
//   (B1) ------------------------------------------------------------------------------------
//   X~'targets: 2 features per wave over the WHOLE series (xt_sums_wave; all 8 waves), no
//   cross-wave reduction; time waves also sum the level / slope increments
//   (B2) ------------------------------------------------------------------------------------
//   normals from LDS, emission of    right-hand side replay, flips,   gammas / permutation /
//   draw it-1, disturbance scales,   sigma^2_obs(it) -> LDS           x_0 normals of it+1
//   prior-simulation scan (1st half)
//   (Bs) sigma^2_obs published --------------------------------------------------------------
//   MATRIX phase of the draw         weights(it) replay, scale        normals of it+1 ...
//   ((A, C, J) chunk + in-wave scan) draws of it-1, scalar outputs
//   (B3) weights published = barrier of the matrix scan -------------------------------------
//   covariances / gains, X w,        precompute for it+1 ...          ...
//   forward scan of the means
//   (B4) ------------------------------------------------------------------------------------
//   local means, adjoint scan        ...                              ...
//   (B5) ------------------------------------------------------------------------------------
//   fix-up -> level, slope           ...                              ...
//
// The point of the split: the covariance side of the Kalman filter needs
// sigma^2_obs but not the weights, so it runs WHILE the regression wave draws the weights, and
// the time waves no longer generate random numbers in the window before it.
#pragma once
#include "ci_kernels.h"

namespace ci {

// Schedule of the helper waves (see "scheduling of the helper waves" below), one word so that
// $CI_SCHED_WORD can replace it for experiments without a rebuild:
//   bits 0-1   test mode: 1 = all background work as early as possible, 2 = as late as possible
//   bits 4-5   regression wave: sweeps before (B3)        bits 6-7   ... between (B3) and (B4)
//   bits 8-9   ... left for after (B5)
//   bits 10-11 randomness waves 5, 6: rounds before (B3)   bits 12-13 all: rounds between (B3), (B4)
//   bits 14-15 rounds left for after (B5)
constexpr int SCHED_DEFAULT = (2 << 4) | (1 << 6) | (0 << 8) | (2 << 10) | (2 << 12) | (0 << 14);   // tools/exp_sched.py, cfg2
constexpr int NT8 = 512;              // 4 time waves + regression wave + 3 randomness waves
constexpr int NW8 = 8;
constexpr int PRE_MAXS = 16;          // recorded sweeps / un-sweeps (P <= 16)

// LDS tables written by the regression wave's precompute, read by its next serial section.
struct PreTables {
  double* tsw;      // [PRE_MAXS][16]  sweep s: t_j = A[k_s][j] / pivot of feature j (1 at the pivot)
  double* V;        // [16][16]        the posterior block swept on S, row-major (V[k][j])
  double* rdk;      // [16]            1 / pivot of the sweep of feature j (j in S)
  double* hand;     // [20]            serial section -> weights wave: posterior mean of the weights
                    //                 [16], sigma_obs, then (as ints) S lo, S hi, flag "draw them"
};
// tsw / V / rdk exist twice (parity of the iteration that CONSUMES them): the precompute for
// iteration it + 1 starts as soon as sigma^2_obs(it) is out, while the weights of iteration `it`
// are still being drawn from the tables of `it`.
constexpr size_t PRE_HALF_DOUBLES = PRE_MAXS * 16 + 16 * 16 + 16;
__host__ __device__ constexpr size_t pre_tables_bytes() {
  return sizeof(double) * (2 * PRE_HALF_DOUBLES + 20) + 16;
}
__device__ __forceinline__ PreTables pre_tables_at(unsigned char* base, int parity) {
  PreTables tb;
  double* d = (double*)base + (size_t)(parity & 1) * PRE_HALF_DOUBLES;
  tb.tsw = d; d += PRE_MAXS * 16;
  tb.V = d; d += 16 * 16;
  tb.rdk = d;
  tb.hand = (double*)base + 2 * PRE_HALF_DOUBLES;
  return tb;
}
constexpr int HAND_SCALE = 16, HAND_INTS = 17;   // doubles; ints at 2 * HAND_INTS + {0, 1, 2}

// What the precompute leaves in the regression wave's registers for the next serial section.
struct PreState {
  double c[4];      // swept posterior block (quadrant layout of QCols)
  double diag;
  double rdk;       // lane j: 1 / pivot of the sweep of feature j (j in S)
  unsigned long long S;   // the set it is swept on
  int n_sw;
  int valid;
};

// Sum over the 16 lanes of each row, through the DPP crossbar (row_ror 8, 4, 2, 1), then made
// wave-uniform per row by taking the row's first lane (the four rows hold replicas: same inputs,
// same order, same bits).
template <int CTRL> __device__ __forceinline__ double dpp_ror_add_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16_d(double v) {
  v = dpp_ror_add_d<0x128>(v);   // row_ror:8
  v = dpp_ror_add_d<0x124>(v);   // row_ror:4
  v = dpp_ror_add_d<0x122>(v);   // row_ror:2
  v = dpp_ror_add_d<0x121>(v);   // row_ror:1
  return readlane_d(v, 0);
}

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

// The matrix work of the NEXT serial section: the posterior block for sigma^2 = var_next swept on
// S (ascending), recording per sweep the multipliers t_j = A[k][j] / pivot.  Those ARE the Cholesky
// factor of the included block M_S = L L' (L_jk / L_kk = t_j of sweep k for j > k in S,
// L_kk^2 = pivot_k), which is all the weights draw u = L^-T z needs (back substitution, see
// weights_backsub) -- no second pass of un-sweeps.  The swept block itself goes to LDS row-major
// for the right-hand-side product.  Written as a RESUMABLE sequence of steps (build, one sweep per
// member of S, store): the caller runs each step in whatever interval between two workgroup
// barriers its schedule gives it.
struct PreRun {
  QCols m;
  double rdk;
  unsigned long long pending;
  int n;
  int phase;          // 0 build, 1 sweeps, 2 store, 3 done
};
__device__ __forceinline__ void pre_begin(PreRun& st, unsigned long long S) {
  st.pending = S;
  st.n = 0;
  st.rdk = 0.0;
  st.phase = 0;
}
__device__ __forceinline__ void pre_step(PreRun& st, const RegLds& R, int P, double var_next,
                                         unsigned long long S, int lane, const PreTables& tb,
                                         PreState& ps) {
  const int j = lane & 15, q = lane >> 4;
  const bool live = j < P;
  const int col = live ? j : 0;
  if (st.phase == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * q + r;
      double om = 0.0, xx = 0.0;
      if (live && i < P) {
        om = R.omega[i * P + col];
        xx = R.xtx[i * P + col];
      }
      st.m.c[r] = om * var_next + xx;
      st.m.p[r] = 0.0;
    }
    st.m.diag = live ? R.omega[col * P + col] * var_next + R.xtx[col * P + col] : 1.0;
    st.m.cb = 0.0; st.m.corner = 0.0; st.m.pdiag = 0.0;
    st.phase = st.pending != 0ull ? 1 : 2;
  } else if (st.phase == 1) {
    const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)st.pending) - 1);
    double t, rd;
    sweep_rec(st.m, k, lane, t, rd);
    if (q == 0) tb.tsw[st.n * 16 + j] = t;
    st.rdk = (j == k) ? rd : st.rdk;
    ++st.n;
    st.pending &= st.pending - 1ull;
    if (st.pending == 0ull) st.phase = 2;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) tb.V[(4 * q + r) * 16 + j] = st.m.c[r];
    if (q == 0) tb.rdk[j] = st.rdk;
    ps.n_sw = st.n;
#pragma unroll
    for (int r = 0; r < 4; ++r) ps.c[r] = st.m.c[r];
    ps.diag = st.m.diag;
    ps.rdk = st.rdk;
    ps.S = S;
    ps.valid = 1;
    st.phase = 3;
  }
}

// The weights draw from the recorded sweeps: u = L^-T z by back substitution over the members of S
// in descending order,
//   u_a = z_a / L_aa - sum_{i in S, i > a} (L_ia / L_aa) u_i ,   1 / L_aa = sqrt(1 / pivot_a),
//   L_ia / L_aa = t_i of the sweep of a,
// then w_j = mean_j + sigma_obs u_j.  Lane j carries u_j (each row of 16 lanes a replica) and
// knows its rank within S; step s handles the member of rank s: every lane forms "own increment
// minus the dot product", only the lane of that member keeps it -- neither the pivot's index nor
// a wave-uniform dot product is needed (no v_readlane in the chain).  Same u as the un-sweep
// route of spike_slab_draw_regs up to float64 rounding.  Any wavefront can run it: everything it
// needs is in LDS (the eight-wave kernel gives it to a randomness wave on another SIMD than the
// one the regression wave shares with time wave 0).
__device__ __forceinline__ void weights_backsub(const PreTables& tb, const double* pre, int P,
                                                unsigned long long S, double mean, double new_scale,
                                                int lane, float* w) {
  const int j = lane & 15, q = lane >> 4;
  const bool live = j < P;
  const float zf = reinterpret_cast<const float*>(pre + 24)[live ? j : 0];
  const bool inS = ((S >> j) & 1ull) != 0ull;
  const int pos = __popcll(S & ((1ull << j) - 1ull));
  const int n_sw = __popcll(S);
  const double rsz = (double)__fsqrt_rn((float)tb.rdk[j]) * (double)zf;      // z_j / L_jj
  double tt[PRE_MAXS];
#pragma unroll
  for (int s = 0; s < PRE_MAXS; ++s) tt[s] = tb.tsw[s * 16 + j];
  double u = 0.0;
#pragma unroll
  for (int s = PRE_MAXS - 1; s >= 0; --s) {
    if (s < n_sw) {
      double dot = (inS && pos > s) ? tt[s] * u : 0.0;
      dot = dpp_ror_add_d<0x128>(dot);
      dot = dpp_ror_add_d<0x124>(dot);
      dot = dpp_ror_add_d<0x122>(dot);
      dot = dpp_ror_add_d<0x121>(dot);
      u = (inS && pos == s) ? rsz - dot : u;
    }
  }
  if (q == 0 && live) w[j] = inS ? (float)(mean + new_scale * u) : 0.f;
}

// spike_slab_draw_regs with the matrix sweeps replayed from the precompute (same results).
// `publish(new_scale, mean, S, clean)` is called as soon as sigma_obs is drawn; when it returns true
// the weights are drawn by somebody else (weights_backsub on another wave) and this function ends.
// `publish(new_scale)` is called as soon as sigma_obs is drawn -- before the weights.
template <class PF, class Pub>
__device__ __forceinline__ double spike_slab_draw_pre(const RegLds& R, int P,
                                                      const DevSeriesParams& sp,
                                                      double prev_obs_scale, double g_obs,
                                                      int lane, PriorCarry& pc, const double* pre,
                                                      const PreTables& tb, const PreState& ps,
                                                      const float* red, PF& prof, Pub publish) {
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
  // X~'targets and y'y straight from the sums of xt_sums_wave (R.bvec holds the same values for
  // the fall-back route; reading `red` saves the round trip through it)
  m.cb = live ? (double)red[col] : 0.0;
  m.corner = (double)red[RED_YTY];
  unsigned long long S = ps.S;
  // ---- right-hand side: with V the matrix swept on S (precompute) and b = X~'targets,
  //   b~_j = (j in S ? 0 : b_j) - sum_{k in S} V_kj b_k ,   corner = y'y - sum_{k in S} b_k b~_k
  // (what carrying b through the recorded sweeps gives): every lane forms the product for its
  // own column j from the row-major copy in LDS -- 16 independent loads, no cross-lane traffic;
  // the four rows of 16 lanes compute replicas
  {
    // (rows in groups of four, groups beyond P skipped: the branch is uniform and sits around a
    //  whole group of independent loads)
    double v[16], bk[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (4 * g < P) {
#pragma unroll
        for (int k = 4 * g; k < 4 * g + 4; ++k) {
          v[k] = tb.V[k * 16 + j];
          bk[k] = (double)red[k];        // (red[P .. 15] hold nothing: masked by S below)
        }
      } else {
#pragma unroll
        for (int k = 4 * g; k < 4 * g + 4; ++k) { v[k] = 0.0; bk[k] = 0.0; }
      }
    }
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      acc0 = ((S >> k) & 1ull) ? fma(v[k], bk[k], acc0) : acc0;
      acc1 = ((S >> (k + 1)) & 1ull) ? fma(v[k + 1], bk[k + 1], acc1) : acc1;
    }
    const bool inj = ((S >> j) & 1ull) != 0ull;
    const double bj = m.cb;
    m.cb = live ? (inj ? 0.0 : bj) - (acc0 + acc1) : 0.0;
    m.corner -= row_sum16_d((live && inj) ? bj * m.cb : 0.0);
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
  const double mean = m.cb;
  if (publish(new_scale, mean, S, !dirty)) {
    prof.tick(23);
    return new_scale;
  }
  if (!dirty) {
    weights_backsub(tb, pre, P, S, mean, new_scale, lane, R.w);
  } else {
    // an inclusion flip was accepted in this very iteration: the recorded sweeps are those of the
    // old set -- un-sweep on the fly (the route of spike_slab_draw_regs)
    const float zf = reinterpret_cast<const float*>(pre + 24)[col];
    double mu = 0.0, umine = 0.0;
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
    if (q == 0 && live) R.w[j] = ((S >> j) & 1ull) ? (float)(mean + new_scale * umine) : 0.f;
  }
  wave_sync();
  prof.tick(23);
  return new_scale;
}

// ---- scheduling of the helper waves.  Workgroup barriers are all-or-nothing: a helper (the
// regression wave outside its serial section, a randomness wave) that is still inside a chunk of
// background work -- a Philox round (~2k cycles next to a busy time wave), a sweep (~1k) -- when the
// time waves reach a barrier makes the critical path wait.  Each helper therefore passes barrier k
// once it has finished a fixed share of its units of work (quota_k), chosen from the measured
// lengths of the intervals (profiles/r04_phase_cycles.txt): (B2)-(Bs) is the regression wave's
// critical stretch and gets no rounds; (Bs)-(B3) one in four (the weights instead on wave 7);
// (B3)-(B4), the longest, half of what is left; (B4)-(B5) the rest; nothing after (B5), where the
// iteration turns round quickly.  A self-timing scheduler (each helper measuring the intervals and
// its chunks with s_memtime and asking "does another chunk fit?") was built and measured in round
// 4: the clock reads and the bookkeeping cost more than the waits they removed (7.5 -> 10+ ms).

// ---- LDS layout: every offset but the design's size is a compile-time constant (P <= 16 is
// provisioned as 16), so LDS addresses are immediates instead of live scalar registers.
template <int D, int L> struct Lay8 {
  static constexpr size_t TP = (size_t)NT * L;
  static constexpr size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }
  static constexpr size_t off_ctx = 0;
  static constexpr size_t off_xtx = a16(sizeof(SerialCtx));
  static constexpr size_t off_omega = off_xtx + 16 * 16 * sizeof(double);
  static constexpr size_t off_bvec = off_omega + 16 * 16 * sizeof(double);
  static constexpr size_t off_w = off_bvec + a16(20 * sizeof(double));
  static constexpr size_t off_scal = off_w + 16 * sizeof(float);
  static constexpr size_t off_red = off_scal + 32 * sizeof(float);
  static constexpr size_t off_slots = off_red + 32 * sizeof(float);
  static constexpr size_t off_xlast = off_slots + 3 * NW * 16 * sizeof(float);
  static constexpr size_t off_gam = off_xlast + a16((size_t)NT * D * sizeof(float));
  static constexpr size_t off_pre = off_gam + a16((8 + 64 + 4) * sizeof(double));
  static constexpr size_t off_tgv = off_pre + a16(pre_tables_bytes());
  static constexpr size_t off_z = off_tgv + TP * sizeof(float);          // zl, zs, zo
  // L >= 8 (T > 1024): the builds that spill.  Two per-thread arrays that live from one iteration
  // into the next leave the registers: X w waits in LDS for the emission of its draw (PARK), the
  // running sum of the predictor in its output array in HBM -- L registers each during the draw.
  static constexpr bool PARK = L >= 8;
  static constexpr size_t off_xw = off_z + 3 * TP * sizeof(float);
  static constexpr size_t off_y = off_xw + (PARK ? TP * sizeof(float) : 0);     // the observations (PARK)
  static constexpr size_t off_x = off_y + (PARK ? TP * sizeof(float) : 0);
  static __host__ __device__ constexpr size_t total(int P) { return off_x + (size_t)P * TP * sizeof(float); }
};
// indices into `scal` beyond enum Scal: prior moments, then the x_0 normals of even / odd iterations
enum Scal8 { SC8_INIT_LOC = 8, SC8_INIT_VAR = 9, SC8_INIT_SVAR = 10, SC8_ZINIT = 12 };

// Profile slots of the instrumented build (ci_session_profile; 32 int64):
//   time thread 0:      0 = targets .. (B2)   2 = window work   1 = wait at (Bs)   24 = matrix chunk +
//                       in-wave scan   25 = wait (B3) + cross-wave + local   3 = X w + prior path
//                       27 = forward chunk + in-wave scan   5 = (B4) + finish   6 = local means +
//                       adjoint in-wave scan   28 = (B5) + finish   7 = fix-up
//   regression lane 0:  16 = until (B2) done   20 = gather   21 = rhs   22 = flips + sigma^2
//                       18 = wait at (Bs)   23 = weights + scale draws   17 = wait at (B3)
//                       19 = precompute steps (pure)   29 = its waits at (B4) (B5)
//   randomness wave 5:  30 = work   31 = waits at barriers
// XG: the design stays in global memory (L2) -- series too long for a copy in LDS beside the
// randomness buffers (T > ~1500 with a dozen columns, every T > 2048).
template <int D, int L, bool PROF = false, bool XG = false>
__global__ __launch_bounds__(NT8) void gibbs_kernel8(KArgs a) {
  using PF = typename std::conditional<PROF, Prof, NoProf>::type;
  using LY = Lay8<D, L>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int series = blockIdx.x / a.C, chain = blockIdx.x % a.C;
  const int T = a.T, P = a.P;
  constexpr int TPAD = NT * L;
  SerialCtx* cx = (SerialCtx*)(smem + LY::off_ctx);
  float* scal = (float*)(smem + LY::off_scal);
  float* red = (float*)(smem + LY::off_red);
  float* slots = (float*)(smem + LY::off_slots);
  float* xlast = (float*)(smem + LY::off_xlast);
  float* wls = (float*)(smem + LY::off_w);
  float* tgv = (float*)(smem + LY::off_tgv);
  float* zb = (float*)(smem + LY::off_z);
  float* xwb = (float*)(smem + LY::off_xw);       // (PARK builds only)
  float* ybuf = (float*)(smem + LY::off_y);       // (PARK builds only)
  double* gam = (double*)(smem + LY::off_gam);
  const size_t chain_lin = (size_t)series * a.C + chain;
  const int n_iter = a.W + a.S;

  Rng rng;
  rng.k0 = stream_key0(a.seed0, a.series_stream_base, series);
  rng.k1 = stream_key1(a.seed1, a.series_stream_base, series);
  rng.chain = (uint32_t)(a.chain_offset + chain);

  const float* yg = a.y + (size_t)series * T;
  const uint8_t* mg = a.mask + (size_t)series * T;
  const float* Xg = a.Xt + (size_t)series * P * T;
  // the design as the sums / products read it: the LDS copy (rows of TPAD floats), or the
  // global rows of T floats (float4 reads need T % 4 == 0 and 16-byte aligned rows)
  const float* Xs = XG ? Xg : (const float*)(smem + LY::off_x);
  const bool xwide = L % 4 == 0 && (T & 3) == 0 && (reinterpret_cast<uintptr_t>(Xg) & 15) == 0;
  RegLds R;
  R.xtx = (double*)(smem + LY::off_xtx);
  R.omega = (double*)(smem + LY::off_omega);
  R.aug[0] = R.aug[1] = R.pri[0] = R.pri[1] = R.chol = R.zv = R.uperm = nullptr;   // (LDS block: unused)
  R.nz = R.perm = R.idx = nullptr;
  R.bvec = (double*)(smem + LY::off_bvec);
  R.w = wls;
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
    cx->P = P; cx->T = T; cx->D = D; cx->W = a.W; cx->S = a.S; cx->n_iter = n_iter;
    cx->prof = nullptr;
    scal[SC_OBS_DK] = (float)cx->sp.obs_scale0;
    scal[SC8_INIT_LOC] = (float)cx->sp.init_level_loc;
    scal[SC8_INIT_VAR] = (float)(cx->sp.init_level_scale * cx->sp.init_level_scale);
    scal[SC8_INIT_SVAR] = (float)(cx->sp.init_slope_scale * cx->sp.init_slope_scale);
  }
  {
    for (int e = tid; e < P * P; e += NT8) {
      R.xtx[e] = a.xtx[(size_t)series * P * P + e];
      R.omega[e] = a.omega[(size_t)series * P * P + e];
    }
    if constexpr (!XG) {
      float* xw_ = (float*)(smem + LY::off_x);
      for (int j = 0; j < P; ++j)
        for (int t = tid; t < TPAD; t += NT8) xw_[j * TPAD + t] = (t < T) ? Xg[(size_t)j * T + t] : 0.f;
    }
    if (tid < 16) wls[tid] = 0.f;                   // weights = 0            :575-578
  }
  __syncthreads();
  unsigned char* pre_base = smem + LY::off_pre;

  if (wave > NW) {
    // ================================ randomness waves =========================================
    // Everything random the other waves consume, one iteration ahead, into LDS:
    //   zb[0..2][TPAD]  level / slope / observation disturbances of the draw of the NEXT
    //                   iteration, chunk c (= Philox call c: 4 normals) at floats 4c .. 4c + 3;
    //   gam / pre       gamma variates (wave 5) and the regression block's permutation ranks,
    //                   flip uniforms and weight normals (wave 6), double-buffered by parity;
    //   scal[SC8_ZINIT] the x_0 normals (wave 7), double-buffered by parity.
    const int e = wave - NW - 1;                       // 0, 1, 2
    constexpr int NTASK = (D == 2) ? 3 : 2;            // zl, [zs,] zo (the predictive normals are
                                                       // drawn by the time waves while they wait for
                                                       // sigma^2_obs)
    constexpr int ROUNDS = NTASK * L;                  // wave-rounds of 64 chunks
    // contiguous, equal shares
    constexpr int N0 = ROUNDS / 3, N1 = (ROUNDS - N0) / 2;
    const int k_lo = e == 0 ? 0 : (e == 1 ? N0 : N0 + N1);
    const int k_hi = e == 0 ? N0 : (e == 1 ? N0 + N1 : ROUNDS);
    const int n_my = k_hi - k_lo;
    PF eprof;
    eprof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && wave == NW + 1 && lane == 0);
    auto normals_round = [&](int k, uint32_t it_for) __attribute__((always_inline)) {
      // the disturbances of the draw of iteration it_for; task-major: all chunks of a kind, then
      // the next kind
      const int task = k / L, blk = k - task * L;
      const int slot = (D == 2) ? task : (task == 0 ? 0 : 2);            // D = 1 has no slope row
      const uint32_t site = slot == 0 ? SITE_PRIOR_LEVEL : (slot == 1 ? SITE_PRIOR_SLOPE : SITE_PRIOR_OBS);
      const int c = 64 * blk + lane;
      const U4 r = site_call(rng, it_for, site, 0, (uint32_t)c);
      float z[4];
      normals4(r, z);
      *reinterpret_cast<float4*>(zb + (size_t)slot * TPAD + 4 * c) = make_float4(z[0], z[1], z[2], z[3]);
    };
    auto role_work = [&](int it_next) __attribute__((always_inline)) {
      if (it_next > n_iter) return;
      if (e == 0) {
        serial_gammas<1>(cx, it_next, lane, gam + 4 * (it_next & 1));
      } else if (e == 1) {
        if (it_next < n_iter) spike_slab_randoms(rng, (uint32_t)it_next, P, lane, gam + 8 + 32 * (it_next & 1));
      } else if (it_next < n_iter) {
        // (by all lanes, stored by the first D: see the four-wave kernel)
        float zi[1];
        fill_normals<1>(rng, (uint32_t)it_next, SITE_PRIOR_INIT, 0, (uint32_t)lane, zi);
        asm volatile("" : "+v"(zi[0]));
        if (lane < D) scal[SC8_ZINIT + 2 * (it_next & 1) + lane] = zi[0];
      }
    };
    // iteration 0's randomness
    role_work(0);
    for (int k = k_lo; k < k_hi; ++k) normals_round(k, 0u);
    eprof.tick(30);
    // rounds finished on arrival at (B3), (B4), (B5): see "scheduling of the helper waves" above.
    // $CI_SCHED_WORD modes 1 / 2 (everything right after (Bs) / everything after (B5)) exist for the
    // timing-independence test.
    const int sched = (a.dbg & 0xFFFF) ? (a.dbg & 0xFFFF) : SCHED_DEFAULT;
    const int mode = sched & 3;
    const int scale = L >= 4 ? L / 4 : 1;                // rounds grow with the steps per thread
    const int c_3 = e == 2 ? 0 : ((sched >> 10) & 3) * scale, c_4 = ((sched >> 12) & 3) * scale,
              c_a = ((sched >> 14) & 3) * scale;
    const int r_3 = mode == 1 ? n_my : (mode == 2 ? 0 : (c_3 < n_my ? c_3 : n_my));
    const int r_4 = mode == 1 ? n_my : (mode == 2 ? 0 : (r_3 + c_4 < n_my ? r_3 + c_4 : n_my));
    const int r_5 = mode == 2 ? 0 : (n_my - c_a > r_4 ? n_my - c_a : r_4);
    for (int it = 0; it <= n_iter; ++it) {
      __syncthreads();                                   // (B1)
      eprof.tick(31);
      xt_sums_wave<L, 2, XG>(tgv, Xs, TPAD, P, 2 * wave, wave == NW8 - 1, red, lane, T, xwide);
      eprof.tick(30);
      __syncthreads();                                   // (B2)
      eprof.tick(31);
      role_work(it + 1);
      eprof.tick(30);
      __syncthreads();                                   // (Bs)
      eprof.tick(31);
      if (e == 2 && it < n_iter) {
        // the weights of iteration `it`, when the serial section asks for it (the regression wave
        // shares its SIMD with time wave 0, which is on the critical path from here to (B3))
        const PreTables tb = pre_tables_at(pre_base, it);
        const int* hi_ = reinterpret_cast<const int*>(tb.hand + HAND_INTS);
        if (hi_[2] != 0) {
          const unsigned long long S = ((unsigned long long)(unsigned)hi_[1] << 32) | (unsigned)hi_[0];
          weights_backsub(tb, gam + 8 + 32 * (it & 1), P, S, tb.hand[lane & 15], tb.hand[HAND_SCALE],
                          lane, wls);
        }
      }
      // the rounds of normals for iteration it + 1: after (Bs) -- the time waves have read this
      // iteration's normals in the window before it -- and before (B1) of the next iteration
      int r = 0;
      if (it < n_iter)
        for (; r < r_3; ++r) normals_round(k_lo + r, (uint32_t)(it + 1));
      eprof.tick(30);
      __syncthreads();                                   // (B3)
      eprof.tick(31);
      if (it == n_iter) break;
      for (; r < r_4; ++r) normals_round(k_lo + r, (uint32_t)(it + 1));
      eprof.tick(30);
      __syncthreads();                                   // (B4)
      eprof.tick(31);
      for (; r < r_5; ++r) normals_round(k_lo + r, (uint32_t)(it + 1));
      eprof.tick(30);
      __syncthreads();                                   // (B5)
      eprof.tick(31);
      for (; r < n_my; ++r) normals_round(k_lo + r, (uint32_t)(it + 1));
      eprof.tick(30);
    }
    return;
  }

  if (wave == NW) {
    // ================================ regression wave ==========================================
    PriorCarry pc;
    pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;
    PreState ps;
    ps.valid = 0; ps.S = 0ull; ps.n_sw = 0; ps.diag = 1.0; ps.rdk = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) ps.c[r] = 0.0;
    PF rprof;
    rprof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && lane == 0);
    double obs_scale = cx->sp.obs_scale0, level_scale = cx->sp.level_scale0, slope_scale = cx->sp.slope_scale0;
    const DevSeriesParams sp = cx->sp;
    float* o_obs = a.out_obs;
    float* o_ls = a.out_level_scale;
    float* o_ss = a.out_slope_scale;
    float* o_w = a.out_weights;
    const int sched = (a.dbg & 0xFFFF) ? (a.dbg & 0xFFFF) : SCHED_DEFAULT;
    const int mode = sched & 3;
    for (int it = 0; it <= n_iter; ++it) {
      __syncthreads();                                   // (B1) targets in LDS
      // from here to sigma^2_obs this wave is the critical path of the iteration and shares its
      // SIMD with time wave 0, which has slack until (Bs): win the issue arbitration while it lasts
      __builtin_amdgcn_s_setprio(3);
      xt_sums_wave<L, 2, XG>(tgv, Xs, TPAD, P, 2 * wave, false, red, lane, T, xwide);
      __syncthreads();                                   // (B2) all sums complete
      rprof.tick(16);
      const PreTables tb = pre_tables_at(pre_base, it);
      if (lane < P + 1) R.bvec[lane] = (double)red[lane < P ? lane : RED_YTY];
      const float wprev = lane < P ? wls[lane] : 0.f;    // the previous draw's weights (stored below)
      const double* gm = gam + 4 * (it & 1);
      const double g_level = gm[0], g_slope = gm[1], g_obs = gm[2];
      const double emit_obs = obs_scale;
      rprof.tick(20);
      // sigma_obs(it) to the time waves -- and, on the replay route with no flip accepted in this
      // iteration, the weights draw to randomness wave 7 (it runs on another SIMD; from here to
      // (B3) time wave 0, which shares this wave's SIMD, is on the critical path, this wave is not)
      auto publish4 = [&](double ns, double mean, unsigned long long S, bool clean) __attribute__((always_inline)) {
        if (lane == 0) scal[SC_OBS_DK] = (float)ns;
        if (lane < 16) tb.hand[lane] = mean;
        if (lane == 0) {
          tb.hand[HAND_SCALE] = ns;
          int* hi_ = reinterpret_cast<int*>(tb.hand + HAND_INTS);
          hi_[0] = (int)(unsigned)(S & 0xFFFFFFFFull);
          hi_[1] = (int)(unsigned)(S >> 32);
          hi_[2] = clean ? 1 : 0;
        }
        rprof.tick(22);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();                                 // (Bs) sigma^2_obs(it) published
        rprof.tick(18);
        return clean;
      };
      auto publish1 = [&](double ns) __attribute__((always_inline)) { (void)publish4(ns, 0.0, 0ull, false); };
      if (it < n_iter) {
        if (ps.valid && ps.S == pc.S) {
          obs_scale = spike_slab_draw_pre(R, P, sp, obs_scale, g_obs, lane, pc, gam + 8 + 32 * (it & 1), tb,
                                          ps, red, rprof, publish4);
        } else {
          NoProf np;
          wave_sync();                                   // R.bvec written above
          obs_scale = spike_slab_draw_regs(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, np, pc,
                                           gam + 8 + 32 * (it & 1), publish1);
        }
      } else {
        publish1(obs_scale);
      }
      rprof.tick(23);
      // level / slope scales of iteration it - 1 (the time waves formed the same values for their
      // draw already) and its scalar outputs: nobody waits for them, they follow (B3)
      auto after_b3 = [&]() __attribute__((always_inline)) {
        if (it > 0) {
          auto clipped_scale = [](double scale, double ss, double g, double ub) {
            const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
            return s < ub ? s : ub;
          };
          double ssl = 0.0, sss = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            ssl += (double)red[RED_INC + 2 * w];
            sss += (double)red[RED_INC + 2 * w + 1];
          }
          level_scale = clipped_scale(sp.level_scale, ssl, g_level, sp.level_ub);
          if (D == 2) slope_scale = clipped_scale(sp.slope_scale, sss, g_slope, sp.slope_ub);
          const int s = it - 1 - a.W;
          if (s >= 0) {
            const size_t o = chain_lin * a.S + s;
            if (lane == 0) {
              if (o_obs) o_obs[o] = (float)emit_obs;
              if (o_ls) o_ls[o] = (float)level_scale;
              if (o_ss) o_ss[o] = (float)(D == 2 ? slope_scale : 0.0);
            }
            if (o_w && lane < P) o_w[o * P + lane] = wprev;
          }
        }
      };
      if (it == n_iter) {
        __syncthreads();                                 // (B3)
        after_b3();
        break;
      }
      // Next iteration's matrix work starts HERE: it needs sigma^2_obs(it) and the included set,
      // not the weights (its tables are those of the other parity: the weights of this iteration
      // are still being drawn from this iteration's).  Each step runs in the first interval that
      // (B3) -- weights(it) published --, (B4), (B5) are passed in between.
      const PreTables tbn = pre_tables_at(pre_base, it + 1);
      const double var_next = obs_scale * obs_scale;
      PreRun st;
      pre_begin(st, pc.S);
      // steps (build, one sweep per member of the set, store) finished before (B3), (B4), (B5): the
      // build and p_3 sweeps (this wave waits there anyway once the weights are drawn elsewhere, but
      // its SIMD partner, time wave 0, is in its densest stretch), p_4 more, all but p_a; the store
      // of the tables comes last
      const int n_sw = __popcll(pc.S);
      const int p_3 = (sched >> 4) & 3, p_4 = (sched >> 6) & 3, p_a = (sched >> 8) & 3;
      const int s_3 = mode == 1 ? n_sw + 2 : (mode == 2 ? 0 : 1 + (p_3 < n_sw ? p_3 : n_sw));
      const int s_4 = mode == 1 ? n_sw + 2 : (mode == 2 ? 0 : (s_3 + p_4 < 1 + n_sw ? s_3 + p_4 : 1 + n_sw));
      const int s_5 = mode == 1 ? n_sw + 2 : (mode == 2 ? 0 : (1 + n_sw - p_a > s_4 ? 1 + n_sw - p_a : s_4));
      int step = 0;
      for (; st.phase != 3 && step < s_3; ++step) pre_step(st, R, P, var_next, pc.S, lane, tbn, ps);
      rprof.tick(19);
      __syncthreads();                                   // (B3) weights(it) published
      rprof.tick(17);
      after_b3();
      for (; st.phase != 3 && step < s_4; ++step) pre_step(st, R, P, var_next, pc.S, lane, tbn, ps);
      rprof.tick(19);
      __syncthreads();                                   // (B4)
      rprof.tick(29);
      for (; st.phase != 3 && step < s_5; ++step) pre_step(st, R, P, var_next, pc.S, lane, tbn, ps);
      rprof.tick(19);
      __syncthreads();                                   // (B5)
      rprof.tick(29);
      for (; st.phase != 3; ++step) pre_step(st, R, P, var_next, pc.S, lane, tbn, ps);
      rprof.tick(19);
    }
    return;
  }

  // ==================================== time waves =============================================
  // above the regression wave's precompute and the randomness waves (priority 0), below the
  // serial section (3)
  __builtin_amdgcn_s_setprio(1);
  const int t0 = tid * L;
  float yv[L];
  uint32_t maskbits = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    const bool in = t < T;
    yv[l] = in ? yg[t] : 0.f;
    const bool m = in ? (mg[t] != 0) : true;
    if (m) { maskbits |= (1u << l); yv[l] = 0.f; }
  }
  if constexpr (LY::PARK) store_targets<L>(ybuf, t0, yv);       // read back where they are used
  const float init_loc = scal[SC8_INIT_LOC], init_var = scal[SC8_INIT_VAR], init_svar = scal[SC8_INIT_SVAR];
  const DevSeriesParams* spp = &cx->sp;
  const double sp_level_scale0 = spp->level_scale0, sp_slope_scale0 = spp->slope_scale0;
  const double sp_level_scale = spp->level_scale, sp_slope_scale = spp->slope_scale;
  const double sp_level_ub = spp->level_ub, sp_slope_ub = spp->slope_ub;
  float lev[L], slp[L], xw[L], pm_acc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { lev[l] = 0.f; slp[l] = 0.f; xw[l] = 0.f; pm_acc[l] = 0.f; }  // :580-581
  float* o_level = a.out_level ? a.out_level + chain_lin * a.S * T : nullptr;
  float* o_slope = a.out_slope ? a.out_slope + chain_lin * a.S * T : nullptr;
  float* o_traj = a.out_traj ? a.out_traj + chain_lin * a.S * T : nullptr;
  PF prof;
  prof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && tid == 0);
  for (int it = 0; it <= n_iter; ++it) {
    // ---- targets (they use the CURRENT level) to LDS; the last owned state for the neighbour
    {
      float tg[L], yq[L];
      if constexpr (LY::PARK) {
        lds_row_load<L>(ybuf + t0, yq);
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l) yq[l] = yv[l];
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const bool obs = ((maskbits >> l) & 1u) == 0u;
        tg[l] = obs ? (yq[l] - lev[l]) : 0.f;
      }
      store_targets<L>(tgv, t0, tg);
      xlast[tid * D] = lev[L - 1];
      if constexpr (D == 2) xlast[tid * D + 1] = slp[L - 1];
    }
    __syncthreads();                                       // (B1)
    xt_sums_wave<L, 2, XG>(tgv, Xs, TPAD, P, 2 * wave, false, red, lane, T, xwide);
    {
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
        red[RED_INC + 2 * wave] = s1;
        red[RED_INC + 2 * wave + 1] = s2;
      }
    }
    // sigma_obs of iteration it-1's regression draw: the noise scale of its predictive trajectory
    // (read before the regression wave publishes this iteration's, which happens after (B2))
    const float so_prev = scal[SC_OBS_DK];
    __syncthreads();                                       // (B2)
    prof.tick(0);

    // ---- window: the normals of this iteration from LDS, emission of draw it-1, this iteration's
    // disturbance scales, first half of the prior-simulation scan
    float zl[L], zs[L], zo[L];
    const float* zit = zb + t0;
    lds_row_load<L>(zit, zl);
    if constexpr (D == 2) {
      lds_row_load<L>(zit + TPAD, zs);
    } else {
#pragma unroll
      for (int l = 0; l < L; ++l) zs[l] = 0.f;
    }
    lds_row_load<L>(zit + 2 * TPAD, zo);
    if (it > a.W) {
      const int s = it - 1 - a.W;
      float zp[L];
      fill_normals<L>(rng, (uint32_t)(it - 1), SITE_PRED, 0, (uint32_t)t0, zp);
      float tr[L];
      if constexpr (LY::PARK) {
        // X w of draw it-1 from LDS; the predictor's running sum in place in its output array (the
        // same sequence of float additions as the register copy of the other builds: same bits)
        float xwp[L], acc[L];
        lds_row_load<L>(xwb + t0, xwp);
        float* pm = a.out_pred_mean ? a.out_pred_mean + chain_lin * T : nullptr;
        const bool vec = (T & 3) == 0;
        if (pm != nullptr && s > 0) {
          if (vec) {
#pragma unroll
            for (int q = 0; q < L / 4; ++q) {
              const int t = t0 + 4 * q;
              const float4 v = t < T ? *(const float4*)(pm + t) : make_float4(0.f, 0.f, 0.f, 0.f);
              acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
            }
          } else {
#pragma unroll
            for (int l = 0; l < L; ++l) acc[l] = t0 + l < T ? pm[t0 + l] : 0.f;
          }
        } else {
#pragma unroll
          for (int l = 0; l < L; ++l) acc[l] = 0.f;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float loc = lev[l] + xwp[l];
          acc[l] += loc;
          tr[l] = fmaf(so_prev, zp[l], loc);
        }
        if (pm != nullptr) {
          if (vec) {
#pragma unroll
            for (int q = 0; q < L / 4; ++q) {
              const int t = t0 + 4 * q;
              if (t < T) *(float4*)(pm + t) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
          } else {
#pragma unroll
            for (int l = 0; l < L; ++l)
              if (t0 + l < T) pm[t0 + l] = acc[l];
          }
        }
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float loc = lev[l] + xw[l];
          pm_acc[l] += loc;
          tr[l] = fmaf(so_prev, zp[l], loc);
        }
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
    // The level / slope disturbance scales of this iteration -- the draw the regression wave makes
    // too (same expression, same inputs: the increments' sums in `red`, complete since (B2), and
    // the gamma variates of the randomness wave)
    Vec<D> sig;
    PElem<D> pincl = pelem_identity<D>();
    {
      auto clipped_scale = [](double scale, double ss, double g, double ub) {
        const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
        return s < ub ? s : ub;
      };
      double level_scale = sp_level_scale0, slope_scale = sp_slope_scale0;
      if (it > 0) {
        const double* gm = gam + 4 * (it & 1);
        double ssl = 0.0, sss = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          ssl += (double)red[RED_INC + 2 * w];
          sss += (double)red[RED_INC + 2 * w + 1];
        }
        level_scale = clipped_scale(sp_level_scale, ssl, gm[0], sp_level_ub);
        if constexpr (D == 2) slope_scale = clipped_scale(sp_slope_scale, sss, gm[1], sp_slope_ub);
      }
      sig.v[0] = (float)level_scale;
      if constexpr (D == 2) sig.v[1] = (float)slope_scale;
      if (it < n_iter) pincl = dk_prior_begin<D, L>(sig, zl, zs, slots, lane, wave);
    }
    const bool publish = a.progress != nullptr && it > a.W &&
                         ((it - a.W) % a.progress_every == 0 || it - a.W == a.S);
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows are in L2
    prof.tick(2);
    __syncthreads();                                       // (Bs) sigma^2_obs(it) in LDS
    prof.tick(1);
    if (publish) {
      // the time waves' stores of the first `done` rows were issued before (Bs); make them visible
      // to the copy engines and publish (the scalars / weights of these draws are copied after
      // the kernel has ended)
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.progress + chain_lin, (unsigned int)(it - a.W), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (it == n_iter) {
      __syncthreads();                                     // (B3)
      break;
    }

    // ---- matrix phase of the draw of iteration it: needs sigma^2_obs, not the weights
    DkModel<D> md;
    {
      const float so = scal[SC_OBS_DK];
      md.H = so * so;
      md.sig = sig;
      md.a1 = vzero<D>();
      md.a1.v[0] = init_loc;
      md.p1.v[0] = init_var;
      if constexpr (D == 2) md.p1.v[1] = init_svar;
    }
    Vec<D> q;
#pragma unroll
    for (int i = 0; i < D; ++i) q.v[i] = md.sig.v[i] * md.sig.v[i];
    const PElem<D> ppre = dk_prior_finish<D>(pincl, slots, lane, wave);   // ((Bs) was its barrier)
    const FMElem<D> fme = dk_matrix_chunk<D, L>(md, q, maskbits, tid);
    const FMElem<D> mincl = dk_matrix_scan_begin<D>(fme, slots + NW * 16, lane, wave);
    prof.tick(24);
    __syncthreads();                                       // (B3) weights(it) in LDS; matrix scan
    float Pin[D * (D + 1) / 2];
    dk_matrix_scan_finish<D>(mincl, slots + NW * 16, lane, wave, Pin);
    DkMats<D, L> km;
    dk_matrix_local<D, L>(md, q, maskbits, tid, Pin, km);
    prof.tick(25);

    // ---- residual, data phase
    float resid[L];
#pragma unroll
    for (int l = 0; l < L; ++l) xw[l] = 0.f;
    {
      float wv[16];
      lds_row_load<16>(wls, wv);           // the weights vector is padded to 16 floats
      if constexpr (!XG) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int jj = j < P ? j : P - 1;
          const float wj = j < P ? wv[j] : 0.f;
          float xr[L];
          lds_row_load<L>(Xs + jj * TPAD + t0, xr);
#pragma unroll
          for (int l = 0; l < L; ++l) xw[l] = fmaf(xr[l], wj, xw[l]);
        }
      } else {
        // the streamed design: rows in batches of independent loads, and only the rows of the
        // INCLUDED features (a zero weight adds an exact zero: the same sums, bit for bit, as above)
        constexpr int RB = L >= 16 ? 2 : (L >= 8 ? 4 : 8);
        auto stream = [&](auto load_row) __attribute__((always_inline)) {
          unsigned long long todo = __ballot(lane < P && wls[lane < P ? lane : 0] != 0.f);
          while (todo != 0ull) {
            float xr[RB][L], wj[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
              const bool have = todo != 0ull;
              const int j = have ? __ffsll((long long)todo) - 1 : 0;
              todo &= todo - 1ull;
              wj[u] = have ? wls[j] : 0.f;
              load_row(j, xr[u]);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u)
#pragma unroll
              for (int l = 0; l < L; ++l) xw[l] = fmaf(xr[u][l], wj[u], xw[l]);
          }
        };
        if (xwide) {
          if constexpr (L % 4 == 0)
            stream([&](int j, float (&xr)[L]) { global_row_load_wide<L>(Xg + (size_t)j * T, t0, T, xr); });
        } else {
          stream([&](int j, float (&xr)[L]) { global_row_load_scalar<L>(Xg + (size_t)j * T, t0, T, xr); });
        }
      }
    }
    if constexpr (LY::PARK) {
      float yq[L];
      lds_row_load<L>(ybuf + t0, yq);
#pragma unroll
      for (int l = 0; l < L; ++l) resid[l] = yq[l] - xw[l];
    } else {
#pragma unroll
      for (int l = 0; l < L; ++l) resid[l] = yv[l] - xw[l];
    }
    if constexpr (LY::PARK) store_targets<L>(xwb, t0, xw);     // read back by the emission of this draw
    const Vec<D> a1e = dk_initial_mean<D>(md, rng, (uint32_t)it, tid, scal + SC8_ZINIT + 2 * (it & 1));
    Vec<D> xp[L];
    float ytil[L];
    dk_prior_path<D, L>(md, ppre, resid, zl, zs, zo, xp, ytil);
    prof.tick(3);
    const AElem<D> fe = dk_fwd_chunk<D, L>(km, a1e, ytil, maskbits, tid);
    const AElem<D> fincl = aff_fwd_begin<D>(fe, slots, lane, wave);
    prof.tick(27);
    __syncthreads();                                       // (B4)
    const Vec<D> mf = aff_fwd_finish<D>(fincl, slots, lane, wave);
    prof.tick(5);
    Vec<D> ap[L];
    float vf[L];
    const AElem<D> be = dk_fwd_local_bwd_chunk<D, L>(km, a1e, ytil, maskbits, tid, mf, ap, vf);
    const AElem<D> bincl = aff_bwd_begin<D>(be, slots + 2 * NW * 16, lane, wave);
    prof.tick(6);
    __syncthreads();                                       // (B5)
    const Vec<D> rr = aff_bwd_finish<D>(bincl, slots + 2 * NW * 16, lane, wave);
    prof.tick(28);
    Vec<D> x[L];
    dk_bwd_fixup<D, L>(km, rr, ap, vf, xp, x);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      lev[l] = x[l].v[0];
      if constexpr (D == 2) slp[l] = x[l].v[1];
    }
    prof.tick(7);
  }

  if (a.out_pred_mean) {
    const float inv = 1.0f / (float)(a.S > 0 ? a.S : 1);
    float* pm = a.out_pred_mean + chain_lin * T;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int t = t0 + l;
      if constexpr (LY::PARK) {
        if (t < T) pm[t] = (a.S > 0 ? pm[t] : 0.f) * inv;      // (this thread's own sums, written above)
      } else {
        if (t < T) pm[t] = pm_acc[l] * inv;
      }
    }
  }
}

}  // namespace ci
