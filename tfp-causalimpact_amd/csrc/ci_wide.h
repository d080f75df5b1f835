// ci_wide.h -- time-parallel Gibbs kernel for trend + one seasonal block, any series length (up to 65536 steps).
//
// Model: LocalLevel / LocalLinearTrend (TR = 1 / 2 trend dims) + one
// tfp.sts.Seasonal(num_seasons = NS, constrain_mean_effect_to_zero=True) block carried in the
// oracle's (NS-1)-effect coordinates (oracle/ci_oracle.c: apply_transition, propagate_cov,
// initial_moments), state dim d = TR + NS - 1 <= 8.  This is BASELINE config "T=10000,
// 50 covariates + Seasonal(num_seasons=7)"  and every daily series
// with a weekly cycle.
//
// One 256-thread workgroup per (series, chain).  Thread i owns the Lc = ceil(T/256) (rounded up
// to 4) consecutive steps [i Lc, (i+1) Lc).  Every recursion of the Durbin-Koopman draw
// (oracle: ci_oracle_dk_draw) is cut the same way:
//   per-thread pass over the chunk  ->  block scan of 256 chunk elements  ->  per-thread pass
// so the sequential depth is O(Lc + log 256) instead of T:
//   * prior simulation x+   : elements (steps, season changes mod NS, offset vector);
//   * Kalman filter         : Sarkka & Garcia-Fernandez (2021) elements (A, b, C, eta, J), d x d
//                             dense; a chunk's element is built by an O(d^2)-per-step recursion
//                             (not by d^3 combines), only the 256-element scan pays d^3;
//   * backward recursion r  : affine maps (M, c), d x d dense;
//   * x~_t = x+_t + a_t + P_t r_{t-1} in the last per-thread pass.
// Per-step intermediates (y~, x+ + a_t, P_t, K_t, v_t/F_t) live in a per-chain HBM workspace laid
// out [step-in-chunk][field][thread] so that every access is a fully coalesced 1 KiB row and each
// thread only ever reads what it wrote.  The regression block, scale draws and random streams are
// the ones of ci_kernels.h (same sites, same counters), so draws agree with the oracle per
// random number.
#pragma once
#include "ci_kernels.h"
#include "ci_seasonal.h"   // SArgs, DevSeasonalParams (declarations only in this TU)
#include "ci_bigp.h"       // the regression draw of the BIGP builds (53+ design columns)

namespace ci {

#ifndef CI_LDS
#define CI_LDS __attribute__((address_space(3)))
#define CI_GLB __attribute__((address_space(1)))
#endif
typedef float ci_f4v __attribute__((ext_vector_type(4)));

constexpr int WIDE_MAX_LC = 256;      // T <= 65536
constexpr int XR = 16;              // design rows per pass of the X'targets / X w loops

template <int TR, int NS> struct WDim {
  static constexpr int D = TR + NS - 1;
  static constexpr int O = TR;          // first effect of the seasonal block
  static constexpr int N1 = NS - 1;
  static constexpr int NPS = D * (D + 1) / 2;
  // per-step private fields: y~ | K_t | v_t / F_t | r_{t-1}.  (The covariance P_t and the
  // simulated path x+_t are NOT stored: the smoothed path comes from the forward recursion
  // x^_{t+1} = T x^_t + Q_t r_t of the fast state smoother (de Jong 1989; Koopman 1993) started at
  // the chunk's predicted moments, and x+ is re-simulated from the counter-based stream -- 16
  // floats per step at d = 7 instead of 44.)
  static constexpr int F_YT = 0, F_KF = 1, F_VF = 1 + D, F_RS = 2 + D;
  static constexpr int NF = 2 + 2 * D;
};

// The Durbin-Koopman draw over the cluster's workgroups (ci_wide_quad.h): its exchange region.
constexpr int DK_V = 8;               // virtual workgroups (64 quads of lanes each) per chain
constexpr int DK_CH = 64 * DK_V;      // chunks of the series: its grid, fixed by T alone
constexpr int DK_NWI = 4 * DK_V;      // wavefronts of the chain's grid
constexpr int DK_EF = 72;             // floats per lane of a published filtering element (d <= 8: 68)
constexpr int DK_EA = 32;             //                         ... backward map (d <= 8: 24)
constexpr int DK_EP = 12;             // floats of a published prior-simulation element (d <= 8: 10)
constexpr int DK_ST = 20;             // per chunk: 3 sums of squares, first state (8), last state (8)
constexpr int DK_VS = 80;             // floats per lane parked between phases when one workgroup runs several virtual ones
constexpr int DK_NF = 9;              // per-step workspace: y~ -> v/F, K_t -> r_{t-1} (8 floats)
__host__ __device__ inline size_t wide_dk_floats() {
  return (size_t)DK_NWI * 4 * DK_EF + (size_t)DK_NWI * 4 * DK_EA + 512 + (size_t)DK_CH * DK_ST +
         (size_t)DK_V * NT * DK_VS;
}
// steps per chunk of the grid (a multiple of 4: whole 16-byte accesses)
__host__ __device__ inline int wide_quad_steps(int T) {
  const int lc = (T + DK_CH - 1) / DK_CH;
  return lc < 4 ? 4 : (lc + 3) & ~3;
}
// floats of HBM workspace per chain
__host__ __device__ inline size_t wide_workspace_floats(int D, int Lc) {
  (void)D;
  const size_t TP = (size_t)DK_CH * Lc;
  return (6 + DK_NF) * TP + TP / 2 + wide_dk_floats();   // 6 shared T-arrays, per-step fields, mask + change bytes, exchange
}

// ---- x <- T_t x and friends (oracle: apply_transition / apply_transition_T) ---------------
template <int TR, int NS>
__device__ __forceinline__ void w_season_shift(Vec<TR + NS - 1>& x) {
  constexpr int O = TR, N1 = NS - 1;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N1; ++i) s += x.v[O + i];
#pragma unroll
  for (int i = 0; i + 1 < N1; ++i) x.v[O + i] = x.v[O + i + 1];
  x.v[O + N1 - 1] = -s;
}
template <int TR, int NS>
__device__ __forceinline__ void w_apply(Vec<TR + NS - 1>& x, bool ch) {
  if constexpr (TR == 2) x.v[0] += x.v[1];
  if (ch) w_season_shift<TR, NS>(x);
}
template <int TR, int NS>
__device__ __forceinline__ void w_apply_t(Vec<TR + NS - 1>& x, bool ch) {
  constexpr int O = TR, N1 = NS - 1;
  if constexpr (TR == 2) x.v[1] += x.v[0];
  if (ch) {
    const float last = x.v[O + N1 - 1];
#pragma unroll
    for (int j = N1 - 1; j >= 1; --j) x.v[O + j] = x.v[O + j - 1] - last;
    x.v[O] = -last;
  }
}
// A <- T A
template <int TR, int NS>
__device__ __forceinline__ void w_left(Mat<TR + NS - 1>& A, bool ch) {
  constexpr int D = TR + NS - 1, O = TR, N1 = NS - 1;
  if constexpr (TR == 2) {
#pragma unroll
    for (int j = 0; j < D; ++j) A.m[0][j] += A.m[1][j];
  }
  if (ch) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < N1; ++i) s += A.m[O + i][j];
#pragma unroll
      for (int i = 0; i + 1 < N1; ++i) A.m[O + i][j] = A.m[O + i + 1][j];
      A.m[O + N1 - 1][j] = -s;
    }
  }
}
// A <- A T'
template <int TR, int NS>
__device__ __forceinline__ void w_right_t(Mat<TR + NS - 1>& A, bool ch) {
  constexpr int D = TR + NS - 1, O = TR, N1 = NS - 1;
  if constexpr (TR == 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) A.m[i][0] += A.m[i][1];
  }
  if (ch) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < N1; ++j) s += A.m[i][O + j];
#pragma unroll
      for (int j = 0; j + 1 < N1; ++j) A.m[i][O + j] = A.m[i][O + j + 1];
      A.m[i][O + N1 - 1] = -s;
    }
  }
}
// A <- T' A
template <int TR, int NS>
__device__ __forceinline__ void w_left_t(Mat<TR + NS - 1>& A, bool ch) {
  constexpr int D = TR + NS - 1, O = TR, N1 = NS - 1;
  if constexpr (TR == 2) {
#pragma unroll
    for (int j = 0; j < D; ++j) A.m[1][j] += A.m[0][j];
  }
  if (ch) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float last = A.m[O + N1 - 1][j];
#pragma unroll
      for (int r = N1 - 1; r >= 1; --r) A.m[O + r][j] = A.m[O + r - 1][j] - last;
      A.m[O][j] = -last;
    }
  }
}

struct WideScal {
  float H, so, sl, ss, sdn;   // obs variance / scale, level, slope scales, drift scale / NS
  float ql, qs, qd;           // level, slope variances, (drift scale / NS)^2
};

// P <- T P T' + Q_t  (oracle: propagate_cov)
template <int TR, int NS>
__device__ __forceinline__ void w_cov_predict(Mat<TR + NS - 1>& P, bool ch, const WideScal& sc) {
  constexpr int O = TR, N1 = NS - 1;
  w_left<TR, NS>(P, ch);
  w_right_t<TR, NS>(P, ch);
  P.m[0][0] += sc.ql;
  if constexpr (TR == 2) P.m[1][1] += sc.qs;
  if (ch) {
#pragma unroll
    for (int i = 0; i < N1; ++i)
#pragma unroll
      for (int j = 0; j < N1; ++j) P.m[O + i][O + j] += sc.qd;
  }
  symmetrize(P);
}

// P <- T P T' + Q_t on a PACKED upper triangle (symidx): the per-thread passes are VALU-issue bound
// at one wave per SIMD, so the covariance is never expanded to d x d.  Trend block first
// ([[1,1],[0,1]] on rows/columns 0,1), then the companion shift of the seasonal block in closed
// form: entries move up-left by one, the last row/column is minus the row sums, the corner is the
// total sum.
template <int TR, int NS>
__device__ __forceinline__ void w_cov_predict_sym(float (&C)[(TR + NS - 1) * (TR + NS) / 2], bool ch,
                                                  const WideScal& sc) {
  constexpr int D = TR + NS - 1, O = TR, N1 = NS - 1;
  auto S = [](int i, int j) constexpr { return symidx<D>(i, j); };
  if constexpr (TR == 2) {
    // rows/cols (0,1) <- [[1,1],[0,1]] . [[1,0],[1,1]]
    C[S(0, 0)] += 2.0f * C[S(0, 1)] + C[S(1, 1)];
    C[S(0, 1)] += C[S(1, 1)];
#pragma unroll
    for (int j = 2; j < D; ++j) C[S(0, j)] += C[S(1, j)];
  }
  if (ch) {
    float rs[N1];                 // row sums of the seasonal block
    float cross[TR];              // row sums of the trend x seasonal block
#pragma unroll
    for (int p = 0; p < N1; ++p) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < N1; ++k) a += C[S(O + p, O + k)];
      rs[p] = a;
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < N1; ++k) a += C[S(r, O + k)];
      cross[r] = a;
    }
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < N1; ++p) tot += rs[p];
    // shifted entries (ascending order: every source lies after its destination)
#pragma unroll
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int q = 0; q + 1 < N1; ++q) C[S(r, O + q)] = C[S(r, O + q + 1)];
      C[S(r, O + N1 - 1)] = -cross[r];
    }
#pragma unroll
    for (int p = 0; p + 1 < N1; ++p) {
#pragma unroll
      for (int q = p; q + 1 < N1; ++q) C[S(O + p, O + q)] = C[S(O + p + 1, O + q + 1)];
      C[S(O + p, O + N1 - 1)] = -rs[p + 1];
    }
    C[S(O + N1 - 1, O + N1 - 1)] = tot;
#pragma unroll
    for (int p = 0; p < N1; ++p)
#pragma unroll
      for (int q = p; q < N1; ++q) C[S(O + p, O + q)] += sc.qd;
  }
  C[S(0, 0)] += sc.ql;
  if constexpr (TR == 2) C[S(1, 1)] += sc.qs;
}

// ---- prior-simulation element: x_out = Phi x_in + s, Phi fixed by (steps, changes mod NS) ----
template <int D> struct WPElem {
  float k;
  int m;
  Vec<D> s;
};
template <int TR, int NS>
__device__ __forceinline__ WPElem<TR + NS - 1> wpelem_combine(const WPElem<TR + NS - 1>& e1,
                                                              const WPElem<TR + NS - 1>& e2) {
  WPElem<TR + NS - 1> r;
  r.k = e1.k + e2.k;
  int m = e1.m + e2.m;
  r.m = m >= NS ? m - NS : m;
  r.s = e1.s;
  if constexpr (TR == 2) r.s.v[0] = fmaf(e2.k, r.s.v[1], r.s.v[0]);
#pragma unroll 1
  for (int i = 0; i < e2.m; ++i) w_season_shift<TR, NS>(r.s);
#pragma unroll
  for (int i = 0; i < TR + NS - 1; ++i) r.s.v[i] += e2.s.v[i];
  return r;
}

// ---- block scans with the operator kept in ROLLED loops (a d = 8 combine is ~6k instructions;
// unrolling the 6 wave levels would blow the instruction cache) -----------------------------
template <class E, class Op>
__device__ __forceinline__ E block_scan_excl_fwd_rolled(const E& tot, Op op, const E& ident,
                                                        float* slots, int lane, int wave) {
  constexpr int N = sizeof(E) / 4;
  E incl = tot;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    const E o = shfl_up_e(incl, off);
    if (lane >= off) incl = op(o, incl);
  }
  if (lane == 63) lds_store_e(slots + wave * N, incl);
  __syncthreads();
  E ex = shfl_up_e(incl, 1);
  if (lane == 0) ex = ident;
  // prefix of the earlier waves' totals followed by ex: one rolled loop, one operator instance
  E acc = (wave == 0) ? ex : lds_load_e<E>(slots);
#pragma unroll 1
  for (int ww = 1; ww <= wave; ++ww) {
    const E nxt = (ww < wave) ? lds_load_e<E>(slots + ww * N) : ex;
    acc = op(acc, nxt);
  }
  return acc;
}
template <class E, class Op>
__device__ __forceinline__ E block_scan_excl_bwd_rolled(const E& tot, Op op, const E& ident,
                                                        float* slots, int lane, int wave) {
  constexpr int N = sizeof(E) / 4;
  E incl = tot;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    const E o = shfl_down_e(incl, off);
    if (lane + off < 64) incl = op(incl, o);
  }
  if (lane == 0) lds_store_e(slots + wave * N, incl);
  __syncthreads();
  E ex = shfl_down_e(incl, 1);
  if (lane == 63) ex = ident;
  // ex o W_{wave+1} o ... o W_{NW-1}   (op(outer, inner): the LAST wave's map acts first)
  E acc = ex;
#pragma unroll 1
  for (int ww = wave + 1; ww < NW; ++ww) acc = op(acc, lds_load_e<E>(slots + ww * N));
  return acc;
}

// (the layout of the <= 52-column builds, kept apart from the BIGP one below: the kernel's code
// generation is sensitive to what the layout struct carries)
struct WLayoutS {
  size_t xtx, omega, aug0, aug1, pri0, pri1, chol, bvec, zv, uperm, nz, perm, idx, w, scal, red,
      st, bpre, gsum, big0, total;
};
// The small arrays first, the regression block's matrices from `big0` on: a DK worker that is
// neither the main workgroup nor the sweeper never touches the matrices, and keeps the per-step
// workspace of its share of the Durbin-Koopman draw there (wide_dk_lds_bytes, ci_wide_quad.h).
__host__ __device__ inline WLayoutS make_wlayout_small(int P, int D) {
  WLayoutS l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const int Pp = P > 0 ? P : 1;
  const bool big = P > 16;    // the LDS-resident regression block is only used for P > 16
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.zv = take(sizeof(double) * Pp);
  l.uperm = take(sizeof(double) * Pp);
  l.nz = take(sizeof(int) * Pp);
  l.perm = take(sizeof(int) * Pp);
  l.idx = take(sizeof(int) * Pp);
  l.w = take(sizeof(float) * (Pp > 16 ? Pp : 16));
  l.scal = take(sizeof(float) * 16);
  l.red = take(sizeof(float) * NW * ((Pp > 16 ? Pp : 16) + 4));
  (void)D;
  l.st = take(sizeof(double) * 8);      // serial wave -> block: previous sigma_obs, gamma variate; [2..5] gamma variates drawn ahead
  l.bpre = take(sizeof(double) * BLOCK_PRE_DOUBLES);   // the regression block's randomness (block_randoms) drawn ahead
  l.gsum = take(sizeof(double) * NW * 64);   // quarter sums of the segment partials
  o = (o + 127) & ~(size_t)127;
  l.big0 = o;
  l.xtx = take(sizeof(double) * Pp * Pp);
  l.omega = take(sizeof(double) * Pp * Pp);
  l.aug0 = take(big ? sizeof(double) * block_matrix_doubles(Pp + 1) : 16);
  l.aug1 = take(16);
  l.pri0 = take(big ? sizeof(double) * block_matrix_doubles(Pp) : 16);
  l.pri1 = take(16);
  l.chol = take(big ? sizeof(double) * block_chol_doubles(Pp) : 16);   // recorded pivot rows / Cholesky + staging
  l.total = o;
  return l;
}
struct WLayout {
  size_t xtx, omega, aug0, aug1, pri0, pri1, chol, bvec, zv, uperm, nz, perm, idx, w, scal, red,
      st, bpre, gsum, big0, total;
  // BIGP builds (P > MAXP; spike_slab_draw_big_wg): aug0 = the packed swept matrix, pri0 = the packed
  // swept prior block when it fits (pm_lds; else the chain's workspace), the sweeps' (i, j) table,
  // the draw's row buffers; gs = doubles per wave of `gsum`
  size_t ijtab, trow;
  int pm_lds, gs;
};
// The small arrays first, the regression block's matrices from `big0` on: a DK worker that is
// neither the main workgroup nor the sweeper never touches the matrices, and keeps the per-step
// workspace of its share of the Durbin-Koopman draw there (wide_dk_lds_bytes, ci_wide_quad.h).
__host__ __device__ inline WLayout make_wlayout(int P, int D) {
  WLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const int Pp = P > 0 ? P : 1;
  const bool bigp = P > MAXP;
  const bool big = P > 16 && !bigp;    // the LDS-resident regression block of 17..MAXP columns
  l.gs = bigp ? ((Pp + 1 + 63) & ~63) : 64;
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.zv = take(sizeof(double) * Pp);
  l.uperm = take(sizeof(double) * Pp);
  l.nz = take(sizeof(int) * Pp);
  l.perm = take(sizeof(int) * Pp);
  l.idx = take(sizeof(int) * Pp);
  l.w = take(sizeof(float) * (Pp > 16 ? Pp : 16));
  l.scal = take(sizeof(float) * 16);
  l.red = take(sizeof(float) * NW * ((Pp > 16 ? Pp : 16) + 4));
  (void)D;
  l.st = take(sizeof(double) * 8);      // serial wave -> block: previous sigma_obs, gamma variate; [2..5] gamma variates drawn ahead
  l.bpre = take(sizeof(double) * BLOCK_PRE_DOUBLES);   // the regression block's randomness (block_randoms) drawn ahead
  l.gsum = take(sizeof(double) * NW * l.gs);   // quarter sums of the segment partials
  l.trow = 0;
  if (bigp) l.trow = take(sizeof(double) * bigp_trow_doubles(Pp));
  o = (o + 127) & ~(size_t)127;
  l.big0 = o;
  l.ijtab = 0; l.pm_lds = 0;
  if (bigp) {
    // X'X and Omega stay where the setup kernel left them (global memory)
    l.xtx = take(16); l.omega = take(16);
    l.aug0 = take(sizeof(double) * bigp_packed(Pp + 1));
    l.aug1 = take(16);
    l.ijtab = take(sizeof(unsigned) * bigp_packed(Pp + 1));
    const size_t p_bytes = sizeof(double) * bigp_packed(Pp);
    l.pm_lds = o + p_bytes + 64 <= 160 * 1024 - 512 ? 1 : 0;
    l.pri0 = take(l.pm_lds ? p_bytes : 16);
    l.pri1 = take(16);
    l.chol = take(16);
    l.total = o;
    return l;
  }
  l.xtx = take(sizeof(double) * Pp * Pp);
  l.omega = take(sizeof(double) * Pp * Pp);
  l.aug0 = take(big ? sizeof(double) * block_matrix_doubles(Pp + 1) : 16);
  l.aug1 = take(16);
  l.pri0 = take(big ? sizeof(double) * block_matrix_doubles(Pp) : 16);
  l.pri1 = take(16);
  l.chol = take(big ? sizeof(double) * block_chol_doubles(Pp) : 16);   // recorded pivot rows / Cholesky + staging
  l.total = o;
  return l;
}
// floats of the per-chain cluster message `cw`: weights [P], emission scale, then (from cw0) sigma_obs
// as a double and the four scales of the draw
__host__ __device__ inline int wide_cw0(int P) { return P > MAXP ? ((P + 4) & ~3) : 56; }
__host__ __device__ inline int wide_cw_floats(int P) { return P > MAXP ? wide_cw0(P) + 8 : 64; }
// doubles of the per-chain buffer `cv`: the matrix swept ahead (17..MAXP columns), or the BIGP builds'
// workspace -- the swept prior block when it has no room in LDS, then the included block's factor
__host__ __device__ inline size_t wide_cv_doubles(int P) {
  return P > MAXP ? 2 * (size_t)P * P + 16 : presweep_doubles(P);
}
// LDS a DK worker needs to keep its share of the draw's per-step rows (K_t / r_{t-1}: 8 floats, y~ /
// v/F: 1 float, for 64 chunks of Lc steps) and the 72 floats per lane parked between the phases
__host__ __device__ inline size_t wide_dk_lds_bytes(int Lc) {
  return sizeof(float) * ((size_t)DK_VS * NT + (size_t)Lc * 64 * 9);
}

}  // namespace ci
#include "ci_wide_quad.h"
namespace ci {

// ------------------------------------------------------------------------------------
// Clusters: several workgroups (CUs) per chain.  A few chains of a long series leave most of
// the GPU idle while one CU streams the design matrix at its own latency-bound ~45 GB/s, so the
// phases that are independent across time -- X~'targets, the emission of the previous draw, X w
// -- are shared between `cluster` workgroups: workgroup 0 of a chain ("main") runs the whole
// iteration, the others only their share of those phases -- and the first of them also sweeps the
// NEXT iteration's regression matrix while main is in the Durbin-Koopman draw (presweep_block).
// Handshakes are per-chain counters in
// HBM (release: every thread fences, barrier, one atomic store; acquire: one thread spins, fences,
// barrier); the workgroups of a chain are placed on ONE XCD (dispatch is round-robin over the 8
// XCDs), so they share its L2.  Reductions are over FIXED segments of NT chunks (1024 steps)
// whatever the cluster size, summed in segment order: every cluster size gives the same bits.
// ------------------------------------------------------------------------------------
// `light`: every workgroup of the chain was found on the same XCD (cl_same_xcd), so stores only
// have to reach the shared L2 (the vector L1 is write-through: wait for them) and the reader only
// has to drop its L1; otherwise the release also writes the L2 back (agent scope).
__device__ __forceinline__ void cl_publish(int* flag, int value, int tid, bool light) {
  // (round 6, ADVICE) the workgroup-scope release emits no s_waitcnt on gfx950: drain this wave's
  // stores to the shared L2 explicitly, so the flag below cannot overtake them.
  if (light) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else __threadfence();
  __syncthreads();
  // (round 5) light: a RELAXED store -- the data is in the shared L2 once s_waitcnt vmcnt(0), the
  // workgroup-scope release above, has returned; a RELEASE at agent scope adds an L2 write-back
  // ("agent" spans the XCDs, each with its own L2).  Measured on cfg4: same kernel time and the same
  // WRITE_SIZE (1.9 GB per launch either way: the counter sees every write that leaves the L2 for
  // the fabric, and this part's L2 forwards the chains' workspace writes whatever the scope).
  if (tid == 0) {
    if (light) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// waits until flags[0..n) have all reached `value`
__device__ __forceinline__ void cl_wait(int* flags, int n, int value, int tid) {
  if (tid == 0) {
    for (int r = 0; r < n; ++r)
      while (__hip_atomic_load(flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value)
        __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
constexpr int CL_INTS = 160;             // handshake ints per chain, in 128-byte lines by who polls what
// line 0: flags (written once per iteration, polled by many); line 1: per-role flags; line 2: check-in
// slots (start only); line 3: arrival counters (atomic adds, never polled); line 4: the DK barrier's flag
enum ClusterFlag { CL_LATENTS = 0, CL_WEIGHTS = 1, CL_MODE = 2, CL_V = 3, CL_SCALES = 5, CL_PARTIAL = 32, CL_XW = 48,
                   CL_XCC = 64, CL_DK = 96, CL_LATCNT = 97, CL_DKFLAG = 128 };   // PARTIAL / XW / XCC + role (< 16)
// Assembling a cluster.  The handshakes below spin, so a cluster may only run when ALL its
// workgroups are resident -- which the host sizes the launch for, but cannot guarantee (another
// stream or process may hold CUs).  So the cluster is agreed on at the start, with time-outs:
// helpers check in (their XCD id + 1); main claims them one by one (compare-and-swap) and
// publishes the mode -- 2: all here, all on main's XCD (L2-local handshakes), 1: all here, mixed
// XCDs (agent-scope releases), 3: someone missing, main runs alone (every cluster size gives the
// same bits, so nothing else changes).  A helper main has not claimed after 5 ms withdraws with
// the same compare-and-swap, so neither side ever waits for a workgroup that is not running.
__device__ __forceinline__ int cl_assemble(int* csync, int role, int G, int tid) {
  __shared__ int mode_s;
  if (tid == 0) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 15) + 1;
    const long long t0 = wall_clock64();                 // 100 MHz
    int mode = 0;
    if (role > 0) {
      int* slot = csync + CL_XCC + role;
      __hip_atomic_store(slot, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        mode = __hip_atomic_load(csync + CL_MODE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode != 0) break;
        if (wall_clock64() - t0 > 500000) {
          int expect = xcc;
          if (__hip_atomic_compare_exchange_strong(slot, &expect, -1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)) {
            mode = 3;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(8);
      }
    } else {
      bool all = true, same = true;
      for (int r = 1; r < G && all; ++r) {
        int* slot = csync + CL_XCC + r;
        int v;
        while ((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 &&
               wall_clock64() - t0 < 100000)
          __builtin_amdgcn_s_sleep(8);
        int expect = v;
        if (v <= 0 || !__hip_atomic_compare_exchange_strong(slot, &expect, v + 64, __ATOMIC_RELAXED,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          all = false;
        else
          same = same && v == xcc;
      }
      mode = all ? (same ? 2 : 1) : 3;
      __hip_atomic_store(csync + CL_MODE, mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    mode_s = mode;
  }
  __syncthreads();
  return mode_s;
}

// ------------------------------------------------------------------------------------
// the persistent Gibbs kernel (same iteration structure as gibbs_kernel / the oracle's
// ci_oracle_fit_gibbs; gibbs_sampler.fit_with_gibbs_sampling called at causalimpact_lib.py:365)
// ------------------------------------------------------------------------------------
// BIGP: the build for 53+ design columns (its own instantiation: the kernels for <= 52 columns are
// what they were) -- X'X / Omega in global memory, the regression draw of ci_bigp.h by the main
// workgroup, no sweeper, wider cluster message and partial-sum rows.
template <int TR, int NS, bool BIGP = false>
__global__ __launch_bounds__(NT) void gibbs_wide_kernel(SArgs a) {
  using W = WDim<TR, NS>;
  constexpr int D = W::D, O = W::O, N1 = W::N1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const KArgs& g = a.k;
  const int T = g.T, P = g.P, Lc = a.Lc;
  const int TP = DK_CH * Lc;
  // workgroup -> (chain, role): the workgroups of one chain share an XCD (ids equal mod 8)
  const int GL = a.cluster;                   // workgroups per chain in this launch
  int chain_id = blockIdx.x, role = 0;
  if (GL > 1) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    role = slot % GL;
    chain_id = (slot / GL) * 8 + xcd;
    if (chain_id >= g.B * g.C) return;
  }
  const int series = chain_id / g.C, chain = chain_id % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  typedef typename std::conditional<BIGP, WLayout, WLayoutS>::type Lay;
  Lay lay;
  if constexpr (BIGP) lay = make_wlayout(P, D); else lay = make_wlayout_small(P, D);
  RegLds R;
  R.xtx = (double*)(smem + lay.xtx); R.omega = (double*)(smem + lay.omega);
  R.aug[0] = (double*)(smem + lay.aug0); R.aug[1] = (double*)(smem + lay.aug1);
  R.pri[0] = (double*)(smem + lay.pri0); R.pri[1] = (double*)(smem + lay.pri1);
  R.chol = (double*)(smem + lay.chol); R.bvec = (double*)(smem + lay.bvec);
  R.zv = (double*)(smem + lay.zv); R.uperm = (double*)(smem + lay.uperm);
  R.nz = (int*)(smem + lay.nz); R.perm = (int*)(smem + lay.perm); R.idx = (int*)(smem + lay.idx);
  R.w = (float*)(smem + lay.w);
  if constexpr (BIGP) {      // X'X and Omega where the setup kernel left them
    R.bvec = bp_opaque(R.bvec); R.zv = bp_opaque(R.zv); R.uperm = bp_opaque(R.uperm);
    R.nz = bp_opaque(R.nz); R.perm = bp_opaque(R.perm); R.idx = bp_opaque(R.idx);
    R.aug[0] = bp_opaque(R.aug[0]); R.aug[1] = R.aug[0]; R.pri[0] = bp_opaque(R.pri[0]); R.pri[1] = R.pri[0];
    R.xtx = const_cast<double*>(g.xtx + (size_t)series * P * P);
    R.omega = const_cast<double*>(g.omega + (size_t)series * P * P);
  }
  float* scal = (float*)(smem + lay.scal);
  float* red = (float*)(smem + lay.red);
  double* st = (double*)(smem + lay.st);
  double* gsum = (double*)(smem + lay.gsum);
  double* bpre = (double*)(smem + lay.bpre);
  const int RS = (P > 16 ? P : 16) + 4;

  // per-chain HBM workspace
  float* ws = a.ws + chain_lin * wide_workspace_floats(D, Lc);
  float* tgw = ws;                  // targets y - level - seasonal (observed steps)
  float* residw = tgw + TP;         // y - X w
  float* levw = residw + TP;
  float* slpw = levw + TP;
  float* seaw = slpw + TP;
  float* xww = seaw + TP;           // X w
  float* wsp = xww + TP;            // per-step fields of the draw: y~ -> v/F [Lc][NT], K_t -> r_{t-1} [Lc][NT][8]
  uint8_t* mskp = (uint8_t*)(wsp + (size_t)DK_NF * TP);   // mask, padded with 1
  uint8_t* cbp = mskp + TP;                               // season-change flags, padded with 0
  float* dkx = (float*)(cbp + TP);                        // exchange region of the draw (wide_dk_floats())

  const DevSeriesParams& sp = g.sp[series];       // (by reference: a copy holds ~50 scalar registers through the whole kernel)
  const DevSeasonalParams& ss = a.ssp[series];
  Rng rng{stream_key0(g.seed0, g.series_stream_base, series), stream_key1(g.seed1, g.series_stream_base, series), (uint32_t)(g.chain_offset + chain)};
  const float* yg = g.y + (size_t)series * T;
  const float* Xg = g.Xt + (size_t)series * P * T;
  const float* chol1 = a.p1_chol + (size_t)series * D * D;

  // ---- phases shared by the workgroups of a cluster -------------------------------------------
  const bool vec4 = (T & 3) == 0;            // (clusters need it; the host checks)
  const int n4 = T >> 2;                     // chunks of 4 steps
  const int nseg = (n4 + NT - 1) / NT;
  int* csync = a.csync + chain_lin * CL_INTS;
  if (GL > 1 && role > 0 && role == a.cluster_drop) return;     // (test knob: a helper that never ran)
  const int cmode = GL > 1 ? cl_assemble(csync, role, GL, tid) : 3;
  if (cmode == 3 && role > 0) return;         // the cluster did not assemble: main runs alone
  const int G = cmode == 3 ? 1 : GL;          // workgroups actually sharing this chain
  const bool light = cmode == 2;
  float* cpart = a.cpart + chain_lin * (size_t)nseg * NW * RS;
  int GS = 64;                          // (constants in the <= 52-column builds)
  if constexpr (BIGP) GS = lay.gs;
  const int CW0 = BIGP ? wide_cw0(P) : 56;
  float* cw = a.cw + chain_lin * (BIGP ? wide_cw_floats(P) : 64);
  double* cv = a.cv + chain_lin * (BIGP ? wide_cv_doubles(P) : presweep_doubles(P));
  if constexpr (BIGP) R.chol = cv + (size_t)P * P;      // (the included block's factor when it has no room in LDS)
  const int clo = (int)((long long)(TP >> 2) * role / G), chi = (int)((long long)(TP >> 2) * (role + 1) / G);
  const int n_iter = g.W + g.S;
  // The Durbin-Koopman draw runs on Gd workgroups of the cluster (ci_wide_quad.h); with sixteen
  // workgroups the tenth (role 9) sweeps the next iteration's regression matrix meanwhile, and main
  // imports it while it waits for the draw.  (Eight workgroups: all eight in the draw and no sweep
  // ahead -- 8 x 1 virtual workgroup -- measured at 32 chains against four in the draw plus a
  // sweeper, 4 x 2 virtual workgroups: the draw is the longer pole.)
  const int Gd = G >= DK_V ? DK_V : G;
  const int sweep_role = G == 16 ? 9 : -1;
  // with sixteen workgroups main is NOT a DK worker (roles 1..8 are): the draw's first phase needs the
  // disturbance scales only and runs on them while main is still in the regression block
  const int dw0 = G == 16 ? 1 : 0;
  const bool dk_worker = role >= dw0 && role < dw0 + Gd;
  const bool early_a = G == 16;
  DkSync dsy;
  dsy.cnt = csync + CL_DK; dsy.flag = csync + CL_DKFLAG; dsy.latcnt = csync + CL_LATCNT; dsy.latflag = csync + CL_LATENTS;
  dsy.Gd = Gd; dsy.epoch = 0; dsy.cluster = G > 1; dsy.light = light;
  // (the draw's per-step rows in LDS: on the DK workers that are helpers -- they never touch the
  // regression block's matrices -- when every worker runs ONE virtual workgroup (clusters of sixteen
  // or eight) and the host found room)
  const bool dk_lds = a.dk_lds != 0 && (G == 16 || (G == 8 && role > 0));   // (eight: main is a DK worker and keeps its matrices)
  DkCtx dk;
  dk.lv = (CI_LDS float*)(smem + lay.big0);
  dk.lkr = dk.lv + DK_VS * NT;
  dk.lyv = dk.lkr + Lc * 64 * 8;
  dk.T = T; dk.Lc = Lc; dk.resid = residw; dk.msk = mskp; dk.cbv = cbp;
  dk.yv = wsp; dk.kr = wsp + TP; dk.levw = levw; dk.slpw = slpw; dk.seaw = seaw; dk.xb = dkx;
  dk.chol1 = chol1; dk.a1_loc = (float)sp.init_level_loc;
  dk.p1l = (float)(sp.init_level_scale * sp.init_level_scale);
  dk.p1s = (float)(sp.init_slope_scale * sp.init_slope_scale);
  dk.p1e = (float)(ss.init_seasonal_scale * ss.init_seasonal_scale);

  const bool sweeper = !BIGP && role > 0 && role == sweep_role && P > 16;
  Prof prof;
  // phase budget: main's phases from chain 0's main workgroup, the draw's from its first DK worker
  prof.start(g.prof, g.prof != nullptr && tid == 0 && chain_id == 0 && (role == 0 || role == dw0));
  // (1) targets y - level - seasonal of segment `sa` (NT chunks of 4 steps each), their squares (`yy`)
  // and X~'targets for the columns [jlo, jhi): four wave partials per column and segment, to
  // cpart[seg][wave][.].  One wave per SIMD here: memory latency is hidden by bytes in flight, not by
  // occupancy -- the next pass's rows are requested before this pass's are reduced.  (Round 6: the
  // two-segments-per-pass form this replaces gave a role with ONE segment -- every role of a cluster of
  // sixteen at cfg4 -- an idle second lane of work whose wave reductions were half of the phase.)
  auto segment_part_sums = [&](int sa, int jlo, int jhi, bool yy) {
    const int c4 = sa * NT + tid;
    const bool ha = c4 < n4;
    const int ca = ha ? c4 : 0;
    float4 tg;
    {
      const float4 y4 = *reinterpret_cast<const float4*>(yg + 4 * ca);
      const float4 l4 = *reinterpret_cast<const float4*>(levw + 4 * ca);
      const float4 s4 = *reinterpret_cast<const float4*>(seaw + 4 * ca);
      const uint32_t mk = ha ? *reinterpret_cast<const uint32_t*>(mskp + 4 * ca) : 0xFFFFFFFFu;
      tg.x = (mk & 0xFFu) ? 0.f : y4.x - l4.x - s4.x;
      tg.y = (mk & 0xFF00u) ? 0.f : y4.y - l4.y - s4.y;
      tg.z = (mk & 0xFF0000u) ? 0.f : y4.z - l4.z - s4.z;
      tg.w = (mk & 0xFF000000u) ? 0.f : y4.w - l4.w - s4.w;
    }
    float* outa = cpart + (size_t)sa * NW * RS;
    constexpr int XS = 8;           // rows per pass (two passes' rows are live at once)
    float4 xn[XS];
    if (jlo < jhi) {
#pragma unroll
      for (int q = 0; q < XS; ++q)
        xn[q] = *reinterpret_cast<const float4*>(Xg + (size_t)(jlo + q < jhi ? jlo + q : jhi - 1) * T + 4 * ca);
    }
    for (int j0 = jlo; j0 < jhi; j0 += XS) {
      float4 xv[XS];
#pragma unroll
      for (int q = 0; q < XS; ++q) xv[q] = xn[q];
      if (j0 + XS < jhi) {
#pragma unroll
        for (int q = 0; q < XS; ++q) {
          const int j = j0 + XS + q < jhi ? j0 + XS + q : jhi - 1;
          xn[q] = *reinterpret_cast<const float4*>(Xg + (size_t)j * T + 4 * ca);
        }
      }
#pragma unroll
      for (int q = 0; q < XS; ++q) {
        const float pa = xv[q].x * tg.x + xv[q].y * tg.y + xv[q].z * tg.z + xv[q].w * tg.w;
        const float wa = wave_sum_dpp(pa);
        if (lane == 0 && j0 + q < jhi) outa[wave * RS + j0 + q] = wa;
      }
    }
    if (yy) {
      float ya = 0.f;
      ya = fmaf(tg.x, tg.x, ya); ya = fmaf(tg.y, tg.y, ya);
      ya = fmaf(tg.z, tg.z, ya); ya = fmaf(tg.w, tg.w, ya);
      const float wa = wave_sum_dpp(ya);
      if (lane == 0) outa[wave * RS + RS - 4] = wa;
    }
  };
  // The work items of one role.  A segment's columns are cut into `ns` parts when that shortens the
  // longest role (ten segments on sixteen workgroups: three parts, two items of 17 columns per role
  // instead of one of 51); who sums a (segment, column) pair never changes its value.
  auto role_segment_sums = [&]() {
    int ns = 1;
    {
      int best = 0x7fffffff;
      for (int c = 1; c <= 4; ++c) {
        const int per_role = (nseg * c + G - 1) / G;
        const int cost = per_role * ((P + c - 1) / c + 4);       // rows + what loading the targets costs
        if (cost < best) { best = cost; ns = c; }
      }
    }
    const int fs = (P + ns - 1) / ns;
    for (int w = role; w < nseg * ns; w += G) {
      const int sa = w / ns, part = w - sa * ns;
      const int jlo = part * fs, jhi = (part + 1) * fs < P ? (part + 1) * fs : P;
      segment_part_sums(sa, jlo < P ? jlo : P, jhi, part == 0);
    }
  };

  // (3) latents and posterior-predictive trajectory of iteration it - 1, chunks [lo, hi)
  auto emit_range = [&](int it, float so, int lo, int hi) {
    const int s = it - 1 - g.W;
    const size_t o = chain_lin * g.S + s;
    const size_t row = o * T;
    // 4 steps per thread: whole 16-byte accesses when the rows are 16-byte aligned (T % 4 == 0;
    // the shared T-arrays are padded to a multiple of 4)
    const int cend = hi < (T + 3) / 4 ? hi : (T + 3) / 4;
    for (int c = lo + tid; c < cend; c += NT) {
      float zp[4];
      normals4(site_call(rng, (uint32_t)(it - 1), SITE_PRED, 0, (uint32_t)c), zp);
      if (vec4) {
        const float4 lv = *reinterpret_cast<const float4*>(levw + 4 * c);
        const float4 sv = *reinterpret_cast<const float4*>(seaw + 4 * c);
        const float4 xv = *reinterpret_cast<const float4*>(xww + 4 * c);
        const float4 loc = make_float4(lv.x + sv.x + xv.x, lv.y + sv.y + xv.y, lv.z + sv.z + xv.z,
                                       lv.w + sv.w + xv.w);
        const size_t at = row + 4 * (size_t)c;
        // (the draws are written once and never read by the kernel: streamed past the L2's
        // retention, which the chain's design matrix and workspace need)
        auto put = [](float* p, float x, float y, float z, float w_) {
          __builtin_nontemporal_store(ci_f4v{x, y, z, w_}, reinterpret_cast<ci_f4v*>(p));
        };
        if (g.out_level) put(g.out_level + at, lv.x, lv.y, lv.z, lv.w);
        if (g.out_slope && TR == 2) {
          const float4 sl4 = *reinterpret_cast<const float4*>(slpw + 4 * c);
          put(g.out_slope + at, sl4.x, sl4.y, sl4.z, sl4.w);
        }
        if (a.out_seasonal) put(a.out_seasonal + at, sv.x, sv.y, sv.z, sv.w);
        if (g.out_traj)
          put(g.out_traj + at, fmaf(so, zp[0], loc.x), fmaf(so, zp[1], loc.y), fmaf(so, zp[2], loc.z),
              fmaf(so, zp[3], loc.w));
        if (g.out_pred_mean) {
          float4* pm = reinterpret_cast<float4*>(g.out_pred_mean + chain_lin * T + 4 * c);
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          if (s != 0) acc = *pm;                 // running sum, scaled at the end
          *pm = make_float4(acc.x + loc.x, acc.y + loc.y, acc.z + loc.z, acc.w + loc.w);
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = 4 * c + q;
        if (t < T) {
          const float lv = levw[t], sv = seaw[t];
          const float loc = lv + sv + xww[t];
          if (g.out_level) g.out_level[row + t] = lv;
          if (g.out_slope && TR == 2) g.out_slope[row + t] = slpw[t];
          if (a.out_seasonal) a.out_seasonal[row + t] = sv;
          if (g.out_traj) g.out_traj[row + t] = fmaf(so, zp[q], loc);
          if (g.out_pred_mean) {
            float* pm = g.out_pred_mean + chain_lin * T + t;   // running sum, scaled at the end
            *pm = (s == 0 ? 0.f : *pm) + loc;
          }
        }
      }
    }
  };

  // (4) X w and the residual of chunks [lo, hi) (T % 4 == 0): only the INCLUDED features' rows are
  // streamed (a zero weight contributes an exact zero)
  auto xw_range_small = [&](int lo, int hi) {
    const unsigned long long included = __ballot(lane < P && R.w[lane < P ? lane : 0] != 0.f);
    for (int c4 = lo + tid; c4 < hi; c4 += NT) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f), yv = s;
      if (4 * c4 < T) {
        for (unsigned long long todo = included; todo != 0ull;) {
          float4 xv[XR];
          float wj[XR];
#pragma unroll
          for (int q = 0; q < XR; ++q) {
            const bool have = todo != 0ull;
            const int j = have ? __ffsll((long long)todo) - 1 : 0;
            todo &= todo - 1ull;
            wj[q] = have ? R.w[j] : 0.f;
            xv[q] = *reinterpret_cast<const float4*>(Xg + (size_t)j * T + 4 * c4);
          }
#pragma unroll
          for (int q = 0; q < XR; ++q) {
            s.x = fmaf(xv[q].x, wj[q], s.x); s.y = fmaf(xv[q].y, wj[q], s.y);
            s.z = fmaf(xv[q].z, wj[q], s.z); s.w = fmaf(xv[q].w, wj[q], s.w);
          }
        }
        const float4 y4 = *reinterpret_cast<const float4*>(yg + 4 * c4);
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(mskp + 4 * c4);
        yv.x = (mk & 0xFFu) ? 0.f : y4.x; yv.y = (mk & 0xFF00u) ? 0.f : y4.y;
        yv.z = (mk & 0xFF0000u) ? 0.f : y4.z; yv.w = (mk & 0xFF000000u) ? 0.f : y4.w;
      }
      *reinterpret_cast<float4*>(xww + 4 * c4) = s;
      *reinterpret_cast<float4*>(residw + 4 * c4) = make_float4(yv.x - s.x, yv.y - s.y, yv.z - s.z, yv.w - s.w);
    }
  };

  // ... BIGP builds: the included features in blocks of 64 columns
  auto xw_range_big = [&](int lo, int hi) {
    // (the included features in blocks of 64 columns: BIGP builds; one mask otherwise)
    constexpr int NBLK = BIGP ? 8 : 1;
    unsigned long long included[NBLK];
    if constexpr (BIGP) {
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int j = 64 * b + lane;
        included[b] = __ballot(j < P && R.w[j < P ? j : 0] != 0.f);
      }
    } else {
      included[0] = __ballot(lane < P && R.w[lane < P ? lane : 0] != 0.f);
    }
    for (int c4 = lo + tid; c4 < hi; c4 += NT) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f), yv = s;
      if (4 * c4 < T) {
        auto block = [&](unsigned long long mask, int base) __attribute__((always_inline)) {
          for (unsigned long long todo = mask; todo != 0ull;) {
            float4 xv[XR];
            float wj[XR];
#pragma unroll
            for (int q = 0; q < XR; ++q) {
              const bool have = todo != 0ull;
              const int j = have ? base + __ffsll((long long)todo) - 1 : 0;
              todo &= todo - 1ull;
              wj[q] = have ? R.w[j] : 0.f;
              xv[q] = *reinterpret_cast<const float4*>(Xg + (size_t)j * T + 4 * c4);
            }
#pragma unroll
            for (int q = 0; q < XR; ++q) {
              s.x = fmaf(xv[q].x, wj[q], s.x); s.y = fmaf(xv[q].y, wj[q], s.y);
              s.z = fmaf(xv[q].z, wj[q], s.z); s.w = fmaf(xv[q].w, wj[q], s.w);
            }
          }
        };
        if constexpr (BIGP) {
          for (int b = 0; b < NBLK; ++b) block(included[b], 64 * b);
        } else {
          block(included[0], 0);
        }
        const float4 y4 = *reinterpret_cast<const float4*>(yg + 4 * c4);
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(mskp + 4 * c4);
        yv.x = (mk & 0xFFu) ? 0.f : y4.x; yv.y = (mk & 0xFF00u) ? 0.f : y4.y;
        yv.z = (mk & 0xFF0000u) ? 0.f : y4.z; yv.w = (mk & 0xFF000000u) ? 0.f : y4.w;
      }
      *reinterpret_cast<float4*>(xww + 4 * c4) = s;
      *reinterpret_cast<float4*>(residw + 4 * c4) = make_float4(yv.x - s.x, yv.y - s.y, yv.z - s.z, yv.w - s.w);
    }
  };

  auto xw_range = [&](int lo, int hi) {
    if constexpr (BIGP) xw_range_big(lo, hi);
    else xw_range_small(lo, hi);
  };

  // ---- helper workgroup: its share of phases (1), (3), (4); the sweeper (sixteen workgroups: role 9)
  // also prepares the next iteration's regression matrix; the DK workers take part in the draw below
  auto helper_iteration = [&](int it) {
    cl_wait(csync + CL_LATENTS, 1, it + 1, tid);
    prof.tick(19);
    role_segment_sums();
    cl_publish(csync + CL_PARTIAL + role, it + 1, tid, light);
    if constexpr (!BIGP) prof.tick(8);       // (DK worker's budget) its segments of X'targets
    if (early_a && dk_worker && it < n_iter) {
      // the prior simulation of this iteration's draw, as soon as main has drawn the scales
      cl_wait(csync + CL_SCALES, 1, it + 1, tid);
      WideScal se;
      se.so = 0.f; se.H = 0.f;
      se.sl = cw[CW0 + 3]; se.ql = se.sl * se.sl;
      se.ss = cw[CW0 + 4]; se.qs = se.ss * se.ss;
      se.sdn = cw[CW0 + 5] * (1.0f / (float)NS); se.qd = se.sdn * se.sdn;
      if (dk_lds) wide_dk_quad<TR, NS, true>(se, dk, rng, (uint32_t)it, role - dw0, dsy, tid, prof, true, false);
      else wide_dk_quad<TR, NS, false>(se, dk, rng, (uint32_t)it, role - dw0, dsy, tid, prof, true, false);
    }
    cl_wait(csync + CL_WEIGHTS, 1, it + 1, tid);
    if constexpr (!BIGP) prof.tick(14);      // ... waiting for main's serial section (scales, regression draw; BIGP builds: slots 11-15 are the draw's)
    if (tid < P) R.w[tid] = cw[tid];
    const float so = cw[P];
    __syncthreads();
    if (it > g.W) emit_range(it, so, clo, chi);
    if constexpr (!BIGP) prof.tick(15);      // ... its share of the emission
    if (it < n_iter) xw_range(clo, chi);
    cl_publish(csync + CL_XW + role, it + 1, tid, light);
    if constexpr (!BIGP) prof.tick(13);      // ... its share of X w
    if (sweeper && it + 1 < n_iter) {
      // iteration it + 1 sweeps Omega s2 + X'X on the features that are in now, s2 this
      // iteration's observation-noise variance: both are in the message just received
      const double so_d = *reinterpret_cast<const double*>(cw + CW0);
      const bool all_in = sp.nonzero_prob >= 1.0;
      const unsigned long long nzmask = __ballot(lane < P && (all_in || R.w[lane < P ? lane : 0] != 0.f));
      presweep_block(R, P, so_d * so_d, nzmask, false, tid);
      presweep_export(R, P, nzmask, cv, tid);
      cl_publish(csync + CL_V, it + 1, tid, light);
    }
  };

  double n_changes = 0.0;
  double obs_scale = sp.obs_scale0, level_scale = sp.level_scale0, slope_scale = sp.slope_scale0;
  double drift = ss.drift_scale0[0];
  float ssl = 0.f, sss = 0.f, ssd = 0.f;
  PriorCarry pc;
  pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;
  if (role > 0) {
    if (sweeper) {
      for (int e = tid; e < P * P; e += NT) {
        R.xtx[e] = g.xtx[(size_t)series * P * P + e];
        R.omega[e] = g.omega[(size_t)series * P * P + e];
      }
      __syncthreads();
    }
  } else {
  float nch = 0.f;
  for (int t = tid; t < TP; t += NT) {
    const bool in = t < T;
    mskp[t] = in ? (g.mask[(size_t)series * T + t] != 0 ? 1 : 0) : 1;
    const uint8_t c = in ? (a.season_change[t] != 0 ? 1 : 0) : 0;
    cbp[t] = c;
    if (t + 1 < T && c) nch += 1.f;
    levw[t] = 0.f; slpw[t] = 0.f; seaw[t] = 0.f; xww[t] = 0.f; residw[t] = 0.f; tgw[t] = 0.f;
  }
  if constexpr (!BIGP)
    for (int e = tid; e < P * P; e += NT) {
      R.xtx[e] = g.xtx[(size_t)series * P * P + e];
      R.omega[e] = g.omega[(size_t)series * P * P + e];
    }
  if (tid < 16 || tid < P) R.w[tid] = 0.f;
  {
    const float s = wave_sum_dpp(nch);
    if (lane == 0) red[wave] = s;
  }
  __syncthreads();
  n_changes = (double)(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  if (G > 1) cl_publish(csync + CL_LATENTS, 1, tid, light);      // masks and zeroed latents are in place
  }

  for (int it = 0; it <= n_iter; ++it) {
    WideScal sc;
    if (role > 0) {
      helper_iteration(it);
      if (it == n_iter) break;
      if (!dk_worker) continue;
      // a DK worker: every share of X w / the residual is in, the scales came with the weights
      cl_wait(csync + CL_XW, G, it + 1, tid);
      sc.so = cw[CW0 + 2]; sc.H = sc.so * sc.so;
      sc.sl = cw[CW0 + 3]; sc.ql = sc.sl * sc.sl;
      sc.ss = cw[CW0 + 4]; sc.qs = sc.ss * sc.ss;
      sc.sdn = cw[CW0 + 5] * (1.0f / (float)NS); sc.qd = sc.sdn * sc.sdn;
    } else {
    // ---- (1) targets, y'y, X~'targets (time interleaved over threads: coalesced)
    if (it > 0) dk_stats<TR, NS>(dkx, cbp, T, Lc, tid, ssl, sss, ssd);
    if (vec4) {
      role_segment_sums();
      const float s1 = wave_sum_dpp(ssl), s2 = wave_sum_dpp(sss), s3 = wave_sum_dpp(ssd);
      if (lane == 0) {
        red[wave * RS + RS - 3] = s1;
        red[wave * RS + RS - 2] = s2;
        red[wave * RS + RS - 1] = s3;
      }
      if (G > 1) cl_wait(csync + CL_PARTIAL + 1, G - 1, it + 1, tid);
    } else {
      float yty = 0.f;
      for (int t = tid; t < T; t += NT) {
        float tg = 0.f;
        if (!mskp[t]) tg = yg[t] - levw[t] - seaw[t];
        tgw[t] = tg;
        yty = fmaf(tg, tg, yty);
      }
      __syncthreads();      // tgw is read back through another thread mapping below
      for (int j0 = 0; j0 < P; j0 += XR) {
        float acc[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) acc[q] = 0.f;
        {
          for (int tb = tid; tb < T; tb += 4 * NT) {
            float tg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int t = tb + u * NT;
              tg[u] = t < T ? tgw[t] : 0.f;
            }
            float xv[XR][4];
#pragma unroll
            for (int q = 0; q < XR; ++q) {
              const int j = j0 + q < P ? j0 + q : P - 1;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int t = tb + u * NT;
                xv[q][u] = Xg[(size_t)j * T + (t < T ? t : T - 1)];
              }
            }
#pragma unroll
            for (int q = 0; q < XR; ++q)
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[q] = fmaf(xv[q][u], tg[u], acc[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < XR; ++q) {
          const float s = wave_sum_dpp(acc[q]);
          if (lane == 0 && j0 + q < P) red[wave * RS + j0 + q] = s;
        }
      }
      const float s0 = wave_sum_dpp(yty), s1 = wave_sum_dpp(ssl), s2 = wave_sum_dpp(sss),
                  s3 = wave_sum_dpp(ssd);
      if (lane == 0) {
        red[wave * RS + RS - 4] = s0;
        red[wave * RS + RS - 3] = s1;
        red[wave * RS + RS - 2] = s2;
        red[wave * RS + RS - 1] = s3;
      }
    }
    __syncthreads();      // (G == 1: the partials are this workgroup's own stores, same CU)
    if (vec4) {
      // X~'targets and y'y: wave w adds its quarter of the (segment, wave) partials in order; the
      // serial section adds the four quarters -- a fixed tree, whatever the cluster size
      const int ne = nseg * NW, e0 = ne * wave / NW, e1 = ne * (wave + 1) / NW;
      if constexpr (BIGP) {
        for (int j = lane; j <= P; j += 64) {
          const int src = j < P ? j : RS - 4;
          double sq = 0.0;
          for (int e = e0; e < e1; ++e) sq += (double)cpart[(size_t)e * RS + src];
          gsum[wave * GS + j] = sq;
        }
      } else if (lane <= P) {
        const int src = lane < P ? lane : RS - 4;
        double sq = 0.0;
        for (int e = e0; e < e1; ++e) sq += (double)cpart[(size_t)e * RS + src];
        gsum[wave * 64 + lane] = sq;
      }
      __syncthreads();
    }
    prof.tick(0);

    // ---- (2) serial section (wave 0): scales of iteration it-1, regression draw of iteration it
    if (wave == 0) {
      for (int j = lane; j < P + 4; j += 64) {
        const int src = j < P ? j : RS - 4 + (j - P);
        double s = 0.0;
        if (vec4 && j <= P) {
#pragma unroll
          for (int w = 0; w < NW; ++w) s += gsum[w * GS + j];
        } else {
#pragma unroll
          for (int w = 0; w < NW; ++w) s += (double)red[w * RS + src];
        }
        R.bvec[j] = s;
      }
      wave_sync();
      double emit_obs = obs_scale;
      if (it > 0) {
        const uint32_t pit = (uint32_t)(it - 1);
        // (the gamma variates of these draws were drawn at the end of the previous pass: draw_ahead)
        level_scale = scale_from_gamma(sp.level_scale, sp.level_ub, R.bvec[P + 1], st[2]);
        if constexpr (TR == 2)
          slope_scale = scale_from_gamma(sp.slope_scale, sp.slope_ub, R.bvec[P + 2], st[3]);
        {
          const double gk = st[4];
          const double sd = (double)__fsqrt_rn((float)((ss.drift_scale + 0.5 * R.bvec[P + 3]) * fast_rcp(gk)));
          drift = sd < ss.drift_ub ? sd : ss.drift_ub;
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
            if (g.out_slope_scale) g.out_slope_scale[o] = (float)(TR == 2 ? slope_scale : 0.0);
            if (a.out_drift) a.out_drift[o] = (float)drift;
          }
          if (g.out_weights)
            for (int j = lane; j < P; j += 64) g.out_weights[o * P + j] = R.w[j];
        }
      }
      if (early_a && it < n_iter && lane == 0) {
        // the DK workers start the draw's prior simulation on these while the regression is drawn
        cw[CW0 + 3] = (float)level_scale; cw[CW0 + 4] = (float)slope_scale; cw[CW0 + 5] = (float)drift;
        if (light) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(csync + CL_SCALES, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_store(csync + CL_SCALES, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (P > 0 && it < n_iter) {
        const double g_obs = it > 0 ? st[5]
                                    : gamma_wave(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
        if (P <= 16) {
          obs_scale = spike_slab_draw_regs(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, pc);
        } else if (lane == 0) {       // drawn by the whole workgroup below
          st[0] = obs_scale;
          st[1] = g_obs;
        }
      }
      if (lane == 0) {
        scal[0] = (float)obs_scale;
        scal[1] = (float)emit_obs;
        scal[2] = (float)level_scale;
        scal[3] = (float)slope_scale;
        scal[4] = (float)drift;
      }
    }
    __syncthreads();
    if constexpr (BIGP) {
      if (it < n_iter) {
        // 53+ columns: spike_slab_draw_big_wg (ci_bigp.h) by this workgroup -- the packed swept matrix,
        // its (i, j) table and the row buffers in LDS, the swept prior block too when it fits
        typedef CI_GLB double* GD;
        CI_LDS double* Al = (CI_LDS double*)(smem + lay.aug0);
        CI_LDS unsigned* ijt = (CI_LDS unsigned*)(smem + lay.ijtab);
        CI_LDS double* trow = (CI_LDS double*)(smem + lay.trow);
        if (lay.pm_lds)
          obs_scale = spike_slab_draw_big_wg<NT>(R, Al, (CI_LDS double*)(smem + lay.pri0), trow, ijt, R.w, P, sp, st[0],
                                                 st[1], rng, (uint32_t)it, tid, it == 0, prof, 9, 11);
        else
          obs_scale = spike_slab_draw_big_wg<NT>(R, Al, (GD)cv, trow, ijt, R.w, P, sp, st[0], st[1], rng,
                                                 (uint32_t)it, tid, it == 0, prof, 9, 11);
        if (tid == 0) scal[0] = (float)obs_scale;
        __syncthreads();
      }
    } else if (P > 16 && it < n_iter) {
      // P > 16: the regression draw with its (P+1)^2 sweeps spread over all four waves
      prof.tick(9);
      // a helper of the cluster swept the matrix, and main copied it into LDS while it waited for
      // the previous draw (end of the loop body)
      const bool prepared = sweep_role > 0 && it > 0;
      obs_scale = spike_slab_draw_block(R, P, sp, st[0], st[1], rng, (uint32_t)it, tid, it == 0, &prof,
                                        true, prepared ? cv : nullptr, 4, it > 0 ? bpre : nullptr, prepared);
      if (tid == 0) scal[0] = (float)obs_scale;
      __syncthreads();
      prof.tick(10);
    }
    prof.tick(1);

    // the cluster's helpers take their share of (3) and (4) from here
    if (G > 1) {
      if (tid < P) cw[tid] = R.w[tid];
      if (tid == 0) {
        cw[P] = scal[1];
        *reinterpret_cast<double*>(cw + CW0) = obs_scale;
        cw[CW0 + 2] = scal[0]; cw[CW0 + 3] = scal[2]; cw[CW0 + 4] = scal[3]; cw[CW0 + 5] = scal[4];    // the DK workers' scales
      }
      cl_publish(csync + CL_WEIGHTS, it + 1, tid, light);
    }
    // ---- (3) emit iteration it-1: latents and the posterior-predictive trajectory
    if (it > g.W) emit_range(it, scal[1], clo, chi);
    prof.tick(2);
    if (it == n_iter) {
      if (G > 1) cl_wait(csync + CL_XW + 1, G - 1, it + 1, tid);
      break;
    }
    __syncthreads();     // (3) reads xww through a different thread mapping than (4) writes it

    // ---- (4) X w and the residual
    if (vec4) {
      xw_range(clo, chi);
    } else {
      for (int tb = tid; tb < TP; tb += 4 * NT) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < P; j0 += XR) {
          float xv[XR][4], wj[XR];
#pragma unroll
          for (int q = 0; q < XR; ++q) {
            const int j = j0 + q < P ? j0 + q : P - 1;
            wj[q] = j0 + q < P ? R.w[j] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int t = tb + u * NT;
              xv[q][u] = Xg[(size_t)j * T + (t < T ? t : T - 1)];
            }
          }
#pragma unroll
          for (int q = 0; q < XR; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = fmaf(xv[q][u], wj[q], s[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = tb + u * NT;
          if (t < TP) {
            const float yv = (t < T && !mskp[t]) ? yg[t] : 0.f;
            const float sv = t < T ? s[u] : 0.f;
            xww[t] = sv;
            residw[t] = yv - sv;
          }
        }
      }
    }
    if (G > 1) {
      cl_publish(csync + CL_XW, it + 1, tid, light);        // this workgroup's share of the residual
      cl_wait(csync + CL_XW + 1, G - 1, it + 1, tid);
    }
    sc.so = scal[0]; sc.H = sc.so * sc.so;
    sc.sl = scal[2]; sc.ql = sc.sl * sc.sl;
    sc.ss = scal[3]; sc.qs = sc.ss * sc.ss;
    sc.sdn = scal[4] * (1.0f / (float)NS); sc.qd = sc.sdn * sc.sdn;
    __syncthreads();
    prof.tick(3);
    }
    // the Durbin-Koopman draw, by the cluster's DK workers together; its last worker raises the
    // latents flag to it + 2
    if (dk_worker) {
      if (dk_lds) wide_dk_quad<TR, NS, true>(sc, dk, rng, (uint32_t)it, role - dw0, dsy, tid, prof, !early_a, true);
      else wide_dk_quad<TR, NS, false>(sc, dk, rng, (uint32_t)it, role - dw0, dsy, tid, prof, !early_a, true);
    }
    // Main, before it waits for the latents (with sixteen workgroups it is idle through the whole
    // draw): everything of the next serial section that needs no data -- the gamma variates of this
    // iteration's scale draws and of the next sigma^2_obs, the regression block's visiting order,
    // flip uniforms and weight normals.  Same sites, same counters: the same numbers as drawn in place.
    if (role == 0) {
      if (wave == 0) {
        const uint32_t pit = (uint32_t)it;
        const double g_lev = gamma_wave(sp.level_conc + 0.5 * (double)(T - 1), rng, pit, SITE_LEVEL_SCALE, 0, lane);
        double g_slp = 1.0;
        if constexpr (TR == 2)
          g_slp = gamma_wave(sp.slope_conc + 0.5 * (double)(T - 1), rng, pit, SITE_SLOPE_SCALE, 0, lane);
        const double g_drf = gamma_wave(ss.drift_conc + 0.5 * n_changes, rng, pit, SITE_DRIFT_SCALE, 0, lane);
        const double g_obn = gamma_wave(sp.obs_conc + 0.5 * sp.n_obs, rng, pit + 1u, SITE_OBSVAR, 0, lane);
        if (lane == 0) { st[2] = g_lev; st[3] = g_slp; st[4] = g_drf; st[5] = g_obn; }
      } else if (!BIGP && wave == 1 && P > 16) {
        block_randoms_store(block_randoms(rng, (uint32_t)it + 1u, P, lane), bpre, lane);
      }
    }
    if constexpr (!BIGP) {
      if (role == 0 && sweep_role > 0 && P > 16 && it + 1 < n_iter) {
        // ... and the matrix the sweeper prepared for the next regression draw (36 KB through L2 at
        // P = 51: 7k cycles of the serial section when copied there)
        cl_wait(csync + CL_V, 1, it + 1, tid);
        const bool all_in = sp.nonzero_prob >= 1.0;
        const unsigned long long nzmask = __ballot(lane < P && (all_in || R.w[lane < P ? lane : 0] != 0.f));
        presweep_import<NT>(R, P, nzmask, cv, tid);
      }
    }
    if (role == 0 && G > 1) cl_wait(csync + CL_LATENTS, 1, it + 2, tid);
  }
  if (role > 0) return;
  __syncthreads();     // the running sums were accumulated through the emission's thread mapping
  if (g.out_pred_mean) {
    const float inv = 1.0f / (float)(g.S > 0 ? g.S : 1);
    for (int t = tid; t < T; t += NT) g.out_pred_mean[chain_lin * T + t] *= inv;
  }
}

}  // namespace ci
