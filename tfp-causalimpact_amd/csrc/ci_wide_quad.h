// ci_wide_quad.h -- the Durbin-Koopman draw of the trend + seasonal kernel (ci_wide.h), spread over
// the workgroups of the chain's cluster with every chunk of the series carried by a QUAD of lanes.
//
// Round 6.  Rounds 2-5 ran this draw on ONE workgroup, one lane per chunk: 437k of the iteration's
// 556k cycles at cfg4, VALU-issue bound on that CU's four SIMDs (a combine of two 119-float
// filtering elements is ~2.8k dependent FMAs on one lane; the per-step passes carry 7 x 7 matrices
// through 40 steps per lane) while the cluster's other CUs waited.  Now:
//   * the grid is unchanged -- 512 chunks of Lc = ceil(T / 512) steps (rounded to 4), fixed by T alone --
//     but a chunk belongs to the four lanes of a quad: matrices split by columns, operands from the
//     other lanes through DPP broadcasts (ci_quad.h).  1024 lanes = four VIRTUAL workgroups of 64
//     quads; the cluster's first Gd = 4, 2 or 1 workgroups ("DK workers") take 1, 2 or 4 of them
//     each, one after the other, with the little state that crosses a hand-over parked in L2.
//     Which workgroup runs a virtual workgroup changes nothing in the arithmetic: every cluster size
//     gives the same bits.
//   * every scan is two-level with ONE hand-over between workgroups: Kogge-Stone over the 16 quads of
//     a wavefront (ds_bpermute), the 16 wave totals of the chain published to L2, a cluster barrier,
//     then every wavefront scans the 16 totals again on its own (4 more levels) and picks its prefix.
//     9 combine latencies of ~3k cycles instead of 9-10 of 12k.
//   * the per-step passes cost ~1/2.5 of the one-lane versions: the transition acts on the rows of a
//     column-split matrix (inside a lane), rank-one updates take their vectors from DPP broadcasts,
//     only the congruence T C T' moves columns between lanes; the random numbers of a 4-step block
//     are drawn once per quad (one site per lane) instead of once per lane.
//   * per-step workspace: 9 floats per step (y~ -> v/F, K_t -> r_{t-1} in place) instead of 16.
// Same passes, same formulas, same random-number sites as wide_dk_draw (oracle: ci_oracle_dk_draw).
#pragma once
#include "ci_quad.h"
// (included from the middle of ci_wide.h: WDim, WideScal, WPElem, the DK_* sizes are defined there)

namespace ci {

struct DkSync {
  int* cnt;            // arrival counter of the DK workers (monotone; nobody polls it)
  int* flag;           // epoch of the last completed barrier, written by its last arriver (own cache line)
  int* latcnt;         // arrivals at the END of a draw
  int* latflag;        // the cluster's "latents of iteration it are in place" flag (CL_LATENTS)
  int Gd, epoch;
  bool cluster, light;
};
// everything this workgroup wrote is in L2 (one XCD) / written back (several) before tid 0 arrives
__device__ __forceinline__ void dk_release(const DkSync& s) {
  if (s.light) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    __threadfence();
  }
  __syncthreads();
}
// Barrier of the chain's DK workers with release / acquire of everything written before it.  An
// arrival is one atomic add whose RESULT tells the last arriver, and only that workgroup writes the
// flag the others poll -- on its own cache line: polling the arrival counter itself made the adds
// queue behind the polls (measured: 5.7k cycles for 8 workgroups, 10.1k for 16, empty barrier).
__device__ __forceinline__ void dk_barrier(DkSync& s, int tid) {
  if (!s.cluster) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return;
  }
  ++s.epoch;
  dk_release(s);
  if (tid == 0) {
    const int old = s.light ? __hip_atomic_fetch_add(s.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : __hip_atomic_fetch_add(s.cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == s.epoch * s.Gd) {
      if (s.light) __hip_atomic_store(s.flag, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(s.flag, s.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(s.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s.epoch)
        __builtin_amdgcn_s_sleep(1);
    }
    // Acquire: the CU's vector L1 drops its stale lines.  ONE wave's invalidate serves the whole
    // workgroup (the L1 belongs to the CU); the fence on every wave cost 1.5k cycles more per barrier
    // at 8 workgroups, 3k at 16 (tools/bench_cluster_barrier2.hip, with a stale-read check).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}
// End of a draw: arrive, and the last worker raises the cluster's latents flag to `value`.  Nobody
// waits here -- every workgroup of the cluster waits for that flag where it next needs the latents.
__device__ __forceinline__ void dk_finish(DkSync& s, int value, int done_draws, int tid) {
  if (!s.cluster) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return;
  }
  dk_release(s);
  if (tid == 0) {
    const int old = s.light ? __hip_atomic_fetch_add(s.latcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : __hip_atomic_fetch_add(s.latcnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == done_draws * s.Gd) {
      if (s.light) __hip_atomic_store(s.latflag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(s.latflag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- the transition on a d-vector held in registers (w_apply / w_apply_t of ci_wide.h on arrays) --
template <int TR, int NS, class E> __device__ __forceinline__ void qw_apply(E (&x)[TR + NS - 1], bool ch) {
  constexpr int O = TR, N1 = NS - 1;
  if constexpr (TR == 2) x[0] += x[1];
  if (ch) {
    E s = x[O];
#pragma unroll
    for (int i = 1; i < N1; ++i) s += x[O + i];
#pragma unroll
    for (int i = 0; i + 1 < N1; ++i) x[O + i] = x[O + i + 1];
    x[O + N1 - 1] = -s;
  }
}
template <int TR, int NS, class E> __device__ __forceinline__ void qw_apply_t(E (&x)[TR + NS - 1], bool ch) {
  constexpr int O = TR, N1 = NS - 1;
  if constexpr (TR == 2) x[1] += x[0];
  if (ch) {
    const E last = x[O + N1 - 1];
#pragma unroll
    for (int j = N1 - 1; j >= 1; --j) x[O + j] = x[O + j - 1] - last;
    x[O] = -last;
  }
}

// per-lane constants of the covariance prediction
template <int D> struct QScal {
  static constexpr int H = (D + 3) / 4;
  typename QPk<H>::T qd_own;   // qd where the own column is a seasonal one
  typename QPk<H>::T ql_own, qs_own;   // ql at (row 0, column 0), qs at (row 1, column 1): on their owners' first slot
  bool keep0;              // slot 0 holds a trend column (does not move in the seasonal shift)
  bool inject;             // this lane's last slot is "column D": receives minus the row sums before the shift
  bool last_own;           // (d % 4 == 0) this lane's last slot is column D - 1
};
template <int TR, int NS>
__device__ __forceinline__ QScal<TR + NS - 1> make_qscal(const WideScal& sc, int q) {
  constexpr int D = TR + NS - 1, O = TR, H = (D + 3) / 4;
  QScal<D> s;
  typedef typename QPk<H>::T PT;
  s.ql_own = qp_make<PT>([&](int h) { return (h == 0 && q == 0) ? sc.ql : 0.f; });
  s.qs_own = qp_make<PT>([&](int h) { return (TR == 2 && h == 0 && q == 1) ? sc.qs : 0.f; });
  s.qd_own = qp_make<PT>([&](int h) { return (q + 4 * h >= O && q + 4 * h < D) ? sc.qd : 0.f; });
  s.keep0 = q < O;
  s.inject = (D % 4 != 0) && q == (D & 3);
  s.last_own = (D % 4 == 0) && q == 3;
  return s;
}

// C <- T C T' + Q_t on a column-split symmetric matrix (oracle: propagate_cov; w_cov_predict_sym).
// Rows: the transition on every own column (in registers).  Columns: the trend pair through one
// broadcast; the seasonal shift "column j <- column j + 1, last <- minus the row sums" moves every
// slot one lane down the quad (lane 3 takes lane 0's NEXT slot); the row sums of T C are T applied to
// the row sums of C, and those are the own columns' sums over the seasonal rows (symmetry).
template <int TR, int NS>
__device__ __forceinline__ void qc_predict(QMat<TR + NS - 1>& C, bool ch, const QScal<TR + NS - 1>& qs,
                                           int q) {
  constexpr int D = TR + NS - 1, O = TR, H = (D + 3) / 4;
  typedef typename QPk<H>::T PT;
  float rs[D];
  if (ch) {
    QVec<D> cs;
    cs.v = C.r[O];
#pragma unroll
    for (int r = O + 1; r < D; ++r) cs.v += C.r[r];
    q_rep(cs, rs);
    qw_apply<TR, NS>(rs, true);
  }
  qw_apply<TR, NS>(C.r, ch);
  if constexpr (TR == 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float t = q_bc(qp_get(C.r[i], 0), 1);
      qp_set(C.r[i], 0, qp_get(C.r[i], 0) + ((q == 0) ? t : 0.f));
    }
  }
  if (ch) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float v[H + 1], n[H + 1];
#pragma unroll
      for (int h = 0; h < H; ++h) v[h] = qp_get(C.r[i], h);
      if constexpr (D % 4 != 0) v[H - 1] = qs.inject ? -rs[i] : v[H - 1];
#pragma unroll
      for (int h = 0; h < H; ++h) n[h] = q_next(v[h]);
      n[H] = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float take = (q == 3) ? n[h + 1] : n[h];
        if constexpr (D % 4 == 0)
          if (h == H - 1) take = qs.last_own ? -rs[i] : take;
        qp_set(C.r[i], h, (h == 0 && qs.keep0) ? qp_get(C.r[i], h) : take);
      }
    }
#pragma unroll
    for (int i = O; i < D; ++i) C.r[i] += qs.qd_own;
  }
  C.r[0] += qs.ql_own;
  if constexpr (TR == 2) C.r[1] += qs.qs_own;
}

// ---- everything one chain's draw needs ------------------------------------------------------------
struct DkCtx {
  int T, Lc;
  const float* resid;
  const uint8_t* msk;
  const uint8_t* cbv;
  float* kr;           // [Lc][256 chunks][8]: K_t, then r_{t-1} in place
  float* yv;           // [Lc][256 chunks]:    y~_t, then v_t / F_t in place
  float* levw;
  float* slpw;
  float* seaw;
  float* xb;           // exchange region (wide_dk_floats())
  CI_LDS float* lv;    // LW builds: what a lane parks between the phases [DK_EF][256 lanes],
  CI_LDS float* lkr;   //   K_t, then r_{t-1} in place [Lc][64 chunks of this workgroup][8],
  CI_LDS float* lyv;   //   y~_t, then v_t / F_t in place [Lc][64]
  const float* chol1;  // lower Cholesky factor of the prior covariance of x_0 (d x d)
  float a1_loc;        // prior mean of the level
  float p1l, p1s, p1e; // prior variances: level, slope, seasonal effects
};

// One Durbin-Koopman draw.  Called by the chain's DK workers (role 0 .. Gd-1) with the same
// arguments (`role` = index among them); contains 3 dk_barrier()s and ends with dk_finish().
// do_a / do_rest: phase A (prior simulation: needs the disturbance scales only, so a cluster whose
// main workgroup is not a DK worker runs it while the regression is still being drawn) and phases
// B-D; the hand-over between them is in L2 either way.  Leaves the latents in levw / slpw / seaw and, per chunk, the
// sums of squared increments + first / last state in the exchange region (dk_stats reads them).
// LW: the per-step rows and the parked state in this workgroup's LDS (one virtual workgroup per
// worker: every lane only reads back what its own quad wrote) instead of the chain's workspace in
// L2 / HBM -- same arithmetic, same bits.
template <int TR, int NS, bool LW, class P>
__device__ __forceinline__ void wide_dk_quad(const WideScal& sc, const DkCtx& x, const Rng& rng,
                                             uint32_t iter, int role, DkSync& sy, int tid, P& prof,
                                             bool do_a, bool do_rest) {
  using W = WDim<TR, NS>;
  constexpr int D = W::D, O = W::O, N1 = W::N1, H = (D + 3) / 4;
  constexpr int EF = (int)(sizeof(QFElem<D>) / 4), EA = (int)(sizeof(QAElem<D>) / 4);
  static_assert(EF <= DK_EF && EA <= DK_EA && D + 2 <= DK_EP, "exchange slots");
  const int lane = tid & 63, wave = tid >> 6, q = tid & 3, qi = lane >> 2;
  const int T = x.T, Lc = x.Lc;
  const int nv = DK_V / sy.Gd, v0 = role * nv;
  const bool park = true;                      // state crosses the hand-overs through L2 (LW: LDS), never in registers
  // virtual workgroups whose chunks start before the end of the series; the others are skipped and
  // their wavefronts' totals read as the identity
  const int nvact = (T + 64 * Lc - 1) / (64 * Lc);
  const int nwact = 4 * nvact;
  float* xbF = x.xb;
  float* xbA = xbF + (size_t)DK_NWI * 4 * DK_EF;
  float* xbP = xbA + (size_t)DK_NWI * 4 * DK_EA;
  float* stat = xbP + 512;
  float* vst = stat + (size_t)DK_CH * DK_ST;
  const QScal<D> qs = make_qscal<TR, NS>(sc, q);
  typedef typename std::conditional<LW, CI_LDS float*, float*>::type FP;
  typedef typename std::conditional<LW, CI_LDS ci_f4v*, ci_f4v*>::type F4P;
  const int cl = tid >> 2;                     // LW: the chunk's index inside this workgroup
  auto vslot = [&](int v, int f) -> FP {
    if constexpr (LW) { (void)v; return x.lv + (f * NT + tid); }
    else return vst + ((size_t)(v * DK_VS + f) * NT + tid);
  };
  auto krow = [&](int step, int c) -> FP {     // the 8 floats of step `step` of chunk c
    if constexpr (LW) { (void)c; return x.lkr + (step * 64 + cl) * 8; }
    else return x.kr + ((size_t)step * DK_CH + c) * 8;
  };
  auto yvp = [&](int step, int c) -> FP {
    if constexpr (LW) { (void)c; return x.lyv + (step * 64 + cl); }
    else return x.yv + ((size_t)step * DK_CH + c);
  };
  auto ld4 = [](FP p) -> float4 { const ci_f4v t = *(F4P)p; return make_float4(t.x, t.y, t.z, t.w); };
  auto st4 = [](FP p, float a, float b, float c, float d) { *(F4P)p = ci_f4v{a, b, c, d}; };
  auto at4 = [](const float (&z)[4], int s) { return s == 0 ? z[0] : s == 1 ? z[1] : s == 2 ? z[2] : z[3]; };
  // the four random-number sites of a 4-step block, one per lane of the quad
  const uint32_t my_site = q == 0 ? (uint32_t)SITE_PRIOR_LEVEL : q == 1 ? (uint32_t)SITE_PRIOR_SEAS
                           : q == 2 ? (uint32_t)SITE_PRIOR_OBS : (uint32_t)SITE_PRIOR_SLOPE;
  // x+ one step on: x <- T x + noise
  auto xplus_step = [&](float (&xp)[D], bool ch, float zl, float zs, float zk) {
    qw_apply<TR, NS>(xp, ch);
    xp[0] = fmaf(sc.sl, zl, xp[0]);
    if constexpr (TR == 2) xp[1] = fmaf(sc.ss, zs, xp[1]);
    if (ch) {
      const float dz = sc.sdn * zk;
#pragma unroll
      for (int i = 0; i < N1; ++i) xp[O + i] -= dz;
    }
  };

  // moments of x_1 with the simulated x+_1 folded into the mean (see dk_draw in ci_kernels.h): on
  // every lane of the quad that owns chunk 0
  auto prior_moments = [&](float (&m1)[D], QMat<D>& P1) {
    float z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float z1[1];
      fill_normals<1>(rng, iter, SITE_PRIOR_INIT, 0, (uint32_t)i, z1);
      z[i] = z1[0];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float s = i == 0 ? x.a1_loc : 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) s = fmaf(x.chol1[i * D + j], z[j], s);
      m1[i] = s;
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
      P1.r[i] = qp_make<typename QMat<D>::T>([&](int h) {
        const int j = q + 4 * h;
        float p = 0.f;
        if (i == 0 && j == 0) p = x.p1l;
        if (TR == 2 && i == 1 && j == 1) p = x.p1s;
        if (i >= O && j >= O && j < D) p = x.p1e * ((i == j ? 1.f : 0.f) - 1.f / (float)NS);
        return p;
      });
  };

  // per-virtual-workgroup state that lives in registers when nv == 1
  float xpre[D];            // x+ at the chunk start
  float a0[D];              // predicted mean at the chunk start
  QMat<D> P0;               // predicted covariance at the chunk start
  QAElem<D> aex;            // in-wave exclusive suffix of the backward scan

  prof.tick(19);     // (whatever this workgroup did since its last stamp is not the draw's)
  // =================== phase A: prior simulation, chunk elements, in-wave scan ====================
  if (do_a) {
#pragma unroll 1
  for (int v = v0; v < v0 + nv; ++v) {
    if (v >= nvact) continue;
    const int c = 64 * v + (tid >> 2), t0 = c * Lc, wi = 4 * v + wave;
    WPElem<D> pe;
    {
      float s[D];
#pragma unroll
      for (int i = 0; i < D; ++i) s[i] = 0.f;
      int m = 0;
#pragma unroll 1
      for (int g4 = 0; g4 < Lc; g4 += 4) {
        const int t4 = t0 + g4;
        float z[4];
        normals4(site_call(rng, iter, my_site, 0, (uint32_t)(t4 >> 2)), z);
        const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(x.cbv + t4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
          xplus_step(s, ch, q_bc(z[u], 0), q_bc(z[u], 3), q_bc(z[u], 1));
          m += ch ? 1 : 0;
        }
      }
      pe.k = (float)Lc;
      pe.m = m % NS;
#pragma unroll
      for (int i = 0; i < D; ++i) pe.s.v[i] = s[i];
    }
    WPElem<D> incl = pe;
#pragma unroll 1
    for (int off = 1; off < 16; off <<= 1) {
      const WPElem<D> o = q_shfl_up(incl, off);
      if (qi >= off) incl = wpelem_combine<TR, NS>(o, incl);
    }
    if (qi == 15 && q == 0) {
      float* p = xbP + wi * DK_EP;
      p[0] = incl.k; p[1] = (float)incl.m;
#pragma unroll
      for (int i = 0; i < D; ++i) p[2 + i] = incl.s.v[i];
    }
    WPElem<D> pex = q_shfl_up(incl, 1);
    if (qi == 0) { pex.k = 0.f; pex.m = 0; pex.s = vzero<D>(); }
    {
      *vslot(v, 0) = pex.k; *vslot(v, 1) = (float)pex.m;
#pragma unroll
      for (int i = 0; i < D; ++i) *vslot(v, 2 + i) = pex.s.v[i];
    }
  }
  prof.tick(20);
  dk_barrier(sy, tid);
  prof.tick(16);
  }
  if (!do_rest) return;

  // =================== phase B: x+ and y~, chunk filtering elements, in-wave scan ==================
#pragma unroll 1
  for (int v = v0; v < v0 + nv; ++v) {
    if (v >= nvact) continue;
    const int c = 64 * v + (tid >> 2), t0 = c * Lc, wi = 4 * v + wave;
    WPElem<D> pex;
    {
      pex.k = *vslot(v, 0); pex.m = (int)*vslot(v, 1);
#pragma unroll
      for (int i = 0; i < D; ++i) pex.s.v[i] = *vslot(v, 2 + i);
    }
    {
      // Everything before this wavefront: lane l of each half-wave loads the total of wavefront l
      // (skipped ones: the identity), a Kogge-Stone over the 32 lanes, and the prefix is read off lane
      // wi - 1.  (A serial walk over the earlier totals -- up to 31 dependent L2 round trips -- made
      // the last DK worker enter its element pass 18k cycles after the first.)
      WPElem<D> e;
      {
        const int w = lane & 31;
        e.k = 0.f; e.m = 0; e.s = vzero<D>();
        if (w < nwact) {
          const float* p = xbP + w * DK_EP;
          e.k = p[0]; e.m = (int)p[1];
#pragma unroll
          for (int i = 0; i < D; ++i) e.s.v[i] = p[2 + i];
        }
      }
#pragma unroll 1
      for (int off = 1; off < 32; off <<= 1) {
        const WPElem<D> o = q_perm(e, (lane & 31) >= off ? lane - off : lane);
        if ((lane & 31) >= off) e = wpelem_combine<TR, NS>(o, e);
      }
      WPElem<D> acc = q_perm(e, wi > 0 ? wi - 1 : 0);
      if (wi == 0) { acc.k = 0.f; acc.m = 0; acc.s = vzero<D>(); }
      acc = wpelem_combine<TR, NS>(acc, pex);
#pragma unroll
      for (int i = 0; i < D; ++i) xpre[i] = acc.s.v[i];
    }
    // the chunk's element in the column-split layout; b and eta on every lane while it is built
    QMat<D> A = qm_eye<D>(q), C = qm_zero<D>(), J = qm_zero<D>();
    float b[D], eta[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { b[i] = 0.f; eta[i] = 0.f; }
    if (c == 0) {
      // the prior as the first element: x_1 ~ N(a_1 + chol(P_1) z, P_1)
      A = qm_zero<D>();
      prior_moments(b, C);
    }
    {
      float xp[D];
#pragma unroll
      for (int i = 0; i < D; ++i) xp[i] = xpre[i];
      // (one wave per SIMD: nothing hides an L2 round trip but a request made a block ahead)
      float4 nr4 = *reinterpret_cast<const float4*>(x.resid + t0);
      uint32_t nmk = *reinterpret_cast<const uint32_t*>(x.msk + t0);
      uint32_t ncb = *reinterpret_cast<const uint32_t*>(x.cbv + t0);
#pragma unroll 1
      for (int g4 = 0; g4 < Lc; g4 += 4) {
        const int t4 = t0 + g4;
        const float4 r4 = nr4;
        const uint32_t mk4 = nmk, cb4 = ncb;
        if (g4 + 4 < Lc) {
          nr4 = *reinterpret_cast<const float4*>(x.resid + t4 + 4);
          nmk = *reinterpret_cast<const uint32_t*>(x.msk + t4 + 4);
          ncb = *reinterpret_cast<const uint32_t*>(x.cbv + t4 + 4);
        }
        float z[4];
        normals4(site_call(rng, iter, my_site, 0, (uint32_t)(t4 >> 2)), z);
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
          const bool obs = ((mk4 >> (8 * u)) & 0xFFu) == 0u;
          const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
          const float zm = at4(z, u);
          const float zl = q_bc(zm, 0), zk = q_bc(zm, 1), zo = q_bc(zm, 2), zs = q_bc(zm, 3);
          const float ru = u == 0 ? r4.x : u == 1 ? r4.y : u == 2 ? r4.z : r4.w;
          const float yt = ru - (xp[0] + xp[O] + sc.so * zo);
          if (q == 0) *yvp(g4 + u, c) = yt;
          if (obs) {
            // fold y~_t into (A, b, C, eta, J)
            QVec<D> za, cz;
            za.v = A.r[0] + A.r[O];
            cz.v = C.r[0] + C.r[O];
            float zar[D], czr[D];
            q_rep(za, zar);
            q_rep(cz, czr);
            const float zb = b[0] + b[O];
            const float Sv = czr[0] + czr[O] + sc.H;
            const float rS = __builtin_amdgcn_rcpf(Sv);
            const float e = (yt - zb) * rS;
#pragma unroll
            for (int i = 0; i < D; ++i) {
              eta[i] = fmaf(zar[i], e, eta[i]);
              b[i] = fmaf(czr[i], e, b[i]);
              const float ki = -(czr[i] * rS), zi = zar[i] * rS;
              A.r[i] = ki * za.v + A.r[i];
              J.r[i] = zi * za.v + J.r[i];
              C.r[i] = ki * cz.v + C.r[i];
            }
          }
          // time update t -> t + 1
          qw_apply<TR, NS>(A.r, ch);
          qw_apply<TR, NS>(b, ch);
          qc_predict<TR, NS>(C, ch, qs, q);
          xplus_step(xp, ch, zl, zs, zk);
        }
      }
    }
    QFElem<D> fe;
    fe.A = A; fe.C = C; fe.J = J;
    fe.AT = q_transpose(A, q);
    fe.b = q_own<D>(b, q);
    fe.eta = q_own<D>(eta, q);
    prof.tick(21);
    QFElem<D> incl = fe;
#pragma unroll 1
    for (int off = 1; off < 16; off <<= 1) {
      const QFElem<D> o = q_shfl_up(incl, off);
      if (qi >= off) incl = qf_combine<D>(o, incl, q);
    }
    if (qi == 15) {
      float* p = xbF + (size_t)(wi * 4 + q) * DK_EF;
      const Arr<QFElem<D>> a = __builtin_bit_cast(Arr<QFElem<D>>, incl);
#pragma unroll
      for (int i = 0; i < EF; ++i) p[i] = a.f[i];
    }
    QFElem<D> fex = q_shfl_up(incl, 1);
    if (qi == 0) fex = qf_identity<D>(q);
    {
      // the in-wave prefix waits in L2 for the scan of the wave totals in every configuration: 60
      // registers that would otherwise sit (spilled) under the 32-total scan
      const Arr<QFElem<D>> a = __builtin_bit_cast(Arr<QFElem<D>>, fex);
#pragma unroll
      for (int i = 0; i < EF; ++i) *vslot(v, i) = a.f[i];
    }
    if (park) {
#pragma unroll
      for (int i = 0; i < D; ++i) *vslot(v, EF + i) = xpre[i];
    }
    prof.tick(22);
  }
  dk_barrier(sy, tid);
  prof.tick(17);

  // =================== phase C: prefixes, local filter (gains), backward chunk maps =================
  {
    // Every wavefront scans the 32 wave totals of the chain on its own: quad j loads totals 2j and
    // 2j + 1, combines them, and the 16 pairs go through one more Kogge-Stone (totals of skipped
    // wavefronts are the identity).
    QFElem<D> tot0, tot;
    {
      auto load_total = [&](int w) {
        if (w >= nwact) return qf_identity<D>(q);
        const float* p = xbF + (size_t)(w * 4 + q) * DK_EF;
        Arr<QFElem<D>> a;
#pragma unroll
        for (int i = 0; i < EF; ++i) a.f[i] = p[i];
        return __builtin_bit_cast(QFElem<D>, a);
      };
      tot0 = load_total(2 * qi);
      tot = qf_combine<D>(tot0, load_total(2 * qi + 1), q);
    }
#pragma unroll 1
    for (int off = 1; off < 16; off <<= 1) {
      const QFElem<D> o = q_shfl_up(tot, off);
      if (qi >= off) tot = qf_combine<D>(o, tot, q);
    }
    // the predicted moments at every chunk start (the totals die here: nothing of the scan stays
    // live through the per-step passes below)
#pragma unroll 1
    for (int v = v0; v < v0 + nv; ++v) {
      if (v >= nvact) continue;
      const int c = 64 * v + (tid >> 2), wi = 4 * v + wave;
      QFElem<D> fex;
      {
        Arr<QFElem<D>> a;
#pragma unroll
        for (int i = 0; i < EF; ++i) a.f[i] = *vslot(v, i);
        fex = __builtin_bit_cast(QFElem<D>, a);
      }
      // everything before wavefront wi: the pairs before its own, and for an odd wi the even total
      QFElem<D> wp = q_shfl_from(tot, wi >= 2 ? (wi >> 1) - 1 : 0, q);
      if (wi < 2) wp = qf_identity<D>(q);
      if (wi & 1) wp = qf_combine<D>(wp, q_shfl_from(tot0, wi >> 1, q), q);
      const QFElem<D> pre = qf_combine<D, true>(wp, fex, q);
      {
        QVec<D> bb = pre.b;
        q_rep(bb, a0);
        P0 = pre.C;
      }
      if (c == 0) prior_moments(a0, P0);        // nothing before the first chunk: the prior itself
      if (park) {
#pragma unroll
        for (int i = 0; i < D; ++i) *vslot(v, i) = a0[i];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int i = 0; i < D; ++i) *vslot(v, 8 + h * D + i) = qp_get(P0.r[i], h);
      }
    }
    prof.tick(23);
#pragma unroll 1
    for (int v = v0; v < v0 + nv; ++v) {
      if (v >= nvact) continue;
      const int c = 64 * v + (tid >> 2), t0 = c * Lc, wi = 4 * v + wave;
      if (park) {
#pragma unroll
        for (int i = 0; i < D; ++i) { a0[i] = *vslot(v, i); xpre[i] = *vslot(v, EF + i); }
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int i = 0; i < D; ++i) qp_set(P0.r[i], h, *vslot(v, 8 + h * D + i));
      }
      // local Kalman filter from the predicted moments at the chunk start: K_t, v_t / F_t
      {
        float am[D];
#pragma unroll
        for (int i = 0; i < D; ++i) am[i] = a0[i];
        QMat<D> Pc = P0;
        float nyt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nyt[u] = *yvp(u, c);
        uint32_t nmk = *reinterpret_cast<const uint32_t*>(x.msk + t0);
        uint32_t ncb = *reinterpret_cast<const uint32_t*>(x.cbv + t0);
#pragma unroll 1
        for (int g4 = 0; g4 < Lc; g4 += 4) {
          const int t4 = t0 + g4;
          const uint32_t mk4 = nmk, cb4 = ncb;
          float yt4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) yt4[u] = nyt[u];
          if (g4 + 4 < Lc) {       // the next block's rows, requested before this block's stores
#pragma unroll
            for (int u = 0; u < 4; ++u) nyt[u] = *yvp(g4 + 4 + u, c);
            nmk = *reinterpret_cast<const uint32_t*>(x.msk + t4 + 4);
            ncb = *reinterpret_cast<const uint32_t*>(x.cbv + t4 + 4);
          }
#pragma unroll 1
          for (int u = 0; u < 4; ++u) {
            const bool obs = ((mk4 >> (8 * u)) & 0xFFu) == 0u;
            const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
            const float yt = at4(yt4, u);
            float vf = 0.f;
            float kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i] = 0.f;
            if (obs) {
              QVec<D> pz;
              pz.v = Pc.r[0] + Pc.r[O];
              float pzr[D];
              q_rep(pz, pzr);
              const float Fv = pzr[0] + pzr[O] + sc.H;
              const float rF = __builtin_amdgcn_rcpf(Fv);
              const float vv = yt - (am[0] + am[O]);
              vf = vv * rF;
#pragma unroll
              for (int i = 0; i < D; ++i) {
                kf[i] = pzr[i] * rF;
                am[i] = fmaf(kf[i], vv, am[i]);
                Pc.r[i] = (pz.v * (-pzr[i])) * rF + Pc.r[i];
              }
            }
            const FP kp = krow(g4 + u, c);
            if (q == 0) st4(kp, kf[0], kf[1], kf[2], kf[3]);
            if (q == 1) st4(kp + 4, kf[4], kf[5], kf[6], kf[7]);
            if (q == 2) *yvp(g4 + u, c) = vf;
            qw_apply<TR, NS>(am, ch);
            qc_predict<TR, NS>(Pc, ch, qs, q);
          }
        }
      }
      prof.tick(24);
      // backward recursion r <- T' r ; r += Z'(v/F - K'r): the chunk's map
      QAElem<D> ae = qa_identity<D>(q);
      // (the quad's own stores of K_t / v/F above are read back below by ALL its lanes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      struct KV4 { float4 a[4], b[4]; float vf[4]; uint32_t mk, cb; };
      auto load_kv = [&](int g4, KV4& o) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const FP kp = krow(g4 + u, c);
          o.a[u] = ld4(kp);
          o.b[u] = ld4(kp + 4);
          o.vf[u] = *yvp(g4 + u, c);
        }
        o.mk = *reinterpret_cast<const uint32_t*>(x.msk + t0 + g4);
        o.cb = *reinterpret_cast<const uint32_t*>(x.cbv + t0 + g4);
      };
      KV4 nxt;
      load_kv(Lc - 4, nxt);
#pragma unroll 1
      for (int g4 = Lc - 4; g4 >= 0; g4 -= 4) {
        const KV4 cur = nxt;
        if (g4 >= 4) load_kv(g4 - 4, nxt);
        const uint32_t mk4 = cur.mk, cb4 = cur.cb;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
          const bool obs = ((mk4 >> (8 * u)) & 0xFFu) == 0u;
          const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
          qw_apply_t<TR, NS>(ae.M.r, ch);
          qw_apply_t<TR, NS>(ae.c, ch);
          if (obs) {
            const float kf[8] = {cur.a[u].x, cur.a[u].y, cur.a[u].z, cur.a[u].w,
                                 cur.b[u].x, cur.b[u].y, cur.b[u].z, cur.b[u].w};
            const float vf = cur.vf[u];
            float kc = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i) kc = fmaf(kf[i], ae.c[i], kc);
            const float add = vf - kc;
            ae.c[0] += add;
            ae.c[O] += add;
            {
              typename QMat<D>::T kr = ae.M.r[0] * kf[0];
#pragma unroll
              for (int i = 1; i < D; ++i) kr = ae.M.r[i] * kf[i] + kr;
              ae.M.r[0] -= kr;
              ae.M.r[O] -= kr;
            }
          }
        }
      }
      prof.tick(25);
      QAElem<D> incl = ae;
#pragma unroll 1
      for (int off = 1; off < 16; off <<= 1) {
        const QAElem<D> o = q_shfl_down(incl, off);
        if (qi + off < 16) incl = qa_compose<D>(incl, o, q);
      }
      if (qi == 0) {
        float* p = xbA + (size_t)(wi * 4 + q) * DK_EA;
        const Arr<QAElem<D>> a = __builtin_bit_cast(Arr<QAElem<D>>, incl);
#pragma unroll
        for (int i = 0; i < EA; ++i) p[i] = a.f[i];
      }
      aex = q_shfl_down(incl, 1);
      if (qi == 15) aex = qa_identity<D>(q);
      if (park) {
        const Arr<QAElem<D>> a = __builtin_bit_cast(Arr<QAElem<D>>, aex);
#pragma unroll
        for (int i = 0; i < EA; ++i) *vslot(v, i) = a.f[i];
#pragma unroll
        for (int i = 0; i < D; ++i) { *vslot(v, EA + i) = xpre[i]; *vslot(v, EA + 8 + i) = a0[i]; }
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int i = 0; i < D; ++i) *vslot(v, EA + 16 + h * D + i) = qp_get(P0.r[i], h);
      }
    }
  }
  dk_barrier(sy, tid);
  prof.tick(18);

  // =================== phase D: r through the chunk, forward reconstruction, statistics ============
  {
    QAElem<D> tot1, tot;
    {
      auto load_total = [&](int w) {
        if (w >= nwact) return qa_identity<D>(q);
        const float* p = xbA + (size_t)(w * 4 + q) * DK_EA;
        Arr<QAElem<D>> a;
#pragma unroll
        for (int i = 0; i < EA; ++i) a.f[i] = p[i];
        return __builtin_bit_cast(QAElem<D>, a);
      };
      tot1 = load_total(2 * qi + 1);
      tot = qa_compose<D>(load_total(2 * qi), tot1, q);
    }
#pragma unroll 1
    for (int off = 1; off < 16; off <<= 1) {
      const QAElem<D> o = q_shfl_down(tot, off);
      if (qi + off < 16) tot = qa_compose<D>(tot, o, q);
    }
    // r at every chunk's end: the maps of the wave's later chunks applied to what the later waves leave
    float rsuf[D];
#pragma unroll 1
    for (int v = v0; v < v0 + nv; ++v) {
      if (v >= nvact) continue;
      const int wi = 4 * v + wave;
      if (park) {
        Arr<QAElem<D>> a;
#pragma unroll
        for (int i = 0; i < EA; ++i) a.f[i] = *vslot(v, i);
        aex = __builtin_bit_cast(QAElem<D>, a);
      }
      // everything after wavefront wi: the pairs after its own, and for an even wi the odd total
      QAElem<D> ws = q_shfl_from(tot, (wi >> 1) < 15 ? (wi >> 1) + 1 : 15, q);
      if ((wi >> 1) == 15) ws = qa_identity<D>(q);
      if (!(wi & 1)) ws = qa_compose<D>(q_shfl_from(tot1, wi >> 1, q), ws, q);
#pragma unroll
      for (int i = 0; i < D; ++i) rsuf[i] = aex.c[i];
      q_mv_acc(aex.M, ws.c, q, rsuf);
      if (park) {
#pragma unroll
        for (int i = 0; i < D; ++i) *vslot(v, i) = rsuf[i];
      }
    }
    prof.tick(26);
#pragma unroll 1
    for (int v = v0; v < v0 + nv; ++v) {
      if (v >= nvact) continue;
      const int c = 64 * v + (tid >> 2), t0 = c * Lc;
      if (park) {
#pragma unroll
        for (int i = 0; i < D; ++i) { rsuf[i] = *vslot(v, i); xpre[i] = *vslot(v, EA + i); a0[i] = *vslot(v, EA + 8 + i); }
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int i = 0; i < D; ++i) qp_set(P0.r[i], h, *vslot(v, EA + 16 + h * D + i));
      }
      // (5a) r through the chunk, backward; r_{t-1} takes the place of K_t
      {
        float r[D];
#pragma unroll
        for (int i = 0; i < D; ++i) r[i] = rsuf[i];
        // The rows of a block are requested ONE BLOCK AHEAD; r_{t-1} then overwrites K_t in place.  No
        // hazard: a row is stored after the step that consumed its loaded copy, and vmcnt returns in
        // order -- every load of the block (issued by all four lanes in one instruction) has landed
        // before the first of its stores is issued.
        struct KR4 { float4 a[4], b[4]; float vf[4]; uint32_t mk, cb; };
        auto load_kr = [&](int g4, KR4& o) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const FP kp = krow(g4 + u, c);
            o.a[u] = ld4(kp);
            o.b[u] = ld4(kp + 4);
            o.vf[u] = *yvp(g4 + u, c);
          }
          o.mk = *reinterpret_cast<const uint32_t*>(x.msk + t0 + g4);
          o.cb = *reinterpret_cast<const uint32_t*>(x.cbv + t0 + g4);
        };
        KR4 nx;
        load_kr(Lc - 4, nx);
#pragma unroll 1
        for (int g4 = Lc - 4; g4 >= 0; g4 -= 4) {
          const KR4 cu = nx;
          if (g4 >= 4) load_kr(g4 - 4, nx);
          const uint32_t mk4 = cu.mk, cb4 = cu.cb;
#pragma unroll
          for (int u = 3; u >= 0; --u) {
            const bool obs = ((mk4 >> (8 * u)) & 0xFFu) == 0u;
            const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
            qw_apply_t<TR, NS>(r, ch);
            if (obs) {
              const float kf[8] = {cu.a[u].x, cu.a[u].y, cu.a[u].z, cu.a[u].w, cu.b[u].x, cu.b[u].y, cu.b[u].z, cu.b[u].w};
              float kr = 0.f;
#pragma unroll
              for (int i = 0; i < D; ++i) kr = fmaf(kf[i], r[i], kr);
              const float add = cu.vf[u] - kr;
              r[0] += add;
              r[O] += add;
            }
            float r8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r8[i] = i < D ? r[i < D ? i : 0] : 0.f;
            const FP kp = krow(g4 + u, c);
            if (q == 0) st4(kp, r8[0], r8[1], r8[2], r8[3]);
            if (q == 1) st4(kp + 4, r8[4], r8[5], r8[6], r8[7]);
          }
        }
        // x^ at the chunk start: a + P r_{t0 - 1}
        q_mv_acc(P0, r, q, a0);
      }
      prof.tick(27);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // (5b) forward: x^_{t+1} = T x^_t + Q_t r_t; x+ re-simulated; x~ = x^ + x+; statistics
      {
        float ssl = 0.f, sss = 0.f, ssd = 0.f;
        auto stats = [&](const float (&xt)[D], const float (&xn)[D], bool ch) {
          float dl = xn[0] - xt[0];
          if constexpr (TR == 2) {
            dl -= xt[1];
            const float ds = xn[1] - xt[1];
            sss = fmaf(ds, ds, sss);
          }
          ssl = fmaf(dl, dl, ssl);
          if (ch) {
            float w;
            if constexpr (NS >= 3) w = (float)NS * (xt[O + 1] - xn[O]);
            else w = -2.0f * (xn[O] + xt[O]);
            ssd = fmaf(w, w, ssd);
          }
        };
        float xh[D], xp[D], xprev[D], xfirst[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { xh[i] = a0[i]; xp[i] = xpre[i]; xprev[i] = 0.f; xfirst[i] = 0.f; }
        bool chprev = false;
        auto load_r = [&](int g4, float4 (&ra)[4], float4 (&rb)[4]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int sidx = g4 + u + 1;          // r_t of step t = the row stored for step t + 1
            if (sidx < Lc) {
              const FP kp = krow(sidx, c);
              ra[u] = ld4(kp);
              rb[u] = ld4(kp + 4);
            } else {
              float r8[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) r8[i] = i < D ? rsuf[i < D ? i : 0] : 0.f;
              ra[u] = make_float4(r8[0], r8[1], r8[2], r8[3]);
              rb[u] = make_float4(r8[4], r8[5], r8[6], r8[7]);
            }
          }
        };
        float4 rna[4], rnb[4];
        load_r(0, rna, rnb);
#pragma unroll 1
        for (int g4 = 0; g4 < Lc; g4 += 4) {
          const int t4 = t0 + g4;
          float4 rca[4], rcb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { rca[u] = rna[u]; rcb[u] = rnb[u]; }
          if (g4 + 4 < Lc) load_r(g4 + 4, rna, rnb);
          float z[4];
          normals4(site_call(rng, iter, my_site, 0, (uint32_t)(t4 >> 2)), z);
          const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(x.cbv + t4);
          float lev4 = 0.f, slp4 = 0.f, sea4 = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t4 + u;
            const bool ch = ((cb4 >> (8 * u)) & 0xFFu) != 0u;
            float xt[D];
#pragma unroll
            for (int i = 0; i < D; ++i) xt[i] = xh[i] + xp[i];
            if (q == u) { lev4 = xt[0]; sea4 = xt[O]; if constexpr (TR == 2) slp4 = xt[1]; }
            if (g4 + u == 0) {
#pragma unroll
              for (int i = 0; i < D; ++i) xfirst[i] = xt[i];
            } else if (t < T) {
              stats(xprev, xt, chprev);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) xprev[i] = xt[i];
            chprev = ch;
            const float rn[8] = {rca[u].x, rca[u].y, rca[u].z, rca[u].w, rcb[u].x, rcb[u].y, rcb[u].z, rcb[u].w};
            qw_apply<TR, NS>(xh, ch);
            xh[0] = fmaf(sc.ql, rn[0], xh[0]);
            if constexpr (TR == 2) xh[1] = fmaf(sc.qs, rn[1], xh[1]);
            if (ch) {
              float sr = 0.f;
#pragma unroll
              for (int i = 0; i < N1; ++i) sr += rn[O + i];
              const float add = sc.qd * sr;
#pragma unroll
              for (int i = 0; i < N1; ++i) xh[O + i] += add;
            }
            xplus_step(xp, ch, q_bc(z[u], 0), q_bc(z[u], 3), q_bc(z[u], 1));
          }
          // lane u of the quad holds step t4 + u: one 4-byte store per lane, 16 bytes per quad
          if (t4 + q < T) {
            x.levw[t4 + q] = lev4;
            if constexpr (TR == 2) x.slpw[t4 + q] = slp4;
            x.seaw[t4 + q] = sea4;
          }
        }
        if (q == 0) {
          float* sp = stat + (size_t)c * DK_ST;
          sp[0] = ssl; sp[1] = sss; sp[2] = ssd;
#pragma unroll
          for (int i = 0; i < D; ++i) { sp[4 + i] = xfirst[i]; sp[12 + i] = xprev[i]; }
        }
      }
      prof.tick(28);
    }
  }
  dk_finish(sy, (int)iter + 2, (int)iter + 1, tid);
  prof.tick(29);
}

// The statistics of the draw for the scale updates: thread i adds chunk i's sums and the increment
// across the boundary to chunk i + 1 (from the first / last states the draw left per chunk).
template <int TR, int NS>
__device__ __forceinline__ void dk_stats(const float* xb, const uint8_t* cbv, int T, int Lc, int tid,
                                         float& ssl, float& sss, float& ssd) {
  constexpr int D = TR + NS - 1, O = TR;
  const float* stat = xb + (size_t)DK_NWI * 4 * DK_EF + (size_t)DK_NWI * 4 * DK_EA + 512;
  ssl = 0.f; sss = 0.f; ssd = 0.f;
#pragma unroll 1
  for (int c = tid; c < DK_CH; c += NT) {
    if (c * Lc >= T) break;                      // chunks past the end of the series were skipped
    const float* sp = stat + (size_t)c * DK_ST;
    ssl += sp[0]; sss += sp[1]; ssd += sp[2];
    const int t = c * Lc + Lc - 1;
    if (t + 1 < T) {
      const float* sn = sp + DK_ST;
      float xt[D], xn[D];
#pragma unroll
      for (int i = 0; i < D; ++i) { xt[i] = sp[12 + i]; xn[i] = sn[4 + i]; }
      float dl = xn[0] - xt[0];
      if constexpr (TR == 2) {
        dl -= xt[1];
        const float ds = xn[1] - xt[1];
        sss = fmaf(ds, ds, sss);
      }
      ssl = fmaf(dl, dl, ssl);
      if (cbv[t] != 0) {
        float w;
        if constexpr (NS >= 3) w = (float)NS * (xt[O + 1] - xn[O]);
        else w = -2.0f * (xn[O] + xt[O]);
        ssd = fmaf(w, w, ssd);
      }
    }
  }
}

}  // namespace ci
