// ci_wide_score.h -- TIME-PARALLEL Kalman log-likelihood and score for trend + ONE seasonal block of
// 2-7 seasons (state d = TR + NS - 1 <= 8, the oracle's (NS-1)-effect coordinates) and for trend-only
// series beyond the register-resident scans (T > 4096: an inert block), any length up to 65536.
// SURVEY.md section 8 row H (extension; oracle: ci_oracle_loglik_score / hmc_target).
//
// Same chunking as the Durbin-Koopman draw of ci_wide.h -- thread i of a 256-thread workgroup owns
// Lc consecutive steps; every recursion is per-thread pass -> block scan of 256 chunk elements ->
// per-thread pass:
//   * filter: chunk elements (A, b, C, eta, J) folded step by step in O(d^2), scan with the
//     Sarkka combine, local pass leaving K_t, v_t/F_t, 1/F_t per step in the HBM workspace and
//     this thread's share of  l = -1/2 sum_obs (log 2 pi + log F_t + v_t^2 / F_t);
//   * smoothing adjoints r_t (vector) and N_t (symmetric matrix) TOGETHER: one element
//     (M, c, S) per chunk carries  r -> M r + c  and  N -> M N M' + S  (the N map's matrix is the
//     r map's: M_t = (I - K_t Z)' T_t'), built backwards in O(d^2) per step (T' S T in closed form
//     for the companion shift, then the rank-two update -z u' - u z' + z z' (K'u + 1/F),
//     u = S K), composed by a suffix scan (one congruence + one product per combine);
//   * local backward pass from the chunk's suffix: e_t = v_t/F_t - K_t'(T'r_t),
//     D_t = 1/F_t + K_t'(T'N_t T)K_t, and the score sums
//       dl/dsigma_obs = sigma_obs sum (e_t^2 - D_t),  dl/dsigma_level = sigma_level sum (r_t[0]^2 - N_t[00]),
//       slope likewise, dl/dsigma_drift = sigma_drift sum_{changes} ((1'r_t)^2 - 1'N_t 1) / NS^2,
//     e_t to the workspace for  dl/dbeta = X'e.
// One evaluation per workgroup; ~75 us at T = 1000 with a weekly block (the sequential route of
// ci_score_seq.h: 3 ms).
#pragma once
#include "ci_wide.h"
#include "ci_score_seq.h"      // HmcSeqArgs, SeqScoreArgs (argument structs shared with the sequential route)

namespace ci {

// (M, c, S): r -> M r + c ;  N -> M N M' + S  (S packed upper triangle).  outer acts after inner.
template <int D> struct RNW {
  Mat<D> M;
  Vec<D> c;
  float S[D * (D + 1) / 2];
};
template <int D> __device__ __forceinline__ RNW<D> rnw_identity() {
  RNW<D> e;
  e.M = meye<D>();
  e.c = vzero<D>();
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) e.S[i] = 0.f;
  return e;
}
template <int D> __device__ __forceinline__ RNW<D> rnw_compose(const RNW<D>& o, const RNW<D>& i) {
  RNW<D> r;
  r.M = mm(o.M, i.M);
  r.c = vadd(mv(o.M, i.c), o.c);
  Mat<D> t;                                   // Mo Si
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      float sv = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) sv = fmaf(o.M.m[a][k], i.S[symidx<D>(k, b)], sv);
      t.m[a][b] = sv;
    }
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) {
      float sv = o.S[symidx<D>(a, b)];
#pragma unroll
      for (int k = 0; k < D; ++k) sv = fmaf(t.m[a][k], o.M.m[b][k], sv);
      r.S[symidx<D>(a, b)] = sv;
    }
  return r;
}

// S <- T' S T on a packed upper triangle (the transpose twin of w_cov_predict_sym's T S T').
// Trend block: T' = [[1,0],[1,1]].  Companion shift of the seasonal block (w_apply_t):
// (T'r)_0 = -r_last, (T'r)_j = r_{j-1} - r_last, so with s_a = S[a][last], tot = S[last][last]:
//   S'_{pq} = S_{p-1,q-1} - s_{p-1} - s_{q-1} + tot   (index -1: the term is absent),
// cross terms with a trend row r:  S'_{r,q} = S_{r,q-1} - S_{r,last}.
template <int TR, int NS>
__device__ __forceinline__ void w_cov_backward_sym(float (&C)[(TR + NS - 1) * (TR + NS) / 2], bool ch) {
  constexpr int D = TR + NS - 1, O = TR, N1 = NS - 1;
  auto S = [](int i, int j) constexpr { return symidx<D>(i, j); };
  if constexpr (TR == 2) {
    // rows / cols (0, 1) <- [[1,0],[1,1]] . [[1,1],[0,1]]
    C[S(1, 1)] += 2.0f * C[S(0, 1)] + C[S(0, 0)];
    C[S(0, 1)] += C[S(0, 0)];
#pragma unroll
    for (int j = 2; j < D; ++j) C[S(1, j)] += C[S(0, j)];
  }
  if (ch) {
    float sl[N1], cr[TR];
#pragma unroll
    for (int p = 0; p < N1; ++p) sl[p] = C[S(O + p, O + N1 - 1)];
#pragma unroll
    for (int r = 0; r < TR; ++r) cr[r] = C[S(r, O + N1 - 1)];
    const float tot = sl[N1 - 1];
    // descending order: every source lies before its destination
#pragma unroll
    for (int p = N1 - 1; p >= 1; --p) {
#pragma unroll
      for (int q = N1 - 1; q >= p; --q)
        C[S(O + p, O + q)] = C[S(O + p - 1, O + q - 1)] - sl[p - 1] - sl[q - 1] + tot;
    }
#pragma unroll
    for (int q = N1 - 1; q >= 1; --q) C[S(O, O + q)] = tot - sl[q - 1];
    C[S(O, O)] = tot;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int q = N1 - 1; q >= 1; --q) C[S(r, O + q)] = C[S(r, O + q - 1)] - cr[r];
      C[S(r, O)] = -cr[r];
    }
  }
}

struct WScoreSums { float ll, gH, gl, gs, gd; };

// The score of one parameter set by the whole workgroup.  resid [TP] (0 where masked), msk / cbv
// [TP] bytes (mask padded with 1, season-change flags padded with 0), wsp: private per-step fields
// [Lc][NF][NT] (slot F_YT holds 1/F_t here), ew [TP] receives e_t.  fslots: NW * 119 floats at
// d = 7, rslots: NW * 84.  Returns this THREAD's partial sums (the caller reduces them).
// Contains 2 __syncthreads().
template <int TR, int NS>
__device__ __forceinline__ WScoreSums wide_loglik_score(const WideScal& sc, const Vec<TR + NS - 1>& a1,
                                                        const Mat<TR + NS - 1>& P1, int T, int Lc,
                                                        const float* __restrict__ resid,
                                                        const uint8_t* __restrict__ msk,
                                                        const uint8_t* __restrict__ cbv,
                                                        float* __restrict__ wsp, float* __restrict__ ew,
                                                        bool want_grad, int tid, int lane, int wave,
                                                        float* fslots, float* rslots) {
  using W = WDim<TR, NS>;
  constexpr int D = W::D, O = W::O, N1 = W::N1, NF = W::NF;
  const int t0 = tid * Lc;
  auto at4 = [](const float4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };
  WScoreSums out;
  out.ll = 0.f; out.gH = 0.f; out.gl = 0.f; out.gs = 0.f; out.gd = 0.f;

  // ---- (1) the chunk's filtering element from the residuals (wide_dk_draw (2) without x+)
  FElemS<D> fe = felems_identity<D>();
  if (tid == 0) {
    fe.A = mzero<D>();
    fe.b = a1;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) fe.C[symidx<D>(i, j)] = P1.m[i][j];
  }
#pragma unroll 1
  for (int g4 = 0; g4 < Lc; g4 += 4) {
    const int t4 = t0 + g4;
    const float4 r4 = *reinterpret_cast<const float4*>(resid + t4);
    const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(msk + t4);
    const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(cbv + t4);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
      const bool ch = ((cb4 >> (8 * q)) & 0xFFu) != 0u;
      if (obs) {
        const float yt = at4(r4, q);
        float za[D], cz[D];
#pragma unroll
        for (int j = 0; j < D; ++j) za[j] = fe.A.m[0][j] + fe.A.m[O][j];
#pragma unroll
        for (int i = 0; i < D; ++i) cz[i] = fe.C[symidx<D>(i, 0)] + fe.C[symidx<D>(i, O)];
        const float zb = fe.b.v[0] + fe.b.v[O];
        const float rS = __builtin_amdgcn_rcpf(cz[0] + cz[O] + sc.H);
        const float e = (yt - zb) * rS;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          fe.eta.v[i] = fmaf(za[i], e, fe.eta.v[i]);
          fe.b.v[i] = fmaf(cz[i], e, fe.b.v[i]);
          const float ki = cz[i] * rS, zi = za[i] * rS;
#pragma unroll
          for (int j = 0; j < D; ++j) fe.A.m[i][j] = fmaf(-ki, za[j], fe.A.m[i][j]);
#pragma unroll
          for (int j = i; j < D; ++j) {
            fe.J[symidx<D>(i, j)] = fmaf(zi, za[j], fe.J[symidx<D>(i, j)]);
            fe.C[symidx<D>(i, j)] = fmaf(-ki, cz[j], fe.C[symidx<D>(i, j)]);
          }
        }
      }
      w_left<TR, NS>(fe.A, ch);
      w_apply<TR, NS>(fe.b, ch);
      w_cov_predict_sym<TR, NS>(fe.C, ch, sc);
    }
  }
  const FElemS<D> fpre = block_scan_excl_fwd_rolled(
      fe, [](const FElemS<D>& x, const FElemS<D>& y) { return felems_combine(x, y); },
      felems_identity<D>(), fslots, lane, wave);

  // ---- (2) local filter from the chunk's predicted moments: K_t, v_t/F_t, 1/F_t; l
  {
    Vec<D> am;
    float Ps[W::NPS];
    if (tid == 0) {
      am = a1;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) Ps[symidx<D>(i, j)] = P1.m[i][j];
    } else {
      am = fpre.b;
#pragma unroll
      for (int i = 0; i < W::NPS; ++i) Ps[i] = fpre.C[i];
    }
#pragma unroll 1
    for (int g4 = 0; g4 < Lc; g4 += 4) {
      const int t4 = t0 + g4;
      const float4 r4 = *reinterpret_cast<const float4*>(resid + t4);
      const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(msk + t4);
      const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(cbv + t4);
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
        const bool ch = ((cb4 >> (8 * q)) & 0xFFu) != 0u;
        float* wl = wsp + (size_t)(g4 + q) * NF * NT + tid;
        float vf = 0.f, rF = 0.f;
        float kf[D];
#pragma unroll
        for (int i = 0; i < D; ++i) kf[i] = 0.f;
        if (obs) {
          float pz[D];
#pragma unroll
          for (int i = 0; i < D; ++i) pz[i] = Ps[symidx<D>(i, 0)] + Ps[symidx<D>(i, O)];
          const float Fv = pz[0] + pz[O] + sc.H;
          rF = __builtin_amdgcn_rcpf(Fv);
          const float v = at4(r4, q) - (am.v[0] + am.v[O]);
          vf = v * rF;
          out.ll -= 0.5f * (1.8378770664093453f + __logf(Fv) + v * vf);
#pragma unroll
          for (int i = 0; i < D; ++i) {
            kf[i] = pz[i] * rF;
            am.v[i] = fmaf(kf[i], v, am.v[i]);
#pragma unroll
            for (int j = i; j < D; ++j)
              Ps[symidx<D>(i, j)] = fmaf(-(pz[i] * pz[j]), rF, Ps[symidx<D>(i, j)]);
          }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) wl[(W::F_KF + i) * NT] = kf[i];
        wl[W::F_VF * NT] = vf;
        wl[W::F_YT * NT] = rF;
        w_apply<TR, NS>(am, ch);
        w_cov_predict_sym<TR, NS>(Ps, ch, sc);
      }
    }
  }
  if (!want_grad) return out;

  // ---- (3) the chunk's (M, c, S), built backwards step by step in O(d^2)
  RNW<D> re = rnw_identity<D>();
#pragma unroll 1
  for (int g4 = Lc - 4; g4 >= 0; g4 -= 4) {
    const int t4 = t0 + g4;
    const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(msk + t4);
    const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(cbv + t4);
#pragma unroll 1
    for (int q = 3; q >= 0; --q) {
      const int t = t4 + q;
      const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
      // the transition t -> t+1 exists for t + 1 < T; beyond the series everything is masked and
      // unchanged (cbv is padded with 0), so applying T' there is the identity on r = 0, N = 0
      const bool ch = ((cb4 >> (8 * q)) & 0xFFu) != 0u;
      const float* wl = wsp + (size_t)(g4 + q) * NF * NT + tid;
      if (t + 1 < T) {
        w_left_t<TR, NS>(re.M, ch);
        w_apply_t<TR, NS>(re.c, ch);
        w_cov_backward_sym<TR, NS>(re.S, ch);
      }
      if (obs) {
        float kf[D];
#pragma unroll
        for (int i = 0; i < D; ++i) kf[i] = wl[(W::F_KF + i) * NT];
        const float vf = wl[W::F_VF * NT], rF = wl[W::F_YT * NT];
        float kc = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) kc = fmaf(kf[i], re.c.v[i], kc);
        const float add = vf - kc;
        re.c.v[0] += add;
        re.c.v[O] += add;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          float kr = 0.f;
#pragma unroll
          for (int i = 0; i < D; ++i) kr = fmaf(kf[i], re.M.m[i][j], kr);
          re.M.m[0][j] -= kr;
          re.M.m[O][j] -= kr;
        }
        float u[D];
        float s = rF;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          float ui = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) ui = fmaf(re.S[symidx<D>(i, j)], kf[j], ui);
          u[i] = ui;
          s = fmaf(kf[i], ui, s);
        }
        // S <- S - z u' - u z' + z z' s,  z = e_0 + e_O
#pragma unroll
        for (int j = 0; j < D; ++j) {
          re.S[symidx<D>(0, j)] -= u[j];
          if (j >= O) re.S[symidx<D>(O, j)] -= u[j];
        }
#pragma unroll
        for (int i = 0; i <= O; ++i) re.S[symidx<D>(i, O)] -= u[i];
        re.S[symidx<D>(0, 0)] -= u[0];
        re.S[symidx<D>(0, 0)] += s;
        re.S[symidx<D>(0, O)] += s;
        re.S[symidx<D>(O, O)] += s;
      }
    }
  }
  const RNW<D> rsuf = block_scan_excl_bwd_rolled(
      re, [](const RNW<D>& o, const RNW<D>& i) { return rnw_compose(o, i); }, rnw_identity<D>(), rslots,
      lane, wave);

  // ---- (4) local backward pass: r_t, N_t through the chunk, e_t, the score sums
  {
    Vec<D> r = rsuf.c;                  // the later chunks' maps applied to r = 0, N = 0
    float Ns[W::NPS];
#pragma unroll
    for (int i = 0; i < W::NPS; ++i) Ns[i] = rsuf.S[i];
#pragma unroll 1
    for (int g4 = Lc - 4; g4 >= 0; g4 -= 4) {
      const int t4 = t0 + g4;
      const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(msk + t4);
      const uint32_t cb4 = *reinterpret_cast<const uint32_t*>(cbv + t4);
      float e4[4];
#pragma unroll 1
      for (int q = 3; q >= 0; --q) {
        const int t = t4 + q;
        const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
        const bool ch = ((cb4 >> (8 * q)) & 0xFFu) != 0u;
        const float* wl = wsp + (size_t)(g4 + q) * NF * NT + tid;
        if (t + 1 < T) {
          // disturbance of the transition t -> t+1 (r, N here are r_t, N_t)
          out.gl += r.v[0] * r.v[0] - Ns[symidx<D>(0, 0)];
          if constexpr (TR == 2) out.gs += r.v[1] * r.v[1] - Ns[symidx<D>(1, 1)];
          if (ch) {
            float sr = 0.f, sn = 0.f;
#pragma unroll
            for (int i = 0; i < N1; ++i) {
              sr += r.v[O + i];
#pragma unroll
              for (int j = 0; j < N1; ++j) sn += Ns[symidx<D>(O + i, O + j)];
            }
            out.gd += sr * sr - sn;
          }
          w_apply_t<TR, NS>(r, ch);
          w_cov_backward_sym<TR, NS>(Ns, ch);
        }
        float et = 0.f;
        if (obs) {
          float kf[D];
#pragma unroll
          for (int i = 0; i < D; ++i) kf[i] = wl[(W::F_KF + i) * NT];
          const float vf = wl[W::F_VF * NT], rF = wl[W::F_YT * NT];
          float kr = 0.f;
#pragma unroll
          for (int i = 0; i < D; ++i) kr = fmaf(kf[i], r.v[i], kr);
          et = vf - kr;
          float u[D];
          float dt = rF;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            float ui = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) ui = fmaf(Ns[symidx<D>(i, j)], kf[j], ui);
            u[i] = ui;
            dt = fmaf(kf[i], ui, dt);
          }
          out.gH += et * et - dt;
          r.v[0] += et;
          r.v[O] += et;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            Ns[symidx<D>(0, j)] -= u[j];
            if (j >= O) Ns[symidx<D>(O, j)] -= u[j];
          }
#pragma unroll
          for (int i = 0; i <= O; ++i) Ns[symidx<D>(i, O)] -= u[i];
          Ns[symidx<D>(0, 0)] -= u[0];
          Ns[symidx<D>(0, 0)] += dt;
          Ns[symidx<D>(0, O)] += dt;
          Ns[symidx<D>(O, O)] += dt;
        }
        e4[q] = et;
      }
      *reinterpret_cast<float4*>(ew + t4) = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
  }
  return out;
}

// ------------------------------------------------------------------------------------
// Workgroup-level evaluation: residual y - X beta, score, block reductions, X'e.
// ------------------------------------------------------------------------------------
struct WideScoreArgs {
  SeqScoreArgs q;        // data (theta / out_* as in the sequential route); q.K = 1, or 0 for a trend-only
                         // series on an inert block; q.ws unused
  int Lc;
  float* ws;             // [E or C, wide_score_ws_floats(D, Lc)]
};
__host__ __device__ inline size_t wide_score_ws_floats(int D, int Lc) {
  const size_t TP = (size_t)NT * Lc;
  return (2 + (size_t)(2 + 2 * D)) * TP + TP / 2;      // resid, e, private fields, mask + change bytes
}
template <int D> __host__ __device__ inline size_t wide_score_lds_floats(int P) {
  return (size_t)NW * (sizeof(FElemS<D>) / 4) + (size_t)NW * (sizeof(RNW<D>) / 4) + (size_t)NW * (P + 8);
}

template <int TR, int NS> struct WideScoreFn {
  using W = WDim<TR, NS>;
  static constexpr int D = W::D, O = W::O, N1 = W::N1;
  const SeqScoreArgs* q;
  int Lc, tid, lane, wave;
  float *residw, *ew, *wsp, *fslots, *rslots, *red;
  uint8_t *mskp, *cbp;
  Vec<D> a1;
  Mat<D> P1;

  __device__ __forceinline__ void init(const SeqScoreArgs* q_, int Lc_, float* ws, float* lds, int tid_) {
    q = q_; Lc = Lc_; tid = tid_; lane = tid_ & 63; wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int T = q->T, TP = NT * Lc;
    residw = ws; ew = residw + TP; wsp = ew + TP;
    mskp = (uint8_t*)(wsp + (size_t)W::NF * TP); cbp = mskp + TP;
    fslots = lds; rslots = fslots + NW * (sizeof(FElemS<D>) / 4); red = rslots + NW * (sizeof(RNW<D>) / 4);
    for (int t = tid; t < TP; t += NT) {
      const bool in = t < T;
      mskp[t] = in ? (q->mask[t] != 0 ? 1 : 0) : 1;
      cbp[t] = (in && q->K > 0 && q->season_change[t] != 0) ? 1 : 0;
    }
    a1 = vzero<D>();
    a1.v[0] = q->a1;
    P1 = mzero<D>();
    P1.m[0][0] = q->p10;
    if constexpr (TR == 2) P1.m[1][1] = q->p11;
    const float p1e = q->K > 0 ? q->p1e : 0.f;       // (an inert block has zero prior variance)
#pragma unroll
    for (int i = 0; i < N1; ++i)
#pragma unroll
      for (int j = 0; j < N1; ++j) P1.m[O + i][O + j] = p1e * ((i == j ? 1.f : 0.f) - 1.f / (float)NS);
    __syncthreads();
  }

  // dev / gdev: (sigma_obs, sigma_level, sigma_slope, sigma_drift[K], beta[P])
  __device__ __forceinline__ void eval(const double* dev, double* gdev, double* ll_out, bool want_grad) {
    const int T = q->T, P = q->P, K = q->K, TP = NT * Lc, ob = 3 + K;
    const int n4 = TP >> 2;
    // ---- residual (time interleaved over threads: coalesced 16-byte accesses).  The rows of the
    // design are read UNCONDITIONALLY (a load inside a lane-predicated branch is waited for at the
    // end of the branch: one L2 round trip per load), four features per batch; `wide` (T % 4 == 0,
    // 16-byte aligned rows) is chosen outside the loops.
    const bool wide = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(q->Xt) & 15) == 0;
    auto row4_wide = [&](int j, int t) { return *reinterpret_cast<const float4*>(q->Xt + (size_t)j * T + t); };
    auto row4_scalar = [&](int j, int t) {            // t < T; entries beyond the end read as 0
      const float* r = q->Xt + (size_t)j * T;
      float4 x;
      x.x = r[t];
      const float x1 = r[t + 1 < T ? t + 1 : T - 1], x2 = r[t + 2 < T ? t + 2 : T - 1],
                  x3 = r[t + 3 < T ? t + 3 : T - 1];
      x.y = t + 1 < T ? x1 : 0.f; x.z = t + 2 < T ? x2 : 0.f; x.w = t + 3 < T ? x3 : 0.f;
      return x;
    };
    auto residual_pass = [&](auto row4) {
      for (int c4 = tid; c4 < n4; c4 += NT) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const int t = 4 * c4;
        if (t < T) {
          float xs[4] = {0.f, 0.f, 0.f, 0.f};
          for (int j0 = 0; j0 < P; j0 += 4) {
            float4 xr[4];
            float bj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u < P ? j0 + u : P - 1;
              bj[u] = j0 + u < P ? (float)dev[ob + j] : 0.f;
              xr[u] = row4(j, t);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              xs[0] = fmaf(xr[u].x, bj[u], xs[0]); xs[1] = fmaf(xr[u].y, bj[u], xs[1]);
              xs[2] = fmaf(xr[u].z, bj[u], xs[2]); xs[3] = fmaf(xr[u].w, bj[u], xs[3]);
            }
          }
          float yv[4];
          bool ok[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {               // clamped, unconditional (see above)
            const int tu = t + u < T ? t + u : T - 1;
            yv[u] = q->y[tu];
            ok[u] = t + u < T && mskp[tu] == 0;
          }
          s.x = ok[0] ? yv[0] - xs[0] : 0.f;
          s.y = ok[1] ? yv[1] - xs[1] : 0.f;
          s.z = ok[2] ? yv[2] - xs[2] : 0.f;
          s.w = ok[3] ? yv[3] - xs[3] : 0.f;
        }
        *reinterpret_cast<float4*>(residw + 4 * c4) = s;
      }
    };
    if (wide) residual_pass(row4_wide); else residual_pass(row4_scalar);
    WideScal sc;
    sc.so = (float)dev[0]; sc.H = sc.so * sc.so;
    sc.sl = (float)dev[1]; sc.ql = sc.sl * sc.sl;
    sc.ss = (float)dev[2]; sc.qs = sc.ss * sc.ss;
    const float sd = K > 0 ? (float)dev[3] : 0.f;
    sc.sdn = sd * (1.0f / (float)NS); sc.qd = sc.sdn * sc.sdn;
    __syncthreads();
    const WScoreSums sm = wide_loglik_score<TR, NS>(sc, a1, P1, T, Lc, residw, mskp, cbp, wsp, ew, want_grad,
                                                    tid, lane, wave, fslots, rslots);
    // ---- block sums
    const int RSN = P + 8;
    {
      const float v0 = wave_sum_dpp(sm.ll), v1 = wave_sum_dpp(sm.gH), v2 = wave_sum_dpp(sm.gl),
                  v3 = wave_sum_dpp(sm.gs), v4 = wave_sum_dpp(sm.gd);
      if (lane == 0) {
        red[wave * RSN + 0] = v0; red[wave * RSN + 1] = v1; red[wave * RSN + 2] = v2;
        red[wave * RSN + 3] = v3; red[wave * RSN + 4] = v4;
      }
    }
    __syncthreads();                 // e_t of every step is in the workspace
    if (want_grad) {
      // d l / d beta_j = sum_t x_jt e_t, four features per pass over time (e_t = 0 beyond T and
      // where y is missing)
      auto gradient_pass = [&](auto row4) {
        for (int j0 = 0; j0 < P; j0 += 4) {
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
          for (int c4 = tid; c4 < n4; c4 += NT) {
            const int t = 4 * c4;
            if (t < T) {
              const float4 e4 = *reinterpret_cast<const float4*>(ew + t);
              float4 xr[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) xr[u] = row4(j0 + u < P ? j0 + u : P - 1, t);
#pragma unroll
              for (int u = 0; u < 4; ++u)
                acc[u] = fmaf(xr[u].w, e4.w, fmaf(xr[u].z, e4.z, fmaf(xr[u].y, e4.y, fmaf(xr[u].x, e4.x, acc[u]))));
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float w = wave_sum_dpp(acc[u]);
            if (lane == 0 && j0 + u < P) red[wave * RSN + 8 + j0 + u] = w;
          }
        }
      };
      if (wide) gradient_pass(row4_wide); else gradient_pass(row4_scalar);
    }
    __syncthreads();
    if (tid < RSN) {
      double sv = 0.0;
      for (int w = 0; w < NW; ++w) sv += (double)red[w * RSN + tid];
      if (tid == 0) *ll_out = sv;
      else if (want_grad) {
        if (tid == 1) gdev[0] = dev[0] * sv;
        else if (tid == 2) gdev[1] = dev[1] * sv;
        else if (tid == 3) gdev[2] = TR == 2 ? dev[2] * sv : 0.0;
        else if (tid == 4) { if (K > 0) gdev[3] = dev[3] * sv / (double)(NS * NS); }
        else if (tid >= 8) gdev[ob + (tid - 8)] = sv;
      }
    }
    __syncthreads();
  }

  __device__ __forceinline__ void operator()(const double* dev, double* gdev, double* ll_out) {
    eval(dev, gdev, ll_out, true);
  }
};

// E evaluations, one workgroup each: theta rows [3 + K + P] in the device layout.
template <int TR, int NS>
__global__ __launch_bounds__(NT) void wide_score_kernel(WideScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  constexpr int D = TR + NS - 1;
  const int ev = blockIdx.x, dim = 3 + a.q.K + a.q.P;
  WideScoreFn<TR, NS> fn;
  fn.init(&a.q, a.Lc, a.ws + (size_t)ev * wide_score_ws_floats(D, a.Lc), (float*)smem_w, threadIdx.x);
  fn.eval(a.q.theta + (size_t)ev * dim, a.q.out_grad ? a.q.out_grad + (size_t)ev * dim : nullptr,
          a.q.out_ll + ev, a.q.out_grad != nullptr);
}

// The HMC fit over the time-parallel score: hmc_drive of ci_score_seq.h, one workgroup per chain.
struct HmcWideArgs {
  HmcSeqArgs h;          // h.q.ws unused
  int Lc;
  float* ws;             // [C, wide_score_ws_floats(D, Lc)]
};
template <int TR, int NS>
__global__ __launch_bounds__(NT) void hmc_wide_kernel(HmcWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  constexpr int D = TR + NS - 1;
  WideScoreFn<TR, NS> fn;
  float* lds = (float*)(smem_w + ((sizeof(double) * hmc_seq_dbl_count() + 15) & ~(size_t)15));
  fn.init(&a.h.q, a.Lc, a.ws + (size_t)blockIdx.x * wide_score_ws_floats(D, a.Lc), lds, threadIdx.x);
  hmc_drive(a.h, smem_w, fn);
}
template <int D> __host__ __device__ inline size_t hmc_wide_lds_bytes(int P) {
  return ((sizeof(double) * hmc_seq_dbl_count() + 15) & ~(size_t)15) + sizeof(float) * wide_score_lds_floats<D>(P);
}

}  // namespace ci
