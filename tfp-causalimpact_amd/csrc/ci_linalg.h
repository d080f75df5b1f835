// ci_linalg.h -- register-resident d x d algebra (d = 1, 2) and the two associative
// operators the time-parallel Kalman recursions are built from.
//
// The reference runs LinearGaussianStateSpaceModel.forward_filter /
// posterior_marginals as three sequential length-T loops inside every Gibbs
// iteration.  Here each recursion is an associative scan:
//   * filtering  : elements (A, b, C, eta, J), Sarkka & Garcia-Fernandez (2021),
//                  "Temporal parallelization of Bayesian smoothers", eqs. (10)-(12);
//   * smoothing  : affine maps r_{t-1} = M_t r_t + c_t of the Durbin-Koopman /
//                  de Jong backward recursion.
// Everything is fully unrolled so the arrays live in VGPRs (no scratch).
#pragma once
#include <hip/hip_runtime.h>

namespace ci {

template <int D> struct Vec { float v[D]; };
template <int D> struct Mat { float m[D][D]; };

template <int D> __device__ __forceinline__ Vec<D> vzero() {
  Vec<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) r.v[i] = 0.f;
  return r;
}
template <int D> __device__ __forceinline__ Mat<D> mzero() {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) r.m[i][j] = 0.f;
  return r;
}
template <int D> __device__ __forceinline__ Mat<D> meye() {
  Mat<D> r = mzero<D>();
#pragma unroll
  for (int i = 0; i < D; ++i) r.m[i][i] = 1.f;
  return r;
}
template <int D> __device__ __forceinline__ Mat<D> mm(const Mat<D>& a, const Mat<D>& b) {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(a.m[i][k], b.m[k][j], s);
      r.m[i][j] = s;
    }
  return r;
}
// a * b'
template <int D> __device__ __forceinline__ Mat<D> mmt(const Mat<D>& a, const Mat<D>& b) {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(a.m[i][k], b.m[j][k], s);
      r.m[i][j] = s;
    }
  return r;
}
// a' * b
template <int D> __device__ __forceinline__ Mat<D> mtm(const Mat<D>& a, const Mat<D>& b) {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(a.m[k][i], b.m[k][j], s);
      r.m[i][j] = s;
    }
  return r;
}
template <int D> __device__ __forceinline__ Vec<D> mv(const Mat<D>& a, const Vec<D>& x) {
  Vec<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = fmaf(a.m[i][k], x.v[k], s);
    r.v[i] = s;
  }
  return r;
}
// a' * x
template <int D> __device__ __forceinline__ Vec<D> mtv(const Mat<D>& a, const Vec<D>& x) {
  Vec<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = fmaf(a.m[k][i], x.v[k], s);
    r.v[i] = s;
  }
  return r;
}
template <int D> __device__ __forceinline__ Mat<D> madd(const Mat<D>& a, const Mat<D>& b) {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
template <int D> __device__ __forceinline__ Vec<D> vadd(const Vec<D>& a, const Vec<D>& b) {
  Vec<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <int D> __device__ __forceinline__ Vec<D> vsub(const Vec<D>& a, const Vec<D>& b) {
  Vec<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) r.v[i] = a.v[i] - b.v[i];
  return r;
}
template <int D> __device__ __forceinline__ void symmetrize(Mat<D>& a) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i + 1; j < D; ++j) {
      const float s = 0.5f * (a.m[i][j] + a.m[j][i]);
      a.m[i][j] = s;
      a.m[j][i] = s;
    }
}
__device__ __forceinline__ Mat<1> minv(const Mat<1>& a) {
  Mat<1> r;
  r.m[0][0] = 1.0f / a.m[0][0];
  return r;
}
__device__ __forceinline__ Mat<2> minv(const Mat<2>& a) {
  const float det = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
  const float rd = 1.0f / det;
  Mat<2> r;
  r.m[0][0] = a.m[1][1] * rd;
  r.m[0][1] = -a.m[0][1] * rd;
  r.m[1][0] = -a.m[1][0] * rd;
  r.m[1][1] = a.m[0][0] * rd;
  return r;
}

// General d x d inverse: Gauss-Jordan with partial pivoting, fully unrolled so that every
// register index is static (row swaps are select chains).  Used for (I + C J) in the filtering
// operator when d > 2.
template <int D> __device__ __forceinline__ Mat<D> minv(const Mat<D>& M) {
  Mat<D> a = M, inv = meye<D>();
#pragma unroll
  for (int c = 0; c < D; ++c) {
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
      const bool sw = fabsf(a.m[r][c]) > fabsf(a.m[c][c]);
#pragma unroll
      for (int j = c; j < D; ++j) {
        const float u = a.m[c][j], v = a.m[r][j];
        a.m[c][j] = sw ? v : u;
        a.m[r][j] = sw ? u : v;
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float u = inv.m[c][j], v = inv.m[r][j];
        inv.m[c][j] = sw ? v : u;
        inv.m[r][j] = sw ? u : v;
      }
    }
    const float rp = 1.0f / a.m[c][c];
#pragma unroll
    for (int j = c + 1; j < D; ++j) a.m[c][j] *= rp;
#pragma unroll
    for (int j = 0; j < D; ++j) inv.m[c][j] *= rp;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = a.m[r][c];
#pragma unroll
      for (int j = c + 1; j < D; ++j) a.m[r][j] = fmaf(-f, a.m[c][j], a.m[r][j]);
#pragma unroll
      for (int j = 0; j < D; ++j) inv.m[r][j] = fmaf(-f, inv.m[c][j], inv.m[r][j]);
    }
  }
  return inv;
}

// ---- transition of the trend block: LocalLevel (D=1) T=[1]; LocalLinearTrend (D=2)
// T=[[1,1],[0,1]]  (tfp.sts.LocalLevel / LocalLinearTrend state-space models).
template <int D> __device__ __forceinline__ Mat<D> trans_mat() {
  Mat<D> t = meye<D>();
  if constexpr (D == 2) t.m[0][1] = 1.f;
  return t;
}
template <int D> __device__ __forceinline__ Vec<D> trans_apply(const Vec<D>& x) {
  Vec<D> r = x;
  if constexpr (D == 2) r.v[0] = x.v[0] + x.v[1];
  return r;
}
// T P T' + Q, Q = diag(q)
template <int D>
__device__ __forceinline__ Mat<D> trans_cov(const Mat<D>& p, const Vec<D>& q) {
  Mat<D> r;
  if constexpr (D == 1) {
    r.m[0][0] = p.m[0][0] + q.v[0];
  } else {
    const float s01 = p.m[0][1] + p.m[1][1];
    r.m[0][0] = p.m[0][0] + p.m[0][1] + p.m[1][0] + p.m[1][1] + q.v[0];
    r.m[0][1] = 0.5f * (s01 + p.m[1][0] + p.m[1][1]);
    r.m[1][0] = r.m[0][1];
    r.m[1][1] = p.m[1][1] + q.v[1];
  }
  return r;
}

// ---------------------------------------------------------------------------------
// Filtering element: p(x_k | y_{i:k}, x_{i-1}) = N(A x_{i-1} + b, C),
//                    p(y_{i:k} | x_{i-1})     ~ N_I(eta, J)   (information form).
// ---------------------------------------------------------------------------------
template <int D> struct FElem {
  Mat<D> A;
  Vec<D> b;
  Mat<D> C;
  Vec<D> eta;
  Mat<D> J;
};

template <int D> __device__ __forceinline__ FElem<D> felem_identity() {
  FElem<D> e;
  e.A = meye<D>();
  e.b = vzero<D>();
  e.C = mzero<D>();
  e.eta = vzero<D>();
  e.J = mzero<D>();
  return e;
}

// e1 covers the earlier steps, e2 the later ones.
template <int D>
__device__ __forceinline__ FElem<D> felem_combine(const FElem<D>& e1, const FElem<D>& e2) {
  FElem<D> r;
  const Mat<D> M = madd(meye<D>(), mm(e1.C, e2.J));   // I + C1 J2
  const Mat<D> Mi = minv(M);
  const Mat<D> G = mm(Mi, e1.A);                       // (I + C1 J2)^-1 A1
  const Mat<D> A2Mi = mm(e2.A, Mi);
  r.A = mm(e2.A, G);
  r.b = vadd(mv(A2Mi, vadd(e1.b, mv(e1.C, e2.eta))), e2.b);
  r.C = madd(mmt(mm(A2Mi, e1.C), e2.A), e2.C);
  // (I + J2 C1)^-1 = Mi'  because C1 and J2 are symmetric  =>  A1' (I + J2 C1)^-1 = G'
  r.eta = vadd(mtv(G, vsub(e2.eta, mv(e2.J, e1.b))), e1.eta);
  r.J = madd(mtm(G, mm(e2.J, e1.A)), e1.J);
  symmetrize(r.C);
  symmetrize(r.J);
  return r;
}

// ---- the same element with C and J stored as packed upper triangles, and a combine that
// (i) never forms (I + C1 J2)^-1: one Gauss-Jordan elimination (no pivoting -- I + C1 J2 with
//     C1, J2 positive semi-definite has eigenvalues >= 1; checked against pivoted LAPACK
//     inverses in float32, tools/proto_wide_scan.py) solves for W^-1 [A1 | C1 | b1 + C1 eta2];
// (ii) computes only the upper triangles of the two symmetric results.
// ~2.8k FMAs at d = 7 instead of ~4.2k instructions, and 119 instead of 161 floats to shuffle.
template <int D> struct FElemS {
  Mat<D> A;
  Vec<D> b;
  float C[D * (D + 1) / 2];
  Vec<D> eta;
  float J[D * (D + 1) / 2];
};
template <int D> __host__ __device__ constexpr int symidx(int i, int j) {
  return i <= j ? i * D - i * (i - 1) / 2 + (j - i) : j * D - j * (j - 1) / 2 + (i - j);
}
template <int D> __device__ __forceinline__ FElemS<D> felems_pack(const FElem<D>& e) {
  FElemS<D> r;
  r.A = e.A; r.b = e.b; r.eta = e.eta;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      r.C[symidx<D>(i, j)] = 0.5f * (e.C.m[i][j] + e.C.m[j][i]);
      r.J[symidx<D>(i, j)] = 0.5f * (e.J.m[i][j] + e.J.m[j][i]);
    }
  return r;
}
template <int D> __device__ __forceinline__ FElemS<D> felems_identity() {
  return felems_pack(felem_identity<D>());
}
template <int D>
__device__ __forceinline__ FElemS<D> felems_combine(const FElemS<D>& e1, const FElemS<D>& e2) {
  // W = I + C1 J2 ; right-hand sides RA = A1, RC = C1, u = b1 + C1 eta2
  Mat<D> W, RA = e1.A, RC;
  Vec<D> u;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float ui = e1.b.v[i];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = (i == j) ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(e1.C[symidx<D>(i, k)], e2.J[symidx<D>(k, j)], s);
      W.m[i][j] = s;
      RC.m[i][j] = e1.C[symidx<D>(i, j)];
      ui = fmaf(e1.C[symidx<D>(i, j)], e2.eta.v[j], ui);
    }
    u.v[i] = ui;
  }
  // J2 A1 and eta2 - J2 b1 before the elimination: A1, b1 and J2 are dead while it runs (the
  // combine sits at the register limit for d = 8)
  Mat<D> T2;
  Vec<D> w = e2.eta;
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float sj = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) sj = fmaf(e2.J[symidx<D>(i, k)], e1.A.m[k][j], sj);
      T2.m[i][j] = sj;
      w.v[i] = fmaf(-e2.J[symidx<D>(i, j)], e1.b.v[j], w.v[i]);
    }
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float rp = __builtin_amdgcn_rcpf(W.m[c][c]);   // 1 ulp: ample for float32 moments
#pragma unroll
    for (int j = c + 1; j < D; ++j) W.m[c][j] *= rp;
#pragma unroll
    for (int j = 0; j < D; ++j) { RA.m[c][j] *= rp; RC.m[c][j] *= rp; }
    u.v[c] *= rp;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = W.m[r][c];
#pragma unroll
      for (int j = c + 1; j < D; ++j) W.m[r][j] = fmaf(-f, W.m[c][j], W.m[r][j]);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        RA.m[r][j] = fmaf(-f, RA.m[c][j], RA.m[r][j]);
        RC.m[r][j] = fmaf(-f, RC.m[c][j], RC.m[r][j]);
      }
      u.v[r] = fmaf(-f, u.v[c], u.v[r]);
    }
  }
  FElemS<D> r;
  r.A = mm(e2.A, RA);
  r.b = vadd(mv(e2.A, u), e2.b);
  {
    const Mat<D> T1 = mm(e2.A, RC);               // A2 W^-1 C1
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        float s = e2.C[symidx<D>(i, j)];
#pragma unroll
        for (int k = 0; k < D; ++k) s = fmaf(T1.m[i][k], e2.A.m[j][k], s);
        r.C[symidx<D>(i, j)] = s;
      }
  }
  // r.J = G' (J2 A1) + J1 and r.eta = G' (eta2 - J2 b1) + eta1, G = W^-1 A1 = RA
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float sv = e1.eta.v[i];
#pragma unroll
    for (int k = 0; k < D; ++k) sv = fmaf(RA.m[k][i], w.v[k], sv);
    r.eta.v[i] = sv;
#pragma unroll
    for (int j = i; j < D; ++j) {
      float t = e1.J[symidx<D>(i, j)];
#pragma unroll
      for (int k = 0; k < D; ++k) t = fmaf(RA.m[k][i], T2.m[k][j], t);
      r.J[symidx<D>(i, j)] = t;
    }
  }
  return r;
}


// ---- the MATRIX part (A, C, J) of the filtering element on its own.  The combine of
// (A, b, C, eta, J) never feeds (b, eta) back into (A, C, J): the matrix part is closed under
// composition and depends on the model's variances and the missing-data pattern only -- not on
// the data.  The Gibbs kernels exploit that twice: the covariance side of the Kalman filter
// (P_t, gains, 1/F_t) is scanned as soon as sigma^2_obs is known, while the regression weights
// -- hence the data of the filter -- are still being drawn; and the data side shrinks to two
// scans of affine maps of the MEAN (forward: filtered means, backward: the smoothing adjoint)
// whose matrices come out of the matrix pass.
template <int D> struct FMElem {
  Mat<D> A;
  float C[D * (D + 1) / 2];
  float J[D * (D + 1) / 2];
};
template <int D> __device__ __forceinline__ FMElem<D> fmelem_identity() {
  FMElem<D> r;
  r.A = meye<D>();
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) { r.C[i] = 0.f; r.J[i] = 0.f; }
  return r;
}
// e1 covers the earlier steps, e2 the later ones (felems_combine without its vector parts).
template <int D>
__device__ __forceinline__ FMElem<D> fmelem_combine(const FMElem<D>& e1, const FMElem<D>& e2) {
  Mat<D> W, RA = e1.A, RC;
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = (i == j) ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(e1.C[symidx<D>(i, k)], e2.J[symidx<D>(k, j)], s);
      W.m[i][j] = s;
      RC.m[i][j] = e1.C[symidx<D>(i, j)];
    }
  }
  Mat<D> T2;                                         // J2 A1
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float sj = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) sj = fmaf(e2.J[symidx<D>(i, k)], e1.A.m[k][j], sj);
      T2.m[i][j] = sj;
    }
  }
  // W^-1 [A1 | C1] by unpivoted Gauss-Jordan (W = I + C1 J2, C1 and J2 positive semi-definite:
  // eigenvalues >= 1)
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float rp = __builtin_amdgcn_rcpf(W.m[c][c]);
#pragma unroll
    for (int j = c + 1; j < D; ++j) W.m[c][j] *= rp;
#pragma unroll
    for (int j = 0; j < D; ++j) { RA.m[c][j] *= rp; RC.m[c][j] *= rp; }
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = W.m[r][c];
#pragma unroll
      for (int j = c + 1; j < D; ++j) W.m[r][j] = fmaf(-f, W.m[c][j], W.m[r][j]);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        RA.m[r][j] = fmaf(-f, RA.m[c][j], RA.m[r][j]);
        RC.m[r][j] = fmaf(-f, RC.m[c][j], RC.m[r][j]);
      }
    }
  }
  FMElem<D> r;
  r.A = mm(e2.A, RA);
  {
    const Mat<D> T1 = mm(e2.A, RC);                 // A2 W^-1 C1
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        float s = e2.C[symidx<D>(i, j)];
#pragma unroll
        for (int k = 0; k < D; ++k) s = fmaf(T1.m[i][k], e2.A.m[j][k], s);
        r.C[symidx<D>(i, j)] = s;
      }
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      float t = e1.J[symidx<D>(i, j)];
#pragma unroll
      for (int k = 0; k < D; ++k) t = fmaf(RA.m[k][i], T2.m[k][j], t);
      r.J[symidx<D>(i, j)] = t;
    }
  return r;
}

// T M and M T' for the trend transition (D = 1: identity)
template <int D> __device__ __forceinline__ Mat<D> trans_left(const Mat<D>& m) {
  Mat<D> r = m;
  if constexpr (D == 2) {
#pragma unroll
    for (int j = 0; j < D; ++j) r.m[0][j] = m.m[0][j] + m.m[1][j];
  }
  return r;
}
template <int D> __device__ __forceinline__ Mat<D> trans_right_t(const Mat<D>& m) {
  Mat<D> r = m;
  if constexpr (D == 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) r.m[i][0] = m.m[i][0] + m.m[i][1];
  }
  return r;
}
// T' r
template <int D> __device__ __forceinline__ Vec<D> trans_t_apply(const Vec<D>& r) {
  Vec<D> u = r;
  if constexpr (D == 2) u.v[1] = r.v[0] + r.v[1];
  return u;
}

// Affine map r_out = M r_in + c (backward smoothing recursion).
template <int D> struct AElem {
  Mat<D> M;
  Vec<D> c;
};
template <int D> __device__ __forceinline__ AElem<D> aelem_identity() {
  AElem<D> e;
  e.M = meye<D>();
  e.c = vzero<D>();
  return e;
}
// outer after inner: (outer o inner)(r) = outer.M (inner.M r + inner.c) + outer.c
template <int D>
__device__ __forceinline__ AElem<D> aelem_compose(const AElem<D>& outer, const AElem<D>& inner) {
  AElem<D> r;
  r.M = mm(outer.M, inner.M);
  r.c = vadd(mv(outer.M, inner.c), outer.c);
  return r;
}

// Prior-simulation element: x_out = T^k x_in + s  (constant transition T).
template <int D> struct PElem {
  float k;
  Vec<D> s;
};
template <int D> __device__ __forceinline__ PElem<D> pelem_identity() {
  PElem<D> e;
  e.k = 0.f;
  e.s = vzero<D>();
  return e;
}
template <int D>
__device__ __forceinline__ PElem<D> pelem_combine(const PElem<D>& e1, const PElem<D>& e2) {
  PElem<D> r;
  r.k = e1.k + e2.k;
  r.s = e1.s;
  if constexpr (D == 2) r.s.v[0] = fmaf(e2.k, e1.s.v[1], e1.s.v[0]);  // T^k2 s1
  r.s = vadd(r.s, e2.s);
  return r;
}

}  // namespace ci
