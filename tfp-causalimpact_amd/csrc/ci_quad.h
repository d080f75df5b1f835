// ci_quad.h -- small dense algebra (d <= 8) carried by the FOUR LANES OF A QUAD.
//
// The time-parallel trend + seasonal kernel (ci_wide.h) used to give every chunk of the series to
// ONE lane: a combine of two Sarkka filtering elements (A, b, C, eta, J) at d = 7 is ~2.8k dependent
// FMAs on that lane, and the 256-chunk scan pays 9-10 of them in sequence -- 117k of the iteration's
// 556k cycles, on a CU that is VALU-issue bound while seven CUs of the chain's cluster idle
// (round-5 profile).  Here a chunk belongs to a QUAD: every d x d matrix is split by COLUMNS over the
// four lanes (lane q holds columns q and q + 4, all rows in registers), and the one operand of a
// product that lives on another lane comes in through the DPP crossbar (`quad_perm` broadcast, folded
// into the multiply-add: no LDS, no extra instruction).  The rule that makes everything work:
//
//     in every multiply-add at most ONE operand is foreign, and its (row, column) is a compile-time
//     constant; the other operand and the destination are addressed by (own slot, constant index).
//
//   K(XY)  <- X foreign, K(Y) local:        R[.][j] = sum_k X[.][k] Y[k][j],  j own
//   a symmetric matrix is its own transpose: K(C) doubles as the rows of C
//   A is carried twice, K(A) and K(A') (= its rows): the products A2 Y A2' and A1' S A1 need both
//
// One combine is ~1.1k VALU operations per lane at d = 7 (2.6x fewer than on one lane); the same
// layout takes the per-step recursions (the transition T acts on the ROWS of a column-split matrix,
// i.e. inside a lane's registers; only the congruence T C T' moves columns across lanes).
#pragma once
#include "ci_kernels.h"   // Arr<E>, Vec / Mat, the one-lane elements of ci_linalg.h

namespace ci {

// ---- the crossbar ------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float q_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// x of lane k of this quad (k must fold to a constant: call sites are fully unrolled)
__device__ __forceinline__ float q_bc(float x, int k) {
  switch (k & 3) {
    case 0: return q_dpp<0x00>(x);
    case 1: return q_dpp<0x55>(x);
    case 2: return q_dpp<0xAA>(x);
    default: return q_dpp<0xFF>(x);
  }
}
// sum over the quad, on every lane
__device__ __forceinline__ float q_sum(float x) {
  x += q_dpp<0xB1>(x);      // quad_perm [1,0,3,2]
  x += q_dpp<0x4E>(x);      // quad_perm [2,3,0,1]
  return x;
}
// x of the NEXT lane of the quad (lane 3 receives lane 0's)
__device__ __forceinline__ float q_next(float x) { return q_dpp<0x39>(x); }   // quad_perm [1,2,3,0]

// ---- column-split storage -----------------------------------------------------------------------
// The H = ceil(d / 4) columns a lane owns are kept as ONE value per matrix row: a float for d <= 4, a
// 2-vector for d <= 8 -- every row operation is then a packed instruction (v_pk_fma_f32 /
// v_pk_add_f32 / v_pk_mul_f32: two columns per issue slot).
typedef float qf2 __attribute__((ext_vector_type(2)));
template <int H> struct QPk;
template <> struct QPk<1> { typedef float T; };
template <> struct QPk<2> { typedef qf2 T; };
__device__ __forceinline__ float qp_get(float v, int) { return v; }
__device__ __forceinline__ float qp_get(qf2 v, int h) { return h == 0 ? v.x : v.y; }
__device__ __forceinline__ void qp_set(float& v, int, float x) { v = x; }
__device__ __forceinline__ void qp_set(qf2& v, int h, float x) { if (h == 0) v.x = x; else v.y = x; }
__device__ __forceinline__ float qp_hsum(float v) { return v; }
__device__ __forceinline__ float qp_hsum(qf2 v) { return v.x + v.y; }
template <class T> __device__ __forceinline__ T qp_splat(float x);
template <> __device__ __forceinline__ float qp_splat<float>(float x) { return x; }
template <> __device__ __forceinline__ qf2 qp_splat<qf2>(float x) { qf2 r = {x, x}; return r; }
// per-lane pack: g(h) for every own slot
template <class T, class F> __device__ __forceinline__ T qp_make(F g) {
  T r = qp_splat<T>(0.f);
#pragma unroll
  for (int h = 0; h < (int)(sizeof(T) / 4); ++h) qp_set(r, h, g(h));
  return r;
}

template <int D> struct QMat {
  static constexpr int H = (D + 3) / 4;
  typedef typename QPk<H>::T T;
  T r[D];                 // r[i] = (M[i][q], M[i][q + 4]); columns >= D hold zeros
};
template <int D> struct QVec {
  static constexpr int H = (D + 3) / 4;
  typedef typename QPk<H>::T T;
  T v;                    // (x[q], x[q + 4])
};
template <int D> __device__ __forceinline__ QMat<D> qm_zero() {
  QMat<D> m;
#pragma unroll
  for (int i = 0; i < D; ++i) m.r[i] = qp_splat<typename QMat<D>::T>(0.f);
  return m;
}
template <int D> __device__ __forceinline__ QMat<D> qm_eye(int q) {
  QMat<D> m;
#pragma unroll
  for (int i = 0; i < D; ++i)
    m.r[i] = qp_make<typename QMat<D>::T>([&](int h) { return (q + 4 * h == i) ? 1.f : 0.f; });
  return m;
}
template <int D> __device__ __forceinline__ QVec<D> qv_zero() {
  QVec<D> x;
  x.v = qp_splat<typename QVec<D>::T>(0.f);
  return x;
}
// M[i][j] / x[k] wherever they live (i, j, k constants after unrolling)
template <int D> __device__ __forceinline__ float q_at(const QMat<D>& M, int i, int j) {
  return q_bc(qp_get(M.r[i], j >> 2), j & 3);
}
template <int D> __device__ __forceinline__ float q_vat(const QVec<D>& x, int k) {
  return q_bc(qp_get(x.v, k >> 2), k & 3);
}
// a replicated vector -> its own entries (lane-dependent index: a select chain)
template <int D> __device__ __forceinline__ QVec<D> q_own(const float (&rep)[D], int q) {
  QVec<D> x;
  x.v = qp_make<typename QVec<D>::T>([&](int h) {
    float s = (4 * h < D) ? rep[(4 * h < D) ? 4 * h : 0] : 0.f;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float c = (4 * h + k < D) ? rep[(4 * h + k < D) ? 4 * h + k : 0] : 0.f;
      s = (q == k) ? c : s;
    }
    return s;
  });
  return x;
}
template <int D> __device__ __forceinline__ void q_rep(const QVec<D>& x, float (&rep)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) rep[i] = q_vat(x, i);
}

// R = X Y + R0:  X foreign (any column-split matrix), Y and R0 local
template <int D>
__device__ __forceinline__ QMat<D> q_mm_add(const QMat<D>& X, const QMat<D>& Y, const QMat<D>& R0) {
  QMat<D> m = R0;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int i = 0; i < D; ++i) m.r[i] = q_at(X, i, k) * Y.r[k] + m.r[i];
  return m;
}
// R = X' Y + R0:  X foreign, read transposed
template <int D>
__device__ __forceinline__ QMat<D> q_mtm_add(const QMat<D>& X, const QMat<D>& Y, const QMat<D>& R0) {
  QMat<D> m = R0;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int i = 0; i < D; ++i) m.r[i] = q_at(X, k, i) * Y.r[k] + m.r[i];
  return m;
}

// Transpose inside the quad: K(M) -> K(M').  4 x 4 blocks through two butterfly stages (partner
// lane q ^ 1, then q ^ 2), 16 operations per block.
__device__ __forceinline__ void q_transpose4(float (&x)[4], int q) {
  const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
#pragma unroll
  for (int c = 0; c < 4; c += 2) {        // lanes q ^ 1 swap x[c | 1] (even lane) with x[c] (odd lane)
    const float send = b0 ? x[c] : x[c + 1];
    const float got = q_dpp<0xB1>(send);
    if (b0) x[c] = got; else x[c + 1] = got;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {           // lanes q ^ 2 swap x[c + 2] (low lane) with x[c] (high lane)
    const float send = b1 ? x[c] : x[c + 2];
    const float got = q_dpp<0x4E>(send);
    if (b1) x[c] = got; else x[c + 2] = got;
  }
}
template <int D> __device__ __forceinline__ QMat<D> q_transpose(const QMat<D>& M, int q) {
  constexpr int H = QMat<D>::H;
  QMat<D> m = qm_zero<D>();
#pragma unroll
  for (int rb = 0; rb < H; ++rb)
#pragma unroll
    for (int cb = 0; cb < H; ++cb) {
      // block of rows 4 rb .. 4 rb + 3, columns 4 cb .. 4 cb + 3: lane q holds its column 4 cb + q
      float x[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = (4 * rb + c < D) ? qp_get(M.r[(4 * rb + c < D) ? 4 * rb + c : 0], cb) : 0.f;
      q_transpose4(x, q);
      // now x[c] = M[4 rb + q][4 cb + c] = M'[4 cb + c][4 rb + q]: column 4 rb + q of M', rows 4 cb + c
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (4 * cb + c < D) qp_set(m.r[4 * cb + c], rb, x[c]);
    }
  return m;
}

// ---- filtering element (Sarkka & Garcia-Fernandez 2021), quad-split ----------------------------
template <int D> struct QFElem {
  QMat<D> A, AT, C, J;      // K(A), K(A'), and the symmetric C, J
  QVec<D> b, eta;
};
template <int D> __device__ __forceinline__ QFElem<D> qf_identity(int q) {
  QFElem<D> e;
  e.A = qm_eye<D>(q); e.AT = e.A; e.C = qm_zero<D>(); e.J = e.C;
  e.b = qv_zero<D>(); e.eta = e.b;
  return e;
}
// e1 covers the earlier steps, e2 the later ones (felems_combine of ci_linalg.h, same formulas):
//   W = I + C1 J2;  G = W^-1 A1, Y = W^-1 C1, z = W^-1 (b1 + C1 eta2)   (one Gauss-Jordan, no pivoting)
//   A = A2 G;  b = A2 z + b2;  C = A2 Y A2' + C2;  eta = G'(eta2 - J2 b1) + eta1;  J = G' J2 A1 + J1
// STATE_ONLY: only b and C of the result (the predicted moments at a chunk start); of e1 only b and
// C are read then.
template <int D, bool STATE_ONLY = false>
__device__ __forceinline__ QFElem<D> qf_combine(const QFElem<D>& e1, const QFElem<D>& e2, int q) {
  QMat<D> W = q_mm_add(e1.C, e2.J, qm_eye<D>(q));
  QMat<D> G = e1.A, Y = e1.C;
  QVec<D> u = e1.b;
#pragma unroll
  for (int k = 0; k < D; ++k) u.v = e1.C.r[k] * q_vat(e2.eta, k) + u.v;      // C1[own][k] = C1[k][own]
  float z[D];
  q_rep(u, z);
  QMat<D> T2;
  QVec<D> w = e2.eta;
  if constexpr (!STATE_ONLY) {
    T2 = q_mm_add(e2.J, e1.A, qm_zero<D>());                               // J2 A1
#pragma unroll
    for (int k = 0; k < D; ++k) w.v = e2.J.r[k] * (-q_vat(e1.b, k)) + w.v;     // eta2 - J2 b1
  }
  // Gauss-Jordan on the columns each lane owns; the multipliers W[r][c] come from the owner of
  // column c (read before that lane's own update of the same register)
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float rp = __builtin_amdgcn_rcpf(q_at(W, c, c));
    W.r[c] = W.r[c] * rp;
    if constexpr (!STATE_ONLY) G.r[c] = G.r[c] * rp;
    Y.r[c] = Y.r[c] * rp;
    z[c] *= rp;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = -q_at(W, r, c);
      W.r[r] = f * W.r[c] + W.r[r];
      if constexpr (!STATE_ONLY) G.r[r] = f * G.r[c] + G.r[r];
      Y.r[r] = f * Y.r[c] + Y.r[r];
      z[r] = fmaf(f, z[c], z[r]);
    }
  }
  QFElem<D> e;
  // b = A2 z + b2 (own rows of A2 = columns of A2')
  e.b = e2.b;
#pragma unroll
  for (int k = 0; k < D; ++k) e.b.v = e2.AT.r[k] * z[k] + e.b.v;
  // C = (A2 Y) A2' + C2:  column j own needs row j of A2 = column j of A2'
  {
    const QMat<D> T1 = q_mm_add(e2.A, Y, qm_zero<D>());
    e.C = q_mm_add(T1, e2.AT, e2.C);
  }
  if constexpr (STATE_ONLY) {
    e.A = e2.A; e.AT = e2.AT; e.J = e2.J; e.eta = e2.eta;
    return e;
  }
  e.A = q_mm_add(e2.A, G, qm_zero<D>());                // A2 G
  // K(A') : A'[i][j] = A[j][i] = sum_k A2[j][k] G[k][i], j own
  e.AT = q_mtm_add(G, e2.AT, qm_zero<D>());
  e.J = q_mtm_add(G, T2, e1.J);                         // G' (J2 A1) + J1
  e.eta = e1.eta;
#pragma unroll
  for (int k = 0; k < D; ++k) e.eta.v = G.r[k] * q_vat(w, k) + e.eta.v;   // G[k][own]
  return e;
}

// ---- backward affine map r_in = M r_out + c, quad-split (M by columns, c on every lane) ----------
template <int D> struct QAElem {
  QMat<D> M;
  float c[D];
};
template <int D> __device__ __forceinline__ QAElem<D> qa_identity(int q) {
  QAElem<D> e;
  e.M = qm_eye<D>(q);
#pragma unroll
  for (int i = 0; i < D; ++i) e.c[i] = 0.f;
  return e;
}
// M x for a replicated x: every lane adds its own columns' share, the quad sums
template <int D>
__device__ __forceinline__ void q_mv_acc(const QMat<D>& M, const float (&xr)[D], int q, float (&acc)[D]) {
  const QVec<D> xo = q_own<D>(xr, q);
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] += q_sum(qp_hsum(M.r[i] * xo.v));
}
// (outer o inner)(r) = outer.M (inner.M r + inner.c) + outer.c
template <int D>
__device__ __forceinline__ QAElem<D> qa_compose(const QAElem<D>& outer, const QAElem<D>& inner, int q) {
  QAElem<D> e;
  e.M = q_mm_add(outer.M, inner.M, qm_zero<D>());
#pragma unroll
  for (int i = 0; i < D; ++i) e.c[i] = outer.c[i];
  q_mv_acc(outer.M, inner.c, q, e.c);
  return e;
}

// ---- conversions between one-lane and quad-split elements (tests, prior element) -----------------
template <int D> __device__ __forceinline__ QMat<D> qm_from(const Mat<D>& M, int q) {
  QMat<D> m;
#pragma unroll
  for (int i = 0; i < D; ++i)
    m.r[i] = qp_make<typename QMat<D>::T>([&](int h) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * h + k < D) s = (q == k) ? M.m[i][4 * h + k] : s;
      return s;
    });
  return m;
}

// Every float of a struct moved between lanes: ONE source address per lane, then the ds_bpermutes
// back to back (the LDS crossbar pipelines them; __shfl_up recomputes and range-checks the source
// lane per value and waits for each result: ~75 cycles per float measured, 60 floats per element).
// A lane whose source would fall outside the wavefront reads its own value (callers mask it).
template <class E> __device__ __forceinline__ E q_perm(const E& e, int src_lane) {
  const int addr = src_lane << 2;
  Arr<E> a = __builtin_bit_cast(Arr<E>, e);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i)
    a.f[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a.f[i])));
  return __builtin_bit_cast(E, a);
}
template <class E> __device__ __forceinline__ E q_shfl_up(const E& e, int quads) {
  const int lane = (int)__lane_id();
  const int src = lane - 4 * quads;
  return q_perm(e, src < 0 ? lane : src);
}
template <class E> __device__ __forceinline__ E q_shfl_down(const E& e, int quads) {
  const int lane = (int)__lane_id();
  const int src = lane + 4 * quads;
  return q_perm(e, src > 63 ? lane : src);
}
// lane (4 * quad + q) of the wave, for every float
template <class E> __device__ __forceinline__ E q_shfl_from(const E& e, int quad, int q) {
  return q_perm(e, 4 * quad + q);
}

}  // namespace ci
