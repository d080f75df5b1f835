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
template <int D> struct QMat {
  static constexpr int H = (D + 3) / 4;
  float m[H][D];          // m[h][i] = M[i][q + 4h]; columns >= D hold zeros
};
template <int D> struct QVec {
  static constexpr int H = (D + 3) / 4;
  float v[H];             // v[h] = x[q + 4h]
};
template <int D> __device__ __forceinline__ QMat<D> qm_zero() {
  QMat<D> r;
#pragma unroll
  for (int h = 0; h < QMat<D>::H; ++h)
#pragma unroll
    for (int i = 0; i < D; ++i) r.m[h][i] = 0.f;
  return r;
}
template <int D> __device__ __forceinline__ QMat<D> qm_eye(int q) {
  QMat<D> r;
#pragma unroll
  for (int h = 0; h < QMat<D>::H; ++h)
#pragma unroll
    for (int i = 0; i < D; ++i) r.m[h][i] = (q + 4 * h == i) ? 1.f : 0.f;
  return r;
}
template <int D> __device__ __forceinline__ QVec<D> qv_zero() {
  QVec<D> r;
#pragma unroll
  for (int h = 0; h < QVec<D>::H; ++h) r.v[h] = 0.f;
  return r;
}
// M[i][j] / x[k] wherever they live (i, j, k constants after unrolling)
template <int D> __device__ __forceinline__ float q_at(const QMat<D>& M, int i, int j) {
  return q_bc(M.m[j >> 2][i], j & 3);
}
template <int D> __device__ __forceinline__ float q_vat(const QVec<D>& x, int k) {
  return q_bc(x.v[k >> 2], k & 3);
}
// a replicated vector -> its own entries (lane-dependent index: a select chain)
template <int D> __device__ __forceinline__ QVec<D> q_own(const float (&rep)[D], int q) {
  QVec<D> r;
#pragma unroll
  for (int h = 0; h < QVec<D>::H; ++h) {
    float s = (4 * h < D) ? rep[4 * h] : 0.f;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float c = (4 * h + k < D) ? rep[(4 * h + k < D) ? 4 * h + k : 0] : 0.f;
      s = (q == k) ? c : s;
    }
    r.v[h] = s;
  }
  return r;
}
template <int D> __device__ __forceinline__ void q_rep(const QVec<D>& x, float (&rep)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) rep[i] = q_vat(x, i);
}

// R = X Y + R0:  X foreign (any column-split matrix), Y and R0 local
template <int D>
__device__ __forceinline__ QMat<D> q_mm_add(const QMat<D>& X, const QMat<D>& Y, const QMat<D>& R0) {
  QMat<D> r = R0;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float f = q_at(X, i, k);
#pragma unroll
      for (int h = 0; h < QMat<D>::H; ++h) r.m[h][i] = fmaf(f, Y.m[h][k], r.m[h][i]);
    }
  return r;
}
// R = X' Y + R0:  X foreign, read transposed
template <int D>
__device__ __forceinline__ QMat<D> q_mtm_add(const QMat<D>& X, const QMat<D>& Y, const QMat<D>& R0) {
  QMat<D> r = R0;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float f = q_at(X, k, i);
#pragma unroll
      for (int h = 0; h < QMat<D>::H; ++h) r.m[h][i] = fmaf(f, Y.m[h][k], r.m[h][i]);
    }
  return r;
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
  QMat<D> r = qm_zero<D>();
#pragma unroll
  for (int rb = 0; rb < H; ++rb)
#pragma unroll
    for (int cb = 0; cb < H; ++cb) {
      // block of rows 4 rb .. 4 rb + 3, columns 4 cb .. 4 cb + 3: lane q holds its column 4 cb + q
      float x[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = (4 * rb + c < D) ? M.m[cb][(4 * rb + c < D) ? 4 * rb + c : 0] : 0.f;
      q_transpose4(x, q);
      // now x[c] = M[4 rb + q][4 cb + c] = M'[4 cb + c][4 rb + q]: column 4 rb + q of M', rows 4 cb + c
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (4 * cb + c < D) r.m[rb][4 * cb + c] = x[c];
    }
  return r;
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
// STATE_ONLY: only b and C of the result (the predicted moments at a chunk start).
template <int D, bool STATE_ONLY = false>
__device__ __forceinline__ QFElem<D> qf_combine(const QFElem<D>& e1, const QFElem<D>& e2, int q) {
  constexpr int H = QMat<D>::H;
  QMat<D> W = q_mm_add(e1.C, e2.J, qm_eye<D>(q));
  QMat<D> G = e1.A, Y = e1.C;
  QVec<D> u = e1.b;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float f = q_vat(e2.eta, k);
#pragma unroll
    for (int h = 0; h < H; ++h) u.v[h] = fmaf(e1.C.m[h][k], f, u.v[h]);     // C1[own][k] = C1[k][own]
  }
  float z[D];
  q_rep(u, z);
  QMat<D> T2;
  QVec<D> w = e2.eta;
  if constexpr (!STATE_ONLY) {
    T2 = q_mm_add(e2.J, e1.A, qm_zero<D>());                               // J2 A1
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float f = q_vat(e1.b, k);
#pragma unroll
      for (int h = 0; h < H; ++h) w.v[h] = fmaf(-e2.J.m[h][k], f, w.v[h]);  // eta2 - J2 b1
    }
  }
  // Gauss-Jordan on the columns each lane owns; the multipliers W[r][c] come from the owner of
  // column c (read before that lane's own update of the same register)
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float rp = __builtin_amdgcn_rcpf(q_at(W, c, c));
#pragma unroll
    for (int h = 0; h < H; ++h) {
      W.m[h][c] *= rp;
      if constexpr (!STATE_ONLY) G.m[h][c] *= rp;
      Y.m[h][c] *= rp;
    }
    z[c] *= rp;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = q_at(W, r, c);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        W.m[h][r] = fmaf(-f, W.m[h][c], W.m[h][r]);
        if constexpr (!STATE_ONLY) G.m[h][r] = fmaf(-f, G.m[h][c], G.m[h][r]);
        Y.m[h][r] = fmaf(-f, Y.m[h][c], Y.m[h][r]);
      }
      z[r] = fmaf(-f, z[c], z[r]);
    }
  }
  QFElem<D> r;
  // b = A2 z + b2 (own rows of A2 = columns of A2')
  r.b = e2.b;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int h = 0; h < H; ++h) r.b.v[h] = fmaf(e2.AT.m[h][k], z[k], r.b.v[h]);
  // C = (A2 Y) A2' + C2:  column j own needs row j of A2 = column j of A2'
  {
    const QMat<D> T1 = q_mm_add(e2.A, Y, qm_zero<D>());
    r.C = q_mm_add(T1, e2.AT, e2.C);
  }
  if constexpr (STATE_ONLY) {
    r.A = e2.A; r.AT = e2.AT; r.J = e2.J; r.eta = e2.eta;
    return r;
  }
  r.A = q_mm_add(e2.A, G, qm_zero<D>());                // A2 G
  // K(A') : A'[i][j] = A[j][i] = sum_k A2[j][k] G[k][i], j own
  r.AT = qm_zero<D>();
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const float f = q_at(G, k, i);
#pragma unroll
      for (int h = 0; h < H; ++h) r.AT.m[h][i] = fmaf(f, e2.AT.m[h][k], r.AT.m[h][i]);
    }
  r.J = q_mtm_add(G, T2, e1.J);                         // G' (J2 A1) + J1
  r.eta = e1.eta;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float f = q_vat(w, k);
#pragma unroll
    for (int h = 0; h < H; ++h) r.eta.v[h] = fmaf(G.m[h][k], f, r.eta.v[h]);   // G[k][own]
  }
  return r;
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
// (outer o inner)(r) = outer.M (inner.M r + inner.c) + outer.c
template <int D>
__device__ __forceinline__ QAElem<D> qa_compose(const QAElem<D>& outer, const QAElem<D>& inner, int q) {
  constexpr int H = QMat<D>::H;
  QAElem<D> r;
  r.M = q_mm_add(outer.M, inner.M, qm_zero<D>());
  const QVec<D> ci = q_own<D>(inner.c, q);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float p = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) p = fmaf(outer.M.m[h][i], ci.v[h], p);   // own columns' share of row i
    r.c[i] = outer.c[i] + q_sum(p);
  }
  return r;
}

// ---- conversions between one-lane and quad-split elements (tests, prior element) -----------------
template <int D> __device__ __forceinline__ QMat<D> qm_from(const Mat<D>& M, int q) {
  QMat<D> r;
#pragma unroll
  for (int h = 0; h < QMat<D>::H; ++h)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * h + k < D) s = (q == k) ? M.m[i][4 * h + k] : s;
      r.m[h][i] = s;
    }
  return r;
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
