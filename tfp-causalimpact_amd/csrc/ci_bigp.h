// ci_bigp.h -- the spike-and-slab regression draw for MORE than MAXP (52) design columns, by a whole
// workgroup: the arithmetic of spike_slab_draw_big (ci_kernels.h; oracle: ci_oracle.c
// spike_slab_step; gibbs_sampler._resample_weights at causalimpact_lib.py:365) with the swept
// matrices as packed triangles in LDS.  Shared by the time-parallel general kernel
// (ci_seasonal_tp.h) and the BIGP builds of the trend + one-block kernel (ci_wide.h).
#pragma once
#include "ci_kernels.h"

namespace ci {

#ifndef CI_LDS
#define CI_LDS __attribute__((address_space(3)))
#define CI_GLB __attribute__((address_space(1)))
#endif
// generic -> LDS: the LDS offset is the low half of the generic address (no null-check sequence)
template <class T> __device__ __forceinline__ CI_LDS T* bp_lds(const void* generic) {
  return (CI_LDS T*)(unsigned)(unsigned long long)generic;
}
// A generic pointer into LDS whose origin the optimiser may not look through (see tp_opaque,
// ci_seasonal_tp.h: the null check of an inferred flat -> local cast is lowered to an illegal
// "V_CMP_NE_U32 0, $src_shared_base" in some instantiations)
template <class T> __device__ __forceinline__ T* bp_opaque(T* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (T*)v;
}
// hand-off inside one workgroup (LDS and, through the CU's one vector L1, global memory)
__device__ __forceinline__ void bp_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// doubles of LDS behind `trow`: two saved pivot rows (+ divided by their pivots), the weights' solve,
// the posterior means
__host__ __device__ constexpr size_t bigp_trow_doubles(int P) { return 6 * ((size_t)P + 1); }
// entries of the packed upper triangle of an m x m matrix
__host__ __device__ constexpr size_t bigp_packed(int m) { return (size_t)m * (m + 1) / 2; }

// ------------------------------------------------------------------------------------
// spike_slab_draw_big (ci_kernels.h: any number of covariates, every O(P^2) array in the chain's HBM
// workspace) executed by the WHOLE workgroup.  On one wavefront the draw costs 1.1M cycles per
// iteration at P = 101 (28 sweeps of two 100 x 100 float64 matrices through L2 by 64 lanes) -- slower
// than the oracle on one host core.  Here every wavefront takes the same decisions from the same
// state (the proposals of 64 visiting positions are evaluated redundantly: they are cheap), and
// everything that walks a matrix -- building it, the sweeps, the Cholesky factor of the included
// block -- is spread over all NTH threads, a workgroup barrier where the one-wavefront form has a
// wave-level one.  Entry for entry the arithmetic of spike_slab_draw_big / sweep_big (a sweep is
// one pass here: the pivot row and column are written in the same pass as the general entries).
// Every thread returns the same new observation-noise scale.
// ------------------------------------------------------------------------------------
template <int NTH, class PA, class PP>
__device__ __noinline__ double spike_slab_draw_big_wg(const RegLds& R, PA A, PP Pm, CI_LDS double* trow,
                                                         CI_LDS unsigned* ijtab, float* w, int P,
                                                         const DevSeriesParams& sp, double prev_obs_scale,
                                                         double g_obs, const Rng& rng, uint32_t iter,
                                                         int tid, bool first, Prof& prof, int slot_in,
                                                         int slot0) {
  const int lane = tid & 63, wv = tid >> 6;
  constexpr int NWV_ = NTH / 64;
  // phase budget: what came before goes to `slot_in`, this draw's five phases to slot0 ... slot0 + 4
  prof.tick(slot_in);
  const int n = P + 1;
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  // (round 6) Everything the draw reads many times per sweep lives in LDS: the saved pivot rows, the
  // active-set flags, the visiting order, the right-hand side of the weights' solve.  They used to
  // sit in the chain's HBM workspace with the matrices (bigp_point): every row of every sweep then
  // waited for an L2 round trip on its pivot-row entry -- 18k cycles per sweep at P = 101, 546k per
  // iteration.
  CI_LDS double* ta = trow;                // saved pivot row of A   [n]
  CI_LDS double* tp = trow + n;            // saved pivot row of Pm  [n]
  CI_LDS double* zv = trow + 2 * n;        // normals -> solution of the weights' solve; posterior means behind it
  CI_LDS double* mean = trow + 3 * n;
  CI_LDS double* tra = trow + 4 * n;       // the pivot rows divided by their pivots
  CI_LDS double* trp = trow + 5 * n;
  CI_LDS double* uperm = bp_lds<double>(R.uperm);
  CI_LDS int* nz = bp_lds<int>(R.nz);
  CI_LDS int* perm = bp_lds<int>(R.perm);
  CI_LDS int* idx = bp_lds<int>(R.idx);
  CI_GLB const double* omega = (CI_GLB const double*)R.omega;
  CI_GLB const double* xtx = (CI_GLB const double*)R.xtx;
  CI_LDS const double* bvec = bp_lds<double>(R.bvec);
  // The two swept matrices are SYMMETRIC and stay so under the sweep operator: only the upper
  // triangle is stored, COLUMN by column -- entry (i, j), i <= j, at j (j + 1) / 2 + i -- so that the
  // P x P prior block's layout is a prefix of the (P + 1) x (P + 1) one (half the LDS: both matrices
  // fit next to the wavefronts' areas up to P ~ 125; half the entries to sweep).  A sweep is ONE FLAT
  // LOOP over the entries, thread e, e + NTH, ...: perfectly balanced, four entries in flight, and
  // the entry's (i, j) comes from a table in LDS (`ijtab`, rebuilt per call: it lives in the overlay)
  // -- the row-by-wavefront / column-by-lane walk it replaces spent 37 instructions per entry on
  // index arithmetic, tail masks and the pivot row / column special cases (13k cycles per sweep at
  // P = 101).  Without room for the table (`ijtab` null) the pair is recomputed from e.
  const int EA = (n * (n + 1)) >> 1, EP = (P * (P + 1)) >> 1;
  auto ij_of = [](int e, int& i, int& j) {
    j = (int)((__fsqrt_rn((float)(8 * e + 1)) - 1.0f) * 0.5f);
    if (((j * (j + 1)) >> 1) > e) --j;
    if ((((j + 1) * (j + 2)) >> 1) <= e) ++j;
    i = e - ((j * (j + 1)) >> 1);
  };
  if (ijtab)
    for (int e = tid; e < EA; e += NTH) {
      int i, j;
      ij_of(e, i, j);
      ijtab[e] = (unsigned)i | ((unsigned)j << 16);
    }
  auto sym_at = [](int i, int j) { return ((j * (j + 1)) >> 1) + i; };     // i <= j
  // every walk below: four entries per thread in flight; body(e, i, j) after the loads of `pre`
  // (the table-or-not choice is made OUTSIDE the loop: inside it every table read was followed by its
  // own wait and a branch)
  constexpr int FE_U = 8;      // entries per thread in flight (4: 6.3k cycles per sweep at P = 101)
  auto for_entries_t = [&](auto tab, int E, auto pre, auto body) {
    for (int e0 = tid; e0 < E; e0 += FE_U * NTH) {
      int ii[FE_U], jj[FE_U];
      if constexpr (decltype(tab)::value) {
        unsigned v[FE_U];
#pragma unroll
        for (int u = 0; u < FE_U; ++u) v[u] = ijtab[e0 + u * NTH < E ? e0 + u * NTH : E - 1];
#pragma unroll
        for (int u = 0; u < FE_U; ++u) { ii[u] = (int)(v[u] & 0xFFFFu); jj[u] = (int)(v[u] >> 16); }
      } else {
#pragma unroll
        for (int u = 0; u < FE_U; ++u) ij_of(e0 + u * NTH < E ? e0 + u * NTH : E - 1, ii[u], jj[u]);
      }
#pragma unroll
      for (int u = 0; u < FE_U; ++u) pre(u, e0 + u * NTH < E ? e0 + u * NTH : E - 1, ii[u], jj[u]);
#pragma unroll
      for (int u = 0; u < FE_U; ++u)
        if (e0 + u * NTH < E) body(u, e0 + u * NTH, ii[u], jj[u]);
    }
  };
  auto for_entries = [&](int E, auto pre, auto body) {
    if (ijtab) for_entries_t(std::true_type{}, E, pre, body);
    else for_entries_t(std::false_type{}, E, pre, body);
  };
  if (ijtab) bp_barrier();
  {
    double om[FE_U], xx[FE_U];
    for_entries(EA,
                [&](int u, int, int i, int j) {
                  const int ic = i < P ? i : P - 1, jc = j < P ? j : P - 1;
                  om[u] = omega[ic * P + jc]; xx[u] = xtx[ic * P + jc];
                },
                [&](int u, int e, int i, int j) {
                  A[e] = j < P ? om[u] * prev_var + xx[u] : bvec[i];      // border column j = P: b_i, corner: b_P
                });
    if (first)
      for_entries(EP, [&](int u, int, int i, int j) { om[u] = omega[i * P + j]; },
                  [&](int u, int e, int, int) { Pm[e] = om[u]; });
  }
  for (int j = tid; j < P; j += NTH) {
    nz[j] = all_in ? 1 : (w[j] != 0.f ? 1 : 0);
    if (!all_in) uperm[j] = uniform_d(rng, iter, SITE_PERM, 0, (uint32_t)j);
  }
  bp_barrier();
  // t: the saved pivot row, tr = t / pivot (both [m], LDS).  The general entries in the flat loop;
  // the pivot row / column (m entries, disjoint from what the flat loop writes) by thread j.
  auto sweep_one_wg = [&](auto M, int m, int E, CI_LDS const double* t, CI_LDS const double* tr, int k, double sgn) {
    const double rd = 1.0 / t[k];
    double mv[FE_U], ti[FE_U], tj[FE_U];
    for_entries(E,
                [&](int u, int e, int i, int j) { mv[u] = M[e]; ti[u] = tr[i]; tj[u] = t[j]; },
                [&](int u, int e, int i, int j) {
                  if (i != k && j != k) M[e] = mv[u] - ti[u] * tj[u];
                });
    for (int j = tid; j < m; j += NTH)
      M[j <= k ? ((k * (k + 1)) >> 1) + j : ((j * (j + 1)) >> 1) + k] = (j == k) ? -rd : sgn * t[j] * rd;
  };
  auto sweep_both = [&](int k, bool reverse, bool with_prior) {
    const double sgn = reverse ? -1.0 : 1.0;
    {
      const double rda = 1.0 / A[sym_at(k, k)];
      for (int j = tid; j < n; j += NTH) {
        const double v = j >= k ? A[sym_at(k, j)] : A[sym_at(j, k)];
        ta[j] = v; tra[j] = v * rda;
      }
    }
    if (with_prior) {
      const double rdp = 1.0 / Pm[sym_at(k, k)];
      for (int j = tid; j < P; j += NTH) {
        const double v = j >= k ? Pm[sym_at(k, j)] : Pm[sym_at(j, k)];
        tp[j] = v; trp[j] = v * rdp;
      }
    }
    bp_barrier();
    sweep_one_wg(A, n, EA, ta, tra, k, sgn);
    if (with_prior) sweep_one_wg(Pm, P, EP, tp, trp, k, sgn);
    bp_barrier();
  };
  for (int j0 = 0; j0 < P; j0 += 64) {
    unsigned long long todo = __ballot(j0 + lane < P && nz[j0 + lane < P ? j0 + lane : 0] != 0);
    for (; todo != 0ull; todo &= todo - 1ull) sweep_both(j0 + __ffsll((long long)todo) - 1, false, first);
  }
  prof.tick(slot0);       // build + sweep-in of the current model
  if (!all_in) {
    for (int j = tid; j < P; j += NTH) {
      const double uj = uperm[j];
      int rank = 0;
      for (int k = 0; k < P; ++k) {
        const double uk = uperm[k];
        rank += (uk < uj || (uk == uj && k < j)) ? 1 : 0;
      }
      perm[rank] = j;
    }
    bp_barrier();
    const double logit_pi = log(sp.nonzero_prob) - log1p(-sp.nonzero_prob);
    int s_cur = 0;
    while (s_cur < P) {
      const int base = s_cur & ~63;
      const int pos = base + lane;
      bool flip = false;
      if (pos < P && pos >= s_cur) {
        const int j = perm[pos];
        const bool in = nz[j] != 0;
        const double ajj = A[sym_at(j, j)], ajb = A[sym_at(j, P)], corner = A[sym_at(P, P)];
        const double pju = Pm[sym_at(j, j)];
        const double beta_old = sp.obs_scale + 0.5 * corner;
        double delta;
        if (!in) {
          const double beta_new = sp.obs_scale + 0.5 * (corner - ajb * ajb / ajj);
          delta = 0.5 * log(pju * prev_var) - 0.5 * log(ajj) + logit_pi -
                  (a_post - 1.0) * (log(beta_new) - log(beta_old));
        } else {
          const double V = -ajj, Vp = -pju / prev_var;
          const double beta_new = sp.obs_scale + 0.5 * (corner + ajb * ajb / V);
          delta = 0.5 * log(Vp) - 0.5 * log(V) - logit_pi -
                  (a_post - 1.0) * (log(beta_new) - log(beta_old));
        }
        const double u = uniform_d(rng, iter, SITE_FLIP, 0, (uint32_t)pos);
        flip = u < 1.0 / (1.0 + exp(-delta));
      }
      const unsigned long long bal = __ballot(flip);
      if (bal == 0ull) { s_cur = base + 64; continue; }
      const int s_star = base + __ffsll((long long)bal) - 1;
      const int j = perm[s_star];
      const bool in = nz[j] != 0;
      bp_barrier();                       // every wavefront has read the state it decided on
      sweep_both(j, in, true);
      if (tid == 0) nz[j] = in ? 0 : 1;
      bp_barrier();
      s_cur = s_star + 1;
    }
  }
  prof.tick(slot0 + 1);   // visiting order, proposals, accepted flips
  const double beta_post = sp.obs_scale + 0.5 * A[sym_at(P, P)];
  double var = beta_post / g_obs;
  if (var > sp.obs_ub) var = sp.obs_ub;
  const double new_scale = sqrt(var);
  // active set in increasing feature order (every wavefront forms the same list; wave 0 stores it)
  int na = 0;
  for (int j0 = 0; j0 < P; j0 += 64) {
    const int j = j0 + lane;
    const int mynz = j < P ? nz[j] : 0;
    const unsigned long long bal = __ballot(mynz != 0);
    if (mynz && tid < 64) idx[na + __popcll(bal & ((1ull << lane) - 1ull))] = j;
    na += __popcll(bal);
  }
  for (int j = tid; j < P; j += NTH) w[j] = 0.f;
  bp_barrier();
  // The posterior means (border column of the swept matrix) are saved, and the included block's
  // Cholesky factor is built IN PLACE OF A: in LDS whenever A is.
  for (int i = tid; i < na; i += NTH) {
    mean[i] = A[sym_at(idx[i], P)];
    zv[i] = normal_d(rng, iter, SITE_WEIGHTS, 0, (uint32_t)idx[i]);
  }
  bp_barrier();
  CI_LDS double* ldiag = tp;
  auto factor_and_solve = [&](auto Lm) {
    for (int i0 = wv; i0 < na; i0 += 4 * NWV_)
      for (int j = lane; j < na; j += 64) {
        double om[4], xx[4];
        const int fj = idx[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NWV_;
          const int fi = idx[i < na ? i : na - 1];
          om[u] = omega[fi * P + fj]; xx[u] = xtx[fi * P + fj];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NWV_;
          if (i < na) Lm[i * na + j] = om[u] * prev_var + xx[u];
        }
      }
    bp_barrier();
    prof.tick(slot0 + 2); // active set, normals, the included block
    // right-looking, one barrier per column: every thread scales its own operands by 1 / L_kk (the
    // same quotients whoever computes them), column k of L goes to the UPPER triangle (row k) and the
    // diagonal to `tp`, so nothing a concurrent thread still reads is overwritten
    for (int k = 0; k < na; ++k) {
      // 16 x 16 tiles of threads over the trailing block; multiplications by 1 / L_kk (one division
      // per column instead of two per entry: float64 division is ~40 instructions here)
      const double skk = Lm[k * na + k];
      const double dkk = sqrt(skk);
      const double rs = 1.0 / dkk;
      for (int i = k + 1 + (tid >> 4); i < na; i += NTH / 16) {
        const double lik = Lm[i * na + k] * rs;
        for (int j = k + 1 + (tid & 15); j <= i; j += 16) Lm[i * na + j] -= lik * (Lm[j * na + k] * rs);
        if ((tid & 15) == 0) Lm[k * na + i] = lik;
      }
      if (tid == 0) ldiag[k] = dkk;
      bp_barrier();
    }
    prof.tick(slot0 + 3); // Cholesky factor
    // L' u = z by column-oriented back substitution: L_ik sits at row k, column i of the upper triangle
    if (na <= 64) {
      // one wavefront, z_l in lane l's registers, u_i broadcast by v_readlane: no barriers (the same
      // quotients and multiply-subtracts as the loop below)
      if (wv == 0) {
        double z = lane < na ? zv[lane] : 0.0;
        double lnext = (na > 0 && lane < na - 1) ? Lm[lane * na + (na - 1)] : 0.0;
        for (int i = na - 1; i >= 0; --i) {
          const double li = lnext;
          if (i > 0) lnext = lane < i - 1 ? Lm[lane * na + (i - 1)] : 0.0;
          const double ui = readlane_d(z, i) / ldiag[i];
          if (lane == i) z = ui;
          if (lane < i) z -= li * ui;
        }
        if (lane < na) zv[lane] = z;
      }
      bp_barrier();
    } else {
      for (int i = na - 1; i >= 0; --i) {
        const double ui = zv[i] / ldiag[i];
        bp_barrier();
        if (tid == 0) zv[i] = ui;
        for (int k = tid; k < i; k += NTH) zv[k] -= Lm[k * na + i] * ui;
        bp_barrier();
      }
    }
  };
  // (the factor takes the place of the packed A when its na x na entries fit there, else the chain's
  // HBM workspace: only when more than ~70 % of the columns are in the model)
  if (na * na <= ((n * (n + 1)) >> 1)) factor_and_solve(A);
  else factor_and_solve((CI_GLB double*)R.chol);
  for (int i = tid; i < na; i += NTH) w[idx[i]] = (float)(mean[i] + new_scale * zv[i]);
  bp_barrier();
  prof.tick(slot0 + 4);   // back substitution, weights
  return new_scale;
}

}  // namespace ci
