// ci_seasonal_tp.h -- TIME-PARALLEL Gibbs kernel for ANY list of seasonal blocks (and for trend
// models with more design columns than the register / LDS regression blocks hold).
//
// Reference path: causalimpact/causalimpact_lib.py:365-388 (gibbs_sampler.fit_with_gibbs_sampling)
// on the model of :455-500; the reference's own multi-block test model (4 + 7 + 6 seasons) is
// causalimpact_lib_test.py:738-752.  Until round 5 every model other than "trend + one block of
// 2-7 seasons" ran on ci_seasonal.h: one wavefront per chain, sequential in time -- 18.7 ms per
// Gibbs iteration at T = 10^4, i.e. the speed of one host core.
//
// Same arithmetic, cut in time:
//   * the series is cut into N = G x 8 CHUNKS of Lc consecutive steps; a chain owns a CLUSTER of
//     G workgroups (CUs) of 8 wavefronts, one chunk per wavefront;
//   * the state keeps ci_seasonal.h's layout: lane i of a wavefront holds component i of every
//     state-sized vector and ROW i of every D x D matrix (D = 1 + slope + sum of num_seasons <= 32),
//     blocks in their full n-effect form in SLOT coordinates (transition = identity on a block,
//     rank-one drift shocks, observation row e_0 + sum_k e_{off[k] + c_k(t)});
//   * every recursion of the Durbin-Koopman draw becomes
//         per-chunk pass  ->  scan over the chunks  ->  per-chunk pass:
//       - prior simulation x+: chunk sums, prefix by addition;
//       - Kalman filter: each chunk builds its Sarkka & Garcia-Fernandez element (A, b, C, eta, J)
//         by rank-one folds of its steps (tp_build_pass: three row sweeps per step instead of the
//         filter's one); the elements are scanned (Kogge-Stone inside the workgroup, then over the
//         G workgroup totals) with a WAVE-COOPERATIVE combine -- matrix products with one operand
//         broadcast from LDS, Gauss-Jordan on register rows with v_readlane pivots; every chunk
//         then replays the plain filter from its true predicted moments (tp_filter_pass ==
//         seasonal_filter_pass on a range), storing K_t and v_t / F_t;
//       - backward recursion r: the chunk's affine map is M = (I + J P_start)^-1 A' -- the transpose
//         of the closed-loop propagator A (I + P_start J)^-1, read off the chunk's element, no
//         per-step accumulation -- and its offset is one zero-input backward pass; suffix scan of
//         (M, c); second backward pass from the true r at the chunk's end;
//       - forward reconstruction x^_{t+1} = T x^_t + Q_t r_t from x^ = a + P r at the chunk's start.
//   * X~'targets, X w, the emission of the previous draw and the normals are split by chunks too;
//     the regression draw and the scale draws stay on wavefront 0 of the chain's first workgroup
//     (the one-wavefront blocks of ci_kernels.h: registers for P <= 16, LDS to 52, HBM workspace
//     beyond), the other wavefronts wait at a cluster barrier.
// Elements travel through a per-chain HBM workspace (L2-resident); inside a workgroup the hand-off
// is a workgroup barrier, across workgroups a counter per workgroup in HBM (ci_wide.h's scheme:
// L2-local when the cluster sits on one XCD).  If the cluster cannot be assembled (another stream
// holds the CUs) the first workgroup runs all N chunks itself: same bits.
// The random stream is ci_seasonal.h's, site for site: draws agree with the oracle per random
// number (tests/test_gpu_gibbs.py), and with the sequential kernel up to summation order.
#pragma once
#include "ci_kernels.h"
#include "ci_seasonal.h"
#include "ci_bigp.h"

namespace ci {

constexpr int TP_NWV = 4;                  // wavefronts (chunks) per workgroup: ONE per SIMD
constexpr int TP_NT = TP_NWV * 64;
constexpr int TP_MAXG = 32;                // workgroups per chain
constexpr int TP_MAXD = 32;                // widest state
constexpr int TPC_INTS = 64;               // handshake ints per chain: [0,16) barrier | 16 mode | [32,64) check-in, one per role
constexpr int TPC_MODE = 16, TPC_XCC = 32;
static_assert(TPC_XCC + TP_MAXG <= TPC_INTS, "one check-in slot per workgroup of the cluster");
constexpr int TP_STAT = 32;                // floats of statistics / boundary record per chunk

// columns of the register rows: the state rounded up to a multiple of 4 (at least 8).  (Products and
// eliminations of the scan scale with the square of it: 20 instead of 24 columns at D = 18.)
__host__ __device__ inline int tp_nr(int D) { const int n = (D + 3) & ~3; return n < 8 ? 8 : n; }
// row stride of the LDS mirrors: 16-byte rows, = 4 mod 8 floats (conflict-free run-time columns)
__host__ __device__ inline int tp_ds(int NR) { return ((NR + 7) & ~7) + 4; }
__host__ __device__ inline int tp_log2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
// floats of one forward element (rows of [A | C | J | b eta 0 0]) and one backward map ([M | c 0 0 0])
__host__ __device__ inline size_t tp_esz(int NR) { return (size_t)NR * (3 * NR + 4); }
__host__ __device__ inline size_t tp_bsz(int NR) { return (size_t)NR * (NR + 4); }

// Per-chain HBM workspace (bytes from its base).
struct TpLayout {
  size_t yv, lev, slp, xw, ytil, vf, zl, zs, zo, seas, zk, gd, kf, rs, mask, cbits, cidx;   // over time
  size_t e0, ei, et, st, bm, bi, bt, xsum, stat, cpart, cw, big, total;
  int Lc, TP, N, LVI, LVG;
};
__host__ __device__ inline TpLayout make_tplayout(int T, int P, int K, int D, int has_slope, int G) {
  TpLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
  const int N = G * TP_NWV;
  int Lc = (T + N - 1) / N;
  Lc = (Lc + 3) & ~3;
  l.Lc = Lc; l.N = N; l.TP = N * Lc; l.LVI = tp_log2(TP_NWV); l.LVG = tp_log2(G);
  const size_t Tf = sizeof(float) * (size_t)l.TP;
  const int Kp = K > 0 ? K : 1, NR = tp_nr(D);
  l.yv = take(Tf); l.lev = take(Tf); l.slp = take(has_slope ? Tf : 16); l.xw = take(Tf);
  l.ytil = take(Tf); l.vf = take(Tf); l.zl = take(Tf); l.zs = take(has_slope ? Tf : 16); l.zo = take(Tf);
  l.seas = take(Tf * Kp); l.zk = take(Tf * Kp); l.gd = take(Tf * Kp);
  l.kf = take(Tf * D); l.rs = take(Tf * D + 256);
  l.mask = take((size_t)l.TP); l.cbits = take((size_t)l.TP); l.cidx = take((size_t)l.TP * Kp);
  const size_t E = sizeof(float) * tp_esz(NR), Bz = sizeof(float) * tp_bsz(NR);
  l.e0 = take(E * N);                         // the chunks' own elements (kept for the backward maps)
  l.ei = take(E * N * l.LVI);                 // Kogge-Stone levels inside the workgroups
  l.et = take(E * G * (l.LVG + 1));           // ... over the workgroup totals
  l.st = take(sizeof(float) * (size_t)N * NR * (NR + 4));   // chunk-start moments [P rows | a 0 0 0]
  l.bm = take(Bz * N); l.bi = take(Bz * N * l.LVI); l.bt = take(Bz * G * (l.LVG + 1));
  l.xsum = take(sizeof(float) * (size_t)N * 2 * NR);        // x+ chunk sums | x+ at the chunk's start
  l.stat = take(sizeof(float) * (size_t)N * TP_STAT);
  l.cpart = take(sizeof(float) * (size_t)N * ((P + 4) & ~3));
  l.cw = take(sizeof(float) * (size_t)(((P + 3) & ~3) + 32));
  l.big = take(P > MAXP ? bigp_workspace_bytes(P) : 16);
  l.total = o;
  return l;
}

// LDS of one workgroup: per-wavefront areas for the passes, overlaid (never live together) with the
// regression block's buffers of wavefront 0; a few floats shared by the workgroup.
struct TpLds {
  size_t wave0, wave_stride, cm, am, scr, pzv, vb, reg, shared, total;
  size_t big_a, big_p;   // P > MAXP: the swept matrices of the regression block in LDS (0: HBM workspace)
  size_t ijtab;          //   and the (i, j) table of the sweeps' flat entry loop (0: recomputed per entry)
  // regression block (offsets from `reg`)
  size_t xtx, omega, bvec, aug0, pri0, chol, zv, uperm, nz, perm, idx, w, trow;
};
__host__ __device__ inline TpLds make_tplds(int P, int D) {
  TpLds l;
  const int NR = tp_nr(D), DS = tp_ds(NR);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  l.cm = take(sizeof(float) * D * DS);            // covariance rows (mirror of the register rows)
  l.am = take(sizeof(float) * D * DS);            // A' rows (mirror)
  l.scr = take(sizeof(float) * NR * NR);          // broadcast operand of the cooperative products
  l.pzv = take(sizeof(float) * (72 + 72 + SMAXK * 72 + 16));   // C z | A'z | shock vectors | observed columns
  l.vb = take(sizeof(float) * 2 * NR);            // broadcast vectors
  l.wave_stride = o;
  // regression block of wavefront 0 (as make_slayout)
  o = 0;
  const int Pp = P > 0 ? P : 1;
  const bool bigp = P > MAXP, big = P > 16 && !bigp;
  // (X'X and Omega are read where the setup kernel left them, in global memory: constants would not
  // survive the overlay)
  l.xtx = 0; l.omega = 0;
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.aug0 = take(big ? sizeof(double) * sweep_padded((size_t)(Pp + 1) * (Pp + 1)) : 16);
  l.chol = take(big ? sizeof(double) * Pp * Pp : 16);
  l.zv = take(big ? sizeof(double) * Pp : 16);
  // (P > MAXP: the draw's small state in LDS too -- the matrices may live in the HBM workspace, what
  // every row of every sweep reads may not: tp_spike_slab_draw_big_wg)
  l.uperm = take(big || bigp ? sizeof(double) * Pp : 16);
  l.nz = take(big || bigp ? sizeof(int) * Pp : 16);
  l.perm = take(big || bigp ? sizeof(int) * Pp : 16);
  l.idx = take(big || bigp ? sizeof(int) * Pp : 16);
  l.trow = take(bigp ? sizeof(double) * 6 * (Pp + 1) : 16);   // two saved pivot rows (+ divided by their pivots), the weights' solve, the posterior means
  // P > MAXP (spike_slab_draw_big's workspace block): one CU pulls ~20 B/clk out of L2, i.e. a sweep
  // of a 100 x 100 float64 matrix costs 8k cycles there; in LDS it is bandwidth for free.  The
  // augmented matrix A ((P+1)^2 doubles, rebuilt every iteration: overlay) goes to LDS when it
  // fits next to the rest, the swept prior block Pm (P^2 doubles, CARRIED between iterations:
  // outside the overlay) when both fit.
  l.big_a = 0; l.big_p = 0;
  // (upper triangles: tp_spike_slab_draw_big_wg)
  const size_t a_bytes = sizeof(double) * ((size_t)(Pp + 1) * (Pp + 2) / 2), p_bytes = sizeof(double) * ((size_t)Pp * (Pp + 1) / 2);
  const size_t waves = (size_t)TP_NWV * l.wave_stride;
  const size_t fixed = sizeof(float) * (Pp > 16 ? Pp : 16) + 16 + sizeof(float) * 64 + 256;
  // (the sweeps' (i, j) table: one unsigned per stored entry of A; preferred over Pm, which only the
  // accepted flips sweep)
  const size_t t_bytes = sizeof(unsigned) * ((size_t)(Pp + 1) * (Pp + 2) / 2);
  bool a_lds = false, p_lds = false, t_lds = false;
  if (bigp) {
    size_t with_a = (o + a_bytes > waves ? o + a_bytes : waves) + fixed;
    a_lds = with_a <= 156 * 1024;
    if (a_lds) {
      const size_t with_t = (o + a_bytes + t_bytes > waves ? o + a_bytes + t_bytes : waves) + fixed;
      t_lds = with_t <= 156 * 1024;
      if (t_lds) with_a = with_t;
    }
    p_lds = a_lds && with_a + p_bytes <= 156 * 1024;
  }
  if (a_lds) l.big_a = take(a_bytes);
  l.ijtab = 0;
  if (t_lds) l.ijtab = take(t_bytes);
  const size_t reg_bytes = o;
  l.wave0 = 0; l.reg = 0;
  o = waves > reg_bytes ? waves : reg_bytes;
  l.w = take(sizeof(float) * (Pp > 16 ? Pp : 16));      // weights: live across the phases
  // the prior precision swept on the current model is carried from iteration to iteration
  l.pri0 = take(big ? sizeof(double) * sweep_padded((size_t)Pp * Pp) : 16);
  if (p_lds) l.big_p = take(p_bytes);
  l.shared = take(sizeof(float) * 64);
  l.total = o;
  return l;
}

#ifndef CI_SEASONAL_DECL_ONLY
// Pointers with their ADDRESS SPACE in the type: a `float*` that reaches a function through a struct
// or a parameter is a generic pointer to the compiler, and every access through it a FLAT
// instruction (both memory pipelines, both wait counters) -- the first build of this file had 465
// flat loads in the element pass and 1,049 in the combine.  With typed pointers the same accesses
// are ds_read / global_load.
typedef CI_GLB float* TpG;
typedef CI_GLB const float* TpGC;
typedef CI_LDS float* TpL;
typedef CI_LDS const float* TpLC;
typedef CI_GLB uint8_t* TpGB;
typedef CI_GLB const uint8_t* TpGBC;
// generic -> LDS without the null-check sequence of an address-space cast (its lowering hits a
// backend verifier error in one instantiation: "V_CMP_NE_U32 0, $src_shared_base"): the LDS offset
// is the low half of the generic address
template <class T> __device__ __forceinline__ CI_LDS T* tp_lds(const void* generic) {
  return (CI_LDS T*)(unsigned)(unsigned long long)generic;
}
// A generic pointer into LDS whose origin the optimiser may not look through: InferAddressSpaces
// otherwise rewrites the regression block's accesses through R.* (generic pointers by interface)
// as LDS accesses behind flat -> local casts, and in the kernel's larger instantiations the null
// check of such a cast is lowered to an illegal "V_CMP_NE_U32 0, $src_shared_base".  The
// performance-relevant LDS traffic of this file uses typed pointers (TpL) and does not depend on
// the inference.
template <class T> __device__ __forceinline__ T* tp_opaque(T* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (T*)v;
}
__device__ __forceinline__ CI_GLB const ci_f4v* tp_p4(TpGC p) { return (CI_GLB const ci_f4v*)p; }
__device__ __forceinline__ CI_LDS const ci_f4v* tp_p4(TpLC p) { return (CI_LDS const ci_f4v*)p; }
__device__ __forceinline__ CI_GLB ci_f4v* tp_p4w(TpG p) { return (CI_GLB ci_f4v*)p; }
__device__ __forceinline__ CI_LDS ci_f4v* tp_p4w(TpL p) { return (CI_LDS ci_f4v*)p; }
__device__ __forceinline__ CI_GLB const uint32_t* tp_pu(TpGBC p) { return (CI_GLB const uint32_t*)p; }
// ------------------------------------------------------------------------------------
// synchronisation
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void tp_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// LDS hand-off between the lanes of ONE wavefront INSIDE the per-step loops: DS operations of a
// wavefront complete in program order, so a ds_write followed by a ds_read needs no s_waitcnt in
// between (the read's own wait, where its value is used, covers both) -- only the compiler must
// not reorder them.  Each s_waitcnt lgkmcnt(0) in the step was an exposed LDS round trip.
__device__ __forceinline__ void tp_lds_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// hand-off through global memory between the lanes of ONE wavefront
__device__ __forceinline__ void tp_wg_barrier_wave() { wave_sync(); }
// hand-off through global memory inside one workgroup (one CU, one vector L1)
__device__ __forceinline__ void tp_wg_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
struct TpSync {
  int* flags;        // this chain's TPC_INTS counters
  int G, g, epoch;
  bool cluster;      // handshakes with other workgroups
  bool light;        // ... all on this XCD
};
__device__ __forceinline__ void tp_cluster_barrier(TpSync& s, int tid) {
  if (!s.cluster) { tp_wg_barrier(); return; }
  // ONE arrival counter per chain (flags[0], monotone): an arrival is one L2 atomic, the wait polls
  // one location -- polling G per-workgroup flags one after the other cost G dependent L2 round
  // trips (~12k cycles at G = 16, thirteen times per iteration)
  ++s.epoch;
  // One XCD (`light`): the workgroups share the L2, so a store is visible to the others once it has
  // reached L2 -- s_waitcnt vmcnt(0), which the workgroup-scope release is -- and the arrival is a
  // RELAXED atomic.  A release at agent scope would write the XCD's whole L2 back first (MI300-class
  // parts keep one L2 per XCD: "agent" spans them): measured ~30k cycles per barrier, ten barriers
  // per iteration.  Several XCDs: the full agent-scope release.
  // (round 6, ADVICE) the workgroup-scope release compiles to NO wait on gfx950 (the disassembly
  // of the light branch was s_barrier + global_atomic_add only): every wave drains its own stores
  // to the shared L2 explicitly before the barrier, so the arrival cannot overtake them.
  if (s.light) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else __threadfence();
  __syncthreads();
  // (round 6, as dk_barrier of ci_wide_quad.h; tools/bench_cluster_barrier*.hip) The arrival's RESULT
  // tells the last arriver, and only that workgroup writes the flag the others poll -- on another
  // cache line (the check-in slots' line, idle once the cluster is assembled): polling the arrival
  // counter itself made the adds queue behind the polls.  The acquire is ONE wave's L1 invalidate
  // per workgroup (the vector L1 belongs to the CU): a fence on every wave cost 1.5k cycles more per
  // barrier at 8 workgroups, 3k at 16.
  if (tid == 0) {
    int* flag = s.flags + TPC_XCC;
    const int old = s.light ? __hip_atomic_fetch_add(s.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : __hip_atomic_fetch_add(s.flags, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == s.epoch * s.G) {
      if (s.light) __hip_atomic_store(flag, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(flag, s.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s.epoch)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}
// Assembling the cluster (ci_wide.h cl_assemble, for up to TP_MAXG workgroups): helpers check in
// with their XCD id, workgroup 0 claims them and publishes the mode -- 2: all here, one XCD; 1: all
// here, several XCDs; 3: somebody missing, workgroup 0 runs every chunk alone and the helpers leave.
__device__ __forceinline__ int tp_assemble(int* csync, int role, int G, int tid, int* mode_lds) {
  if (tid == 0) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 15) + 1;
    const long long t0 = wall_clock64();                 // 100 MHz
    int mode = 0;
    if (role > 0) {
      int* slot = csync + TPC_XCC + role;
      __hip_atomic_store(slot, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        mode = __hip_atomic_load(csync + TPC_MODE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        int* slot = csync + TPC_XCC + r;
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
      __hip_atomic_store(csync + TPC_MODE, mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *mode_lds = mode;
  }
  __syncthreads();
  return *mode_lds;
}

// ------------------------------------------------------------------------------------
// wave-cooperative linear algebra: lane i holds ROW i of an NR x NR matrix (zeros beyond D)
// ------------------------------------------------------------------------------------
template <int NR> struct TRow { float v[NR]; };

template <int NR> __device__ __forceinline__ TRow<NR> trow_zero() {
  TRow<NR> r;
#pragma unroll
  for (int j = 0; j < NR; ++j) r.v[j] = 0.f;
  return r;
}
// row `lane` of a matrix stored as rows of `stride` floats (16-byte aligned); zeros for lane >= NR
template <int NR, class PT> __device__ __forceinline__ TRow<NR> trow_load(PT base, int stride, int lane) {
  TRow<NR> r = trow_zero<NR>();
  if (lane < NR) {
    const auto p = tp_p4(base + (size_t)lane * stride);
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) {
      const ci_f4v x = p[q];
      r.v[4 * q] = x.x; r.v[4 * q + 1] = x.y; r.v[4 * q + 2] = x.z; r.v[4 * q + 3] = x.w;
    }
  }
  return r;
}
template <int NR, class PT> __device__ __forceinline__ void trow_store(PT base, int stride, int lane, const TRow<NR>& r) {
  if (lane < NR) {
    auto p = tp_p4w(base + (size_t)lane * stride);
#pragma unroll
    for (int q = 0; q < NR / 4; ++q)
      p[q] = ci_f4v{r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]};
  }
}
// the broadcast operand: rows into the wave's LDS scratch (row stride NR)
template <int NR> __device__ __forceinline__ void tscr_put(TpL scr, const TRow<NR>& r, int lane) {
  tp_lds_sync();                       // earlier readers of the scratch are done
  trow_store<NR>(scr, NR, lane, r);
  tp_lds_sync();
}
// C = A B (+ I), rows of B in the scratch
template <int NR> __device__ __forceinline__ TRow<NR> tmul(const TRow<NR>& A, TpLC scr, int D, int lane,
                                                           bool plus_eye = false) {
  TRow<NR> c;
#pragma unroll
  for (int j = 0; j < NR; ++j) c.v[j] = (plus_eye && j == lane) ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    if (k < D) {
      const auto p = tp_p4(scr + k * NR);
      const float a = A.v[k];
#pragma unroll
      for (int q = 0; q < NR / 4; ++q) {
        const ci_f4v b = p[q];
        c.v[4 * q] = fmaf(a, b.x, c.v[4 * q]); c.v[4 * q + 1] = fmaf(a, b.y, c.v[4 * q + 1]);
        c.v[4 * q + 2] = fmaf(a, b.z, c.v[4 * q + 2]); c.v[4 * q + 3] = fmaf(a, b.w, c.v[4 * q + 3]);
      }
    }
  }
  return c;
}
// C = A B', rows of B in the scratch: C[i][j] = row_i(A) . row_j(B)
template <int NR> __device__ __forceinline__ TRow<NR> tmul_t(const TRow<NR>& A, TpLC scr, int D) {
  TRow<NR> c = trow_zero<NR>();
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (j < D) {
      const auto p = tp_p4(scr + j * NR);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NR / 4; ++q) {
        const ci_f4v b = p[q];
        s = fmaf(A.v[4 * q], b.x, s); s = fmaf(A.v[4 * q + 1], b.y, s);
        s = fmaf(A.v[4 * q + 2], b.z, s); s = fmaf(A.v[4 * q + 3], b.w, s);
      }
      c.v[j] = s;
    }
  }
  return c;
}
// transpose through the scratch
template <int NR> __device__ __forceinline__ TRow<NR> ttranspose(TpL scr, const TRow<NR>& r, int lane) {
  tscr_put<NR>(scr, r, lane);
  TRow<NR> t = trow_zero<NR>();
  if (lane < NR) {
#pragma unroll
    for (int k = 0; k < NR; ++k) t.v[k] = scr[k * NR + lane];
  }
  return t;
}
// row . (a lane-distributed vector): the vector goes through the wave's broadcast buffer
template <int NR> __device__ __forceinline__ float tdot(const TRow<NR>& A, TpL vb, float x, int lane) {
  tp_lds_sync();
  if (lane < NR) vb[lane] = x;
  tp_lds_sync();
  float s = 0.f;
  const auto p = tp_p4(vb);
#pragma unroll
  for (int q = 0; q < NR / 4; ++q) {
    const ci_f4v b = p[q];
    s = fmaf(A.v[4 * q], b.x, s); s = fmaf(A.v[4 * q + 1], b.y, s);
    s = fmaf(A.v[4 * q + 2], b.z, s); s = fmaf(A.v[4 * q + 3], b.w, s);
  }
  return s;
}
// Gauss-Jordan without pivoting on register rows: [W | R0 | R1 | u] -> [I | W^-1 R0 | W^-1 R1 | W^-1 u].
// W = I + (psd)(psd): eigenvalues >= 1.  The pivot row travels by v_readlane (uniform values); the
// multiplier of lane c itself is W[c][c] - 1, which turns "row -= f * row_c / W[c][c]" into the
// scaling of the pivot row -- one fused multiply-add per entry for every lane, no select.
template <int NR, int NRHS>
__device__ __forceinline__ void tgauss_jordan(TRow<NR>& W, TRow<NR>& R0, TRow<NR>& R1, float& u, int D, int lane) {
#pragma unroll
  for (int c = 0; c < NR; ++c) {
    if (c < D) {
      const float piv = readlane_f(W.v[c], c);
      float rp = __builtin_amdgcn_rcpf(piv);
      rp = fmaf(fmaf(-piv, rp, 1.0f), rp, rp);
      const float f = W.v[c] - (lane == c ? 1.f : 0.f);
#pragma unroll
      for (int j = c + 1; j < NR; ++j) W.v[j] = fmaf(-f, readlane_f(W.v[j], c) * rp, W.v[j]);
#pragma unroll
      for (int j = 0; j < NR; ++j) R0.v[j] = fmaf(-f, readlane_f(R0.v[j], c) * rp, R0.v[j]);
      if constexpr (NRHS > 1) {
#pragma unroll
        for (int j = 0; j < NR; ++j) R1.v[j] = fmaf(-f, readlane_f(R1.v[j], c) * rp, R1.v[j]);
      }
      u = fmaf(-f, readlane_f(u, c) * rp, u);
    }
  }
}

// ---- the same algebra on BOTH halves of the wavefront: lane l holds columns [h NH, (h + 1) NH) of
// row i = l & 31, h = l >> 5, NH = NR / 2.  Half the registers per matrix and half the instructions
// per product / elimination of the row-per-lane form above (which leaves 32+ lanes idle); a left
// operand is needed as a FULL row (TRow, the same in both halves), results come out in halves, and
// the conversion goes through the LDS scratch the products use anyway.
template <int NH> struct THalf { float v[NH]; };
template <int NH> __device__ __forceinline__ THalf<NH> thalf_zero() {
  THalf<NH> r;
#pragma unroll
  for (int j = 0; j < NH; ++j) r.v[j] = 0.f;
  return r;
}
template <int NR, class PT> __device__ __forceinline__ THalf<NR / 2> hload(PT base, int stride, int lane) {
  constexpr int NH = NR / 2;
  THalf<NH> r = thalf_zero<NH>();
  const int i = lane & 31, h = lane >> 5;
  if (i < NR) {
    const auto p = tp_p4(base + (size_t)i * stride + h * NH);
#pragma unroll
    for (int q = 0; q < NH / 4; ++q) {
      const ci_f4v x = p[q];
      r.v[4 * q] = x.x; r.v[4 * q + 1] = x.y; r.v[4 * q + 2] = x.z; r.v[4 * q + 3] = x.w;
    }
  }
  return r;
}
template <int NR, class PT> __device__ __forceinline__ void hstore(PT base, int stride, int lane, const THalf<NR / 2>& r) {
  constexpr int NH = NR / 2;
  const int i = lane & 31, h = lane >> 5;
  if (i < NR) {
    auto p = tp_p4w(base + (size_t)i * stride + h * NH);
#pragma unroll
    for (int q = 0; q < NH / 4; ++q) p[q] = ci_f4v{r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]};
  }
}
// full row i = lane & 31 (the same registers in both halves)
template <int NR, class PT> __device__ __forceinline__ TRow<NR> fload(PT base, int stride, int lane) {
  return trow_load<NR>(base, stride, lane & 31);
}
template <int NR, class PT> __device__ __forceinline__ float fscalar(PT base, int stride, int off, int lane) {
  const int i = lane & 31;
  return i < NR ? base[(size_t)i * stride + off] : 0.f;
}
template <int NR> __device__ __forceinline__ void hscr_put(TpL scr, const THalf<NR / 2>& r, int lane) {
  tp_lds_sync();
  hstore<NR>(scr, NR, lane, r);
  tp_lds_sync();
}
template <int NR> __device__ __forceinline__ void fscr_put(TpL scr, const TRow<NR>& r, int lane) {
  tp_lds_sync();
  trow_store<NR>(scr, NR, lane, r);        // lanes < NR: row = lane
  tp_lds_sync();
}
// own full row back from the scratch (halves -> full)
template <int NR> __device__ __forceinline__ TRow<NR> fscr_row(TpLC scr, int lane) {
  return trow_load<NR>(scr, NR, lane & 31);
}
// C = A B (+ I): A as full rows, rows of B in the scratch, C in halves
template <int NR> __device__ __forceinline__ THalf<NR / 2> hmul(const TRow<NR>& A, TpLC scr, int D, int lane,
                                                               bool plus_eye = false) {
  constexpr int NH = NR / 2;
  const int i = lane & 31, h = lane >> 5;
  THalf<NH> c;
#pragma unroll
  for (int j = 0; j < NH; ++j) c.v[j] = (plus_eye && h * NH + j == i) ? 1.f : 0.f;
  TpLC sh = scr + h * NH;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    if (k < D) {
      const auto p = tp_p4(sh + k * NR);
      const float a = A.v[k];
#pragma unroll
      for (int q = 0; q < NH / 4; ++q) {
        const ci_f4v b = p[q];
        c.v[4 * q] = fmaf(a, b.x, c.v[4 * q]); c.v[4 * q + 1] = fmaf(a, b.y, c.v[4 * q + 1]);
        c.v[4 * q + 2] = fmaf(a, b.z, c.v[4 * q + 2]); c.v[4 * q + 3] = fmaf(a, b.w, c.v[4 * q + 3]);
      }
    }
  }
  return c;
}
// C = A B': C[i][j] = row_i(A) . row_j(B), j in the lane's half
template <int NR> __device__ __forceinline__ THalf<NR / 2> hmul_t(const TRow<NR>& A, TpLC scr, int lane) {
  constexpr int NH = NR / 2;
  const int h = lane >> 5;
  THalf<NH> c;
#pragma unroll
  for (int jj = 0; jj < NH; ++jj) {
    const auto p = tp_p4(scr + (h * NH + jj) * NR);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) {
      const ci_f4v b = p[q];
      s = fmaf(A.v[4 * q], b.x, s); s = fmaf(A.v[4 * q + 1], b.y, s);
      s = fmaf(A.v[4 * q + 2], b.z, s); s = fmaf(A.v[4 * q + 3], b.w, s);
    }
    c.v[jj] = s;
  }
  return c;
}
template <int NR> __device__ __forceinline__ THalf<NR / 2> htranspose(TpL scr, const THalf<NR / 2>& r, int lane) {
  constexpr int NH = NR / 2;
  hscr_put<NR>(scr, r, lane);
  const int i = lane & 31, h = lane >> 5;
  THalf<NH> t = thalf_zero<NH>();
  if (i < NR) {
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) t.v[jj] = scr[(h * NH + jj) * NR + i];
  }
  return t;
}
// row . x for a per-row scalar x (the same in both halves); the result likewise
template <int NR> __device__ __forceinline__ float hdot(const TRow<NR>& A, TpL vb, float x, int lane) {
  tp_lds_sync();
  if (lane < NR) vb[lane] = x;
  tp_lds_sync();
  float s = 0.f;
  const auto p = tp_p4(vb);
#pragma unroll
  for (int q = 0; q < NR / 4; ++q) {
    const ci_f4v b = p[q];
    s = fmaf(A.v[4 * q], b.x, s); s = fmaf(A.v[4 * q + 1], b.y, s);
    s = fmaf(A.v[4 * q + 2], b.z, s); s = fmaf(A.v[4 * q + 3], b.w, s);
  }
  return s;
}
// Gauss-Jordan without pivoting on half rows: [W | R0 | R1 | u] -> [I | W^-1 R0 | W^-1 R1 | W^-1 u].
// The pivot row is published in LDS (`piv`: 3 NR + 4 floats) and read back as broadcast loads; the
// multiplier W[i][c] sits in half c / NH and reaches the other half by one ds_bpermute; the
// multiplier of row c itself is W[c][c] - 1 (see tgauss_jordan).
template <int NR, int NRHS>
__device__ __forceinline__ void hgauss_jordan(THalf<NR / 2>& W, THalf<NR / 2>& R0, THalf<NR / 2>& R1, float& u,
                                              TpL piv, int D, int lane) {
  constexpr int NH = NR / 2;
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int c = 0; c < NR; ++c) {
    if (c < D) {
      constexpr int dummy = 0; (void)dummy;
      const int hc = c / NH, cc = c % NH;
      const float f0 = __shfl(W.v[cc], i + 32 * hc, 64);
      tp_lds_sync();
      if (i == c) {
#pragma unroll
        for (int q = 0; q < NH / 4; ++q) {
          *tp_p4w(piv + h * NH + 4 * q) = ci_f4v{W.v[4 * q], W.v[4 * q + 1], W.v[4 * q + 2], W.v[4 * q + 3]};
          *tp_p4w(piv + NR + h * NH + 4 * q) = ci_f4v{R0.v[4 * q], R0.v[4 * q + 1], R0.v[4 * q + 2], R0.v[4 * q + 3]};
          if constexpr (NRHS > 1)
            *tp_p4w(piv + 2 * NR + h * NH + 4 * q) = ci_f4v{R1.v[4 * q], R1.v[4 * q + 1], R1.v[4 * q + 2], R1.v[4 * q + 3]};
        }
        if (h == 0) piv[3 * NR] = u;
      }
      tp_lds_sync();
      const float pv = piv[c];
      float rp = __builtin_amdgcn_rcpf(pv);
      rp = fmaf(fmaf(-pv, rp, 1.0f), rp, rp);
      const float f = (f0 - (i == c ? 1.f : 0.f)) * rp;
      TpLC ph = piv + h * NH;
#pragma unroll
      for (int q = 0; q < NH / 4; ++q) {
        const ci_f4v a = *tp_p4(ph + 4 * q);
        W.v[4 * q] = fmaf(-f, a.x, W.v[4 * q]); W.v[4 * q + 1] = fmaf(-f, a.y, W.v[4 * q + 1]);
        W.v[4 * q + 2] = fmaf(-f, a.z, W.v[4 * q + 2]); W.v[4 * q + 3] = fmaf(-f, a.w, W.v[4 * q + 3]);
        const ci_f4v b = *tp_p4(ph + NR + 4 * q);
        R0.v[4 * q] = fmaf(-f, b.x, R0.v[4 * q]); R0.v[4 * q + 1] = fmaf(-f, b.y, R0.v[4 * q + 1]);
        R0.v[4 * q + 2] = fmaf(-f, b.z, R0.v[4 * q + 2]); R0.v[4 * q + 3] = fmaf(-f, b.w, R0.v[4 * q + 3]);
        if constexpr (NRHS > 1) {
          const ci_f4v d = *tp_p4(ph + 2 * NR + 4 * q);
          R1.v[4 * q] = fmaf(-f, d.x, R1.v[4 * q]); R1.v[4 * q + 1] = fmaf(-f, d.y, R1.v[4 * q + 1]);
          R1.v[4 * q + 2] = fmaf(-f, d.z, R1.v[4 * q + 2]); R1.v[4 * q + 3] = fmaf(-f, d.w, R1.v[4 * q + 3]);
        }
      }
      u = fmaf(-f, piv[3 * NR], u);
    }
  }
}

// ---- forward (filtering) elements in the workspace: NR rows of [A | C | J | b eta 0 0] -------------
template <int NR> struct TpElemPtr {
  TpGC p;
  static constexpr int RW = 3 * NR + 4;
  __device__ __forceinline__ TRow<NR> A(int lane) const { return trow_load<NR>(p, RW, lane); }
  __device__ __forceinline__ TRow<NR> C(int lane) const { return trow_load<NR>(p + NR, RW, lane); }
  __device__ __forceinline__ TRow<NR> J(int lane) const { return trow_load<NR>(p + 2 * NR, RW, lane); }
  __device__ __forceinline__ float b(int lane) const { return lane < NR ? p[(size_t)lane * RW + 3 * NR] : 0.f; }
  __device__ __forceinline__ float eta(int lane) const { return lane < NR ? p[(size_t)lane * RW + 3 * NR + 1] : 0.f; }
  // half / full-row forms on both halves of the wavefront
  __device__ __forceinline__ THalf<NR / 2> Ah(int lane) const { return hload<NR>(p, RW, lane); }
  __device__ __forceinline__ THalf<NR / 2> Ch(int lane) const { return hload<NR>(p + NR, RW, lane); }
  __device__ __forceinline__ THalf<NR / 2> Jh(int lane) const { return hload<NR>(p + 2 * NR, RW, lane); }
  __device__ __forceinline__ TRow<NR> Af(int lane) const { return fload<NR>(p, RW, lane); }
  __device__ __forceinline__ TRow<NR> Cf(int lane) const { return fload<NR>(p + NR, RW, lane); }
  __device__ __forceinline__ TRow<NR> Jf(int lane) const { return fload<NR>(p + 2 * NR, RW, lane); }
  __device__ __forceinline__ float bf(int lane) const { return fscalar<NR>(p, RW, 3 * NR, lane); }
  __device__ __forceinline__ float etaf(int lane) const { return fscalar<NR>(p, RW, 3 * NR + 1, lane); }
};
template <int NR>
__device__ __forceinline__ void tp_elem_store(TpG p, int lane, const TRow<NR>& A, const TRow<NR>& C,
                                              const TRow<NR>& J, float b, float eta) {
  trow_store<NR>(p, 3 * NR + 4, lane, A);
  trow_store<NR>(p + NR, 3 * NR + 4, lane, C);
  trow_store<NR>(p + 2 * NR, 3 * NR + 4, lane, J);
  if (lane < NR) *tp_p4w(p + (size_t)lane * (3 * NR + 4) + 3 * NR) = ci_f4v{b, eta, 0.f, 0.f};
}
template <int NR> __device__ __forceinline__ void tp_elem_copy(TpG dst, TpGC src, int lane) {
  const int n4 = (int)(tp_esz(NR) / 4);
  for (int e = lane; e < n4; e += 64) tp_p4w(dst)[e] = tp_p4(src)[e];
}
// (Measured, tools/bench_tp_combine.hip: this row-per-lane form with v_readlane pivots -- no LDS
// round trip inside the elimination -- takes 50k cycles at NR = 24 on one wavefront and 100k with
// all eight of a workgroup in it; the half-row form with LDS-published pivot rows has 40 % fewer
// instructions but two LDS synchronisations per pivot: 75k / 165k.  The half-row form is kept for
// the backward maps, where the product dominates: 8.5k instead of 19k cycles.)
// out = e1 then e2 (e1 covers the earlier steps).  Operands are read from the workspace just in
// time -- at most five matrices are live in registers.  STATE: only (b, C) of the result are
// formed (the predicted moments at a chunk's start), written as rows of [C | b 0 0 0].
template <int NR, bool STATE>
__device__ __noinline__ void tp_combine(TpGC g1, TpGC g2, TpG out, TpL scr, TpL vb,
                                        TpL /*piv*/, int D, int lane) {
  const TpElemPtr<NR> e1{g1}, e2{g2};
  TRow<NR> Y = e1.C(lane);                       // C1, becomes W^-1 C1
  // W = I + C1 J2
  tscr_put<NR>(scr, e2.J(lane), lane);
  TRow<NR> W = tmul<NR>(Y, scr, D, lane, true);
  // u = b1 + C1 eta2
  float u = e1.b(lane) + tdot<NR>(Y, vb, e2.eta(lane), lane);
  TRow<NR> G = trow_zero<NR>(), T2 = trow_zero<NR>();
  float w = 0.f;
  if constexpr (!STATE) {
    G = e1.A(lane);                              // A1, becomes W^-1 A1
    const TRow<NR> J2 = e2.J(lane);
    w = e2.eta(lane) - tdot<NR>(J2, vb, e1.b(lane), lane);     // eta2 - J2 b1
    tscr_put<NR>(scr, G, lane);
    T2 = tmul<NR>(J2, scr, D, lane);             // J2 A1
    tgauss_jordan<NR, 2>(W, Y, G, u, D, lane);
  } else {
    tgauss_jordan<NR, 1>(W, Y, G, u, D, lane);
  }
  const TRow<NR> A2 = e2.A(lane);
  const float bo = e2.b(lane) + tdot<NR>(A2, vb, u, lane);     // A2 W^-1 (b1 + C1 eta2) + b2
  // C = A2 Y A2' + C2
  tscr_put<NR>(scr, Y, lane);
  const TRow<NR> T1 = tmul<NR>(A2, scr, D, lane);
  tscr_put<NR>(scr, A2, lane);
  TRow<NR> Co = tmul_t<NR>(T1, scr, D);
  {
    const TRow<NR> C2 = e2.C(lane);
#pragma unroll
    for (int j = 0; j < NR; ++j) Co.v[j] += C2.v[j];
  }
  if constexpr (STATE) {
    // symmetrised: the replayed filter's sweeps assume P = P'
    const TRow<NR> Ct = ttranspose<NR>(scr, Co, lane);
#pragma unroll
    for (int j = 0; j < NR; ++j) Co.v[j] = 0.5f * (Co.v[j] + Ct.v[j]);
    trow_store<NR>(out, NR + 4, lane, Co);
    if (lane < NR) *tp_p4w(out + (size_t)lane * (NR + 4) + NR) = ci_f4v{bo, 0.f, 0.f, 0.f};
  } else {
    tscr_put<NR>(scr, G, lane);
    const TRow<NR> Ao = tmul<NR>(A2, scr, D, lane);            // A2 G
    const TRow<NR> Gt = ttranspose<NR>(scr, G, lane);
    const float eo = e1.eta(lane) + tdot<NR>(Gt, vb, w, lane); // G'(eta2 - J2 b1) + eta1
    tscr_put<NR>(scr, T2, lane);
    TRow<NR> Jo = tmul<NR>(Gt, scr, D, lane);                  // G' J2 A1
    {
      const TRow<NR> J1 = e1.J(lane);
#pragma unroll
      for (int j = 0; j < NR; ++j) Jo.v[j] += J1.v[j];
    }
    tp_elem_store<NR>(out, lane, Ao, Co, Jo, bo, eo);
  }
}
// chunk-start moments from an element that already starts at the prior (A = 0): rows [C | b 0 0 0]
template <int NR> __device__ __forceinline__ void tp_state_from_elem(TpGC g, TpG out, int lane) {
  const TpElemPtr<NR> e{g};
  trow_store<NR>(out, NR + 4, lane, e.C(lane));
  if (lane < NR) *tp_p4w(out + (size_t)lane * (NR + 4) + NR) = ci_f4v{e.b(lane), 0.f, 0.f, 0.f};
}

// ---- backward maps in the workspace: NR rows of [M | c 0 0 0]; (outer o inner)(r) = Mo (Mi r + ci) + co
template <int NR>
__device__ __noinline__ void tp_bcompose(TpGC outer, TpGC inner, TpG out, TpL scr, TpL vb,
                                         int D, int lane) {
  if constexpr (NR % 8 == 0) {
    const TRow<NR> Mo = fload<NR>(outer, NR + 4, lane);
    const float co = fscalar<NR>(outer, NR + 4, NR, lane);
    const float ci = fscalar<NR>(inner, NR + 4, NR, lane);
    hscr_put<NR>(scr, hload<NR>(inner, NR + 4, lane), lane);
    const THalf<NR / 2> M = hmul<NR>(Mo, scr, D, lane);
    const float c = co + hdot<NR>(Mo, vb, ci, lane);
    hstore<NR>(out, NR + 4, lane, M);
    if (lane < NR) *tp_p4w(out + (size_t)lane * (NR + 4) + NR) = ci_f4v{c, 0.f, 0.f, 0.f};
  } else {
    // (widths that do not split into two float4-aligned halves: the row-per-lane form)
    const TRow<NR> Mo = trow_load<NR>(outer, NR + 4, lane);
    const float co = lane < NR ? outer[(size_t)lane * (NR + 4) + NR] : 0.f;
    const float ci = lane < NR ? inner[(size_t)lane * (NR + 4) + NR] : 0.f;
    tscr_put<NR>(scr, trow_load<NR>(inner, NR + 4, lane), lane);
    const TRow<NR> M = tmul<NR>(Mo, scr, D, lane);
    const float c = co + tdot<NR>(Mo, vb, ci, lane);
    trow_store<NR>(out, NR + 4, lane, M);
    if (lane < NR) *tp_p4w(out + (size_t)lane * (NR + 4) + NR) = ci_f4v{c, 0.f, 0.f, 0.f};
  }
}
template <int NR> __device__ __forceinline__ void tp_bcopy(TpG dst, TpGC src, int lane) {
  const int n4 = (int)(tp_bsz(NR) / 4);
  for (int e = lane; e < n4; e += 64) tp_p4w(dst)[e] = tp_p4(src)[e];
}

// ------------------------------------------------------------------------------------
// per-chunk passes (one wavefront, steps [s, e) of the series)
// ------------------------------------------------------------------------------------
struct TpCtx {
  int T, TP, D, DS, K, lane, blk, blk0, pos, nb, boff, has_slope;
  int off[SMAXK], nsz[SMAXK];
  float H, ql, qs, myd2, rnb, so, sl, ssc, mydrift;
  // this wavefront's LDS
  TpL cm, am, scr, pzv, vb;
  // arrays over time (this chain's workspace)
  TpG yv, lev, slp, xw, ytil, vf, zl, zs, zo, seas, zk, gd, kf, rs;
  TpGB msk, cbv, cidx;
};

__device__ __forceinline__ float tp_at4(const ci_f4v& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// x+ through the chunk from `xp` (ci_seasonal.h pass 0 on a range; c_k(t) comes from the static
// table).  WRITE: also y~ = (y - X w) - (Z x+ + sigma_obs z_obs).  Returns x+ after the chunk.
template <bool WRITE>
static __device__ __noinline__ float tp_sim_pass(const TpCtx& cref, int s, int e, float xp) {
  const TpCtx c = cref;
  const int lane = c.lane;
  TpGC zkb = c.zk + (size_t)c.blk0 * c.TP;
  TpGBC cidb = c.cidx + (size_t)c.blk0 * c.TP;
  for (int t4 = s; t4 < e; t4 += 4) {
    const ci_f4v zl4 = *tp_p4(c.zl + t4);
    const ci_f4v zk4 = *tp_p4(zkb + t4);
    ci_f4v zo4 = ci_f4v{0.f, 0.f, 0.f, 0.f}, yv4 = zo4, xw4 = zo4, zs4 = zo4;
    if (WRITE) {
      zo4 = *tp_p4(c.zo + t4);
      yv4 = *tp_p4(c.yv + t4);
      xw4 = *tp_p4(c.xw + t4);
    }
    if (c.has_slope) zs4 = *tp_p4(c.zs + t4);
    const uint32_t cb4 = *tp_pu(c.cbv + t4);
    const uint32_t cw4 = *tp_pu(cidb + t4);
    float yt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t4 + q;
      const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
      yt[q] = 0.f;
      if (WRITE) {
        const bool isz = lane == 0 || (c.blk >= 0 && c.pos == mycur);
        const float zx = wave_sum_dpp(isz ? xp : 0.f);
        yt[q] = (tp_at4(yv4, q) - tp_at4(xw4, q)) - (zx + c.so * tp_at4(zo4, q));
      }
      if (t + 1 < c.T) {
        const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
        float r = xp;
        if (c.has_slope) {
          const float s1 = readlane_f(xp, 1);
          if (lane == 0) r += s1;
        }
        if (lane == 0) r = fmaf(c.sl, tp_at4(zl4, q), r);
        if (c.has_slope && lane == 1) r = fmaf(c.ssc, tp_at4(zs4, q), r);
        if (c.blk >= 0 && ((cb >> c.blk) & 1u)) {
          const float gi = (c.pos == mycur ? 1.f : 0.f) - c.rnb;
          r = fmaf(c.mydrift * gi, tp_at4(zk4, q), r);
        }
        xp = r;
      }
    }
    if (WRITE && lane == 0) *tp_p4w(c.ytil + t4) = ci_f4v{yt[0], yt[1], yt[2], yt[3]};
  }
  return xp;
}

// Prior covariance of x_0 in slot coordinates (c_k(0) = 0): row `lane`.
template <int NR>
__device__ __forceinline__ TRow<NR> tp_prior_row(const TpCtx& c, float p1l, float p1s, float p1e) {
  TRow<NR> r = trow_zero<NR>();
  if (c.lane == 0) r.v[0] = p1l;
  if (c.has_slope && c.lane == 1) r.v[1] = p1s;
#pragma unroll
  for (int u = 0; u < NR; ++u) {
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < c.K && c.blk == k && u >= c.off[k] && u < c.off[k] + c.nsz[k])
        r.v[u] = p1e * ((u - c.off[k] == c.pos ? 1.f : 0.f) - c.rnb);
  }
  return r;
}

// The chunk's filtering element (A, b, C, eta, J): x_e | x_s ~ N(A x_s + b, C) given the chunk's
// observations, and their likelihood as a function of x_s, N^-1(eta, J).  One pass over the steps;
// per step the measurement update is a rank-one sweep of the own rows of C, A' and J, the time
// update the sweep ci_seasonal.h's filter applies to P.  Registers hold the rows; LDS mirrors of C
// and A' serve the reads with a run-time column (C z, A'z).  `first`: the chunk starts at the
// prior, i.e. its element is (0, a_1, P_1, 0, 0) followed by its steps.
template <int NQ>
static __device__ __noinline__ void tp_build_pass(const TpCtx& cref, int s, int e, bool first, float a1e,
                                                   float p1l, float p1s, float p1e, TpG eout) {
  const TpCtx c = cref;        // by value: fields read through the reference are FLAT loads the
                               // optimiser cannot hoist past the LDS / global stores of the loop
  constexpr int NR = 4 * NQ;
  const int lane = c.lane, D = c.D, DS = c.DS, T = c.T;
  const bool slope = c.has_slope != 0, comp = lane < D;
  TpL Cm = c.cm;
  TpL Am = c.am;
  CI_LDS float* Crow = Cm + (comp ? lane : 0) * DS;
  CI_LDS float* Arow = Am + (comp ? lane : 0) * DS;
  TpL pzv = c.pzv;
  CI_LDS float* zav = pzv + 72;
  CI_LDS float* gvk = pzv + 144;
  CI_LDS const float* gmine = gvk + c.blk0 * 72;
  CI_LDS float* gslot = gvk + c.blk0 * 72 + lane;
  TpGBC cidb = c.cidx + (size_t)c.blk0 * c.TP;
  auto ld4 = [](CI_LDS const float* q) -> ci_f4v { return *(CI_LDS const ci_f4v*)q; };
  const float H = c.H, ql = c.ql, qs = c.qs, myd2 = c.myd2, rnb = c.rnb;
  for (int x = lane; x < 144 + SMAXK * 72; x += 64) pzv[x] = 0.f;
  tp_lds_sync();
  float crow[NR], arow[NR], jrow[NR];
  float bi = 0.f, etai = 0.f;
  {
    const TRow<NR> p0 = first ? tp_prior_row<NR>(c, p1l, p1s, p1e) : trow_zero<NR>();
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      crow[u] = p0.v[u];
      arow[u] = (!first && comp && u == lane) ? 1.f : 0.f;
      jrow[u] = 0.f;
    }
    if (first) bi = a1e;
  }
  if (comp) {
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) {
      *(CI_LDS ci_f4v*)(Crow + 4 * q) = ci_f4v{crow[4 * q], crow[4 * q + 1], crow[4 * q + 2], crow[4 * q + 3]};
      *(CI_LDS ci_f4v*)(Arow + 4 * q) = ci_f4v{arow[4 * q], arow[4 * q + 1], arow[4 * q + 2], arow[4 * q + 3]};
    }
    *(CI_LDS ci_f4v*)(Crow + NR) = ci_f4v{0.f, 0.f, 0.f, 0.f};      // the zero column unused zcol entries point at
    *(CI_LDS ci_f4v*)(Arow + NR) = ci_f4v{0.f, 0.f, 0.f, 0.f};
  }
  tp_lds_sync();
  for (int t4 = s; t4 < e; t4 += 4) {
    const ci_f4v yt4 = *tp_p4(c.ytil + t4);
    const uint32_t cb4 = *tp_pu(c.cbv + t4);
    const uint32_t mk4 = *tp_pu(c.msk + t4);
    const uint32_t cw4 = *tp_pu(cidb + t4);
    uint32_t cwk[SMAXK];                              // c_k(t) of EVERY block, 4 steps (static table)
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) cwk[k] = k < c.K ? *tp_pu(c.cidx + (size_t)k * c.TP + t4) : 0u;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int t = t4 + q;
      if (t + 1 >= T) continue;                   // (the last step's update feeds nothing)
      const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
      const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
      const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
      const bool isz = lane == 0 || (c.blk >= 0 && c.pos == mycur);
      const bool mych = c.blk >= 0 && ((cb >> c.blk) & 1u) != 0u;
      float cz = 0.f, za = 0.f, rS = 0.f;
      if (obs) {
        if (comp) {
          // (all sixteen reads issued together: blocks that do not exist read the zero column)
          float cq[SMAXK], aq[SMAXK];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k) {
            const int zc = k < c.K ? c.off[k] + (int)((cwk[k] >> (8 * q)) & 0xFFu) : NR;
            cq[k] = Crow[zc];
            aq[k] = Arow[zc];
          }
          cz = crow[0];
          za = arow[0];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k) { cz += cq[k]; za += aq[k]; }
        }
        const float S = wave_sum_dpp(isz ? cz : 0.f) + H;
        rS = __builtin_amdgcn_rcpf(S);
        rS = fmaf(fmaf(-S, rS, 1.0f), rS, rS);
        const float yq = q == 0 ? yt4.x : q == 1 ? yt4.y : q == 2 ? yt4.z : yt4.w;
        const float inn = (yq - wave_sum_dpp(isz ? bi : 0.f)) * rS;
        etai = fmaf(za, inn, etai);
        bi = fmaf(cz, inn, bi);
      }
      const float gi = mych ? ((c.pos == mycur ? 1.f : 0.f) - rnb) : 0.f;
      if (slope) {                                // b <- T b
        const float b1 = readlane_f(bi, 1);
        if (lane == 0) bi += b1;
      }
      if (!obs && cb == 0u && !slope) {
        if (lane == 0) { crow[0] += ql; Crow[0] = crow[0]; }
        continue;
      }
      if (comp) {
        pzv[lane] = cz;
        zav[lane] = za;
        if (c.blk >= 0) *gslot = gi;
      }
      tp_lds_order();
      if (comp) {
        const float cz1 = slope ? pzv[1] : 0.f;
        // every broadcast row of the step in flight at once (NR - D < 4: no quad is empty except
        // for the trend-only widths, where the spare quad works on zeros)
        ci_f4v pa[NQ], za_[NQ], ga[NQ], qa[NQ];
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          pa[qd] = ld4(pzv + 4 * qd); za_[qd] = ld4(zav + 4 * qd); ga[qd] = ld4(gmine + 4 * qd);
          qa[qd] = slope ? ld4(Cm + DS + 4 * qd) : ci_f4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          const float pj[4] = {pa[qd].x, pa[qd].y, pa[qd].z, pa[qd].w};
          const float zj[4] = {za_[qd].x, za_[qd].y, za_[qd].z, za_[qd].w};
          const float gj[4] = {ga[qd].x, ga[qd].y, ga[qd].z, ga[qd].w};
          if (slope) {
            const float p1[4] = {qa[qd].x, qa[qd].y, qa[qd].z, qa[qd].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float r = fmaf(-(cz * pj[u]), rS, crow[4 * qd + u]);
              if (lane == 0) r += fmaf(-(cz1 * pj[u]), rS, p1[u]);
              crow[4 * qd + u] = fmaf(myd2, gi * gj[u], r);
            }
            if (qd == 0) {
              crow[0] += crow[1];
              if (lane == 1) crow[1] += qs;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              crow[4 * qd + u] = fmaf(myd2, gi * gj[u], fmaf(-(cz * pj[u]), rS, crow[4 * qd + u]));
          }
          if (qd == 0 && lane == 0) crow[0] += ql;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            arow[4 * qd + u] = fmaf(-(za * pj[u]), rS, arow[4 * qd + u]);
            jrow[4 * qd + u] = fmaf(za * zj[u], rS, jrow[4 * qd + u]);
          }
          if (slope && qd == 0) arow[0] += arow[1];     // A <- T A (column 0 of A' += column 1)
          *(CI_LDS ci_f4v*)(Crow + 4 * qd) = ci_f4v{crow[4 * qd], crow[4 * qd + 1], crow[4 * qd + 2], crow[4 * qd + 3]};
          *(CI_LDS ci_f4v*)(Arow + 4 * qd) = ci_f4v{arow[4 * qd], arow[4 * qd + 1], arow[4 * qd + 2], arow[4 * qd + 3]};
        }
      }
      tp_lds_order();
    }
  }
  // A' rows -> A rows, and out
  TRow<NR> At, Cr, Jr;
#pragma unroll
  for (int u = 0; u < NR; ++u) { At.v[u] = comp ? arow[u] : 0.f; Cr.v[u] = comp ? crow[u] : 0.f; Jr.v[u] = comp ? jrow[u] : 0.f; }
  const TRow<NR> A = ttranspose<NR>(c.scr, At, lane);
  tp_elem_store<NR>(eout, lane, A, Cr, Jr, comp ? bi : 0.f, comp ? etai : 0.f);
}

// The Kalman filter on the chunk from its true predicted moments (rows [P | a 0 0 0] in `st`):
// ci_seasonal.h's seasonal_filter_pass on a range.  Stores K_t and v_t / F_t.
template <int NQ>
static __device__ __noinline__ void tp_filter_pass(const TpCtx& cref, int s, int e, TpGC st) {
  const TpCtx c = cref;
  constexpr int NR = 4 * NQ;
  const int lane = c.lane, D = c.D, DS = c.DS, T = c.T;
  const bool slope = c.has_slope != 0, comp = lane < D;
  TpL Pm = c.cm;
  CI_LDS float* Prow = Pm + (comp ? lane : 0) * DS;
  TpL pzv = c.pzv;
  CI_LDS float* gvk = pzv + 144;
  CI_LDS const float* gmine = gvk + c.blk0 * 72;
  CI_LDS float* gslot = gvk + c.blk0 * 72 + lane;
  TpGBC cidb = c.cidx + (size_t)c.blk0 * c.TP;
  auto ld4 = [](CI_LDS const float* q) -> ci_f4v { return *(CI_LDS const ci_f4v*)q; };
  const float H = c.H, ql = c.ql, qs = c.qs, myd2 = c.myd2, rnb = c.rnb;
  for (int x = lane; x < 144 + SMAXK * 72; x += 64) pzv[x] = 0.f;
  tp_lds_sync();
  float prow[NR];
  float am = 0.f;
  {
    const TRow<NR> p0 = trow_load<NR>(st, NR + 4, lane);
#pragma unroll
    for (int u = 0; u < NR; ++u) prow[u] = comp ? p0.v[u] : 0.f;
    if (comp) am = st[(size_t)lane * (NR + 4) + NR];
  }
  if (comp) {
#pragma unroll
    for (int q = 0; q < NR / 4; ++q)
      *(CI_LDS ci_f4v*)(Prow + 4 * q) = ci_f4v{prow[4 * q], prow[4 * q + 1], prow[4 * q + 2], prow[4 * q + 3]};
    *(CI_LDS ci_f4v*)(Prow + NR) = ci_f4v{0.f, 0.f, 0.f, 0.f};
  }
  tp_lds_sync();
  TpG kfw = c.kf + (size_t)s * D + lane;
  for (int t4 = s; t4 < e; t4 += 4) {
    const ci_f4v yt4 = *tp_p4(c.ytil + t4);
    const uint32_t cb4 = *tp_pu(c.cbv + t4);
    const uint32_t mk4 = *tp_pu(c.msk + t4);
    const uint32_t cw4 = *tp_pu(cidb + t4);
    uint32_t cwk[SMAXK];                              // c_k(t) of EVERY block, 4 steps (static table)
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) cwk[k] = k < c.K ? *tp_pu(c.cidx + (size_t)k * c.TP + t4) : 0u;
    float vfq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t4 + q;
      vfq[q] = 0.f;
      if (t >= T) continue;
      const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
      const unsigned cb = (t + 1 < T) ? ((cb4 >> (8 * q)) & 0xFFu) : 0u;
      const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
      const bool isz = lane == 0 || (c.blk >= 0 && c.pos == mycur);
      const bool mych = c.blk >= 0 && ((cb >> c.blk) & 1u) != 0u;
      float kfi = 0.f, rF = 0.f, pz = 0.f;
      if (obs) {
        if (comp) {
          float pq[SMAXK];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            pq[k] = Prow[k < c.K ? c.off[k] + (int)((cwk[k] >> (8 * q)) & 0xFFu) : NR];
          pz = prow[0];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k) pz += pq[k];
        }
        const float F = wave_sum_dpp(isz ? pz : 0.f) + H;
        rF = __builtin_amdgcn_rcpf(F);
        rF = fmaf(fmaf(-F, rF, 1.0f), rF, rF);
        const float v = tp_at4(yt4, q) - wave_sum_dpp(isz ? am : 0.f);
        kfi = pz * rF;
        vfq[q] = v * rF;
        am = fmaf(kfi, v, am);
      }
      const float gi = mych ? ((c.pos == mycur ? 1.f : 0.f) - rnb) : 0.f;
      if (comp) {
        *kfw = kfi;
        pzv[lane] = pz;
        if (c.blk >= 0) *gslot = gi;
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
      tp_lds_order();
      if (comp) {
        const float pz1 = slope ? pzv[1] : 0.f;
        ci_f4v pa[NQ], ga[NQ], qa[NQ];
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          pa[qd] = ld4(pzv + 4 * qd); ga[qd] = ld4(gmine + 4 * qd);
          qa[qd] = slope ? ld4(Pm + DS + 4 * qd) : ci_f4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          const float pj[4] = {pa[qd].x, pa[qd].y, pa[qd].z, pa[qd].w};
          const float gj[4] = {ga[qd].x, ga[qd].y, ga[qd].z, ga[qd].w};
          if (slope) {
            const float p1[4] = {qa[qd].x, qa[qd].y, qa[qd].z, qa[qd].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float r = fmaf(-(pz * pj[u]), rF, prow[4 * qd + u]);
              if (lane == 0) r += fmaf(-(pz1 * pj[u]), rF, p1[u]);
              prow[4 * qd + u] = fmaf(myd2, gi * gj[u], r);
            }
            if (qd == 0) {
              prow[0] += prow[1];
              if (lane == 1) prow[1] += qs;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              prow[4 * qd + u] = fmaf(myd2, gi * gj[u], fmaf(-(pz * pj[u]), rF, prow[4 * qd + u]));
          }
          if (qd == 0 && lane == 0) prow[0] += ql;
          *(CI_LDS ci_f4v*)(Prow + 4 * qd) = ci_f4v{prow[4 * qd], prow[4 * qd + 1], prow[4 * qd + 2], prow[4 * qd + 3]};
        }
      }
      tp_lds_order();
    }
    if (lane == 0) *tp_p4w(c.vf + t4) = ci_f4v{vfq[0], vfq[1], vfq[2], vfq[3]};
  }
}

// Backward recursion over the chunk from r at its end; STORE: rs[t] = r_{t-1}.  Returns r_{s-1}.
template <bool STORE>
static __device__ __noinline__ float tp_backward_pass(const TpCtx& cref, int s, int e, float r) {
  const TpCtx c = cref;
  const int lane = c.lane, D = c.D, T = c.T;
  const bool comp = lane < D;
  TpGBC cidb = c.cidx + (size_t)c.blk0 * c.TP;
  for (int t4 = ((e - 1) & ~3); t4 >= s; t4 -= 4) {
    const ci_f4v vf4 = *tp_p4(c.vf + t4);
    const uint32_t mk4 = *tp_pu(c.msk + t4);
    const uint32_t cw4 = *tp_pu(cidb + t4);
    float kfq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kfq[q] = (comp && t4 + q < e) ? c.kf[(size_t)(t4 + q) * D + lane] : 0.f;
#pragma unroll
    for (int q = 3; q >= 0; --q) {
      const int t = t4 + q;
      if (t >= e) continue;
      if (t + 1 < T) {
        if (c.has_slope) {
          const float r0 = readlane_f(r, 0);
          if (lane == 1) r += r0;
        }
      } else {
        r = 0.f;
      }
      if (((mk4 >> (8 * q)) & 0xFFu) == 0u) {
        const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
        const float kr = wave_sum_dpp(kfq[q] * r);
        if (lane == 0 || (c.blk >= 0 && c.pos == mycur)) r += tp_at4(vf4, q) - kr;
      }
      if (STORE && comp) c.rs[(size_t)t * D + lane] = r;
    }
  }
  return r;
}

// g . r of this lane's block for the shock of the change at step t - 1 (g = e_{slot observed at
// t-1} - 1/n), from a lane-distributed r: every lane of the block gets the value.
__device__ __forceinline__ float tp_shock_dot(const TpCtx& c, float r, int cprev) {
  TpL vb = c.vb;
  tp_lds_sync();
  if (c.lane < c.D) vb[c.lane] = r;
  tp_lds_sync();
  float out = 0.f;
  if (c.blk >= 0) {
    float sb = 0.f;
    for (int q = 0; q < c.nb; ++q) sb += vb[c.boff + q];
    out = vb[c.boff + cprev] - sb * c.rnb;
  }
  return out;
}

// Forward reconstruction over the chunk (ci_seasonal.h pass 3 on a range): x^ from `xh`, x+ from
// `xp`, r_t from rs (r_end beyond the chunk).  Writes level / slope / observed effects, leaves the
// chunk's share of the scale statistics and its boundary records in stat[0..TP_STAT):
//   [0] ss_level [1] ss_slope [2+k] ss_drift_k | [10] first level [11] first slope [12+k] first
//   x~[slot observed at s] | [20] last level [21] last slope [22+k] last x~[slot observed at e]
//   (statistics of the step from e - 1 to e are the next chunk's boundary: formed by the reader).
static __device__ __noinline__ void tp_recon_pass(const TpCtx& cref, int s, int e, float xh, float xp, float r_end,
                                                 TpG stat) {
  const TpCtx c = cref;
  const int lane = c.lane, D = c.D, T = c.T, TP = c.TP;
  const bool comp = lane < D;
  TpGC zkb = c.zk + (size_t)c.blk0 * TP;
  TpGC gdb = c.gd + (size_t)c.blk0 * TP;
  TpGBC cidb = c.cidx + (size_t)c.blk0 * TP;
  // g . r_{e-1} for the step that crosses the chunk's end
  float gd_end = 0.f;
  if (e < T) gd_end = tp_shock_dot(c, r_end, (int)cidb[e - 1]);
  float prev = 0.f, ssl = 0.f, sss = 0.f, ssd = 0.f;
  bool ch_prev = false;
  for (int t4 = s; t4 < e; t4 += 4) {
    const ci_f4v zl4 = *tp_p4(c.zl + t4);
    const ci_f4v zk4 = *tp_p4(zkb + t4);
    ci_f4v zs4 = ci_f4v{0.f, 0.f, 0.f, 0.f};
    if (c.has_slope) zs4 = *tp_p4(c.zs + t4);
    const uint32_t cb4 = *tp_pu(c.cbv + t4);
    const uint32_t cw4 = *tp_pu(cidb + t4);
    float rnq[4], gdq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t1 = t4 + q + 1;
      rnq[q] = t1 < e ? (comp ? c.rs[(size_t)t1 * D + lane] : 0.f) : (t1 < T ? r_end : 0.f);
      gdq[q] = t1 < e ? gdb[t1] : (t1 < T ? gd_end : 0.f);
    }
    float xo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t4 + q;
      xo[q] = 0.f;
      if (t >= e) continue;
      const int mycur = (int)((cw4 >> (8 * q)) & 0xFFu);
      const float xt = xh + xp;
      xo[q] = xt;
      if (t > s) {
        const float pslope = readlane_f(prev, 1);
        if (lane == 0) {
          float dl = xt - prev;
          if (c.has_slope) dl -= pslope;
          ssl = fmaf(dl, dl, ssl);
        }
        if (c.has_slope && lane == 1) { const float ds = xt - prev; sss = fmaf(ds, ds, sss); }
        if (ch_prev && c.pos == mycur) {
          const float w = (float)c.nb * (prev - xt);
          ssd = fmaf(w, w, ssd);
        }
      } else {
        // first record: level, slope, the slot each block observes at s
        const float l0 = readlane_f(xt, 0), l1 = readlane_f(xt, 1);
        if (lane == 0) { stat[10] = l0; stat[11] = c.has_slope ? l1 : 0.f; }
        if (c.blk >= 0 && c.pos == mycur) stat[12 + c.blk] = xt;
      }
      if (c.blk >= 0 && c.pos == mycur) c.seas[(size_t)c.blk * TP + t] = xt;
      prev = xt;
      if (t + 1 < T) {
        const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
        const bool mych = c.blk >= 0 && ((cb >> c.blk) & 1u);
        float h = xh, r = xp;
        if (c.has_slope) {
          const float h1 = readlane_f(xh, 1), s1 = readlane_f(xp, 1);
          if (lane == 0) { h += h1; r += s1; }
        }
        if (lane == 0) { h = fmaf(c.ql, rnq[q], h); r = fmaf(c.sl, tp_at4(zl4, q), r); }
        if (c.has_slope && lane == 1) { h = fmaf(c.qs, rnq[q], h); r = fmaf(c.ssc, tp_at4(zs4, q), r); }
        if (mych) {
          const float dgi = c.mydrift * ((c.pos == mycur ? 1.f : 0.f) - c.rnb);
          h = fmaf(c.mydrift * dgi, gdq[q], h);
          r = fmaf(dgi, tp_at4(zk4, q), r);
        }
        xh = h; xp = r;
        ch_prev = mych;
      }
    }
    if (lane == 0) *tp_p4w(c.lev + t4) = ci_f4v{xo[0], xo[1], xo[2], xo[3]};
    if (c.has_slope && lane == 1) *tp_p4w(c.slp + t4) = ci_f4v{xo[0], xo[1], xo[2], xo[3]};
  }
  // last record: level, slope, and -- for the blocks that change between e - 1 and e -- the slot
  // observed at e (whose entry of x~_{e-1} the next chunk's first step is compared with)
  {
    const float l0 = readlane_f(prev, 0), l1 = readlane_f(prev, 1);
    if (lane == 0) { stat[20] = l0; stat[21] = c.has_slope ? l1 : 0.f; }
    if (c.blk >= 0 && e < T) {
      const int cnext = (int)cidb[e];
      if (c.pos == cnext) stat[22 + c.blk] = prev;
    }
  }
#pragma unroll
  for (int k = 0; k < SMAXK; ++k)
    if (k < c.K) {
      const float tot = wave_sum_dpp(c.blk == k ? ssd : 0.f);
      if (lane == 0) stat[2 + k] = tot;
    }
  const float sl0 = readlane_f(ssl, 0), ss1 = readlane_f(sss, 1);
  if (lane == 0) { stat[0] = sl0; stat[1] = c.has_slope ? ss1 : 0.f; }
}

// (the regression draw of models with more than MAXP design columns: spike_slab_draw_big_wg, ci_bigp.h)

// ------------------------------------------------------------------------------------
// the persistent Gibbs kernel (iteration structure of gibbs_seasonal_kernel / the oracle's
// ci_oracle_fit_gibbs; gibbs_sampler.fit_with_gibbs_sampling called at causalimpact_lib.py:365)
// ------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(TP_NT) void gibbs_seasonal_tp_kernel(SArgs a) {
  constexpr int NR = 4 * NQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const KArgs& g = a.k;
  const int T = g.T, P = g.P, K = a.K;
  // The chunk grid is fixed by the SERIES (G virtual workgroups of TP_NWV chunks: a.Lc carries G,
  // chosen from T alone); the launch decides how many REAL workgroups (a.cluster = Gc, a power of
  // two <= G) share them -- workgroup `role` runs the virtual workgroups [role G/Gc, (role+1) G/Gc).
  // Same chunks, same scan trees, same arithmetic whatever Gc: the draws do not depend on the launch
  // size (bit for bit), and the fall-back "first workgroup alone" is just Gc = 1.
  const int G = a.Lc, Gc = a.cluster;
  // workgroup -> (chain, role): the workgroups of one chain share an XCD (ids equal mod 8)
  int chain_id = blockIdx.x, role = 0;
  if (Gc > 1) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    role = slot % Gc;
    chain_id = (slot / Gc) * 8 + xcd;
  }
  if (chain_id >= g.B * g.C) return;
  const int series = chain_id / g.C, chain = chain_id % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  const int trend = a.has_slope ? 2 : 1;
  int D = trend;
  TpCtx cx;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) {
    cx.off[k] = D; cx.nsz[k] = (k < K) ? a.nseas[k] : 0;
    if (k < K) D += cx.nsz[k];
  }
  const TpLayout L = make_tplayout(T, P, K, D, a.has_slope, G);
  const TpLds LL = make_tplds(P, D);
  const int N = L.N, Lc = L.Lc, TP = L.TP, LVI = L.LVI, LVG = L.LVG;
  unsigned char* wsc = reinterpret_cast<unsigned char*>(a.ws) + chain_lin * a.ws_stride;
  int* csync = a.csync + chain_lin * TPC_INTS;
  int* mode_lds = tp_opaque(reinterpret_cast<int*>(smem + LL.shared));
  double* shd = tp_opaque(reinterpret_cast<double*>(smem + LL.shared + 16));     // wavefront 0 -> workgroup

  // ---- the cluster
  TpSync sy;
  sy.flags = csync; sy.G = Gc; sy.g = role; sy.epoch = 0; sy.cluster = false; sy.light = false;
  int v0 = 0, v1 = G;                 // the virtual workgroups this workgroup runs
  if (Gc > 1) {
    if (a.cluster_drop != 0 && role == a.cluster_drop) return;     // test knob: never checks in
    const int mode = tp_assemble(csync, role, Gc, tid, mode_lds);
    if (mode == 3) {
      if (role > 0) return;
      v0 = 0; v1 = G;                 // alone: every chunk, workgroup barriers only
    } else {
      sy.cluster = true; sy.light = mode == 2;
      const int per = G / Gc; v0 = role * per; v1 = v0 + per;
    }
  }
  const bool is_main = role == 0;

  // ---- pointers
  cx.T = T; cx.TP = TP; cx.D = D; cx.DS = tp_ds(NR); cx.K = K; cx.lane = lane; cx.has_slope = a.has_slope;
  cx.yv = (TpG)(wsc + L.yv); cx.lev = (TpG)(wsc + L.lev); cx.slp = (TpG)(wsc + L.slp);
  cx.xw = (TpG)(wsc + L.xw); cx.ytil = (TpG)(wsc + L.ytil); cx.vf = (TpG)(wsc + L.vf);
  cx.zl = (TpG)(wsc + L.zl); cx.zs = (TpG)(wsc + L.zs); cx.zo = (TpG)(wsc + L.zo);
  cx.seas = (TpG)(wsc + L.seas); cx.zk = (TpG)(wsc + L.zk); cx.gd = (TpG)(wsc + L.gd);
  cx.kf = (TpG)(wsc + L.kf); cx.rs = (TpG)(wsc + L.rs);
  cx.msk = (TpGB)(wsc + L.mask); cx.cbv = (TpGB)(wsc + L.cbits); cx.cidx = (TpGB)(wsc + L.cidx);
  {
    unsigned char* wl = smem + LL.wave0 + (size_t)wave * LL.wave_stride;
    cx.cm = tp_lds<float>(wl + LL.cm); cx.am = tp_lds<float>(wl + LL.am); cx.scr = tp_lds<float>(wl + LL.scr);
    cx.pzv = tp_lds<float>(wl + LL.pzv); cx.vb = tp_lds<float>(wl + LL.vb);
  }
  TpG e0 = (TpG)(wsc + L.e0); TpG ei = (TpG)(wsc + L.ei); TpG et = (TpG)(wsc + L.et);
  TpG stt = (TpG)(wsc + L.st);
  TpG bm = (TpG)(wsc + L.bm); TpG bi = (TpG)(wsc + L.bi); TpG bt = (TpG)(wsc + L.bt);
  TpG xsum = (TpG)(wsc + L.xsum); TpG statv = (TpG)(wsc + L.stat);
  TpG cpart = (TpG)(wsc + L.cpart); TpG cw = (TpG)(wsc + L.cw);
  const size_t ESZ = tp_esz(NR), BSZ = tp_bsz(NR), SSZ = (size_t)NR * (NR + 4);
  const int PR = (P + 4) & ~3;                       // floats of one chunk's X~'targets partials (+ y'y)
  const int CWS = (P + 3) & ~3;                      // cw: weights, then the scalars
  RegLds R;
  R.xtx = const_cast<double*>(g.xtx) + (size_t)series * P * P;
  R.omega = const_cast<double*>(g.omega) + (size_t)series * P * P;
  R.bvec = tp_opaque((double*)(smem + LL.reg + LL.bvec)); R.w = tp_opaque((float*)(smem + LL.w));
  R.aug[0] = tp_opaque((double*)(smem + LL.reg + LL.aug0)); R.aug[1] = R.aug[0];
  R.pri[0] = tp_opaque((double*)(smem + LL.pri0)); R.pri[1] = R.pri[0];
  R.chol = tp_opaque((double*)(smem + LL.reg + LL.chol)); R.zv = tp_opaque((double*)(smem + LL.reg + LL.zv));
  R.uperm = tp_opaque((double*)(smem + LL.reg + LL.uperm));
  R.nz = tp_opaque((int*)(smem + LL.reg + LL.nz)); R.perm = tp_opaque((int*)(smem + LL.reg + LL.perm));
  R.idx = tp_opaque((int*)(smem + LL.reg + LL.idx));
  const bool bigp = P > MAXP;
  if (bigp) {
    // the matrices (and X'X, Omega) in the chain's HBM workspace -- or in LDS where the host found
    // room (big_a / big_p) --, the draw's small state in LDS
    int* nz_l = R.nz; int* perm_l = R.perm; int* idx_l = R.idx; double* uperm_l = R.uperm;
    bigp_point(R, wsc + L.big, P);
    R.nz = nz_l; R.perm = perm_l; R.idx = idx_l; R.uperm = uperm_l;
  }
  const DevSeriesParams sp = g.sp[series];
  const DevSeasonalParams ss = a.ssp[series];
  const Rng rng{stream_key0(g.seed0, g.series_stream_base, series), stream_key1(g.seed1, g.series_stream_base, series),
                (uint32_t)(g.chain_offset + chain)};
  TpGC Xg = (TpGC)(g.Xt + (size_t)series * P * T);
  TpGC chol1 = (TpGC)(a.p1_chol + (size_t)series * a.dred * a.dred);

  // ---- lane roles
  int blk = -1, pos = 0, nb = 1, boff = 0, rbase = 0;
  {
    int rr = trend;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        if (lane >= cx.off[k] && lane < cx.off[k] + cx.nsz[k]) {
          blk = k; pos = lane - cx.off[k]; nb = cx.nsz[k]; boff = cx.off[k]; rbase = rr;
        }
        rr += cx.nsz[k] - 1;
      }
  }
  cx.blk = blk; cx.blk0 = blk >= 0 ? blk : 0; cx.pos = pos; cx.nb = nb; cx.boff = boff;
  cx.rnb = 1.0f / (float)nb;
  const bool comp = lane < D;

  // the chunks of this wavefront: c = v * TP_NWV + wave for v in [v0, v1)
  auto chunk_s = [&](int c) { const int s = c * Lc; return s < T ? s : T; };
  auto chunk_e = [&](int c) { const int e = (c + 1) * Lc; return e < T ? e : T; };

  // ---- stage constants (own chunks): mask, outcome, change bits, c_k(t); zero the latents
  for (int v = v0; v < v1; ++v) {
    const int c = v * TP_NWV + wave, s = c * Lc, e = s + Lc;
    for (int t = s + lane; t < e; t += 64) {
      const bool in = t < T;
      const bool m = in ? g.mask[(size_t)series * T + t] != 0 : true;
      cx.msk[t] = m ? 1 : 0;
      cx.yv[t] = m ? 0.f : g.y[(size_t)series * T + t];
      cx.lev[t] = 0.f; cx.xw[t] = 0.f; cx.ytil[t] = 0.f; cx.vf[t] = 0.f; cx.zl[t] = 0.f; cx.zo[t] = 0.f;
      if (a.has_slope) { cx.slp[t] = 0.f; cx.zs[t] = 0.f; }
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          cx.seas[(size_t)k * TP + t] = 0.f; cx.zk[(size_t)k * TP + t] = 0.f; cx.gd[(size_t)k * TP + t] = 0.f;
          if (in && a.season_change[(size_t)k * T + t]) bits |= 1u << k;
        }
      cx.cbv[t] = (uint8_t)bits;
    }
    // c_k(t) = (changes of block k before t) mod n_k: lane k walks its block through the chunk
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        float cnt = 0.f;
        for (int t = lane; t < s && t < T; t += 64) cnt += a.season_change[(size_t)k * T + t] ? 1.f : 0.f;
        const int before = (int)(wave_sum_dpp(cnt) + 0.5f);
        if (lane == k) {
          int cur = before % cx.nsz[k];
          for (int t = s; t < e; ++t) {
            cx.cidx[(size_t)k * TP + t] = (uint8_t)cur;
            if (t + 1 < T && a.season_change[(size_t)k * T + t]) cur = (cur + 1 == cx.nsz[k]) ? 0 : cur + 1;
          }
        }
      }
    if (K == 0) for (int t = s + lane; t < e; t += 64) cx.cidx[t] = 0;
    if (lane < TP_STAT) statv[(size_t)c * TP_STAT + lane] = 0.f;
  }
  if (is_main && wave == 0)
    for (int j = lane; j < (P > 16 ? P : 16); j += 64) R.w[j] = 0.f;
  double n_changes[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) {
    n_changes[k] = 0.0;
    if (k < K && is_main && wave == 0) {
      float cnt = 0.f;
      for (int t = lane; t + 1 < T; t += 64) cnt += a.season_change[(size_t)k * T + t] ? 1.f : 0.f;
      n_changes[k] = (double)wave_sum_dpp(cnt);
    }
  }
  tp_cluster_barrier(sy, tid);

  double obs_scale = sp.obs_scale0, level_scale = sp.level_scale0, slope_scale = sp.slope_scale0;
  double drift[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) drift[k] = (k < K) ? ss.drift_scale0[k] : 0.0;
  const float p1l = (float)(sp.init_level_scale * sp.init_level_scale);
  const float p1s = (float)(sp.init_slope_scale * sp.init_slope_scale);
  const float p1e = (float)(ss.init_seasonal_scale * ss.init_seasonal_scale);
  Prof prof;
  prof.start(g.prof, g.prof != nullptr && blockIdx.x == 0 && tid == 0);
  PriorCarry pc;
  pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;

  const int n_iter = g.W + g.S;
  for (int it = 0; it <= n_iter; ++it) {
    tp_wg_barrier_wave();             // the draw of it-1 written by other lanes of this wavefront
    // ---- (1) targets of the regression and this chunk's share of X~'targets, y'y
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      float yty = 0.f;
      for (int t = s + lane; t < e; t += 64) {
        float tg = 0.f;
        if (!cx.msk[t]) {
          tg = cx.yv[t] - cx.lev[t];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K) tg -= cx.seas[(size_t)k * TP + t];
        }
        cx.ytil[t] = tg;
        yty = fmaf(tg, tg, yty);
      }
      tp_wg_barrier_wave();
      // four features per round (their loads in flight together), the targets in registers
      {
        const int ta = s + lane, tb = s + lane + 64;
        const float tga = ta < e ? cx.ytil[ta] : 0.f, tgb = tb < e ? cx.ytil[tb] : 0.f;
        const bool wide_chunk = e - s > 128;
        for (int j0 = 0; j0 < P; j0 += 4) {
          float xa[4], xb[4], pj[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < P ? j0 + u : P - 1;
            xa[u] = ta < e ? Xg[(size_t)j * T + ta] : 0.f;
            xb[u] = tb < e ? Xg[(size_t)j * T + tb] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) pj[u] = fmaf(xb[u], tgb, xa[u] * tga);
          if (wide_chunk) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u < P ? j0 + u : P - 1;
              for (int t = s + lane + 128; t < e; t += 64) pj[u] = fmaf(Xg[(size_t)j * T + t], cx.ytil[t], pj[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float sj = wave_sum_dpp(pj[u]);
            if (lane == 0 && j0 + u < P) cpart[(size_t)c * PR + j0 + u] = sj;
          }
        }
      }
      const float s0 = wave_sum_dpp(yty);
      if (lane == 0) cpart[(size_t)c * PR + P] = s0;
    }
    prof.tick(20);
    tp_cluster_barrier(sy, tid);                                            // (A)
    // ---- (2) wavefront 0 of the chain: statistics, scale draws of it-1, regression draw of it
    if (is_main && wave == 0) {
      for (int j = lane; j <= P; j += 64) {
        float sj = 0.f;
        for (int c = 0; c < N; ++c) sj += cpart[(size_t)c * PR + j];       // chunk order: fixed bits
        R.bvec[j] = (double)sj;
      }
      // statistics of the draw of it-1: the chunks' shares + the steps across chunk boundaries
      float ssl = 0.f, sss = 0.f, ssdk[SMAXK];
#pragma unroll
      for (int k = 0; k < SMAXK; ++k) ssdk[k] = 0.f;
      for (int c = lane; c < N; c += 64) {
        TpGC sc = statv + (size_t)c * TP_STAT;
        if (chunk_s(c) >= T) continue;
        ssl += sc[0]; sss += sc[1];
#pragma unroll
        for (int k = 0; k < SMAXK; ++k) if (k < K) ssdk[k] += sc[2 + k];
        if (c > 0) {
          TpGC sp_ = statv + (size_t)(c - 1) * TP_STAT;
          float dl = sc[10] - sp_[20];
          if (a.has_slope) { dl -= sp_[21]; const float ds = sc[11] - sp_[21]; sss = fmaf(ds, ds, sss); }
          ssl = fmaf(dl, dl, ssl);
          const unsigned cb = cx.cbv[chunk_s(c) - 1];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K && ((cb >> k) & 1u)) {
              const float w = (float)cx.nsz[k] * (sp_[22 + k] - sc[12 + k]);
              ssdk[k] = fmaf(w, w, ssdk[k]);
            }
        }
      }
      ssl = wave_sum_dpp(ssl); sss = wave_sum_dpp(sss);
#pragma unroll
      for (int k = 0; k < SMAXK; ++k) if (k < K) ssdk[k] = wave_sum_dpp(ssdk[k]);
      tp_lds_sync();
      if (it > 0) {
        const uint32_t pit = (uint32_t)(it - 1);
        level_scale = scale_draw(sp.level_conc, sp.level_scale, sp.level_ub, (double)(T - 1), (double)ssl,
                                 rng, pit, SITE_LEVEL_SCALE, lane);
        if (a.has_slope)
          slope_scale = scale_draw(sp.slope_conc, sp.slope_scale, sp.slope_ub, (double)(T - 1), (double)sss,
                                   rng, pit, SITE_SLOPE_SCALE, lane);
#pragma unroll
        for (int k = 0; k < SMAXK; ++k)
          if (k < K) {
            const double gk = gamma_wave(ss.drift_conc + 0.5 * n_changes[k], rng, pit, SITE_DRIFT_SCALE, (uint32_t)k, lane);
            const double sd = (double)__fsqrt_rn((float)((ss.drift_scale + 0.5 * (double)ssdk[k]) * fast_rcp(gk)));
            drift[k] = sd < ss.drift_ub ? sd : ss.drift_ub;
          }
        if (P == 0)
          obs_scale = scale_draw(sp.obs_conc, sp.obs_scale, sp.obs_ub, sp.n_obs, R.bvec[P], rng, pit,
                                 SITE_OBS_SCALE, lane);
        const int sidx = it - 1 - g.W;
        if (sidx >= 0) {
          const size_t o = chain_lin * g.S + sidx;
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
        }
      }
      if (lane == 0) cw[CWS + 0] = (float)obs_scale;        // sigma_obs of the draw of it-1 (emission)
      if (it < n_iter && P > 0) {
        const double g_obs = gamma_wave(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
        if (P <= 16)
          obs_scale = spike_slab_draw_regs(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, pc);
        else if (!bigp)
          obs_scale = spike_slab_draw(R, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, prof, it == 0);
        else if (lane == 0) { shd[0] = obs_scale; shd[1] = g_obs; }     // the whole workgroup draws (below)
      }
      wave_sync();
    }
    if (is_main && bigp && it < n_iter) {
      tp_wg_barrier();
      typedef CI_GLB double* GD;
      double ns;
      CI_LDS unsigned* ijt = LL.ijtab ? tp_lds<unsigned>(smem + LL.reg + LL.ijtab) : (CI_LDS unsigned*)0;
      if (LL.big_a && LL.big_p)
        ns = spike_slab_draw_big_wg<TP_NT>(R, tp_lds<double>(smem + LL.reg + LL.big_a), tp_lds<double>(smem + LL.big_p), tp_lds<double>(smem + LL.reg + LL.trow), ijt, R.w, P, sp,
                                              shd[0], shd[1], rng, (uint32_t)it, tid, it == 0, prof, 21, 9);
      else if (LL.big_a)
        ns = spike_slab_draw_big_wg<TP_NT>(R, tp_lds<double>(smem + LL.reg + LL.big_a), (GD)R.pri[0], tp_lds<double>(smem + LL.reg + LL.trow), ijt, R.w, P, sp,
                                              shd[0], shd[1], rng, (uint32_t)it, tid, it == 0, prof, 21, 9);
      else
        ns = spike_slab_draw_big_wg<TP_NT>(R, (GD)R.aug[0], (GD)R.pri[0], tp_lds<double>(smem + LL.reg + LL.trow), ijt, R.w, P, sp,
                                              shd[0], shd[1], rng, (uint32_t)it, tid, it == 0, prof, 21, 9);
      if (wave == 0) obs_scale = ns;
    }
    if (is_main && wave == 0) {
      for (int j = lane; j < P; j += 64) cw[j] = R.w[j];
      if (lane == 0) {
        cw[CWS + 1] = (float)obs_scale; cw[CWS + 2] = (float)level_scale; cw[CWS + 3] = (float)slope_scale;
      }
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K && lane == 0) cw[CWS + 4 + k] = (float)drift[k];
    }
    prof.tick(21);
    tp_cluster_barrier(sy, tid);                                            // (B)
    const float emit_so = cw[CWS + 0];
    cx.so = cw[CWS + 1]; cx.sl = cw[CWS + 2]; cx.ssc = cw[CWS + 3];
    cx.H = cx.so * cx.so; cx.ql = cx.sl * cx.sl; cx.qs = cx.ssc * cx.ssc;
    cx.mydrift = 0.f; cx.myd2 = 0.f;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K && blk == k) { cx.mydrift = cw[CWS + 4 + k]; cx.myd2 = cx.mydrift * cx.mydrift; }
    if (K > 0 && blk < 0) { const float d0 = cw[CWS + 4]; cx.myd2 = d0 * d0; }      // (lanes outside the blocks: as ci_seasonal.h's d2[blk0])
    // ---- (3) emission of the draw of it-1; residual and normals of this iteration (own chunks)
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      const int sidx = it - 1 - g.W;
      if (it > 0 && sidx >= 0) {
        const uint32_t pit = (uint32_t)(it - 1);
        const size_t o = chain_lin * g.S + sidx, row = o * T;
        // (4-step blocks of the UNCLAMPED chunk: an empty chunk at the end of the series must not
        // visit the last block of its predecessor a second time -- the running sum below)
        for (int q4 = ((c * Lc) >> 2) + lane; q4 < (((c + 1) * Lc) >> 2); q4 += 64) {
          float zp[4];
          normals4(site_call(rng, pit, SITE_PRED, 0, (uint32_t)q4), zp);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * q4 + q;
            if (t < e) {
              float loc = cx.lev[t] + cx.xw[t];
#pragma unroll
              for (int k = 0; k < SMAXK; ++k)
                if (k < K) {
                  const float sv = cx.seas[(size_t)k * TP + t];
                  loc += sv;
                  if (a.out_seasonal) a.out_seasonal[(row + t) * K + k] = sv;
                }
              if (g.out_level) g.out_level[row + t] = cx.lev[t];
              if (g.out_slope && a.has_slope) g.out_slope[row + t] = cx.slp[t];
              if (g.out_traj) g.out_traj[row + t] = fmaf(emit_so, zp[q], loc);
              if (g.out_pred_mean) {
                float* pm = g.out_pred_mean + chain_lin * T + t;
                *pm = (sidx == 0 ? 0.f : *pm) + loc;
              }
            }
          }
        }
      }
      if (it == n_iter) continue;
      tp_wg_barrier_wave();           // (the emission has read the previous X w)
      for (int t = s + lane; t < e; t += 64) {
        float sx = 0.f;
        int j = 0;
        for (; j + 8 <= P; j += 8) {
          float xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[u] = Xg[(size_t)(j + u) * T + t];
#pragma unroll
          for (int u = 0; u < 8; ++u) sx = fmaf(xv[u], cw[j + u], sx);
        }
        for (; j < P; ++j) sx = fmaf(Xg[(size_t)j * T + t], cw[j], sx);
        cx.xw[t] = sx;
      }
      for (int q4 = ((c * Lc) >> 2) + lane; q4 < (((c + 1) * Lc) >> 2); q4 += 64) {
        float z4[4];
        normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_LEVEL, 0, (uint32_t)q4), z4);
        *tp_p4w(cx.zl + 4 * q4) = ci_f4v{z4[0], z4[1], z4[2], z4[3]};
        normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_OBS, 0, (uint32_t)q4), z4);
        *tp_p4w(cx.zo + 4 * q4) = ci_f4v{z4[0], z4[1], z4[2], z4[3]};
        if (a.has_slope) {
          normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_SLOPE, 0, (uint32_t)q4), z4);
          *tp_p4w(cx.zs + 4 * q4) = ci_f4v{z4[0], z4[1], z4[2], z4[3]};
        }
#pragma unroll
        for (int k = 0; k < SMAXK; ++k)
          if (k < K) {
            normals4(site_call(rng, (uint32_t)it, SITE_PRIOR_SEAS, (uint32_t)k, (uint32_t)q4), z4);
            *tp_p4w(cx.zk + (size_t)k * TP + 4 * q4) = ci_f4v{z4[0], z4[1], z4[2], z4[3]};
          }
      }
    }
    if (it == n_iter) break;
    tp_wg_barrier_wave();
    prof.tick(22);
    // x+_0 = chol(P_1) z in the oracle's reduced coordinates, folded into the prior mean
    float a1e = 0.f;
    {
      float zi0 = 0.f;
      if (lane < a.dred) {
        float z1[1];
        fill_normals<1>(rng, (uint32_t)it, SITE_PRIOR_INIT, 0, (uint32_t)lane, z1);
        zi0 = z1[0];
      }
      tp_lds_sync();
      if (lane < NR) cx.vb[lane] = zi0;
      tp_lds_sync();
      float x0 = 0.f;
      if (lane < a.dred)
        for (int j = 0; j <= lane; ++j) x0 = fmaf(chol1[lane * a.dred + j], cx.vb[j], x0);
      if (lane < NR) cx.vb[NR + lane] = x0;
      tp_lds_sync();
      TpLC x0r = cx.vb + NR;
      if (lane == 0) a1e = (float)sp.init_level_loc + x0r[0];
      if (a.has_slope && lane == 1) a1e = x0r[1];
      if (blk >= 0) {
        if (pos < nb - 1) a1e = x0r[rbase + pos];
        else { float sx = 0.f; for (int q = 0; q < nb - 1; ++q) sx += x0r[rbase + q]; a1e = -sx; }
      }
      tp_lds_sync();
    }
    // ---- (4) prior simulation: chunk sums, prefix, y~
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      const float xs = tp_sim_pass<false>(cx, s, e, 0.f);
      if (lane < NR) xsum[(size_t)c * 2 * NR + lane] = comp ? xs : 0.f;
    }
    tp_cluster_barrier(sy, tid);                                            // (C)
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      float xp = 0.f;
      for (int cc = 0; cc < c; ++cc) {
        const int ss_ = chunk_s(cc), ee_ = chunk_e(cc);
        const int ntr = (ee_ < T ? ee_ : T - 1) - ss_;        // transitions inside chunk cc
        if (ntr <= 0) continue;
        if (a.has_slope) {
          const float x1 = readlane_f(xp, 1);
          if (lane == 0) xp = fmaf((float)ntr, x1, xp);
        }
        xp += (lane < NR) ? xsum[(size_t)cc * 2 * NR + lane] : 0.f;
      }
      if (lane < NR) xsum[(size_t)c * 2 * NR + NR + lane] = xp;            // x+ at the chunk's start
      (void)tp_sim_pass<true>(cx, s, e, xp);
    }
    tp_wg_barrier_wave();
    prof.tick(23);
    // ---- (5) filtering elements of the chunks, their scan, the chunks' predicted moments
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      tp_build_pass<NQ>(cx, s, e, c == 0, a1e, p1l, p1s, p1e, e0 + (size_t)c * ESZ);
      if (c == 0) {
        const TRow<NR> p0 = tp_prior_row<NR>(cx, p1l, p1s, p1e);
        trow_store<NR>(stt, NR + 4, lane, p0);
        if (lane < NR) *tp_p4w(stt + (size_t)lane * (NR + 4) + NR) = ci_f4v{comp ? a1e : 0.f, 0.f, 0.f, 0.f};
      }
    }
    prof.tick(24);
    tp_wg_barrier();
    for (int l = 0; l < LVI; ++l) {
      for (int v = v0; v < v1; ++v) {
        const int c = v * TP_NWV + wave;
        TpGC src = l == 0 ? e0 : ei + (size_t)(l - 1) * N * ESZ;
        TpG dst = ei + (size_t)l * N * ESZ + (size_t)c * ESZ;
        if (wave >= (1 << l)) tp_combine<NR, false>(src + (size_t)(c - (1 << l)) * ESZ, src + (size_t)c * ESZ, dst, cx.scr, cx.vb, cx.pzv, D, lane);
        else tp_elem_copy<NR>(dst, src + (size_t)c * ESZ, lane);
        if (l == LVI - 1 && wave == TP_NWV - 1) {       // the workgroup's total
          tp_wg_barrier_wave();
          tp_elem_copy<NR>(et + (size_t)v * ESZ, dst, lane);
        }
      }
      tp_wg_barrier();
    }
    prof.tick(29);
    TpGC eincl = ei + (size_t)(LVI - 1) * N * ESZ;
    if (Gc > 1) tp_cluster_barrier(sy, tid); else tp_wg_barrier();          // (D)
    prof.tick(30);
    for (int m = 0; m < LVG; ++m) {
      for (int v = v0; v < v1; ++v) {
        if (wave != (v & (TP_NWV - 1))) continue;
        TpGC src = et + (size_t)m * G * ESZ;
        TpG dst = et + (size_t)(m + 1) * G * ESZ + (size_t)v * ESZ;
        if (v >= (1 << m)) tp_combine<NR, false>(src + (size_t)(v - (1 << m)) * ESZ, src + (size_t)v * ESZ, dst, cx.scr, cx.vb, cx.pzv, D, lane);
        else tp_elem_copy<NR>(dst, src + (size_t)v * ESZ, lane);
      }
      tp_cluster_barrier(sy, tid);
    }
    TpGC etot = et + (size_t)LVG * G * ESZ;
    prof.tick(31);
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave;
      TpG dst = stt + (size_t)c * SSZ;
      if (c == 0) continue;
      if (wave == 0) tp_state_from_elem<NR>(etot + (size_t)(v - 1) * ESZ, dst, lane);
      else if (v == 0) tp_state_from_elem<NR>(eincl + (size_t)(c - 1) * ESZ, dst, lane);
      else tp_combine<NR, true>(etot + (size_t)(v - 1) * ESZ, eincl + (size_t)(c - 1) * ESZ, dst, cx.scr, cx.vb, cx.pzv, D, lane);
    }
    tp_wg_barrier_wave();
    prof.tick(25);
    // ---- (6) the filter replayed from the true moments; backward maps and their suffix scan
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      TpGC st = stt + (size_t)c * SSZ;
      float cvec = 0.f;
      if (s < T) {
        tp_filter_pass<NQ>(cx, s, e, st);
        tp_wg_barrier_wave();
        cvec = tp_backward_pass<false>(cx, s, e, 0.f);
      }
      // M = (I + J P_start)^-1 A'
      TpG bo = bm + (size_t)c * BSZ;
      if constexpr (NR % 8 == 0) {
        THalf<NR / 2> M = thalf_zero<NR / 2>();
        if (s < T) {
          const TpElemPtr<NR> el{e0 + (size_t)c * ESZ};
          hscr_put<NR>(cx.scr, hload<NR>(st, NR + 4, lane), lane);
          THalf<NR / 2> W = hmul<NR>(el.Jf(lane), cx.scr, D, lane, true);
          M = htranspose<NR>(cx.scr, el.Ah(lane), lane);
          THalf<NR / 2> dummy = thalf_zero<NR / 2>();
          float du = 0.f;
          hgauss_jordan<NR, 1>(W, M, dummy, du, cx.pzv, D, lane);
        } else {
#pragma unroll
          for (int u = 0; u < NR / 2; ++u) M.v[u] = ((lane & 31) < D && (lane >> 5) * (NR / 2) + u == (lane & 31)) ? 1.f : 0.f;
        }
        hstore<NR>(bo, NR + 4, lane, M);
      } else {
        TRow<NR> M = trow_zero<NR>();
        if (s < T) {
          const TpElemPtr<NR> el{e0 + (size_t)c * ESZ};
          tscr_put<NR>(cx.scr, trow_load<NR>(st, NR + 4, lane), lane);
          TRow<NR> W = tmul<NR>(el.J(lane), cx.scr, D, lane, true);
          M = ttranspose<NR>(cx.scr, el.A(lane), lane);
          TRow<NR> dummy = trow_zero<NR>();
          float du = 0.f;
          tgauss_jordan<NR, 1>(W, M, dummy, du, D, lane);
        } else {
#pragma unroll
          for (int u = 0; u < NR; ++u) M.v[u] = (comp && u == lane) ? 1.f : 0.f;
        }
        trow_store<NR>(bo, NR + 4, lane, M);
      }
      if (lane < NR) *tp_p4w(bo + (size_t)lane * (NR + 4) + NR) = ci_f4v{comp ? cvec : 0.f, 0.f, 0.f, 0.f};
    }
    prof.tick(26);
    tp_wg_barrier();
    for (int l = 0; l < LVI; ++l) {
      for (int v = v0; v < v1; ++v) {
        const int c = v * TP_NWV + wave;
        TpGC src = l == 0 ? bm : bi + (size_t)(l - 1) * N * BSZ;
        TpG dst = bi + (size_t)l * N * BSZ + (size_t)c * BSZ;
        if (wave + (1 << l) < TP_NWV) tp_bcompose<NR>(src + (size_t)c * BSZ, src + (size_t)(c + (1 << l)) * BSZ, dst, cx.scr, cx.vb, D, lane);
        else tp_bcopy<NR>(dst, src + (size_t)c * BSZ, lane);
        if (l == LVI - 1 && wave == 0) {
          tp_wg_barrier_wave();
          tp_bcopy<NR>(bt + (size_t)v * BSZ, dst, lane);
        }
      }
      tp_wg_barrier();
    }
    TpGC bincl = bi + (size_t)(LVI - 1) * N * BSZ;
    if (Gc > 1) tp_cluster_barrier(sy, tid); else tp_wg_barrier();          // (E)
    // (r at the end of a workgroup's last chunk needs the LATER workgroups' total maps applied to
    // zero: a chain of at most G - 1 matrix-vector products, formed by every wavefront for itself --
    // no further cluster barriers, no products of maps)
    prof.tick(27);
    // ---- (7) r through the chunk from its true end value, the draw, its statistics
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      if (s >= T) continue;
      float r_in = 0.f;                                   // r at the end of this workgroup's last chunk
      for (int vv = G - 1; vv > v; --vv) {
        TpGC tm = bt + (size_t)vv * BSZ;
        if ((size_t)vv * TP_NWV * Lc >= (size_t)T) continue;              // (empty workgroup: identity)
        const TRow<NR> Mv = trow_load<NR>(tm, NR + 4, lane);
        const float cv_ = lane < NR ? tm[(size_t)lane * (NR + 4) + NR] : 0.f;
        r_in = cv_ + tdot<NR>(Mv, cx.vb, r_in, lane);
      }
      float r_end = r_in;
      if (wave + 1 < TP_NWV) {
        TpGC nx = bincl + (size_t)(c + 1) * BSZ;
        const TRow<NR> Mn = trow_load<NR>(nx, NR + 4, lane);
        const float cn = lane < NR ? nx[(size_t)lane * (NR + 4) + NR] : 0.f;
        r_end = cn + tdot<NR>(Mn, cx.vb, r_in, lane);
      }
      if (!comp) r_end = 0.f;
      (void)tp_backward_pass<true>(cx, s, e, r_end);
      tp_wg_barrier_wave();
      // g . r_{t-1} per block over the own steps
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          const float rn = 1.0f / (float)cx.nsz[k];
          for (int t = s + lane; t < e; t += 64) {
            TpGC rr = cx.rs + (size_t)t * D + cx.off[k];
            float sb = 0.f;
            for (int q = 0; q < cx.nsz[k]; ++q) sb += rr[q];
            const int cprev = t > 0 ? (int)cx.cidx[(size_t)k * TP + t - 1] : 0;
            cx.gd[(size_t)k * TP + t] = rr[cprev] - sb * rn;
          }
        }
      tp_wg_barrier_wave();
      // x^ at the chunk's start = a + P r_{s-1}
      TpGC st = stt + (size_t)c * SSZ;
      const TRow<NR> Ps = trow_load<NR>(st, NR + 4, lane);
      const float as = lane < NR ? st[(size_t)lane * (NR + 4) + NR] : 0.f;
      const float rs0 = comp ? cx.rs[(size_t)s * D + lane] : 0.f;
      float xh = as + tdot<NR>(Ps, cx.vb, rs0, lane);
      if (!comp) xh = 0.f;
      const float xp0 = lane < NR ? xsum[(size_t)c * 2 * NR + NR + lane] : 0.f;
      tp_recon_pass(cx, s, e, xh, xp0, r_end, statv + (size_t)c * TP_STAT);
    }
    prof.tick(28);
  }
  tp_wg_barrier_wave();
  if (g.out_pred_mean) {
    const float inv = 1.0f / (float)(g.S > 0 ? g.S : 1);
    for (int v = v0; v < v1; ++v) {
      const int c = v * TP_NWV + wave, s = chunk_s(c), e = chunk_e(c);
      for (int t = s + lane; t < e; t += 64) g.out_pred_mean[chain_lin * T + t] *= inv;
    }
  }
}
#endif  // CI_SEASONAL_DECL_ONLY

}  // namespace ci
