// ci_kernels.h -- the Gibbs hot path as ONE persistent workgroup per (series, chain).
//
// What the reference does per Gibbs iteration, as
// three sequential length-T loops plus a length-P sweep of dynamic-shape Choleskys,
// all dispatched op by op through TensorFlow:
//   (a) spike-and-slab draw of (sigma^2_obs, weights)   spike_and_slab.SpikeSlabSampler
//   (b) residual y - X w
//   (c) Durbin-Koopman draw of the latent path          LGSSM.posterior_sample
//   (d) conjugate inverse-gamma scale draws             gibbs_sampler._resample_scale
// (reference call site causalimpact/causalimpact_lib.py:365-388).
//
// MI355X design (DESIGN.md "Kernel"):
//   * 256 threads (4 wavefronts) own one chain for all W+S iterations: no launches, no
//     host round trips, state in VGPRs / LDS, outputs streamed once to HBM.
//   * time is parallel: thread i owns the L consecutive steps [i*L, (i+1)*L).  The prior
//     simulation, the Kalman filter (Sarkka associative elements) and the backward
//     smoother are block-wide scans: in-wave Kogge-Stone over wavefront shuffles, one LDS
//     hand-off between the 4 waves, local sequential fix-up over the L owned steps.
//   * the P x P regression algebra is wave-cooperative in float64 in LDS: a sweep operator
//     makes each inclusion-flip proposal O(1) reads and each accepted flip one rank-1
//     update, instead of a fresh Cholesky per proposal.
//   * randomness is the specified Philox stream (ci_rng.h), so results do not depend on
//     how chains are spread over devices.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "ci_linalg.h"
#include "ci_rng.h"

namespace ci {

constexpr int NT = 256;   // threads per workgroup
constexpr int NW = 4;     // wavefronts per workgroup
constexpr int MAXP = 52;  // design columns supported by the LDS-resident regression block
constexpr int HMC_MAXP = 128;   // design columns of the log-likelihood / HMC path (row H; round 5: was MAXP)

struct DevSeriesParams {
  double level_conc, level_scale, level_ub;
  double slope_conc, slope_scale, slope_ub;
  double obs_conc, obs_scale, obs_ub;
  double nonzero_prob;
  double init_level_loc, init_level_scale, init_slope_scale;
  double obs_scale0, level_scale0, slope_scale0;
  double n_obs;
};

struct KArgs {
  int T, P, W, S, C, B, chain_offset;
  int series_stream_base;  // >= 0: series b of the launch draws from the Philox streams of series
                           // id (series_stream_base + b) -- distinct random numbers per series;
                           // -1: every series uses the streams of series 0 (CI_FLAG_SHARED_SERIES_STREAMS)
  uint32_t seed0, seed1;
  int x_in_lds;
  const float* y;          // [B,T]   0 where masked
  const uint8_t* mask;     // [B,T]
  const float* Xt;         // [B,P,T] feature-major
  const double* xtx;       // [B,P,P] X~'X~ over observed rows
  const double* omega;     // [B,P,P] 0.01 (XtX/2 + diag(XtX)/2) / T over ALL rows
  const DevSeriesParams* sp;  // [B]
  float* out_obs;          // [B,C,S]
  float* out_level_scale;  // [B,C,S]
  float* out_slope_scale;  // [B,C,S]
  float* out_weights;      // [B,C,S,P]
  float* out_level;        // [B,C,S,T]
  float* out_slope;        // [B,C,S,T]
  float* out_pred_mean;    // [B,C,T]
  float* out_traj;         // [B,C,S,T]
  long long* prof;         // optional [16] per-phase cycle counters (block 0, thread 0)
  // Streaming of results to the host while the fit runs (ci_session_run_streamed): every
  // progress_every retained draws (and after the last one) the workgroup makes its output rows
  // visible to the copy engines (system-scope release) and publishes the number of complete
  // draws of its chain in progress[series * C + chain] (host-coherent pinned memory).
  unsigned int* progress;
  int progress_every;
  int dbg;                 // $CI_SCHED_WORD: replaces the eight-wave kernel's helper-wave schedule word
                           // (ci_kernels8.h SCHED_DEFAULT) -- timing experiments and the
                           // timing-independence test; 0 in production
};

// The random stream of (series, chain): Philox counter word 3 = the global chain id (all 32 bits),
// key = (seed0 ^ h1(series id), seed1 ^ h2(series id)) with h1, h2 two bijective 32-bit mixers
// that fix 0.  Series 0 -- every single-series fit -- keeps the plain seeds, so the result of a
// series does not depend on whether it was fitted alone or as series 0 of a batch.  Both key words
// carry the id: two runs that share seed0 (an int seed s maps to (0, s), causalimpact_lib.py:535-539)
// can only meet on a key if h1(b) == h1(b'), i.e. b == b', and then seed1 must agree too -- a batch
// under seed s and the same batch under seed s' never share a stream.  (Round 4 folded the id
// into seed1 by a raw XOR: series b under seed s aliased series b ^ s ^ s' under seed s'.  Rounds
// 1-3 packed series and chain id into the counter word, 16 bits each.)
__host__ __device__ inline uint32_t stream_mix32(uint32_t h) {   // murmur3 finaliser: bijective, 0 -> 0
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t stream_sid(int series_stream_base, int series) {
  return series_stream_base < 0 ? 0u : (uint32_t)(series_stream_base + series);
}
__host__ __device__ inline uint32_t stream_key0(uint32_t seed0, int series_stream_base, int series) {
  return seed0 ^ stream_mix32(stream_sid(series_stream_base, series));
}
__host__ __device__ inline uint32_t stream_key1(uint32_t seed1, int series_stream_base, int series) {
  return seed1 ^ stream_mix32(stream_sid(series_stream_base, series) * 0x9E3779B9u);
}

// ------------------------------------------------------------------------------------
// wave / block primitives
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Per-phase cycle accounting (s_memtime) for DESIGN.md's time budget; one thread records.
struct Prof {
  long long* p;
  long long t;
  __device__ __forceinline__ void start(long long* ptr, bool active) {
    p = active ? ptr : nullptr;
    if (p) t = clock64();
  }
  __device__ __forceinline__ void tick(int slot) {
    if (p) {
      const long long n = clock64();
      p[slot] += n - t;
      t = n;
    }
  }
};

// The same interface compiled to nothing: the production instantiations of the Gibbs kernel carry
// no profiling state (4 registers and ~25 predicated branches per iteration less; the kernel sits
// at the 256-VGPR limit and the instrumented variant spills 40 bytes per lane).
struct NoProf {
  __device__ __forceinline__ void start(long long*, bool) {}
  __device__ __forceinline__ void tick(int) {}
};

template <class E> struct Arr { float f[sizeof(E) / 4]; };

template <class E> __device__ __forceinline__ E shfl_up_e(const E& e, int off) {
  Arr<E> a = __builtin_bit_cast(Arr<E>, e);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i) a.f[i] = __shfl_up(a.f[i], off, 64);
  return __builtin_bit_cast(E, a);
}
template <class E> __device__ __forceinline__ E shfl_down_e(const E& e, int off) {
  Arr<E> a = __builtin_bit_cast(Arr<E>, e);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i) a.f[i] = __shfl_down(a.f[i], off, 64);
  return __builtin_bit_cast(E, a);
}
template <class E> __device__ __forceinline__ void lds_store_e(float* p, const E& e) {
  const Arr<E> a = __builtin_bit_cast(Arr<E>, e);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i) p[i] = a.f[i];
}
template <class E> __device__ __forceinline__ E lds_load_e(const float* p) {
  Arr<E> a;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i) a.f[i] = p[i];
  return __builtin_bit_cast(E, a);
}

// One DPP move of every float of an element: lanes the control word gives no source keep their own
// value (callers mask those lanes anyway).
template <int CTRL, int ROW_MASK, class E> __device__ __forceinline__ E dpp_move_e(const E& e) {
  Arr<E> a = __builtin_bit_cast(Arr<E>, e);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(E) / 4); ++i)
    a.f[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a.f[i]), __float_as_int(a.f[i]),
                                                        CTRL, ROW_MASK, 0xF, false));
  return __builtin_bit_cast(E, a);
}

// Inclusive scan of one wavefront, earlier lanes first, for any associative op(earlier, later), on
// the DPP crossbar (VALU moves, no LDS round trips): Kogge-Stone inside each row of 16 lanes
// (row_shr 1, 2, 4, 8), then the last lane of row 0 / 2 joins rows 1 / 3 (row_bcast:15) and lane 31
// joins rows 2 and 3 (row_bcast:31).
template <class E, class Op>
__device__ __forceinline__ E wave_scan_incl_fwd(const E& v, Op op, int lane) {
  E incl = v;
  const int r = lane & 15;
  { const E o = dpp_move_e<0x111, 0xF>(incl); if (r >= 1) incl = op(o, incl); }
  { const E o = dpp_move_e<0x112, 0xF>(incl); if (r >= 2) incl = op(o, incl); }
  { const E o = dpp_move_e<0x114, 0xF>(incl); if (r >= 4) incl = op(o, incl); }
  { const E o = dpp_move_e<0x118, 0xF>(incl); if (r >= 8) incl = op(o, incl); }
  { const E o = dpp_move_e<0x142, 0xA>(incl); if (lane & 16) incl = op(o, incl); }
  { const E o = dpp_move_e<0x143, 0xC>(incl); if (lane & 32) incl = op(o, incl); }
  return incl;
}
// Inclusive SUFFIX scan: lane i gets v_i o v_{i+1} o ... o v_63 with op(outer, inner).  Inside the
// rows the same DPP pattern mirrored (row_shl); there is no backward row broadcast, so the two
// steps across rows read the first lane of the next row / of row 2 through v_readlane.
template <class E, class Op>
__device__ __forceinline__ E wave_scan_incl_bwd(const E& v, Op op, int lane) {
  E incl = v;
  const int r = lane & 15;
  { const E o = dpp_move_e<0x101, 0xF>(incl); if (r < 15) incl = op(incl, o); }
  { const E o = dpp_move_e<0x102, 0xF>(incl); if (r < 14) incl = op(incl, o); }
  { const E o = dpp_move_e<0x104, 0xF>(incl); if (r < 12) incl = op(incl, o); }
  { const E o = dpp_move_e<0x108, 0xF>(incl); if (r < 8) incl = op(incl, o); }
  {
    // rows 0 and 2 take the suffix of row 1 / row 3, held by that row's first lane: two fixed
    // source lanes, read through v_readlane (no LDS round trip)
    Arr<E> a = __builtin_bit_cast(Arr<E>, incl);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(E) / 4); ++i) {
      const int v16 = __builtin_amdgcn_readlane(__float_as_int(a.f[i]), 16);
      const int v48 = __builtin_amdgcn_readlane(__float_as_int(a.f[i]), 48);
      a.f[i] = __int_as_float(lane < 32 ? v16 : v48);
    }
    const E o = __builtin_bit_cast(E, a);
    if ((lane & 16) == 0) incl = op(incl, o);
  }
  {
    // rows 0 and 1 take the suffix of rows 2-3 (lane 32)
    Arr<E> a = __builtin_bit_cast(Arr<E>, incl);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(E) / 4); ++i)
      a.f[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.f[i]), 32));
    const E o = __builtin_bit_cast(E, a);
    if ((lane & 32) == 0) incl = op(incl, o);
  }
  return incl;
}

// Exclusive block scan over thread order (thread 0 first).  op(earlier, later).
// Contains exactly one __syncthreads(); `slots` needs NW * sizeof(E)/4 floats and must not
// be rewritten before another barrier has been passed.
// The scan in two halves around its barrier, for callers that own the barrier (ci_kernels5.h runs
// the first half while the regression wave is still in its serial section and lets the
// workgroup's (B3) separate the halves): begin = the in-wave inclusive scan + the wave total to
// `slots`; finish = the exclusive prefix from the earlier waves' totals.
template <class E, class Op>
__device__ __forceinline__ E block_scan_fwd_begin(const E& tot, Op op, float* slots, int lane, int wave) {
  constexpr int N = sizeof(E) / 4;
  const E incl = wave_scan_incl_fwd(tot, op, lane);
  if (lane == 63) lds_store_e(slots + wave * N, incl);
  return incl;
}
template <class E, class Op>
__device__ __forceinline__ E block_scan_fwd_finish(const E& incl, Op op, const E& ident, const float* slots,
                                                   int lane, int wave) {
  constexpr int N = sizeof(E) / 4;
  E ex = dpp_move_e<0x138, 0xF>(incl);        // wave_shr:1
  if (lane == 0) ex = ident;
  if (wave == 0) return ex;
  E wp = lds_load_e<E>(slots);
  for (int ww = 1; ww < wave; ++ww) wp = op(wp, lds_load_e<E>(slots + ww * N));
  return op(wp, ex);
}
template <class E, class Op>
__device__ __forceinline__ E block_scan_excl_fwd(const E& tot, Op op, const E& ident, float* slots,
                                                 int lane, int wave) {
  const E incl = block_scan_fwd_begin(tot, op, slots, lane, wave);
  __syncthreads();
  return block_scan_fwd_finish(incl, op, ident, slots, lane, wave);
}

// Exclusive suffix scan: result for thread i = e_{i+1} o e_{i+2} o ... o e_last, with
// op(outer, inner) (outer is applied after inner; the last thread's element acts first).
template <class E, class Op>
__device__ __forceinline__ E block_scan_excl_bwd(const E& tot, Op op, const E& ident, float* slots,
                                                 int lane, int wave) {
  constexpr int N = sizeof(E) / 4;
  const E incl = wave_scan_incl_bwd(tot, op, lane);
  if (lane == 0) lds_store_e(slots + wave * N, incl);
  __syncthreads();
  E ex = dpp_move_e<0x130, 0xF>(incl);        // wave_shl:1
  if (lane == 63) ex = ident;
  if (wave == NW - 1) return ex;
  E ws = lds_load_e<E>(slots + (wave + 1) * N);
  for (int ww = wave + 2; ww < NW; ++ww) ws = op(ws, lds_load_e<E>(slots + ww * N));
  return op(ex, ws);
}

// L consecutive floats row[t0 .. t0 + L) of a length-T row in global memory, 0 beyond the end --
// straight-line code (clamped addresses, zeroed by selects): the callers hoist every wave-uniform
// choice (LDS copy / wide / scalar) OUT of their loops over rows, because a branch around a load
// makes the compiler wait for it at the join, which serialises the rows of a batch (one L2 round
// trip per row instead of one per batch).  Wide: T % 4 == 0 and a 16-byte aligned row.
template <int L>
__device__ __forceinline__ void global_row_load_wide(const float* __restrict__ row, int t0, int T,
                                                     float (&v)[L]) {
  static_assert(L % 4 == 0, "wide rows need L % 4 == 0");
#pragma unroll
  for (int q = 0; q < L / 4; ++q) {
    const int t = t0 + 4 * q;
    const float4 x = *reinterpret_cast<const float4*>(row + (t < T ? t : T - 4));
    const bool in = t < T;
    v[4 * q] = in ? x.x : 0.f; v[4 * q + 1] = in ? x.y : 0.f;
    v[4 * q + 2] = in ? x.z : 0.f; v[4 * q + 3] = in ? x.w : 0.f;
  }
}
template <int L>
__device__ __forceinline__ void global_row_load_scalar(const float* __restrict__ row, int t0, int T,
                                                       float (&v)[L]) {
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    const float x = row[t < T ? t : T - 1];
    v[l] = t < T ? x : 0.f;
  }
}

// L consecutive floats of an LDS row starting at a 4*L-byte aligned index, as wide loads.
template <int L>
__device__ __forceinline__ void lds_row_load(const float* p, float (&v)[L]) {
  if constexpr (L % 4 == 0) {
#pragma unroll
    for (int q = 0; q < L / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else if constexpr (L == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = p[0];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Same sum through the DPP crossbar (no LDS round trips): inclusive scan inside each row of
// 16 lanes (row_shr 1, 2, 4, 8), then row_bcast15 / row_bcast31 carry the row totals; lane 63
// holds the wave total, returned wave-uniform.  ~6 dependent VALU ops instead of 6
// ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF,
                                                        true));
}
// lane 63 of the result holds the wave total (every lane holds its inclusive prefix)
__device__ __forceinline__ float wave_prefix_dpp(float v) {
  v = dpp_add<0x111, 0xF>(v);
  v = dpp_add<0x112, 0xF>(v);
  v = dpp_add<0x114, 0xF>(v);
  v = dpp_add<0x118, 0xF>(v);
  v = dpp_add<0x142, 0xA>(v);
  v = dpp_add<0x143, 0xC>(v);
  return v;
}
// 16 values per lane -> lane l holds the wave-wide sum of value (l & 15).  Reduce-scatter over
// the 16 lanes of each row: at every step a lane keeps the half of its values whose index bit
// matches its own lane bit and adds the partner's copy of that half (partners: row_ror:8,
// row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2] -- each an involution pairing lanes
// with complementary bit and equal higher bits); then the four rows are added.
template <int CTRL> __device__ __forceinline__ float dpp_take(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_reduce_scatter16(const float (&v)[16], int lane) {
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
  float a[8], b[4], c[2];
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = (b3 ? v[8 + q] : v[q]) + dpp_take<0x128>(b3 ? v[q] : v[8 + q]);
#pragma unroll
  for (int q = 0; q < 4; ++q) b[q] = (b2 ? a[4 + q] : a[q]) + dpp_take<0x141>(b2 ? a[q] : a[4 + q]);
#pragma unroll
  for (int q = 0; q < 2; ++q) c[q] = (b1 ? b[2 + q] : b[q]) + dpp_take<0x4E>(b1 ? b[q] : b[2 + q]);
  float d = (b0 ? c[1] : c[0]) + dpp_take<0xB1>(b0 ? c[0] : c[1]);
  d += __shfl_xor(d, 16, 64);
  d += __shfl_xor(d, 32, 64);
  return d;
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0x111, 0xF>(v);   // row_shr:1
  v = dpp_add<0x112, 0xF>(v);   // row_shr:2
  v = dpp_add<0x114, 0xF>(v);   // row_shr:4
  v = dpp_add<0x118, 0xF>(v);   // row_shr:8
  v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ------------------------------------------------------------------------------------
// X~'targets and y'y with a summation order that does not depend on WHO sums WHAT: feature j's
// sum is  lane l: its float4 chunks l, l + 64, ... of the series in that order, four fused
// multiply-adds per chunk in time order;  then the 64 lane sums through the fixed DPP tree of
// wave_prefix_dpp (lane 63 stores).  Any wavefront can take any feature -- the four-wave kernel
// gives each of its waves 4 features, the eight-wave latency kernel 2 (its regression and
// randomness waves join in) -- and the bits are the same.  No cross-wave reduction, no gather.
// tg: tpad floats in LDS (0 where missing / beyond T); Xs: rows of tpad floats; sums[0..15] the
// features, sums[16] = y'y (from the wave called with with_yty).
// ------------------------------------------------------------------------------------
constexpr int RED_YTY = 16;        // slot of y'y in the new layout of `red`
constexpr int RED_INC = 20;        // then (ss_level, ss_slope) per time wave
// XG: the design is NOT in LDS (long series): rows of T floats in global memory (L2-resident),
// read as float4 when `xwide` (T % 4 == 0 and 16-byte aligned rows), else as guarded scalars;
// chunks beyond the series read as zeros.  Same values, same order => same bits as the LDS variant.
template <int L, int NF, bool XG = false, bool ROLL = false>
__device__ __forceinline__ void xt_sums_wave(const float* tg, const float* Xs, int tpad, int P, int j0,
                                             bool with_yty, float* sums, int lane, int T = 0,
                                             bool xwide = true) {
  float4 tq[L];
#pragma unroll
  for (int c = 0; c < L; ++c) tq[c] = *reinterpret_cast<const float4*>(tg + 4 * (lane + 64 * c));
  float acc[NF];
  auto dot = [&](const float4 (&xq)[L]) {
    float sv = 0.f;
#pragma unroll
    for (int c = 0; c < L; ++c) {
      sv = fmaf(xq[c].x, tq[c].x, sv);
      sv = fmaf(xq[c].y, tq[c].y, sv);
      sv = fmaf(xq[c].z, tq[c].z, sv);
      sv = fmaf(xq[c].w, tq[c].w, sv);
    }
    return sv;
  };
  if constexpr (!XG) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int j = j0 + f;
      const float* row = Xs + (size_t)(j < P ? j : P - 1) * tpad;
      float4 xq[L];
#pragma unroll
      for (int c = 0; c < L; ++c) xq[c] = *reinterpret_cast<const float4*>(row + 4 * (lane + 64 * c));
      acc[f] = dot(xq);
    }
  } else if (xwide) {
    // (the choice of the row source is hoisted out of the feature loop: a branch around a load
    //  makes the compiler wait for it at the join)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int j = j0 + f;
      const float* row = Xs + (size_t)(j < P ? j : P - 1) * T;
      float4 xq[L];
#pragma unroll
      for (int c = 0; c < L; ++c) {
        const int t = 4 * (lane + 64 * c);
        const float4 x = *reinterpret_cast<const float4*>(row + (t < T ? t : T - 4));
        xq[c] = t < T ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      acc[f] = dot(xq);
    }
  } else {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int j = j0 + f;
      const float* row = Xs + (size_t)(j < P ? j : P - 1) * T;
      if constexpr (ROLL) {
        // guarded scalar reads (T % 4 != 0), 4 L per feature: four quads of loads in flight at a time
        // in a ROLLED loop, the targets re-read from LDS -- unrolled, the 4 L loads of every feature
        // were hoisted together and cost the four-wave L = 16 build 2 KB of scratch per lane.  Same
        // products in the same order as dot().  (ROLL: the four-wave long-series build only -- the
        // eight-wave kernel measured 8 % SLOWER at T = 4096 with it, less scratch notwithstanding.)
        float sv = 0.f;
#pragma unroll 1
        for (int c0 = 0; c0 < L; c0 += 4) {
          float4 xq[4], tq4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = 4 * (lane + 64 * (c0 + u));
            const float x0 = row[t < T ? t : T - 1], x1 = row[t + 1 < T ? t + 1 : T - 1];
            const float x2 = row[t + 2 < T ? t + 2 : T - 1], x3 = row[t + 3 < T ? t + 3 : T - 1];
            xq[u] = make_float4(t < T ? x0 : 0.f, t + 1 < T ? x1 : 0.f, t + 2 < T ? x2 : 0.f, t + 3 < T ? x3 : 0.f);
            tq4[u] = *reinterpret_cast<const float4*>(tg + t);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            sv = fmaf(xq[u].x, tq4[u].x, sv);
            sv = fmaf(xq[u].y, tq4[u].y, sv);
            sv = fmaf(xq[u].z, tq4[u].z, sv);
            sv = fmaf(xq[u].w, tq4[u].w, sv);
          }
        }
        acc[f] = sv;
      } else {
        float4 xq[L];
#pragma unroll
        for (int c = 0; c < L; ++c) {
          const int t = 4 * (lane + 64 * c);
          const float x0 = row[t < T ? t : T - 1], x1 = row[t + 1 < T ? t + 1 : T - 1];
          const float x2 = row[t + 2 < T ? t + 2 : T - 1], x3 = row[t + 3 < T ? t + 3 : T - 1];
          xq[c] = make_float4(t < T ? x0 : 0.f, t + 1 < T ? x1 : 0.f, t + 2 < T ? x2 : 0.f, t + 3 < T ? x3 : 0.f);
        }
        acc[f] = dot(xq);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = wave_prefix_dpp(acc[f]);
  if (lane == 63) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      if (j0 + f < P) sums[j0 + f] = acc[f];
  }
  if (with_yty) {
    float yy = 0.f;
#pragma unroll
    for (int c = 0; c < L; ++c) {
      yy = fmaf(tq[c].x, tq[c].x, yy);
      yy = fmaf(tq[c].y, tq[c].y, yy);
      yy = fmaf(tq[c].z, tq[c].z, yy);
      yy = fmaf(tq[c].w, tq[c].w, yy);
    }
    yy = wave_prefix_dpp(yy);
    if (lane == 63) sums[RED_YTY] = yy;
  }
}
// The owned steps' targets (0 where missing) to the shared vector.
template <int L>
__device__ __forceinline__ void store_targets(float* tgv, int t0, const float (&tg)[L]) {
  if constexpr (L % 4 == 0) {
#pragma unroll
    for (int q = 0; q < L / 4; ++q)
      *reinterpret_cast<float4*>(tgv + t0 + 4 * q) = make_float4(tg[4 * q], tg[4 * q + 1], tg[4 * q + 2], tg[4 * q + 3]);
  } else if constexpr (L == 2) {
    *reinterpret_cast<float2*>(tgv + t0) = make_float2(tg[0], tg[1]);
  } else {
    tgv[t0] = tg[0];
  }
}

// ------------------------------------------------------------------------------------
// Durbin-Koopman simulation smoother for the trend block, time-parallel.
// (LinearGaussianStateSpaceModel.posterior_sample reached from
//  gibbs_sampler._resample_latents; oracle: ci_oracle_dk_draw.)
// ------------------------------------------------------------------------------------
template <int D> struct DkModel {
  float H;        // observation-noise variance
  Vec<D> sig;     // state disturbance scales (level, slope)
  Vec<D> a1;      // prior mean of x_0
  Vec<D> p1;      // prior variances of x_0 (diagonal)
};

// ------------------------------------------------------------------------------------
// Time-parallel Kalman filter of the trend block on data ytil (Z = e_0'): associative scan
// over (A, b, C, eta, J), then a local sequential pass that leaves, for every owned step, the
// predicted mean a_t, predicted covariance P_t, gain K_t = P_t Z'/F_t and v_t/F_t (0 where
// masked).  F_t itself is returned in fvar (1 where masked).  `a1e` is the prior mean of x_0.
// Contains one __syncthreads().
// ------------------------------------------------------------------------------------
template <int D, int L, class PF>
__device__ __forceinline__ void kalman_filter_pass(const DkModel<D>& md, const Vec<D>& a1e,
                                                   const Vec<D>& q, const float (&ytil)[L],
                                                   uint32_t maskbits, int tid, int lane, int wave,
                                                   float* slots16, Vec<D> (&ap)[L],
                                                   Mat<D> (&Pp)[L], Vec<D> (&kf)[L], float (&vf)[L],
                                                   float (&fvar)[L], PF& prof) {
  // ---- (2) Kalman filter as an associative scan over (A, b, C, eta, J), C and J packed symmetric.
  // The chunk's element is built step by step: time update (A <- T A, b <- T b, C <- T C T' + Q),
  // then, where y_t is observed, a rank-one fold of the observation (Z = e_0'):
  //   S = C_00 + H ;  e = (y - b_0) / S ;  eta += A_0.' e ;  b += C_.0 e ;
  //   J += A_0.' A_0. / S ;  A -= C_.0 A_0. / S ;  C -= C_.0 C_0. / S
  // -- the product of the one-step elements without forming them.
  FElemS<D> fe = felems_identity<D>();
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const bool obs = ((maskbits >> l) & 1u) == 0u;
    if (tid == 0 && l == 0) {
      // prior element: A = 0, (b, C) = moments of x_0 after its own update
      fe.A = mzero<D>();
      fe.b = a1e;
#pragma unroll
      for (int i = 0; i < D; ++i) fe.C[symidx<D>(i, i)] = md.p1.v[i];
      if (obs) {
        const float F = md.p1.v[0] + md.H;
        const float k0 = md.p1.v[0] / F;
        fe.b.v[0] = fmaf(k0, ytil[l] - a1e.v[0], a1e.v[0]);
        fe.C[symidx<D>(0, 0)] = md.p1.v[0] * md.H / F;
      }
      continue;
    }
    // time update
    if constexpr (D == 2) {
#pragma unroll
      for (int j = 0; j < D; ++j) fe.A.m[0][j] += fe.A.m[1][j];
      fe.b.v[0] += fe.b.v[1];
      fe.C[symidx<D>(0, 0)] = fmaf(2.f, fe.C[symidx<D>(0, 1)], __fadd_rn(fe.C[symidx<D>(0, 0)], fe.C[symidx<D>(1, 1)]));
      fe.C[symidx<D>(0, 1)] += fe.C[symidx<D>(1, 1)];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) fe.C[symidx<D>(i, i)] += q.v[i];
    if (obs) {
      float za[D], cz[D];
#pragma unroll
      for (int j = 0; j < D; ++j) { za[j] = fe.A.m[0][j]; cz[j] = fe.C[symidx<D>(j, 0)]; }
      const float rS = __builtin_amdgcn_rcpf(cz[0] + md.H);
      const float e = (ytil[l] - fe.b.v[0]) * rS;
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
  }
  prof.tick(21);
  const FElemS<D> fpre = block_scan_excl_fwd(
      fe, [](const FElemS<D>& a, const FElemS<D>& b) { return felems_combine(a, b); },
      felems_identity<D>(), slots16, lane, wave);
  prof.tick(5);

  // local sequential Kalman pass over the owned steps (predicted-form quantities kept)
  {
    Vec<D> mf = fpre.b;
    Mat<D> Pf;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) Pf.m[i][j] = fpre.C[symidx<D>(i, j)];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const bool obs = ((maskbits >> l) & 1u) == 0u;
      Vec<D> a;
      Mat<D> P;
      if (tid == 0 && l == 0) {
        a = a1e;
        P = mzero<D>();
#pragma unroll
        for (int i = 0; i < D; ++i) P.m[i][i] = md.p1.v[i];
      } else {
        a = trans_apply(mf);
        P = trans_cov(Pf, q);
      }
      ap[l] = a;
      Pp[l] = P;
      if (obs) {
        const float v = ytil[l] - a.v[0];
        const float Fv = P.m[0][0] + md.H;
        const float rF = __builtin_amdgcn_rcpf(Fv);
        fvar[l] = Fv;
        vf[l] = v * rF;
#pragma unroll
        for (int i = 0; i < D; ++i) kf[l].v[i] = P.m[i][0] * rF;
#pragma unroll
        for (int i = 0; i < D; ++i) mf.v[i] = fmaf(kf[l].v[i], v, a.v[i]);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int j = 0; j < D; ++j) Pf.m[i][j] = fmaf(-kf[l].v[i], P.m[0][j], P.m[i][j]);
        symmetrize(Pf);
      } else {
        vf[l] = 0.f;
        fvar[l] = 1.f;
        kf[l] = vzero<D>();
        mf = a;
        Pf = P;
      }
    }
  }
}

// The 3 L normals thread `owner` consumes in dk_draw (level, slope, observation disturbances of
// its L steps).  They depend on (seed, chain, iteration) only, so the Gibbs kernel draws them
// while wave 0 is in the serial section.
template <int D, int L>
__device__ __forceinline__ void dk_normals(const Rng& rng, uint32_t iter, int owner, float (&zl)[L],
                                           float (&zs)[L], float (&zo)[L]) {
  const uint32_t t0 = (uint32_t)owner * L;
  fill_normals<L>(rng, iter, SITE_PRIOR_LEVEL, 0, t0, zl);
  if constexpr (D == 2) fill_normals<L>(rng, iter, SITE_PRIOR_SLOPE, 0, t0, zs);
  fill_normals<L>(rng, iter, SITE_PRIOR_OBS, 0, t0, zo);
}

// Step (1) of dk_draw, the scan of the prior simulation x <- T x + n_t, in two halves around a
// workgroup barrier: it needs the disturbance scales and the normals only, not the regression
// draw, so the five-wave kernel runs the first half before its (B3).
template <int D, int L>
__device__ __forceinline__ PElem<D> dk_prior_begin(const Vec<D>& sig, const float (&zl)[L],
                                                   const float (&zs)[L], float* slots, int lane,
                                                   int wave) {
  PElem<D> ptot = pelem_identity<D>();
#pragma unroll
  for (int l = 0; l < L; ++l) {
    PElem<D> e;
    e.k = 1.f;
    e.s.v[0] = sig.v[0] * zl[l];
    if constexpr (D == 2) e.s.v[1] = sig.v[1] * zs[l];
    ptot = pelem_combine(ptot, e);
  }
  return block_scan_fwd_begin(
      ptot, [](const PElem<D>& a, const PElem<D>& b) { return pelem_combine(a, b); }, slots, lane, wave);
}
template <int D>
__device__ __forceinline__ PElem<D> dk_prior_finish(const PElem<D>& incl, const float* slots, int lane,
                                                    int wave) {
  return block_scan_fwd_finish(
      incl, [](const PElem<D>& a, const PElem<D>& b) { return pelem_combine(a, b); },
      pelem_identity<D>(), slots, lane, wave);
}

// ------------------------------------------------------------------------------------
// The Durbin-Koopman draw in two phases.
//   MATRIX phase -- needs the variances (sigma^2_obs, the disturbance scales) and the
//   missing-data pattern, NOT the data: scan of the (A, C, J) part of the filtering elements
//   (FMElem), then a local pass that leaves for every owned step the predicted covariance P_t,
//   the gain k_t = P_t Z'/F_t, 1/F_t, and the two 2 x 2 products the data phase scans with:
//     Mf = G_{L-1} ... G_0,   G_t = (I - k_t Z) T      filtered mean:  m_t = G_t m_{t-1} + k_t y_t
//     Mb = N_0 ... N_{L-1},   N_t = (I - Z' k_t') T'   adjoint:  r_{t-1} = N_t r_t + Z' v_t / F_t
//   DATA phase -- two scans of affine maps of d-vectors (forward: the filtered means, backward:
//   the smoothing adjoint r) whose matrix parts are Mf / Mb, each followed by a local pass.
// The five-/eight-wave Gibbs kernels run the matrix phase while the regression wave is still
// drawing the weights (the data of the filter); every kernel calls the same functions, so all of
// them produce the same bits (-ffp-contract=on).
// ------------------------------------------------------------------------------------
template <int D, int L> struct DkMats {
  Mat<D> Pp[L];     // predicted covariance P_t
  Vec<D> kf[L];     // filter gain P_t Z' / F_t (0 where missing)
  float rF[L];      // 1 / F_t (0 where missing)
  Mat<D> Mf, Mb;
};

// The chunk's (A, C, J): time update, then -- where y_t is observed -- the rank-one fold of the
// observation (kalman_filter_pass (2) without its vector parts).
template <int D, int L>
__device__ __forceinline__ FMElem<D> dk_matrix_chunk(const DkModel<D>& md, const Vec<D>& q,
                                                     uint32_t maskbits, int tid) {
  FMElem<D> fe = fmelem_identity<D>();
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const bool obs = ((maskbits >> l) & 1u) == 0u;
    if (tid == 0 && l == 0) {
      // prior element: A = 0, C = covariance of x_0 after its own update
      fe.A = mzero<D>();
#pragma unroll
      for (int i = 0; i < D; ++i) fe.C[symidx<D>(i, i)] = md.p1.v[i];
      if (obs) {
        const float F = md.p1.v[0] + md.H;
        fe.C[symidx<D>(0, 0)] = md.p1.v[0] * md.H / F;
      }
      continue;
    }
    if constexpr (D == 2) {
#pragma unroll
      for (int j = 0; j < D; ++j) fe.A.m[0][j] += fe.A.m[1][j];
      fe.C[symidx<D>(0, 0)] = fmaf(2.f, fe.C[symidx<D>(0, 1)], __fadd_rn(fe.C[symidx<D>(0, 0)], fe.C[symidx<D>(1, 1)]));
      fe.C[symidx<D>(0, 1)] += fe.C[symidx<D>(1, 1)];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) fe.C[symidx<D>(i, i)] += q.v[i];
    if (obs) {
      float za[D], cz[D];
#pragma unroll
      for (int j = 0; j < D; ++j) { za[j] = fe.A.m[0][j]; cz[j] = fe.C[symidx<D>(j, 0)]; }
      const float rS = __builtin_amdgcn_rcpf(cz[0] + md.H);
#pragma unroll
      for (int i = 0; i < D; ++i) {
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
  }
  return fe;
}

// The three block scans of the draw, each in two halves around a workgroup barrier that the CALLER
// places (the eight-wave kernel shares its barriers with the regression and randomness waves).
// begin = the in-wave inclusive scan + the wave total to LDS; finish = what the thread needs from
// everything before (after) its chunk.  Across waves nothing is COMPOSED any more: the earlier
// waves' totals are APPLIED one after the other to the quantity that is carried -- a covariance
// (34 instead of 62 operations per wave), a mean, an adjoint (4 instead of 12) -- which is all the
// local passes start from.  Wave 0 holds x_0, whose element has A = 0 (no predecessor): its total
// maps every incoming state to the same (mean, covariance), so the chain starts from that total.
// slots: NW * 16 floats per scan.
template <int D>
__device__ __forceinline__ void fm_apply_cov(float (&P)[D * (D + 1) / 2], const FMElem<D>& e) {
  // P <- A (I + P J)^-1 P A' + C
  Mat<D> W, RC;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = (i == j) ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(P[symidx<D>(i, k)], e.J[symidx<D>(k, j)], s);
      W.m[i][j] = s;
      RC.m[i][j] = P[symidx<D>(i, j)];
    }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float rp = __builtin_amdgcn_rcpf(W.m[c][c]);
#pragma unroll
    for (int j = c + 1; j < D; ++j) W.m[c][j] *= rp;
#pragma unroll
    for (int j = 0; j < D; ++j) RC.m[c][j] *= rp;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = W.m[r][c];
#pragma unroll
      for (int j = c + 1; j < D; ++j) W.m[r][j] = fmaf(-f, W.m[c][j], W.m[r][j]);
#pragma unroll
      for (int j = 0; j < D; ++j) RC.m[r][j] = fmaf(-f, RC.m[c][j], RC.m[r][j]);
    }
  }
  const Mat<D> T1 = mm(e.A, RC);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      float s = e.C[symidx<D>(i, j)];
#pragma unroll
      for (int k = 0; k < D; ++k) s = fmaf(T1.m[i][k], e.A.m[j][k], s);
      P[symidx<D>(i, j)] = s;
    }
}

template <int D>
__device__ __forceinline__ FMElem<D> dk_matrix_scan_begin(const FMElem<D>& fe, float* slots_m, int lane,
                                                          int wave) {
  const FMElem<D> incl = wave_scan_incl_fwd(
      fe, [](const FMElem<D>& a, const FMElem<D>& b) { return fmelem_combine(a, b); }, lane);
  if (lane == 63) lds_store_e(slots_m + wave * 16, incl);
  return incl;
}
// -> the filtered covariance of the step before this thread's chunk (packed upper triangle)
template <int D>
__device__ __forceinline__ void dk_matrix_scan_finish(const FMElem<D>& incl, const float* slots_m,
                                                      int lane, int wave,
                                                      float (&Pin)[D * (D + 1) / 2]) {
  FMElem<D> ex = dpp_move_e<0x138, 0xF>(incl);        // wave_shr:1
  if (lane == 0) ex = fmelem_identity<D>();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) Pin[i] = ex.C[i];
    return;
  }
  FMElem<D> tot[NW - 1];
#pragma unroll
  for (int u = 0; u < NW - 1; ++u) tot[u] = lds_load_e<FMElem<D>>(slots_m + u * 16);
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) Pin[i] = tot[0].C[i];
#pragma unroll
  for (int u = 1; u < NW - 1; ++u)
    if (wave > u) fm_apply_cov<D>(Pin, tot[u]);
  fm_apply_cov<D>(Pin, ex);
}

template <int D>
__device__ __forceinline__ AElem<D> aff_fwd_begin(const AElem<D>& fe, float* slots_f, int lane, int wave) {
  // op(earlier, later) = later after earlier
  const AElem<D> incl = wave_scan_incl_fwd(
      fe, [](const AElem<D>& a, const AElem<D>& b) { return aelem_compose(b, a); }, lane);
  if (lane == 63) lds_store_e(slots_f + wave * 16, incl);
  return incl;
}
// -> the filtered mean of the step before this thread's chunk
template <int D>
__device__ __forceinline__ Vec<D> aff_fwd_finish(const AElem<D>& incl, const float* slots_f, int lane,
                                                 int wave) {
  AElem<D> ex = dpp_move_e<0x138, 0xF>(incl);
  if (lane == 0) ex = aelem_identity<D>();
  if (wave == 0) return ex.c;
  AElem<D> tot[NW - 1];
#pragma unroll
  for (int u = 0; u < NW - 1; ++u) tot[u] = lds_load_e<AElem<D>>(slots_f + u * 16);
  Vec<D> m = tot[0].c;
#pragma unroll
  for (int u = 1; u < NW - 1; ++u)
    if (wave > u) m = vadd(mv(tot[u].M, m), tot[u].c);
  return vadd(mv(ex.M, m), ex.c);
}

template <int D>
__device__ __forceinline__ AElem<D> aff_bwd_begin(const AElem<D>& be, float* slots_b, int lane, int wave) {
  const AElem<D> incl = wave_scan_incl_bwd(
      be, [](const AElem<D>& o, const AElem<D>& i) { return aelem_compose(o, i); }, lane);
  if (lane == 0) lds_store_e(slots_b + wave * 16, incl);
  return incl;
}
// -> the adjoint r of this thread's last step (r = 0 beyond the end of the series)
template <int D>
__device__ __forceinline__ Vec<D> aff_bwd_finish(const AElem<D>& incl, const float* slots_b, int lane,
                                                 int wave) {
  AElem<D> ex = dpp_move_e<0x130, 0xF>(incl);        // wave_shl:1
  if (lane == 63) ex = aelem_identity<D>();
  if (wave == NW - 1) return ex.c;
  AElem<D> tot[NW];
#pragma unroll
  for (int u = 1; u < NW; ++u) tot[u] = lds_load_e<AElem<D>>(slots_b + u * 16);
  Vec<D> r = tot[NW - 1].c;
#pragma unroll
  for (int u = NW - 2; u >= 1; --u)
    if (wave < u) r = vadd(mv(tot[u].M, r), tot[u].c);
  return vadd(mv(ex.M, r), ex.c);
}

// Local covariance pass from the filtered covariance `Pin` of the step before the chunk: P_t,
// gains, 1/F_t and the chunk's two mean maps.
template <int D, int L>
__device__ __forceinline__ void dk_matrix_local(const DkModel<D>& md, const Vec<D>& q,
                                                uint32_t maskbits, int tid,
                                                const float (&Pin)[D * (D + 1) / 2],
                                                DkMats<D, L>& km) {
  Mat<D> Pf;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) Pf.m[i][j] = Pin[symidx<D>(i, j)];
  // thread 0: x_0 has no predecessor -- its first step wipes any dependence on an incoming mean
  Mat<D> Mf = (tid == 0) ? mzero<D>() : meye<D>();
  Mat<D> Mb = meye<D>();
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const bool obs = ((maskbits >> l) & 1u) == 0u;
    Mat<D> P;
    if (tid == 0 && l == 0) {
      P = mzero<D>();
#pragma unroll
      for (int i = 0; i < D; ++i) P.m[i][i] = md.p1.v[i];
    } else {
      P = trans_cov(Pf, q);
      Mf = trans_left(Mf);
    }
    km.Pp[l] = P;
    Vec<D> k = vzero<D>();
    float rF = 0.f;
    if (obs) {
      rF = __builtin_amdgcn_rcpf(P.m[0][0] + md.H);
#pragma unroll
      for (int i = 0; i < D; ++i) k.v[i] = P.m[i][0] * rF;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pf.m[i][j] = fmaf(-k.v[i], P.m[0][j], P.m[i][j]);
      symmetrize(Pf);
    } else {
      Pf = P;
    }
    km.kf[l] = k;
    km.rF[l] = rF;
    // Mf <- (I - k Z) Mf   (row i loses k_i times row 0; k = 0 where missing)
    {
      float r0[D];
#pragma unroll
      for (int j = 0; j < D; ++j) r0[j] = Mf.m[0][j];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Mf.m[i][j] = fmaf(-k.v[i], r0[j], Mf.m[i][j]);
    }
    // Mb <- Mb (I - Z' k') T'   (column j loses k_j times column 0, then T' from the right)
    {
      Mat<D> X;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) X.m[i][j] = fmaf(-Mb.m[i][0], k.v[j], Mb.m[i][j]);
      Mb = trans_right_t(X);
    }
  }
  km.Mf = Mf;
  km.Mb = Mb;
}

// One step of the adjoint recursion r_{t-1} = (I - Z' k_t')(T' r_t) + Z' v_t / F_t.
template <int D, int L>
__device__ __forceinline__ Vec<D> dk_adjoint_step(const DkMats<D, L>& km, int l, const Vec<D>& r,
                                                  float vfl) {
  Vec<D> u = trans_t_apply(r);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) dot = fmaf(km.kf[l].v[i], u.v[i], dot);
  u.v[0] = (u.v[0] - dot) + vfl;
  return u;
}

// Forward chunk map of the filtered mean (matrix part from the matrix phase, vector part here).
template <int D, int L>
__device__ __forceinline__ AElem<D> dk_fwd_chunk(const DkMats<D, L>& km, const Vec<D>& a1e,
                                                 const float (&ytil)[L], uint32_t maskbits, int tid) {
  Vec<D> c = vzero<D>();
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const bool obs = ((maskbits >> l) & 1u) == 0u;
    const Vec<D> a = (tid == 0 && l == 0) ? a1e : trans_apply(c);
    c = a;
    if (obs) {
      const float v = ytil[l] - a.v[0];
#pragma unroll
      for (int i = 0; i < D; ++i) c.v[i] = fmaf(km.kf[l].v[i], v, a.v[i]);
    }
  }
  AElem<D> e;
  e.M = km.Mf;
  e.c = c;
  return e;
}

// Local pass of the means from the filtered mean `mf` of the step before the chunk: predicted
// means a_t and v_t / F_t (0 where missing); then the chunk's adjoint map.
template <int D, int L>
__device__ __forceinline__ AElem<D> dk_fwd_local_bwd_chunk(const DkMats<D, L>& km, const Vec<D>& a1e,
                                                           const float (&ytil)[L], uint32_t maskbits,
                                                           int tid, Vec<D> mf, Vec<D> (&ap)[L],
                                                           float (&vf)[L]) {
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const bool obs = ((maskbits >> l) & 1u) == 0u;
    const Vec<D> a = (tid == 0 && l == 0) ? a1e : trans_apply(mf);
    ap[l] = a;
    mf = a;
    vf[l] = 0.f;
    if (obs) {
      const float v = ytil[l] - a.v[0];
      vf[l] = v * km.rF[l];
#pragma unroll
      for (int i = 0; i < D; ++i) mf.v[i] = fmaf(km.kf[l].v[i], v, a.v[i]);
    }
  }
  Vec<D> d = vzero<D>();
#pragma unroll
  for (int l = L - 1; l >= 0; --l) d = dk_adjoint_step<D, L>(km, l, d, vf[l]);
  AElem<D> e;
  e.M = km.Mb;
  e.c = d;
  return e;
}

// Fix-up: r of the chunk's last step in, the draw x~_t = a_t + P_t r_{t-1} + x+_t out.
template <int D, int L>
__device__ __forceinline__ void dk_bwd_fixup(const DkMats<D, L>& km, Vec<D> r, const Vec<D> (&ap)[L],
                                             const float (&vf)[L], const Vec<D> (&xp)[L],
                                             Vec<D> (&xout)[L]) {
#pragma unroll
  for (int l = L - 1; l >= 0; --l) {
    r = dk_adjoint_step<D, L>(km, l, r, vf[l]);
    const Vec<D> sm = vadd(ap[l], mv(km.Pp[l], r));
    xout[l] = vadd(sm, xp[l]);
  }
}

// The simulated path of the prior (zero initial state) from the scanned prefix, and the data of
// the filter y~ = resid - y+.
template <int D, int L>
__device__ __forceinline__ void dk_prior_path(const DkModel<D>& md, const PElem<D>& ppre,
                                              const float (&resid)[L], const float (&zl)[L],
                                              const float (&zs)[L], const float (&zo)[L],
                                              Vec<D> (&xp)[L], float (&ytil)[L]) {
  Vec<D> x = ppre.s;
  const float so = __fsqrt_rn(md.H);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    xp[l] = x;
    ytil[l] = resid[l] - fmaf(so, zo[l], x.v[0]);
    x = trans_apply(x);
    x.v[0] = fmaf(md.sig.v[0], zl[l], x.v[0]);
    if constexpr (D == 2) x.v[1] = fmaf(md.sig.v[1], zs[l], x.v[1]);
  }
}

// The prior mean of x_0 with the simulated initial state folded in (see dk_draw).
template <int D>
__device__ __forceinline__ Vec<D> dk_initial_mean(const DkModel<D>& md, const Rng& rng, uint32_t iter,
                                                  int tid, const float* zinit) {
  Vec<D> a1e = md.a1;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float zi[1];
      if (zinit) zi[0] = zinit[i];       // drawn ahead by an idle wave (same site, same index)
      else fill_normals<1>(rng, iter, SITE_PRIOR_INIT, 0, (uint32_t)i, zi);
      a1e.v[i] = fmaf(__fsqrt_rn(md.p1.v[i]), zi[0], md.a1.v[i]);
    }
  }
  return a1e;
}

// The data phase: contains two __syncthreads().  slots_f / slots_b: NW * 16 floats each.
template <int D, int L, class PF>
__device__ __forceinline__ void dk_data_phase(const DkMats<D, L>& km, const Vec<D>& a1e,
                                              const float (&ytil)[L], const Vec<D> (&xp)[L],
                                              uint32_t maskbits, int tid, int lane, int wave,
                                              float* slots_f, float* slots_b, Vec<D> (&xout)[L],
                                              PF& prof) {
  const AElem<D> fe = dk_fwd_chunk<D, L>(km, a1e, ytil, maskbits, tid);
  prof.tick(27);
  const AElem<D> fincl = aff_fwd_begin<D>(fe, slots_f, lane, wave);
  __syncthreads();
  const Vec<D> mf = aff_fwd_finish<D>(fincl, slots_f, lane, wave);
  prof.tick(5);
  Vec<D> ap[L];
  float vf[L];
  const AElem<D> be = dk_fwd_local_bwd_chunk<D, L>(km, a1e, ytil, maskbits, tid, mf, ap, vf);
  prof.tick(6);
  const AElem<D> bincl = aff_bwd_begin<D>(be, slots_b, lane, wave);
  __syncthreads();
  const Vec<D> r = aff_bwd_finish<D>(bincl, slots_b, lane, wave);
  prof.tick(28);
  dk_bwd_fixup<D, L>(km, r, ap, vf, xp, xout);
  prof.tick(7);
}

// slots: 3 regions of NW * 16 floats.  Contains 4 __syncthreads() (3 with the prior simulation
// scanned by the caller).
template <int D, int L, class PF>
__device__ __forceinline__ void dk_draw(const DkModel<D>& md, const float (&resid)[L],
                                        uint32_t maskbits, const Rng& rng, uint32_t iter, int tid,
                                        int lane, int wave, float* slots, Vec<D> (&xout)[L],
                                        PF& prof, const float (&zl)[L], const float (&zs)[L],
                                        const float (&zo)[L], const float* zinit = nullptr,
                                        const PElem<D>* ppre_in = nullptr) {
  Vec<D> q;
#pragma unroll
  for (int i = 0; i < D; ++i) q.v[i] = md.sig.v[i] * md.sig.v[i];

  // ---- (1) simulate x+ from the prior with zero initial mean: a scan of x <- T x + n_t
  // The initial draw x+_0 = chol(P_1) z is NOT propagated through the simulated path (it
  // would grow like t * slope+_0 and cancel against the smoother in float32).  By linearity
  // it is folded into the filter's prior mean instead:
  //   x~ = x+_noise + E[x | y - y+_noise ; prior mean a_1 + x+_0]
  // which is the same draw as the oracle's x+ + E[x | y - y+ ; prior mean a_1].
  const Vec<D> a1e = dk_initial_mean<D>(md, rng, iter, tid, zinit);
  PElem<D> ppre;
  if (ppre_in) {
    ppre = *ppre_in;                     // scanned by the caller (dk_prior_begin / dk_prior_finish)
  } else {
    const PElem<D> incl = dk_prior_begin<D, L>(md.sig, zl, zs, slots, lane, wave);
    __syncthreads();
    ppre = dk_prior_finish<D>(incl, slots, lane, wave);
  }
  Vec<D> xp[L];
  float ytil[L];
  dk_prior_path<D, L>(md, ppre, resid, zl, zs, zo, xp, ytil);
  prof.tick(4);

  // ---- (2) matrix phase: covariances, gains, the chunk's mean maps
  const FMElem<D> fe = dk_matrix_chunk<D, L>(md, q, maskbits, tid);
  prof.tick(24);
  const FMElem<D> mincl = dk_matrix_scan_begin<D>(fe, slots + NW * 16, lane, wave);
  __syncthreads();
  float Pin[D * (D + 1) / 2];
  dk_matrix_scan_finish<D>(mincl, slots + NW * 16, lane, wave, Pin);
  prof.tick(25);
  DkMats<D, L> km;
  dk_matrix_local<D, L>(md, q, maskbits, tid, Pin, km);
  prof.tick(26);

  // ---- (3) data phase: filtered means forward, adjoint backward, fix-up
  // (region 0 of `slots` is free again: the prior scan's totals were read before the barrier of
  //  the matrix scan)
  dk_data_phase<D, L>(km, a1e, ytil, xp, maskbits, tid, lane, wave, slots, slots + 2 * NW * 16, xout,
                      prof);
}

template <int D, int L, class PF>
__device__ __forceinline__ void dk_draw(const DkModel<D>& md, const float (&resid)[L],
                                        uint32_t maskbits, const Rng& rng, uint32_t iter, int tid,
                                        int lane, int wave, float* slots, Vec<D> (&xout)[L],
                                        PF& prof) {
  float zl[L], zs[L], zo[L];
#pragma unroll
  for (int l = 0; l < L; ++l) zs[l] = 0.f;
  dk_normals<D, L>(rng, iter, tid, zl, zs, zo);
  dk_draw<D, L>(md, resid, maskbits, rng, iter, tid, lane, wave, slots, xout, prof, zl, zs, zo);
}

// ------------------------------------------------------------------------------------
// spike-and-slab regression block (wave 0 only, float64 in LDS)
// ------------------------------------------------------------------------------------
struct RegLds {
  double* xtx;     // [P*P]
  double* omega;   // [P*P]
  double* aug[2];  // [(P+1)^2] swept augmented matrix [[M, b],[b', y'y]] ([1] unused)
  double* pri[2];  // [P*P]     swept prior precision
  double* chol;    // [P*P]
  double* bvec;    // [P+4]     reduced X~'targets, y'y, ss_level, ss_slope
  double* zv;      // [P]
  double* uperm;   // [P]
  int* nz;         // [P]
  int* perm;       // [P]
  int* idx;        // [P]
  float* w;        // [P]
};

// One symmetric sweep (or its inverse) of both matrices on pivot k, in place.  The pivot row
// (== pivot column: swept symmetric matrices stay symmetric) is first copied to `tmp`, so no
// lane reads an entry another lane is rewriting.  Entries are walked in LINEAR order, 64 lanes x 8
// per trip, with no predicates at all: the matrices are padded to a multiple of 1024 doubles
// (sweep_padded) and whatever lands in the padding is never read.  General entries get one FMA;
// the pivot row / column are rewritten afterwards (LDS operations of one wave complete in
// program order, so the later stores win).  n, np <= 64; tmp: 128 doubles.
__host__ __device__ inline size_t sweep_padded(size_t entries) { return (entries + 1023) & ~(size_t)1023; }
// LDS doubles behind an m x m matrix of the regression block: enough for the one-wavefront sweeps
// (sweep_padded) and for the workgroup-wide ones (rows padded to 16, sweep_cols_block)
__host__ __device__ inline size_t block_matrix_doubles(int m) {
  const size_t a = sweep_padded((size_t)m * m), b = (((size_t)m + 15) & ~(size_t)15) * m;
  return a > b ? a : b;
}

__device__ __forceinline__ void sweep_one(double* __restrict__ M, int m, const double* __restrict__ t,
                                          int k, double sgn, int lane) {
  const double rd = fast_rcp(t[k]);
  const int qd = 64 / m, rm = 64 - qd * m;     // (i, j) of entry e + 64 from (i, j) of entry e
  int i = 0, j = lane;
  while (j >= m) { j -= m; ++i; }
  const int trips = (m * m + 511) >> 9;
  double* Me = M + lane;
#pragma unroll 1
  for (int it = 0; it < trips; ++it) {
    double mv[8], ti[8], tj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mv[u] = Me[64 * u];
      ti[u] = t[i];
      tj[u] = t[j];
      j += rm; i += qd;
      if (j >= m) { j -= m; ++i; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) Me[64 * u] = mv[u] - (ti[u] * rd) * tj[u];
    Me += 512;
  }
  if (lane < m) {
    const double pv = (lane == k) ? -rd : sgn * t[lane] * rd;
    M[k * m + lane] = pv;
    M[lane * m + k] = pv;
  }
}

__device__ __forceinline__ void sweep_pair(double* A, int n, double* Pm, int np, int k,
                                           bool reverse, int lane, double* tmp) {
  const double sgn = reverse ? -1.0 : 1.0;
  if (lane < n) tmp[lane] = A[k * n + lane];
  if (lane < np) tmp[64 + lane] = Pm[k * np + lane];
  wave_sync();
  sweep_one(A, n, tmp, k, sgn, lane);
  sweep_one(Pm, np, tmp + 64, k, sgn, lane);
  wave_sync();
}

// ---- register-resident regression block (P <= 16) ---------------------------------------
// Quadrant layout: lane = j + 16 q holds rows 4q..4q+3 of column j of the swept augmented
// matrix (c) and of the swept prior precision (p), plus -- replicated over q -- the swept
// X~'targets entry cb_j and the two diagonals.  A sweep on the (wave-uniform) pivot k is 4
// rows of work per lane: the pivot column/row arrive through ds_bpermute / v_readlane, no LDS
// memory and no barrier.  Because every lane tracks its own diagonal, ALL flip proposals are
// evaluated at once; they are re-evaluated only after an accepted flip, which reproduces the
// sequential scan of spike_and_slab._resample_all_features decision for decision.
// (Register indices must be compile-time constants or the compiler demotes the columns to
// scratch: the row-within-quadrant index of the pivot is dispatched through a 4-way switch.)
struct QCols {
  double c[4], p[4];
  double cb, diag, pdiag, corner;
};

__device__ __forceinline__ double bperm_d(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int KR, bool WITH_P>
__device__ __forceinline__ void sweep_q_kr(QCols& m, int k, bool reverse, int lane) {
  const double sgn = reverse ? -1.0 : 1.0;
  const int kq = k >> 2, j = lane & 15, q = lane >> 4;
  const double ckr = m.c[KR], pkr = m.p[KR];
  const double rd = fast_rcp(readlane_d(ckr, k + 16 * kq));
  const double rowk = bperm_d(ckr, j + 16 * kq);      // A[k][j]
  const double cbk = readlane_d(m.cb, k);
  const bool isk = j == k;
  const double t = isk ? 1.0 : rowk * rd;
  double rdp = 0.0, prowk = 0.0, tp = 0.0;
  if constexpr (WITH_P) {
    rdp = fast_rcp(readlane_d(pkr, k + 16 * kq));
    prowk = bperm_d(pkr, j + 16 * kq);
    tp = isk ? 1.0 : prowk * rdp;
  }
  double colk[4], pcolk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {                       // all ds_bpermute in flight together
    colk[r] = bperm_d(m.c[r], k + 16 * q);            // A[4q + r][k]
    if constexpr (WITH_P) pcolk[r] = bperm_d(m.p[r], k + 16 * q);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m.c[r] = isk ? sgn * colk[r] * rd : m.c[r] - colk[r] * t;
    if constexpr (WITH_P) m.p[r] = isk ? sgn * pcolk[r] * rdp : m.p[r] - pcolk[r] * tp;
  }
  m.c[KR] = (q == kq) ? (isk ? -rd : sgn * t) : m.c[KR];       // row k of my column
  m.cb = isk ? sgn * cbk * rd : m.cb - cbk * t;
  m.diag = isk ? -rd : m.diag - rowk * rowk * rd;
  m.corner -= cbk * cbk * rd;
  if constexpr (WITH_P) {
    m.p[KR] = (q == kq) ? (isk ? -rdp : sgn * tp) : m.p[KR];
    m.pdiag = isk ? -rdp : m.pdiag - prowk * prowk * rdp;
  }
}

template <bool WITH_P>
__device__ __forceinline__ void sweep_q(QCols& m, int k, bool reverse, int lane) {
  switch (k & 3) {
    case 0: sweep_q_kr<0, WITH_P>(m, k, reverse, lane); break;
    case 1: sweep_q_kr<1, WITH_P>(m, k, reverse, lane); break;
    case 2: sweep_q_kr<2, WITH_P>(m, k, reverse, lane); break;
    default: sweep_q_kr<3, WITH_P>(m, k, reverse, lane); break;
  }
}

// Swept prior precision carried from one Gibbs iteration to the next.  Omega itself never
// changes (only its scale sigma^2_prev does, handled analytically) and the included set at the
// start of an iteration is the set at the end of the previous one, so the prior block needs a
// sweep only when a flip is accepted -- not a rebuild every iteration.
struct PriorCarry {
  double p[4], pdiag;
  unsigned long long S;
  int valid;
};

// Reverse sweep of the posterior block only (no RHS row, no prior columns), returning the
// lane's coefficient A[k][j] / A[k][k] = V_jk / V_kk.  Used by the weights draw below.
template <int KR>
__device__ __forceinline__ double unsweep_q_kr(QCols& m, int k, int lane) {
  const int kq = k >> 2, j = lane & 15, q = lane >> 4;
  const double ckr = m.c[KR];
  const double rd = fast_rcp(readlane_d(ckr, k + 16 * kq));
  const double rowk = bperm_d(ckr, j + 16 * kq);
  const bool isk = j == k;
  const double t = isk ? 1.0 : rowk * rd;
  double colk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) colk[r] = bperm_d(m.c[r], k + 16 * q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) m.c[r] = isk ? -colk[r] * rd : m.c[r] - colk[r] * t;
  m.c[KR] = (q == kq) ? (isk ? -rd : -t) : m.c[KR];
  m.diag = isk ? -rd : m.diag - rowk * rowk * rd;
  return t;
}
__device__ __forceinline__ double unsweep_q(QCols& m, int k, int lane) {
  switch (k & 3) {
    case 0: return unsweep_q_kr<0>(m, k, lane);
    case 1: return unsweep_q_kr<1>(m, k, lane);
    case 2: return unsweep_q_kr<2>(m, k, lane);
    default: return unsweep_q_kr<3>(m, k, lane);
  }
}

// Visiting order of the flip proposals (stable argsort of P uniforms: rank = step at which
// feature j is visited) and the uniform each proposal is tested against.  All 64 lanes, j = lane & 15.
__device__ __forceinline__ void spike_slab_perm(const Rng& rng, uint32_t iter, int P, int j,
                                                bool live, int& rank, double& uflip) {
  const double uperm = live ? uniform_d(rng, iter, SITE_PERM, 0, (uint32_t)j) : 2.0;
  rank = 0;
  for (int kk = 0; kk < P; ++kk) {
    const double uk = readlane_d(uperm, kk);
    rank += (uk < uperm || (uk == uperm && kk < j)) ? 1 : 0;
  }
  uflip = live ? uniform_d(rng, iter, SITE_FLIP, 0, (uint32_t)rank) : 2.0;
}
// The register-resident block's randomness for iteration `iter`, written to `pre` (LDS, 32
// doubles) by a wave that is not in the serial section.
__device__ __forceinline__ void spike_slab_randoms(const Rng& rng, uint32_t iter, int P, int lane,
                                                   double* pre) {
  const int j = lane & 15;
  const bool live = j < P;
  int rank;
  double uflip;
  spike_slab_perm(rng, iter, P, j, live, rank, uflip);
  float zf[1];
  fill_normals<1>(rng, iter, SITE_WEIGHTS, 0, (uint32_t)(live ? j : 0), zf);
  if (lane < 16) {
    pre[lane] = uflip;
    reinterpret_cast<int*>(pre + 16)[lane] = rank;
    reinterpret_cast<float*>(pre + 24)[lane] = zf[0];
  }
}

// `publish(new_scale)` is called as soon as sigma_obs is drawn, before the weights (the eight-wave
// kernel hands it to the time waves there).
struct NoPublish {
  __device__ __forceinline__ void operator()(double) const {}
};
// EXACT (the float64 kernel, ci_gibbs64.h): logarithms, exponential, square roots and the
// weights' normals in float64 (the float32 kernels take float32 hardware transcendentals where
// 1e-7 does not matter), and the weights go to `wout` (double) instead of R.w (float).
template <class PF, class Pub = NoPublish, bool EXACT = false>
__device__ __forceinline__ double spike_slab_draw_regs(const RegLds& R, int P,
                                                       const DevSeriesParams& sp,
                                                       double prev_obs_scale, double g_obs,
                                                       const Rng& rng, uint32_t iter, int lane,
                                                       PF& prof, PriorCarry& pc,
                                                       const double* pre = nullptr,
                                                       Pub publish = Pub(), double* wout = nullptr) {
  // pre (optional, LDS): this iteration's data-independent randomness, drawn one iteration
  // ahead by an idle wave (spike_slab_randoms): [0,16) flip uniforms by feature, [16,24) visiting
  // ranks (int), [24,32) weight normals (float)
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  const int j = lane & 15, q = lane >> 4;
  const bool live = j < P;
  const int col = live ? j : 0;
  const double* __restrict__ omega = R.omega;
  const double* __restrict__ xtx = R.xtx;
  QCols m;
  const bool fresh = pc.valid == 0;      // first iteration: nothing swept yet
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * q + r;
    double om = 0.0, xx = 0.0;
    if (live && i < P) {
      om = omega[i * P + col];
      xx = xtx[i * P + col];
    }
    m.p[r] = fresh ? om : pc.p[r];       // UNIT-scale prior precision (swept on pc.S)
    m.c[r] = om * prev_var + xx;
  }
  {
    const double od = live ? omega[col * P + col] : 1.0;
    m.pdiag = fresh ? od : pc.pdiag;
    m.diag = live ? od * prev_var + xtx[col * P + col] : 1.0;
  }
  m.cb = live ? R.bvec[col] : 0.0;
  m.corner = R.bvec[P];
  unsigned long long S = 0ull;
  {
    // features included at the end of the previous iteration (== weights != 0): the posterior
    // block is rebuilt (sigma^2_prev and the targets changed) and swept in; the prior block is
    // already swept on exactly this set
    unsigned long long pending = fresh ? (all_in ? __ballot(q == 0 && live) : 0ull) : pc.S;
    const bool sweep_prior_too = fresh;
    for (; pending != 0ull; pending &= pending - 1ull) {
      const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
      if (sweep_prior_too) sweep_q<true>(m, k, false, lane); else sweep_q<false>(m, k, false, lane);
      S |= 1ull << k;
    }
  }
  prof.tick(9);
  if (!all_in) {
    // visiting order = stable argsort of P uniforms (rank_j = step at which feature j is visited)
    int rank = 0;
    double uflip = 2.0;
    if (pre) {
      rank = reinterpret_cast<const int*>(pre + 16)[j];
      uflip = pre[j];
    } else {
      spike_slab_perm(rng, iter, P, j, live, rank, uflip);
    }
    const double logit_pi =
        EXACT ? log(sp.nonzero_prob) - log1p(-sp.nonzero_prob)
              : (double)(__logf((float)sp.nonzero_prob) - __logf((float)(1.0 - sp.nonzero_prob)));
    int s_cur = 0;
    const double inv_prev_var = EXACT ? 1.0 / prev_var : fast_rcp(prev_var);
    for (;;) {
      // every lane evaluates the flip of ITS feature against the current swept state
      const bool in = ((S >> j) & 1ull) != 0ull;
      const double sg = in ? -1.0 : 1.0;
      const double rap = fast_rcp(sg * m.diag);          // 1 / Schur pivot (out) or 1 / V_jj (in)
      const double beta_old = sp.obs_scale + 0.5 * m.corner;
      const double x = -0.5 * sg * m.cb * m.cb * rap * fast_rcp(beta_old);
      // unit-scale prior block: Schur pivots scale with sigma^2_prev, inverse-block entries
      // with 1 / sigma^2_prev
      const double pscale = in ? inv_prev_var : prev_var;
      bool take;
      if constexpr (EXACT) {
        const double delta = 0.5 * log(sg * m.pdiag * pscale * rap) + sg * logit_pi -
                             (a_post - 1.0) * log1p(x);
        take = uflip < 1.0 / (1.0 + exp(-delta));
      } else {
        const double delta = 0.5 * (double)__logf((float)(sg * m.pdiag * pscale * rap)) +
                             sg * logit_pi - (a_post - 1.0) * fast_log1p(x);
        const float prob = 1.0f / (1.0f + __expf(-(float)delta));
        take = uflip < (double)prob;
      }
      const bool acc = live && q == 0 && rank >= s_cur && take;
      // the earliest accepted proposal in visiting order is the one the sequential scan takes
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
    }
  }
  prof.tick(10);
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = m.p[r];
  pc.pdiag = m.pdiag;
  pc.S = S;
  pc.valid = 1;
  const double beta_post = sp.obs_scale + 0.5 * m.corner;
  double var = EXACT ? beta_post / g_obs : beta_post * fast_rcp(g_obs);
  if (var > sp.obs_ub) var = sp.obs_ub;   // InverseGammaWithSampleUpperBound clips the variance
  const double new_scale = EXACT ? sqrt(var) : (double)__fsqrt_rn((float)var);
  publish(new_scale);
  prof.tick(11);

  // weights_S ~ N(mean, var * V), V = M_S^{-1} = -(swept block).  Un-sweep the included
  // features in DESCENDING order: feature a is drawn from its current conditional
  // N(mu_a, V_aa), then the reverse sweep conditions the rest on it (V_rr -= V_ra V_ar / V_aa,
  // mu_r += V_ra / V_aa (u_a - mu_a)).  This is exactly u = L^{-T} z with M_S = L L' (the
  // oracle's Cholesky route): u_n = z_n / L_nn, u_{n-1} | u_n, ... -- but it reuses the swept
  // state instead of factorising M_S and back-substituting.
  float zf[1] = {0.f};
  double zd = 0.0;
  if constexpr (EXACT) {
    zd = normal_d(rng, iter, SITE_WEIGHTS, 0, (uint32_t)col);
  } else {
    if (pre) zf[0] = reinterpret_cast<const float*>(pre + 24)[col];
    else fill_normals<1>(rng, iter, SITE_WEIGHTS, 0, (uint32_t)col, zf);
  }
  const double mean = m.cb;
  double mu = 0.0, umine = 0.0;
  for (unsigned long long mm = S; mm != 0ull;) {
    const int aidx = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)mm));   // descending
    mm &= ~(1ull << aidx);
    const double vaa = -readlane_d(m.diag, aidx);
    const double mua = readlane_d(mu, aidx);
    const double za = EXACT ? readlane_d(zd, aidx)
                            : (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(zf[0]), aidx));
    const double ua = mua + (EXACT ? sqrt(vaa) : (double)__fsqrt_rn((float)vaa)) * za;
    const double t = unsweep_q(m, aidx, lane);
    if (j == aidx) umine = ua; else mu += t * (ua - mua);
  }
  if constexpr (EXACT) {
    if (q == 0 && live) wout[j] = ((S >> j) & 1ull) ? mean + new_scale * umine : 0.0;
  } else {
    if (q == 0 && live) R.w[j] = ((S >> j) & 1ull) ? (float)(mean + new_scale * umine) : 0.f;
  }
  wave_sync();
  prof.tick(12);
  return new_scale;
}

// Draws (sigma^2_obs, weights) for iteration `iter`.  Executed by all 64 lanes of wave 0
// with uniform control flow.  Returns the new observation-noise scale.
// spike_and_slab.SpikeSlabSampler.sample_noise_variance_and_weights with
// experimental_use_weight_adjustment=True (causalimpact_lib.py:387-388).
template <class PF>
__device__ __forceinline__ double spike_slab_draw(const RegLds& R, int P,
                                                  const DevSeriesParams& sp, double prev_obs_scale,
                                                  double g_obs, const Rng& rng, uint32_t iter,
                                                  int lane, PF& prof, bool first) {
  const int n = P + 1;
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  constexpr int cur = 0;
  double* tmp = R.chol;            // free until the final Cholesky (P > 16 => P*P >= 128)
  {
    // augmented matrix [[Omega s + X'X, X'r], [r'X, r'r]] and the scaled prior precision, linear
    // entry order, branch-free (clamped loads + selects; stores past the end land in the padding)
    const int qd = 64 / n, rm = 64 - qd * n;
    int i = 0, j = lane;
    while (j >= n) { j -= n; ++i; }
    for (int e = lane; e < n * n; e += 64) {
      const int ic = i < P ? i : P - 1, jc = j < P ? j : P - 1;
      const double inner = R.omega[ic * P + jc] * prev_var + R.xtx[ic * P + jc];
      const double edge = R.bvec[(i == P && j == P) ? P : (i < j ? i : j)];
      R.aug[0][e] = (i < P && j < P) ? inner : edge;
      j += rm; i += qd;
      if (j >= n) { j -= n; ++i; }
    }
    // The prior precision is kept swept on the CURRENT model at unit scale, from iteration to
    // iteration (the model at the start of an iteration is the one the previous iteration ended
    // with): sweeping s * Omega gives s * (unswept block), (swept block) / s, so only the logs
    // below see the scale.  It is built once.
    if (first)
      for (int e = lane; e < P * P; e += 64) R.pri[0][e] = R.omega[e];
  }
  int nz0 = 0;
  if (lane < P) {
    nz0 = all_in ? 1 : (R.w[lane] != 0.f ? 1 : 0);
    R.nz[lane] = nz0;
    if (!all_in) R.uperm[lane] = uniform_d(rng, iter, SITE_PERM, 0, (uint32_t)lane);
  }
  wave_sync();
  // sweep in the currently included features
  for (unsigned long long todo = __ballot(nz0 != 0); todo; todo &= todo - 1ull) {
    const int k = __ffsll((long long)todo) - 1;
    if (first) {
      sweep_pair(R.aug[0], n, R.pri[0], P, k, false, lane, tmp);
    } else {
      if (lane < n) tmp[lane] = R.aug[0][k * n + lane];
      wave_sync();
      sweep_one(R.aug[0], n, tmp, k, 1.0, lane);
      wave_sync();
    }
  }
  prof.tick(9);
  if (!all_in) {
    // visiting order = stable argsort of P uniforms
    if (lane < P) {
      const double uj = R.uperm[lane];
      int rank = 0;
      for (int k = 0; k < P; ++k) {
        const double uk = R.uperm[k];
        rank += (uk < uj || (uk == uj && k < lane)) ? 1 : 0;
      }
      R.perm[rank] = lane;
    }
    wave_sync();
    // Lane s evaluates the proposal at visiting position s on the CURRENT model; the first
    // position (>= the scan position) that flips is applied and the later ones re-evaluated.
    // That is spike_and_slab._resample_all_features decision for decision, with as many
    // evaluation rounds as accepted flips + 1 instead of P.
    const double logit_pi = log(sp.nonzero_prob) - log1p(-sp.nonzero_prob);
    const int myj = lane < P ? R.perm[lane] : 0;
    const double myu = lane < P ? uniform_d(rng, iter, SITE_FLIP, 0, (uint32_t)lane) : 2.0;
    int s_cur = 0;
    while (true) {
      bool flip = false;
      if (lane < P && lane >= s_cur) {
        const double* A = R.aug[0];
        const bool in = R.nz[myj] != 0;
        const double ajj = A[myj * n + myj], ajb = A[myj * n + P], corner = A[P * n + P];
        const double pju = R.pri[0][myj * P + myj];      // unit scale
        const double beta_old = sp.obs_scale + 0.5 * corner;
        double delta;
        if (!in) {
          const double pjj = pju * prev_var;
          const double beta_new = sp.obs_scale + 0.5 * (corner - ajb * ajb / ajj);
          delta = 0.5 * log(pjj) - 0.5 * log(ajj) + logit_pi -
                  (a_post - 1.0) * (log(beta_new) - log(beta_old));
        } else {
          const double V = -ajj, Vp = -pju / prev_var;
          const double beta_new = sp.obs_scale + 0.5 * (corner + ajb * ajb / V);
          delta = 0.5 * log(Vp) - 0.5 * log(V) - logit_pi -
                  (a_post - 1.0) * (log(beta_new) - log(beta_old));
        }
        flip = myu < 1.0 / (1.0 + exp(-delta));
      }
      const unsigned long long bal = __ballot(flip);
      if (bal == 0ull) break;
      const int s_star = __ffsll((long long)bal) - 1;
      const int j = R.perm[s_star];
      const bool in = R.nz[j] != 0;
      wave_sync();
      sweep_pair(R.aug[0], n, R.pri[0], P, j, in, lane, tmp);
      if (lane == 0) R.nz[j] = in ? 0 : 1;
      wave_sync();
      s_cur = s_star + 1;
    }
  }
  prof.tick(10);
  const double* A = R.aug[cur];
  const double beta_post = sp.obs_scale + 0.5 * A[P * n + P];
  double var = beta_post / g_obs;
  if (var > sp.obs_ub) var = sp.obs_ub;   // InverseGammaWithSampleUpperBound clips the variance
  const double new_scale = sqrt(var);

  // active set in increasing feature order
  const int mynz = (lane < P) ? R.nz[lane] : 0;
  const unsigned long long bal = __ballot(mynz != 0);
  const int na = __popcll(bal);
  if (mynz) R.idx[__popcll(bal & ((1ull << lane) - 1ull))] = lane;
  if (lane < P) R.w[lane] = 0.f;
  wave_sync();
  prof.tick(11);
  // Cholesky of M_S = Omega_S * prev_var + XtX_S  (right-looking, in LDS)
  for (int i = lane >> 4; i < na; i += 4)
    for (int j = lane & 15; j < na; j += 16) {
      const int fi = R.idx[i], fj = R.idx[j];
      R.chol[i * na + j] = R.omega[fi * P + fj] * prev_var + R.xtx[fi * P + fj];
    }
  if (lane < na) R.zv[lane] = normal_d(rng, iter, SITE_WEIGHTS, 0, (uint32_t)R.idx[lane]);
  wave_sync();
  for (int k = 0; k < na; ++k) {
    const double dkk = sqrt(R.chol[k * na + k]);
    wave_sync();
    for (int i = k + lane; i < na; i += 64) R.chol[i * na + k] = (i == k) ? dkk : R.chol[i * na + k] / dkk;
    wave_sync();
    for (int i = k + 1 + (lane >> 4); i < na; i += 4)
      for (int j = k + 1 + (lane & 15); j <= i; j += 16)
        R.chol[i * na + j] -= R.chol[i * na + k] * R.chol[j * na + k];
    wave_sync();
  }
  // solve L' u = z (column-oriented back substitution); u overwrites zv
  for (int i = na - 1; i >= 0; --i) {
    const double ui = R.zv[i] / R.chol[i * na + i];
    wave_sync();
    if (lane == 0) R.zv[i] = ui;
    for (int k = lane; k < i; k += 64) R.zv[k] -= R.chol[i * na + k] * ui;
    wave_sync();
  }
  if (lane < na) {
    const int f = R.idx[lane];
    R.w[f] = (float)(A[f * n + P] + new_scale * R.zv[lane]);
  }
  wave_sync();
  prof.tick(12);
  return new_scale;
}

// ------------------------------------------------------------------------------------
// Any number of design columns (MAXP < P <= MAXP_BIG): the same draw, one wavefront, with every
// O(P^2) array in a per-chain HBM workspace instead of LDS (a chain only reads what it wrote, and
// a workgroup's accesses go through its own L1: program order + wave_sync suffice) and every
// per-feature step looped over the lanes.  The reference has no cap on the number of covariates
// (causalimpact_lib.py:445-453); this is the capability route, not a fast one: a sweep costs
// (P+1)^2 / 64 L2 round trips per lane.  R.xtx / R.omega point at the setup kernel's output,
// R.aug[0] [(P+1)^2], R.pri[0] [P^2], R.chol [P^2 + 2 (P + 1)], R.zv / R.uperm [P], R.nz / R.perm /
// R.idx [P] at the workspace; R.bvec [P + 4] and R.w [P] stay in LDS.
// ------------------------------------------------------------------------------------
constexpr int MAXP_BIG = 512;

__device__ __forceinline__ void sweep_big(double* M, int m, const double* t, int k, double sgn,
                                          int lane) {
  const double rd = 1.0 / t[k];
  for (int e = lane; e < m * m; e += 64) {
    const int i = e / m, j = e - i * m;
    M[e] -= (t[i] * rd) * t[j];
  }
  wave_sync();
  for (int j = lane; j < m; j += 64) {
    const double pv = (j == k) ? -rd : sgn * t[j] * rd;
    M[k * m + j] = pv;
    M[j * m + k] = pv;
  }
  wave_sync();
}

// workspace doubles / ints the big-P regression block needs per chain
__host__ __device__ inline size_t bigp_workspace_bytes(int P) {
  const size_t n = (size_t)P + 1;
  const size_t dbl = n * n + (size_t)P * P + (size_t)P * P + 2 * n + 2 * (size_t)P;
  return (dbl * sizeof(double) + 3 * (size_t)P * sizeof(int) + 255) & ~(size_t)255;
}
__device__ __forceinline__ void bigp_point(RegLds& R, unsigned char* ws, int P) {
  const size_t n = (size_t)P + 1;
  double* d = reinterpret_cast<double*>(ws);
  R.aug[0] = d; R.aug[1] = d; d += n * n;
  R.pri[0] = d; R.pri[1] = d; d += (size_t)P * P;
  R.chol = d; d += (size_t)P * P + 2 * n;        // Cholesky block, then the two saved pivot rows
  R.zv = d; d += P;
  R.uperm = d; d += P;
  int* q = reinterpret_cast<int*>(d);
  R.nz = q; R.perm = q + P; R.idx = q + 2 * P;
}

// `w`: the weights vector (float in the float32 kernels, double in the float64 one, ci_gibbs64.h).
template <class WT>
__device__ __noinline__ double spike_slab_draw_big(const RegLds& R, WT* w, int P,
                                                   const DevSeriesParams& sp, double prev_obs_scale,
                                                   double g_obs, const Rng& rng, uint32_t iter,
                                                   int lane, bool first) {
  const int n = P + 1;
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  double* A = R.aug[0];
  double* Pm = R.pri[0];
  double* ta = R.chol + (size_t)P * P;      // saved pivot row of A   [n]
  double* tp = ta + n;                      // saved pivot row of Pm  [n]
  for (int e = lane; e < n * n; e += 64) {
    const int i = e / n, j = e - i * n;
    double v;
    if (i < P && j < P) v = R.omega[i * P + j] * prev_var + R.xtx[i * P + j];
    else v = R.bvec[(i == P && j == P) ? P : (i < j ? i : j)];
    A[e] = v;
  }
  // the prior precision is kept swept on the current model at unit scale from iteration to
  // iteration (see spike_slab_draw)
  if (first)
    for (int e = lane; e < P * P; e += 64) Pm[e] = R.omega[e];
  for (int j = lane; j < P; j += 64) {
    R.nz[j] = all_in ? 1 : (w[j] != (WT)0 ? 1 : 0);
    if (!all_in) R.uperm[j] = uniform_d(rng, iter, SITE_PERM, 0, (uint32_t)j);
  }
  wave_sync();
  auto sweep_both = [&](int k, bool reverse, bool with_prior) {
    const double sgn = reverse ? -1.0 : 1.0;
    for (int j = lane; j < n; j += 64) ta[j] = A[k * n + j];
    if (with_prior)
      for (int j = lane; j < P; j += 64) tp[j] = Pm[k * P + j];
    wave_sync();
    sweep_big(A, n, ta, k, sgn, lane);
    if (with_prior) sweep_big(Pm, P, tp, k, sgn, lane);
  };
  // sweep in the currently included features (the prior matrix only when it is being built)
  for (int k = 0; k < P; ++k)
    if (R.nz[k]) sweep_both(k, false, first);
  if (!all_in) {
    // visiting order = stable argsort of P uniforms
    for (int j = lane; j < P; j += 64) {
      const double uj = R.uperm[j];
      int rank = 0;
      for (int k = 0; k < P; ++k) {
        const double uk = R.uperm[k];
        rank += (uk < uj || (uk == uj && k < j)) ? 1 : 0;
      }
      R.perm[rank] = j;
    }
    wave_sync();
    const double logit_pi = log(sp.nonzero_prob) - log1p(-sp.nonzero_prob);
    // lanes evaluate the proposals of 64 consecutive visiting positions on the CURRENT model; the
    // first position (>= the scan position) that flips is applied and the later ones re-evaluated
    int s_cur = 0;
    while (s_cur < P) {
      const int base = s_cur & ~63;
      const int pos = base + lane;
      bool flip = false;
      if (pos < P && pos >= s_cur) {
        const int j = R.perm[pos];
        const bool in = R.nz[j] != 0;
        const double ajj = A[j * n + j], ajb = A[j * n + P], corner = A[P * n + P];
        const double pju = Pm[j * P + j];
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
      const int j = R.perm[s_star];
      const bool in = R.nz[j] != 0;
      wave_sync();
      sweep_both(j, in, true);
      if (lane == 0) R.nz[j] = in ? 0 : 1;
      wave_sync();
      s_cur = s_star + 1;
    }
  }
  const double beta_post = sp.obs_scale + 0.5 * A[P * n + P];
  double var = beta_post / g_obs;
  if (var > sp.obs_ub) var = sp.obs_ub;   // the variance is clipped at upper_bound (see spike_slab_draw)
  const double new_scale = sqrt(var);
  // active set in increasing feature order
  int na = 0;
  for (int j0 = 0; j0 < P; j0 += 64) {
    const int j = j0 + lane;
    const int mynz = j < P ? R.nz[j] : 0;
    const unsigned long long bal = __ballot(mynz != 0);
    if (mynz) R.idx[na + __popcll(bal & ((1ull << lane) - 1ull))] = j;
    na += __popcll(bal);
  }
  for (int j = lane; j < P; j += 64) w[j] = (WT)0;
  wave_sync();
  // Cholesky of M_S = Omega_S * prev_var + XtX_S (right-looking), z ~ N(0, I), L' u = z
  for (int i = lane >> 4; i < na; i += 4)
    for (int j = lane & 15; j < na; j += 16) {
      const int fi = R.idx[i], fj = R.idx[j];
      R.chol[i * na + j] = R.omega[fi * P + fj] * prev_var + R.xtx[fi * P + fj];
    }
  for (int i = lane; i < na; i += 64) R.zv[i] = normal_d(rng, iter, SITE_WEIGHTS, 0, (uint32_t)R.idx[i]);
  wave_sync();
  for (int k = 0; k < na; ++k) {
    const double dkk = sqrt(R.chol[k * na + k]);
    wave_sync();
    for (int i = k + lane; i < na; i += 64) R.chol[i * na + k] = (i == k) ? dkk : R.chol[i * na + k] / dkk;
    wave_sync();
    for (int i = k + 1 + (lane >> 4); i < na; i += 4)
      for (int j = k + 1 + (lane & 15); j <= i; j += 16)
        R.chol[i * na + j] -= R.chol[i * na + k] * R.chol[j * na + k];
    wave_sync();
  }
  for (int i = na - 1; i >= 0; --i) {
    const double ui = R.zv[i] / R.chol[i * na + i];
    wave_sync();
    if (lane == 0) R.zv[i] = ui;
    for (int k = lane; k < i; k += 64) R.zv[k] -= R.chol[i * na + k] * ui;
    wave_sync();
  }
  for (int i = lane; i < na; i += 64) {
    const int f = R.idx[i];
    w[f] = (WT)(A[f * n + P] + new_scale * R.zv[i]);
  }
  wave_sync();
  return new_scale;
}

// ------------------------------------------------------------------------------------
// The same regression draw executed by the WHOLE workgroup (P > 16 in the time-parallel kernel,
// where the serial section would otherwise idle three waves for ~0.3M cycles at P = 51).
// Control flow is uniform without any broadcast: every wave evaluates the (cheap) proposals and
// ballots on the same LDS state, so all waves take the same decisions; the (expensive) sweeps
// spread their entries over all 256 threads.  Barriers are __syncthreads(); every thread returns
// the same new observation-noise scale.
// ------------------------------------------------------------------------------------
// Sweep of the symmetric m x m matrix M (LDS, row-major) on pivot k by the whole workgroup, ONE
// barrier per sweep.  Thread (wave w, lane) owns column `lane` and the rows w, w + NW, ...: its
// column's pivot-row entry stays in a register, a row's entry is a broadcast read, and the matrix
// itself is read and written once, lanes on consecutive doubles (conflict-free), eight rows in
// flight per thread.  Who writes what:
//   * general entries (i != k, j != k): the owning thread, before the barrier;
//   * column k (i != k): wave (k + 1) % NW with lane = row, before the barrier -- nobody reads
//     column k during a sweep (the pivot row is read as ROW k);
//   * row k: its owning wave AFTER the barrier (everyone has read it by then), which is
//     sweep_rowk_finish -- the caller runs it before the next sweep's reads of this thread's rows
//     (same threads own row k in every sweep: program order suffices).
// `keep` (wave k % NW, may be null) receives the pivot row as it was before the sweep.
// Arithmetic: the expressions of sweep_one / sweep_one_block, entry for entry.
struct RowK { double pv; int k; bool mine; };
// rows of the LDS allocation behind an m x m matrix that the block sweeps may touch: the row loop
// runs in unguarded groups of NW * 4 rows (whatever lands in the padding rows is never read)
__host__ __device__ constexpr size_t block_rows_padded(int m) { return ((size_t)m + 15) & ~(size_t)15; }
__device__ __forceinline__ RowK sweep_cols_block(double* __restrict__ M, int m, int k, double sgn,
                                                 int lane, int w, double* keep) {
  RowK out; out.pv = 0.0; out.k = k; out.mine = false;
  if (lane >= m) return out;
  const double* t = M + k * m;                    // the pivot row, intact until the barrier
  const double rd = fast_rcp(t[k]);
  const double tj = t[lane];
  out.pv = (lane == k) ? -rd : sgn * tj * rd;
  out.mine = (k & (NW - 1)) == w;
  if (out.mine && keep) keep[lane] = tj;
  if (w == ((k + 1) & (NW - 1)) && lane != k) M[lane * m + k] = sgn * tj * rd;   // column k, row `lane`
  if (lane != k) {
    constexpr int R = 4;
    double* Mc = M + w * m + lane;                // row w, this lane's column; rows advance by NW
    const double* tw = t + w;
    const int step = NW * m;
    const int kslot = out.mine ? (k >> 2) : -1;   // row k = w + NW * kslot of the owning wave
    for (int s0 = 0; NW * s0 + w < m; s0 += R) {
      double mv[R], ti[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        mv[u] = Mc[u * step];
        ti[u] = tw[NW * u];
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (s0 + u != kslot) Mc[u * step] = mv[u] - (ti[u] * rd) * tj;     // (never the pivot row)
      Mc += R * step;
      tw += R * NW;
    }
  }
  return out;
}
__device__ __forceinline__ void sweep_rowk_finish(double* M, int m, const RowK& rk, int lane,
                                                  int skip_col = -1) {
  if (rk.mine && lane < m && lane != skip_col) M[(size_t)rk.k * m + lane] = rk.pv;
}
// Sweeps A (n x n) and, `with_prior`, Pm (np x np) on pivot k; keepA [n] (may be null) receives
// A's pivot row as it was before the sweep: ascending sweeps of a set S record the Cholesky factor
// of the S-block that way (spike_slab_draw_block).  Ends with the matrices complete and a barrier.
__device__ __forceinline__ void sweep_pair_block(double* A, int n, double* Pm, int np, int k,
                                                 bool reverse, bool with_prior, int tid,
                                                 double* keepA) {
  const double sgn = reverse ? -1.0 : 1.0;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RowK ra = sweep_cols_block(A, n, k, sgn, lane, w, keepA);
  RowK rp; rp.mine = false; rp.k = k; rp.pv = 0.0;
  if (with_prior) rp = sweep_cols_block(Pm, np, k, sgn, lane, w, nullptr);
  __syncthreads();
  sweep_rowk_finish(A, n, ra, lane);
  if (with_prior) sweep_rowk_finish(Pm, np, rp, lane);
  __syncthreads();
}
// A run of sweeps of A (n x n, LDS) alone on the features of `todo` in ascending order -- the
// sweep-in of every iteration -- with the matrix held in REGISTERS for the duration: thread
// (a, b) = (tid / NB, tid % NB), NB = ceil(n / 4), owns the 4 x 4 tile of rows 4a.. and columns
// 4b..  A sweep is then: the owners of pivot row k publish it (as it is before the sweep) to row r
// of `rec`, ONE barrier, every thread reads its four column entries, its four row entries and the
// pivot from that row (3 LDS reads, one round trip) and updates its tile with 16 FMAs.  The LDS
// form of the same sweep (sweep_cols_block) moves the whole matrix through LDS per pivot and is
// bound by that; this one by the publish -> barrier -> read -> reciprocal chain (~0.7k cycles).
// Entry for entry the arithmetic of sweep_one: M_ij - (t_i / t_k) t_j, pivot row and column
// t / t_k, pivot -1 / t_k.  rec: [popcount(todo)][REC_LD], the pivot rows -- which are the
// Cholesky factor of the swept block (spike_slab_draw_block).  Ends with A complete and a barrier.
constexpr int REC_LD = 64;      // doubles per recorded pivot row: >= MAXP + 1, a multiple of 16
// one sweep of the register-resident tiles on pivot k = 4 ka + KR (KR static: the pivot's row and
// column inside a tile are compile-time register indices)
// WAVE: the tiles fit one wavefront (n <= 32) and that wavefront runs the sweeps alone: the
// barrier becomes a wave-level fence (LDS operations of one wave complete in order).
template <int KR, bool WAVE>
__device__ __forceinline__ void sweep_tile_step(double (&m)[4][4], int ka, int a, int b, bool live,
                                                double* st) {
  if (live && a == ka) {
    *reinterpret_cast<double2*>(st + 4 * b) = make_double2(m[KR][0], m[KR][1]);
    *reinterpret_cast<double2*>(st + 4 * b + 2) = make_double2(m[KR][2], m[KR][3]);
  }
  if constexpr (WAVE) wave_sync(); else __syncthreads();
  if (!live) return;
  const double2 c01 = *reinterpret_cast<const double2*>(st + 4 * b);
  const double2 c23 = *reinterpret_cast<const double2*>(st + 4 * b + 2);
  const double2 r01 = *reinterpret_cast<const double2*>(st + 4 * a);
  const double2 r23 = *reinterpret_cast<const double2*>(st + 4 * a + 2);
  const double tk = st[4 * ka + KR];
  const double tj[4] = {c01.x, c01.y, c23.x, c23.y};
  const double ti[4] = {r01.x, r01.y, r23.x, r23.y};
  const double rd = fast_rcp(tk);
  double q[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) q[x] = ti[x] * rd;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) m[x][y] = m[x][y] - q[x] * tj[y];
  const bool colk = b == ka, rowk = a == ka;
#pragma unroll
  for (int x = 0; x < 4; ++x) m[x][KR] = colk ? q[x] : m[x][KR];            // column k: t_i / t_k
#pragma unroll
  for (int y = 0; y < 4; ++y) {                                               // row k: t_j / t_k, pivot
    double pv = tj[y] * rd;
    if (y == KR) pv = colk ? -rd : pv;
    m[KR][y] = rowk ? pv : m[KR][y];
  }
}
// `init(i, j)`: the matrix entry before the run (read from A, or built on the spot).
template <bool WAVE = false, class Init>
__device__ __forceinline__ void sweep_run_block(double* A, int n, unsigned long long todo, int tid,
                                                double* rec, Init init) {
  const int NB = (n + 3) >> 2;
  const int a = tid / NB, b = tid - a * NB;
  const bool live = a < NB;
  double m[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int i = 4 * a + x, j = 4 * b + y;
      m[x][y] = (live && i < n && j < n) ? init(i, j) : 0.0;
    }
  double* st = rec;
  for (int ka = 0; ka < NB; ++ka) {
    const unsigned four = (unsigned)(todo >> (4 * ka)) & 15u;
    if (four == 0u) continue;
    if (four & 1u) { sweep_tile_step<0, WAVE>(m, ka, a, b, live, st); st += REC_LD; }
    if (four & 2u) { sweep_tile_step<1, WAVE>(m, ka, a, b, live, st); st += REC_LD; }
    if (four & 4u) { sweep_tile_step<2, WAVE>(m, ka, a, b, live, st); st += REC_LD; }
    if (four & 8u) { sweep_tile_step<3, WAVE>(m, ka, a, b, live, st); st += REC_LD; }
  }
  if (live) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int i = 4 * a + x, j = 4 * b + y;
        if (i < n && j < n) A[i * n + j] = m[x][y];
      }
  }
  if constexpr (WAVE) wave_sync(); else __syncthreads();
}

// LDS doubles behind R.chol: the recorded pivot rows of the sweep-in ([P][REC_LD]) -- or the explicit
// Cholesky factor ([P][P]).
// (+ 128: the pivot-row staging of the one-wave sweeps that carry the prior block)
__host__ __device__ constexpr size_t block_chol_doubles(int P) { return (size_t)P * REC_LD + 128; }

// The data-independent part of the regression block's matrix: [Omega s2 + X'X, 0; 0, 0] swept on
// the features of `nzmask` (the border stays zero).  What is left for the iteration itself is to
// fill the border from X~'targets (spike_slab_draw_block, split mode) -- so this half can be
// computed ahead, by another workgroup of the chain's cluster, as soon as s2 and the active set
// of the previous iteration are known (ci_wide.h).  Ends with a barrier.
__device__ __forceinline__ void presweep_block(const RegLds& R, int P, double prev_var,
                                               unsigned long long nzmask, bool with_prior, int tid) {
  const int n = P + 1;
  {
    int i = tid / n, j = tid - (tid / n) * n;
    const int qd = NT / n, rm = NT - qd * n;
    for (int e = tid; e < n * n; e += NT) {
      const int ic = i < P ? i : P - 1, jc = j < P ? j : P - 1;
      const double inner = R.omega[ic * P + jc] * prev_var + R.xtx[ic * P + jc];
      R.aug[0][e] = (i < P && j < P) ? inner : 0.0;
      j += rm; i += qd;
      if (j >= n) { j -= n; ++i; }
    }
  }
  __syncthreads();
  if (with_prior) {
    int r = 0;
    for (unsigned long long todo = nzmask; todo; todo &= todo - 1ull, ++r)
      sweep_pair_block(R.aug[0], n, R.pri[0], P, __ffsll((long long)todo) - 1, false, true, tid,
                       R.chol + (size_t)r * REC_LD);
  } else {
    const double* A = R.aug[0];
    sweep_run_block<false>(R.aug[0], n, nzmask, tid, R.chol, [&](int i, int j) { return A[i * n + j]; });
  }
}

// What another workgroup needs to take presweep_block's result over: the matrix and the recorded
// pivot rows, (P + 1)^2 + popcount(nzmask) * REC_LD doubles (presweep_doubles(P) at most).
__host__ __device__ constexpr size_t presweep_doubles(int P) {
  return (size_t)(P + 1) * (P + 1) + (size_t)P * REC_LD;
}
__device__ __forceinline__ void presweep_export(const RegLds& R, int P, unsigned long long nzmask,
                                                double* dst, int tid) {
  const int n = P + 1, nrec = __popcll(nzmask) * REC_LD;
  for (int e = tid; e < n * n; e += NT) dst[e] = R.aug[0][e];
  for (int e = tid; e < nrec; e += NT) dst[n * n + e] = R.chol[e];
}

// The matrix another workgroup swept ahead (presweep_export: (P + 1)^2 doubles, then the pivot rows of
// its sweeps) into this workgroup's LDS.  No barrier.
template <int NTH>
__device__ __forceinline__ void presweep_import(const RegLds& R, int P, unsigned long long nzmask,
                                                const double* presweep, int tid) {
  const int n = P + 1;
  double* rec = R.chol;
  const int nrec = __popcll(nzmask) * REC_LD;
  if (((n * n) & 1) == 0 && (REC_LD & 1) == 0) {
    // 16 bytes per lane, four loads in flight (the copy is 36 KB from L2 at P = 51: element by
    // element it was a chain of ~18 dependent round trips)
    const double2* s2 = reinterpret_cast<const double2*>(presweep);
    double2* a2 = reinterpret_cast<double2*>(R.aug[0]);
    double2* r2 = reinterpret_cast<double2*>(rec);
    const int na2 = (n * n) >> 1, nt2 = na2 + (nrec >> 1);
    for (int e0 = tid; e0 < nt2; e0 += 4 * NTH) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTH;
        v[u] = s2[e < nt2 ? e : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTH;
        if (e < na2) a2[e] = v[u];
        else if (e < nt2) r2[e - na2] = v[u];
      }
    }
  } else {
    for (int e = tid; e < n * n; e += NTH) R.aug[0][e] = presweep[e];
    for (int e = tid; e < nrec; e += NTH) rec[e] = presweep[n * n + e];
  }
}

// split: the matrix comes from presweep_block -- computed here, or copied from `presweep`
// ((P + 1)^2 doubles in global memory) when another workgroup prepared it -- and the border is
// filled by a matrix-vector product instead of being carried through the sweeps (the same
// arithmetic whoever swept).  !split: the sweeps run on the bordered matrix (ci_kernels.h kernels).
// The data-independent randomness of the workgroup-wide block's draw for iteration `iter`: the
// visiting order (stable argsort of P uniforms, ranked in registers on their raw 32-bit words --
// u01d is strictly increasing in them -- and inverted by one ds_permute), the flip uniform of every
// visiting position, the weight normal of every feature.  block_randoms_store / _load: 128 doubles
// of LDS ([0,64) uniforms by position, then 64 ints: feature by position, then 64 floats: normals
// by feature) when another wave draws them one iteration ahead.
constexpr int GAM_BLOCK_PRE = 8 + 64 + 4;     // offset (doubles) of the two buffers in the gamma area
constexpr int BLOCK_PRE_DOUBLES = 128;        // one buffer: 64 uniforms, 64 ints, 64 floats
struct BlockRandoms {
  int myj;        // feature visited at step `lane`
  double myu;     // its flip uniform
  float zf;       // the weight normal of feature `lane`
};
__device__ __forceinline__ BlockRandoms block_randoms(const Rng& rng, uint32_t iter, int P, int lane) {
  BlockRandoms b;
  uint32_t rw = 0xffffffffu;
  if (lane < P) {
    const U4 r4 = site_call(rng, iter, SITE_PERM, 0, (uint32_t)lane >> 2);
    const uint32_t c = (uint32_t)lane & 3u;
    rw = c == 0 ? r4.x : c == 1 ? r4.y : c == 2 ? r4.z : r4.w;
  }
  int rank = 0;
  for (int k = 0; k < P; ++k) {
    const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rw, k);
    rank += (rk < rw || (rk == rw && k < lane)) ? 1 : 0;
  }
  if (lane >= P) rank = lane;
  b.myj = __builtin_amdgcn_ds_permute(rank << 2, lane);
  b.myu = lane < P ? uniform_d(rng, iter, SITE_FLIP, 0, (uint32_t)lane) : 2.0;
  float zf[1];
  fill_normals<1>(rng, iter, SITE_WEIGHTS, 0, (uint32_t)lane, zf);
  b.zf = zf[0];
  return b;
}
__device__ __forceinline__ void block_randoms_store(const BlockRandoms& b, double* pre, int lane) {
  pre[lane] = b.myu;
  reinterpret_cast<int*>(pre + 64)[lane] = b.myj;
  reinterpret_cast<float*>(pre + 96)[lane] = b.zf;
}
__device__ __forceinline__ BlockRandoms block_randoms_load(const double* pre, int P, int lane) {
  BlockRandoms b;
  b.myu = lane < P ? pre[lane] : 2.0;
  b.myj = lane < P ? reinterpret_cast<const int*>(pre + 64)[lane] : lane;
  b.zf = reinterpret_cast<const float*>(pre + 96)[lane];
  return b;
}

// WAVE (P + 1 <= 32, !split): ONE wavefront runs the whole draw (`tid` = its lane): the tiles of the
// sweep-in fit its 64 lanes, every barrier is a wave-level fence, the rare sweeps that carry the
// prior block and the rare explicit Cholesky use 64-thread strides -- so the rest of the workgroup
// can do something else meanwhile (gibbs_kernel: emission and the next draw's normals).
template <bool WAVE = false>
__device__ __forceinline__ double spike_slab_draw_block(const RegLds& R, int P,
                                                        const DevSeriesParams& sp,
                                                        double prev_obs_scale, double g_obs,
                                                        const Rng& rng, uint32_t iter, int tid,
                                                        bool first, Prof* prof = nullptr,
                                                        bool split = false,
                                                        const double* presweep = nullptr,
                                                        int slot0 = 4, const double* pre = nullptr,
                                                        bool imported = false) {
  // pre (LDS, may be null): this iteration's block_randoms, drawn ahead by another wave
  // imported: the caller has already copied `presweep` into LDS (presweep_import)
  constexpr int NTH = WAVE ? 64 : NT;          // threads taking part
  auto sync = [] { if constexpr (WAVE) wave_sync(); else __syncthreads(); };
  // a sweep of A and the prior block on pivot k (iteration 0 and accepted flips)
  auto sweep_both = [&](int k, bool reverse, double* keep) {
    if constexpr (WAVE) {
      if (keep && tid < P + 1) keep[tid] = R.aug[0][k * (P + 1) + tid];
      sweep_pair(R.aug[0], P + 1, R.pri[0], P, k, reverse, tid, R.chol + (size_t)P * REC_LD);
    } else {
      sweep_pair_block(R.aug[0], P + 1, R.pri[0], P, k, reverse, true, tid, keep);
    }
  };
  const int lane = tid & 63;
  const int n = P + 1;
  const double prev_var = prev_obs_scale * prev_obs_scale;
  const double a_post = sp.obs_conc + 0.5 * sp.n_obs;
  const bool all_in = sp.nonzero_prob >= 1.0;
  // R.chol: the pivot rows of the sweep-in, one per swept feature in ascending order
  // ([na][REC_LD]); behind them (at [P][REC_LD]) the staging rows of the one-wave sweeps
  double* rec = R.chol;
  bool clean = true;               // rec holds the factor of the final model's block
  if (split) {
    if (first)
      for (int e = tid; e < P * P; e += NTH) R.pri[0][e] = R.omega[e];
  } else if (first) {
    int i = tid / n, j = tid - (tid / n) * n;
    const int qd = NTH / n, rm = NTH - qd * n;
    for (int e = tid; e < n * n; e += NTH) {
      const int ic = i < P ? i : P - 1, jc = j < P ? j : P - 1;
      const double inner = R.omega[ic * P + jc] * prev_var + R.xtx[ic * P + jc];
      const double edge = R.bvec[(i == P && j == P) ? P : (i < j ? i : j)];
      R.aug[0][e] = (i < P && j < P) ? inner : edge;
      j += rm; i += qd;
      if (j >= n) { j -= n; ++i; }
    }
    if (first)
      for (int e = tid; e < P * P; e += NTH) R.pri[0][e] = R.omega[e];
  }
  // per-feature state is computed redundantly by every wave (lane = feature), written once
  int nz0 = 0;
  if (lane < P) nz0 = all_in ? 1 : (R.w[lane] != 0.f ? 1 : 0);
  if (tid < P) R.nz[tid] = nz0;
  if (split || first) sync();
  if (prof) prof->tick(slot0);
  const unsigned long long nzmask = __ballot(nz0 != 0);
  if (!split) {
    if (first) {
      int r = 0;
      for (unsigned long long todo = nzmask; todo; todo &= todo - 1ull, ++r)
        sweep_both(__ffsll((long long)todo) - 1, false, rec + (size_t)r * REC_LD);
    } else {
      // [[Omega s2 + X'X, X'r], [r'X, r'r]] built straight into the sweeps' registers
      const double* om = R.omega; const double* xx = R.xtx; const double* bv = R.bvec;
      sweep_run_block<WAVE>(R.aug[0], n, nzmask, tid, rec, [&](int i, int j) {
        const int ic = i < P ? i : P - 1, jc = j < P ? j : P - 1;
        const double inner = om[ic * P + jc] * prev_var + xx[ic * P + jc];
        const double edge = bv[(i == P && j == P) ? P : (i < j ? i : j)];
        return (i < P && j < P) ? inner : edge;
      });
    }
  } else {
    if (presweep) {     // the swept matrix, then the pivot rows of its sweeps (presweep_export)
      if (!imported) presweep_import<NTH>(R, P, nzmask, presweep, tid);
      __syncthreads();
    } else {
      presweep_block(R, P, prev_var, nzmask, first, tid);
    }
    // border: with V the swept matrix, S the swept set and b = X~'targets,
    //   b~_j = (j in S ? 0 : b_j) - sum_{k in S} V_jk b_k ,   corner = y'y - sum_{k in S} b_k b~_k
    if (tid < 64) {
      double* A = R.aug[0];
      const bool in = ((nzmask >> lane) & 1ull) != 0ull;
      double bt = 0.0;
      if (lane < P) {
        double acc = 0.0;
        for (unsigned long long todo = nzmask; todo; todo &= todo - 1ull) {
          const int k = __ffsll((long long)todo) - 1;
          acc += A[lane * n + k] * R.bvec[k];
        }
        bt = (in ? 0.0 : R.bvec[lane]) - acc;
        A[lane * n + P] = bt;
        A[P * n + lane] = bt;
      }
      double term = (lane < P && in) ? R.bvec[lane] * bt : 0.0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) term += __shfl_xor(term, off, 64);
      if (lane == 0) A[P * n + P] = R.bvec[P] - term;
    }
    __syncthreads();
  }
  if (prof) prof->tick(slot0 + 1);
  if (!all_in) {
    // visiting order, flip uniforms (block_randoms: every wave for itself, or read from `pre`)
    const BlockRandoms br = pre ? block_randoms_load(pre, P, lane) : block_randoms(rng, iter, P, lane);
    const int myj = br.myj;
    const double myu = br.myu;
    const double logit_pi =
        (double)(__logf((float)sp.nonzero_prob) - __logf((float)(1.0 - sp.nonzero_prob)));
    const double inv_prev_var = fast_rcp(prev_var);
    int s_cur = 0;
    while (true) {
      bool flip = false;
      if (lane < P && lane >= s_cur) {
        // the same expression as the register-resident block's (spike_slab_draw_regs)
        const double* A = R.aug[0];
        const bool in = R.nz[myj] != 0;
        const double ajj = A[myj * n + myj], ajb = A[myj * n + P], corner = A[P * n + P];
        const double pju = R.pri[0][myj * P + myj];      // unit scale
        const double sg = in ? -1.0 : 1.0;
        const double rap = fast_rcp(sg * ajj);           // 1 / Schur pivot (out) or 1 / V_jj (in)
        const double beta_old = sp.obs_scale + 0.5 * corner;
        const double x = -0.5 * sg * ajb * ajb * rap * fast_rcp(beta_old);
        const double pscale = in ? inv_prev_var : prev_var;
        const double delta = 0.5 * (double)__logf((float)(sg * pju * pscale * rap)) +
                             sg * logit_pi - (a_post - 1.0) * fast_log1p(x);
        const float prob = 1.0f / (1.0f + __expf(-(float)delta));
        flip = myu < (double)prob;
      }
      const unsigned long long bal = __ballot(flip);     // identical in every wave
      if (prof && prof->p) prof->p[bal == 0ull ? 30 : 31] += 1;   // evaluation rounds / accepted flips
      if (bal == 0ull) break;
      const int s_star = __ffsll((long long)bal) - 1;
      const int j = __builtin_amdgcn_readlane(myj, s_star);
      const bool in = R.nz[j] != 0;
      sync();                                            // everyone has read nz / the matrices
      sweep_both(j, in, nullptr);
      if (tid == 0) R.nz[j] = in ? 0 : 1;
      sync();
      clean = false;
      s_cur = s_star + 1;
    }
  }
  if (prof) prof->tick(slot0 + 2);
  const double* A = R.aug[0];
  const double beta_post = sp.obs_scale + 0.5 * A[P * n + P];
  double var = beta_post * fast_rcp(g_obs);
  if (var > sp.obs_ub) var = sp.obs_ub;   // InverseGammaWithSampleUpperBound clips the variance
  const double new_scale = (double)__fsqrt_rn((float)var);

  const int mynz = (lane < P) ? R.nz[lane] : 0;
  const unsigned long long bal = __ballot(mynz != 0);   // the final model (every wave computes it)
  const int na = __popcll(bal);
  if (clean) {
    // No flip was accepted: the model is the one swept in at the top, in ascending order -- and the
    // pivot rows of those sweeps ARE the Cholesky factor of M_S = Omega_S s2 + X'X_S: with pivot
    // r on feature f_r, rec[r][f_j] = L_jr L_rr for j > r and rec[r][f_r] = L_rr^2.  Solve
    // L' u = z from them by column-oriented back substitution, lane = feature: lane f holds z_f
    // and row rank(f) of rec, sixteen columns of it at a time.
    if (tid < 64) {
      const bool in_s = ((bal >> lane) & 1ull) != 0ull;
      const double* row = rec + (size_t)__popcll(bal & ((1ull << lane) - 1ull)) * REC_LD;
      const double rs = in_s ? fast_rsqrt(row[lane]) : 0.0;          // 1 / L_ff
      float zf[1];
      if (pre) zf[0] = reinterpret_cast<const float*>(pre + 96)[lane];
      else fill_normals<1>(rng, iter, SITE_WEIGHTS, 0, (uint32_t)lane, zf);
      double z = in_s ? (double)zf[0] : 0.0;
      double u = 0.0;
      for (int g0 = (P - 1) & ~15; g0 >= 0; g0 -= 16) {
        double lr[16];
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
          const double2 v = *reinterpret_cast<const double2*>(row + g0 + c);
          lr[c] = in_s ? v.x : 0.0; lr[c + 1] = in_s ? v.y : 0.0;
        }
#pragma unroll
        for (int c = 15; c >= 0; --c) {
          const int g = g0 + c;
          if (!((bal >> g) & 1ull)) continue;
          const double ug = readlane_d(z, g) * readlane_d(rs, g);
          if (lane == g) u = ug;
          if (lane < g) z -= (lr[c] * rs) * ug;          // L[g][f] = rec[rank f][g] / L_ff
        }
      }
      if (lane < P) R.w[lane] = in_s ? (float)(A[lane * n + P] + new_scale * u) : 0.f;
    }
  } else {
    sync();                                  // chol and w are about to be rewritten
    if (tid < 64) {                          // active set in increasing feature order
      if (mynz) R.idx[__popcll(bal & ((1ull << lane) - 1ull))] = lane;
      if (lane < P) R.w[lane] = 0.f;
    }
    sync();
    // M_S = Omega_S * prev_var + XtX_S, then a right-looking Cholesky by the whole workgroup, one
    // barrier per column: the Schur-complement entries (i, j), k < j <= i, take their k-th term
    // (the same products, in the same order, as a left-looking factorisation); column k of L goes
    // to the UPPER triangle (row k) and its diagonal to ldiag, so that nothing a concurrent thread
    // still reads is overwritten.  Reciprocal square roots instead of divisions.
    double* ldiag = R.uperm;                 // (the permutation keys are no longer needed)
    for (int e = tid; e < na * na; e += NTH) {
      const int i = e / na, j = e - i * na;
      const int fi = R.idx[i], fj = R.idx[j];
      R.chol[e] = R.omega[fi * P + fj] * prev_var + R.xtx[fi * P + fj];
    }
    if (tid < na) R.zv[tid] = normal_d(rng, iter, SITE_WEIGHTS, 0, (uint32_t)R.idx[tid]);
    sync();
    for (int k = 0; k < na; ++k) {
      const double skk = R.chol[k * na + k];
      const double rs = fast_rsqrt(skk);
      for (int i = k + 1 + (tid >> 4); i < na; i += NTH / 16) {
        const double lik = R.chol[i * na + k] * rs;
        for (int j = k + 1 + (tid & 15); j <= i; j += 16)
          R.chol[i * na + j] -= lik * (R.chol[j * na + k] * rs);
        if ((tid & 15) == 0) R.chol[k * na + i] = lik;
      }
      if (tid == 0) ldiag[k] = skk * rs;
      sync();
    }
    if (tid < 64) {
      // solve L' u = z (column-oriented back substitution) in registers: lane l holds z_l
      double z = lane < na ? R.zv[lane] : 0.0;
      const double dl = lane < na ? ldiag[lane] : 1.0;
      double u = 0.0;
      double lnext = (na > 0 && lane < na - 1) ? R.chol[lane * na + (na - 1)] : 0.0;
      for (int i = na - 1; i >= 0; --i) {
        const double li = lnext;                          // L[i][lane] (stored at row lane, column i)
        if (i > 0) lnext = lane < i - 1 ? R.chol[lane * na + (i - 1)] : 0.0;
        const double ui = readlane_d(z, i) * fast_rcp(readlane_d(dl, i));
        if (lane == i) u = ui;
        if (lane < i) z -= li * ui;
      }
      if (lane < na) {
        const int f = R.idx[lane];
        R.w[f] = (float)(A[f * n + P] + new_scale * u);
      }
    }
  }
  sync();
  if (prof) prof->tick(slot0 + 3);
  return new_scale;
}

// gibbs_sampler._resample_scale: sqrt(IG(conc + n/2, scale + ss/2)) clipped at the bound.
// (the gamma variate needs the prior and the number of terms only: callers with idle time draw it ahead)
__device__ __forceinline__ double scale_from_gamma(double scale, double ub, double ss, double g) {
  const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
  return s < ub ? s : ub;
}
__device__ __forceinline__ double scale_draw(double conc, double scale, double ub, double n,
                                             double ss, const Rng& rng, uint32_t iter,
                                             uint32_t site, int lane) {
  return scale_from_gamma(scale, ub, ss, gamma_wave(conc + 0.5 * n, rng, iter, site, 0, lane));
}

// ------------------------------------------------------------------------------------
// the persistent Gibbs kernel
// ------------------------------------------------------------------------------------
enum Scal { SC_OBS_DK = 0, SC_OBS_EMIT = 1, SC_LEVEL = 2, SC_SLOPE = 3 };

// Everything the serial (wave 0) section needs, resident in LDS for the whole fit.
struct SerialCtx {
  DevSeriesParams sp;
  double obs_scale, level_scale, slope_scale;   // chain state (scalars)
  RegLds R;
  float* scal;
  const float* red;
  float* out_obs;
  float* out_level_scale;
  float* out_slope_scale;
  float* out_weights;
  size_t chain_lin;
  Rng rng;
  int P, T, D, W, S, n_iter;
  long long* prof;
};

struct LdsLayout {
  size_t off_ctx, off_xtx, off_omega, off_aug0, off_aug1, off_pri0, off_pri1, off_chol, off_bvec,
      off_zv, off_uperm, off_nz, off_perm, off_idx, off_w, off_scal, off_red, off_slots, off_xlast,
      off_tg, off_gam, off_nz0, off_tgv, off_park, off_x, total;
};

__host__ __device__ inline LdsLayout make_layout(int P, int D, int tpad, int x_in_lds) {
  LdsLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const int Pp = P > 0 ? P : 1;
  l.off_ctx = take(sizeof(SerialCtx));
  l.off_xtx = take(sizeof(double) * Pp * Pp);
  l.off_omega = take(sizeof(double) * Pp * Pp);
  l.off_aug0 = take(sizeof(double) * block_matrix_doubles(Pp + 1));
  l.off_aug1 = take(16);      // (sweeps are in place: no second buffer)
  l.off_pri0 = take(sizeof(double) * block_matrix_doubles(Pp));
  l.off_pri1 = take(16);
  l.off_chol = take(sizeof(double) * block_chol_doubles(Pp));
  l.off_bvec = take(sizeof(double) * (Pp + 4));
  l.off_zv = take(sizeof(double) * Pp);
  l.off_uperm = take(sizeof(double) * Pp);
  l.off_nz = take(sizeof(int) * Pp);
  l.off_perm = take(sizeof(int) * Pp);
  l.off_idx = take(sizeof(int) * Pp);
  l.off_w = take(sizeof(float) * (Pp > 16 ? Pp : 16));
  l.off_scal = take(sizeof(float) * 16);
  l.off_red = take(sizeof(float) * NW * ((Pp > 16 ? Pp : 16) + 4));
  l.off_slots = take(sizeof(float) * 3 * NW * 16);
  l.off_xlast = take(sizeof(float) * NT * D);
  l.off_tg = take(sizeof(float) * 16);
  l.off_gam = take(sizeof(double) * (8 + 64 + 4 + 2 * BLOCK_PRE_DOUBLES));   // gamma draws (wave 1) and regression-block randomness
                                                 // (wave 2) handed to the serial wave, double-buffered
  l.off_nz0 = take(sizeof(float) * (4 * 64 * (tpad / NT) + 4));   // wave 0's normals, drawn by waves 1-3
  // targets y - level over time, handed to the waves that sum X~'targets (P <= 16)
  l.off_tgv = take((P > 0 && P <= 16) ? sizeof(float) * (size_t)tpad : 16);
  // eight and more steps per thread (T > 1024): X w and the observations wait in LDS between their
  // uses instead of in registers through every phase of the iteration (round 5, as in ci_kernels8.h)
  l.off_park = take(tpad / NT >= 8 ? 2 * sizeof(float) * (size_t)tpad : 16);
  l.off_x = take(x_in_lds ? sizeof(float) * (size_t)Pp * tpad : 16);
  l.total = o;
  return l;
}

// Serial section of iteration `it` (wave 0, all 64 lanes, uniform control flow):
//   * reduce the per-wave partial sums,
//   * draw the scales of iteration it-1 from the path drawn in it-1 and emit its scalars,
//   * draw (sigma^2_obs, weights) of iteration it.
// PM (regression mode) prunes dead code per instantiation -- the whole per-iteration path must
// stay inside the 64 KB instruction cache: 0 = no regression, 1 = P <= 16 with X in LDS
// (register-resident block), 2 = general (LDS block, X possibly streamed from L2).
// The gamma variates the serial section of iteration `it` consumes (level / slope scale of
// iteration it-1, and sigma^2_obs of iteration it or the observation scale of it-1).  They depend
// on nothing but the shape parameters, so an otherwise idle wave draws them one iteration ahead
// (double-buffered in LDS) while wave 0 is in the serial section: 2.6k cycles off the critical
// path of every Gibbs iteration.  out[0..2] = (g_level, g_slope, g_obs).
template <int PM>
static __device__ __forceinline__ void serial_gammas(const SerialCtx* cx, int it, int lane,
                                                     double* out) {
  const int P = (PM == 0) ? 0 : cx->P, T = cx->T;
  const bool have_prev = it > 0;
  const bool want_obsvar = P > 0 && it < cx->n_iter;
  const uint32_t pit = (uint32_t)(it > 0 ? it - 1 : 0);
  const GammaReq rq_level{cx->sp.level_conc + 0.5 * (double)(T - 1), pit, SITE_LEVEL_SCALE, 0};
  const GammaReq rq_slope{cx->sp.slope_conc + 0.5 * (double)(T - 1), pit, SITE_SLOPE_SCALE, 0};
  const GammaReq rq_obs = want_obsvar
      ? GammaReq{cx->sp.obs_conc + 0.5 * cx->sp.n_obs, (uint32_t)it, SITE_OBSVAR, 0}
      : GammaReq{cx->sp.obs_conc + 0.5 * cx->sp.n_obs, pit, SITE_OBS_SCALE, 0};
  const unsigned active = (have_prev ? 1u : 0u) | ((have_prev && cx->D == 2) ? 2u : 0u) |
                          ((want_obsvar || (have_prev && P == 0)) ? 4u : 0u);
  double g_level = 1.0, g_slope = 1.0, g_obs = 1.0;
  if (active) gamma_wave3(rq_level, rq_slope, rq_obs, active, g_level, g_slope, g_obs, cx->rng, lane);
  if (lane == 0) { out[0] = g_level; out[1] = g_slope; out[2] = g_obs; }
}

template <int PM, class PF, bool NR = false>
static __device__ __forceinline__ void serial_section(SerialCtx* cx, const RegLds& R,
                                                      const float* red, float* scal, int it,
                                                      int lane, PriorCarry& pc,
                                                      const double* gam, const double* pre,
                                                      double* block_st) {
  // (R, red, scal are passed in rather than read from cx: loaded from the LDS context they
  //  would be generic pointers and every access a flat_* instruction)
  const int P = (PM == 0) ? 0 : cx->P, T = cx->T;
  PF prof;
  prof.start(cx->prof, cx->prof != nullptr && lane == 0);
  if constexpr (NR) {
    // xt_sums_wave left the finished sums (one float each); the increments come per time wave
    if (lane < P + 1) R.bvec[lane] = (double)red[lane < P ? lane : RED_YTY];
    if (lane >= 62) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += (double)red[RED_INC + 2 * w + (lane - 62)];
      R.bvec[P + 1 + (lane - 62)] = s;
    }
  } else {
    const int RS = (P > 16 ? P : 16) + 4;
    for (int j = lane; j < P + 3; j += 64) {
      const int src = j < P ? j : RS - 4 + (j - P);
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += (double)red[w * RS + src];
      R.bvec[j] = s;
    }
  }
  wave_sync();
  double obs_scale = cx->obs_scale, level_scale = cx->level_scale, slope_scale = cx->slope_scale;
  double emit_obs = obs_scale;
  // the gamma draws of this section were made by wave 1 during the previous iteration
  // (serial_gammas): level / slope scale of iteration it-1, sigma^2_obs of iteration it
  prof.tick(17);
  const double g_level = gam[0], g_slope = gam[1], g_obs = gam[2];
  prof.tick(18);
  auto clipped_scale = [](double scale, double ss, double g, double ub) {
    const double s = (double)__fsqrt_rn((float)((scale + 0.5 * ss) * fast_rcp(g)));
    return s < ub ? s : ub;
  };
  if (it > 0) {
    level_scale = clipped_scale(cx->sp.level_scale, R.bvec[P + 1], g_level, cx->sp.level_ub);
    if (cx->D == 2)
      slope_scale = clipped_scale(cx->sp.slope_scale, R.bvec[P + 2], g_slope, cx->sp.slope_ub);
    if (P == 0)
      obs_scale = clipped_scale(cx->sp.obs_scale, R.bvec[P], g_obs, cx->sp.obs_ub);
    emit_obs = obs_scale;
    const int s = it - 1 - cx->W;
    if (s >= 0) {
      const size_t o = cx->chain_lin * cx->S + s;
      if (lane == 0) {
        if (cx->out_obs) cx->out_obs[o] = (float)obs_scale;
        if (cx->out_level_scale) cx->out_level_scale[o] = (float)level_scale;
        if (cx->out_slope_scale) cx->out_slope_scale[o] = (float)(cx->D == 2 ? slope_scale : 0.0);
      }
      if (cx->out_weights && lane < P) cx->out_weights[o * P + lane] = R.w[lane];
    }
  }
  prof.tick(8);
  if (P > 0 && it < cx->n_iter) {
    if constexpr (PM == 1)
      obs_scale = spike_slab_draw_regs(R, P, cx->sp, obs_scale, g_obs, cx->rng, (uint32_t)it, lane, prof, pc,
                                       pre);
    else if constexpr (PM == 2) {
      if (P > 16 && P + 1 <= 32) {
        // 17-31 columns: the tiles of the sweep-in fit this wavefront, which draws alone while the
        // other waves emit and generate normals
        Prof* pp = nullptr;                      // (instrumented build: the draw's four phases, slots 9-12)
        if constexpr (std::is_same<PF, Prof>::value) pp = &prof;
        obs_scale = spike_slab_draw_block<true>(R, P, cx->sp, obs_scale, g_obs, cx->rng, (uint32_t)it, lane,
                                                it == 0, pp, false, nullptr, 9,
                                                block_st + 4 + BLOCK_PRE_DOUBLES * (it & 1));
      } else if (P > 16) {     // drawn by the whole workgroup right after this section
        if (lane == 0) { block_st[0] = obs_scale; block_st[1] = g_obs; }
      } else {
        obs_scale = spike_slab_draw(R, P, cx->sp, obs_scale, g_obs, cx->rng, (uint32_t)it, lane, prof, it == 0);
      }
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

template <int D, int L, int PM, bool PROF = false>
#ifndef CI_MIN_WAVES
#define CI_MIN_WAVES 2
#endif
// (PM = 2 -- more than 16 columns -- holds > 80 KB of LDS: one workgroup per CU whatever the
//  registers, so it is compiled for one wave per SIMD and takes its spills in the AGPR half)
__global__ __launch_bounds__(NT, PM == 2 ? 1 : CI_MIN_WAVES) void gibbs_kernel(KArgs a) {
  // PM = 3: the register-resident regression block (PM = 1) with the design STREAMED from L2 --
  // its own instantiation, so that the loops of the LDS-resident build stay what they were (one
  // function holding both cost the 512-series batch 7 %)
  constexpr int RPM = (PM == 3) ? 1 : PM;
  constexpr bool STREAM = PM == 3;
  constexpr bool NEWRED = RPM == 1;       // xt_sums_wave's layout of `red` (shared with the eight-wave kernel)
  using PF = typename std::conditional<PROF, Prof, NoProf>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int series = blockIdx.x / a.C, chain = blockIdx.x % a.C;
  const int T = a.T, P = (RPM == 0) ? 0 : a.P;
  constexpr int TPAD = NT * L;
  const LdsLayout lay = make_layout(P, D, TPAD, a.x_in_lds);
  SerialCtx* cx = (SerialCtx*)(smem + lay.off_ctx);
  float* scal = (float*)(smem + lay.off_scal);
  float* red = (float*)(smem + lay.off_red);
  float* slots = (float*)(smem + lay.off_slots);
  float* xlast = (float*)(smem + lay.off_xlast);
  float* wls = (float*)(smem + lay.off_w);
  const float* Xs = (const float*)(smem + lay.off_x);
  float* tgv = (float*)(smem + lay.off_tgv);
  constexpr bool PARK = L >= 8;
  float* xwb = (float*)(smem + lay.off_park);        // (PARK builds only)
  float* ybuf = xwb + TPAD;
  const size_t chain_lin = (size_t)series * a.C + chain;
  const int RS = (P > 16 ? P : 16) + 4;   // stride of the per-wave partial-sum rows

  Rng rng;
  rng.k0 = stream_key0(a.seed0, a.series_stream_base, series);
  rng.k1 = stream_key1(a.seed1, a.series_stream_base, series);
  rng.chain = (uint32_t)(a.chain_offset + chain);

  // ---- stage the constants of this series
  const float* yg = a.y + (size_t)series * T;
  const uint8_t* mg = a.mask + (size_t)series * T;
  const float* Xg = a.Xt + (size_t)series * P * T;
  // rows of the streamed design can be read as float4: T % 4 == 0 keeps every row 16-byte aligned
  const bool xwide = L % 4 == 0 && (T & 3) == 0 && (reinterpret_cast<uintptr_t>(Xg) & 15) == 0;
  const int t0 = tid * L;
  RegLds R;
  {
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
    cx->prof = (blockIdx.x == 0) ? a.prof : nullptr;
    scal[8] = (float)cx->sp.init_level_loc;
    scal[9] = (float)(cx->sp.init_level_scale * cx->sp.init_level_scale);
    scal[10] = (float)(cx->sp.init_slope_scale * cx->sp.init_slope_scale);
  }
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
  if constexpr (PARK) store_targets<L>(ybuf, t0, yv);
  {
    double* lx = (double*)(smem + lay.off_xtx);
    double* lo = (double*)(smem + lay.off_omega);
    for (int e = tid; e < P * P; e += NT) {
      lx[e] = a.xtx[(size_t)series * P * P + e];
      lo[e] = a.omega[(size_t)series * P * P + e];
    }
    if (a.x_in_lds) {
      float* xw_ = (float*)(smem + lay.off_x);
      for (int j = 0; j < P; ++j)
        for (int t = tid; t < TPAD; t += NT) xw_[j * TPAD + t] = (t < T) ? Xg[(size_t)j * T + t] : 0.f;
    }
    if (tid < P) wls[tid] = 0.f;                   // weights = 0            :575-578
  }
  __syncthreads();
  double* gam = (double*)(smem + lay.off_gam);
  float* nz0 = (float*)(smem + lay.off_nz0);
  if (wave == 1) serial_gammas<RPM>(cx, 0, lane, gam);     // (no draw is active at it = 0 but P > 0's)
  if constexpr (RPM == 1) {
    if (wave == 2) spike_slab_randoms(rng, 0u, P, lane, gam + 8);
  }
  if constexpr (RPM == 2) {     // 17-31 columns: the one-wave draw's randomness comes from wave 2
    if (wave == 2 && P > 16) block_randoms_store(block_randoms(rng, 0u, P, lane), gam + GAM_BLOCK_PRE, lane);
  }
  const float init_loc = scal[8], init_var = scal[9], init_svar = scal[10];

  // chain state
  float lev[L], slp[L], xw[L], pm_acc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { lev[l] = 0.f; slp[l] = 0.f; xw[l] = 0.f; pm_acc[l] = 0.f; }  // :580-581

  float* o_level = a.out_level ? a.out_level + chain_lin * a.S * T : nullptr;
  float* o_slope = a.out_slope ? a.out_slope + chain_lin * a.S * T : nullptr;
  float* o_traj = a.out_traj ? a.out_traj + chain_lin * a.S * T : nullptr;

  const int n_iter = a.W + a.S;
  PriorCarry pc;
  pc.valid = 0;
  pc.S = 0ull;
  pc.pdiag = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;
  PF prof;
  prof.start(a.prof, a.prof != nullptr && blockIdx.x == 0 && tid == 0);
  for (int it = 0; it <= n_iter; ++it) {
    // ---- emit iteration it-1: level / slope / posterior-predictive trajectory
    auto emit = [&](float so, const float* zp_lds) {
      const int s = it - 1 - a.W;
      float zp[L];
      if (zp_lds) {
#pragma unroll
        for (int l = 0; l < L; ++l) zp[l] = zp_lds[l * 64 + lane];
      } else {
        fill_normals<L>(rng, (uint32_t)(it - 1), SITE_PRED, 0, (uint32_t)t0, zp);
      }
      float tr[L];
      if constexpr (PARK) {
        // X w of draw it-1 from LDS; the predictor's running sum in place in its output array (the
        // same sequence of float additions as the register copy: same bits)
        float xwp[L], acc[L];
        lds_row_load<L>(xwb + t0, xwp);
        float* pmo = a.out_pred_mean ? a.out_pred_mean + chain_lin * T : nullptr;
#pragma unroll
        for (int l = 0; l < L; ++l) acc[l] = (pmo != nullptr && s > 0 && t0 + l < T) ? pmo[t0 + l] : 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float loc = lev[l] + xwp[l];
          acc[l] += loc;
          tr[l] = fmaf(so, zp[l], loc);
        }
        if (pmo != nullptr) {
#pragma unroll
          for (int l = 0; l < L; ++l)
            if (t0 + l < T) pmo[t0 + l] = acc[l];
        }
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float loc = lev[l] + xw[l];
          pm_acc[l] += loc;
          tr[l] = fmaf(so, zp[l], loc);
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
    };
    // ---- partial sums over the owned steps (targets use the CURRENT level)
    {
      float tg[L], yq[L];
      float yty = 0.f;
      if constexpr (PARK) {
        lds_row_load<L>(ybuf + t0, yq);
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l) yq[l] = yv[l];
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const bool obs = ((maskbits >> l) & 1u) == 0u;
        tg[l] = obs ? (yq[l] - lev[l]) : 0.f;
        yty = fmaf(tg[l], tg[l], yty);
      }
      prof.tick(15);
      if constexpr (RPM == 1) {
        // register-resident regression block: the targets go to the shared vector; after (B1) every
        // wave sums four features over the WHOLE series (xt_sums_wave: no cross-wave reduction, no
        // gather) -- from the LDS copy of the design, or (PM = 3, long series) from L2
        store_targets<L>(tgv, t0, tg);
        prof.tick(16);
      } else if constexpr (RPM == 2) {
        // 16 features per round: their rows are independent loads (one L2 round trip per round
        // when X streams from L2, instead of one per feature; the kernel is compiled for one wave
        // per SIMD, the rows in flight live in its register half) and their wave sums ONE
        // reduce-scatter.  The row source is chosen outside the loop (see global_row_load_wide).
        auto xt_rounds = [&](auto load_row) {
          if constexpr (L <= 4) {
            // two half-rounds of 8 rows, software-pipelined: the rows of the NEXT round's first half
            // are requested before this round's second half is consumed, so that an L2 round trip
            // runs under the FMAs and the reduce-scatter (32 registers live across the reduce)
            float xa[8][L], xb[8][L];
#pragma unroll
            for (int u = 0; u < 8; ++u) load_row(u < P ? u : P - 1, xa[u]);
            for (int j0 = 0; j0 < P; j0 += 16) {
              float pj[16];
#pragma unroll
              for (int u = 0; u < 8; ++u) load_row(j0 + 8 + u < P ? j0 + 8 + u : P - 1, xb[u]);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                float sv = 0.f;
#pragma unroll
                for (int l = 0; l < L; ++l) sv = fmaf(xa[u][l], tg[l], sv);
                pj[u] = sv;
              }
              if (j0 + 16 < P) {
#pragma unroll
                for (int u = 0; u < 8; ++u) load_row(j0 + 16 + u < P ? j0 + 16 + u : P - 1, xa[u]);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                float sv = 0.f;
#pragma unroll
                for (int l = 0; l < L; ++l) sv = fmaf(xb[u][l], tg[l], sv);
                pj[8 + u] = sv;
              }
              const float tot = wave_reduce_scatter16(pj, lane);
              if (lane < 16 && j0 + lane < P) red[wave * RS + j0 + lane] = tot;
            }
          } else {
          for (int j0 = 0; j0 < P; j0 += 16) {
            float pj[16];
            constexpr int XB = L <= 4 ? 16 : (L == 8 ? 8 : 4);   // rows in flight: at most 64 registers
#pragma unroll
            for (int h = 0; h < 16 / XB; ++h) {
              float xr[XB][L];
#pragma unroll
              for (int u = 0; u < XB; ++u) load_row(j0 + XB * h + u < P ? j0 + XB * h + u : P - 1, xr[u]);
#pragma unroll
              for (int u = 0; u < XB; ++u) {
                float sv = 0.f;
#pragma unroll
                for (int l = 0; l < L; ++l) sv = fmaf(xr[u][l], tg[l], sv);
                pj[XB * h + u] = sv;
              }
            }
            const float tot = wave_reduce_scatter16(pj, lane);
            if (lane < 16 && j0 + lane < P) red[wave * RS + j0 + lane] = tot;
          }
          }
        };
        if (a.x_in_lds) {
          xt_rounds([&](int j, float (&xr)[L]) { lds_row_load<L>(Xs + j * TPAD + t0, xr); });
        } else if (xwide) {
          if constexpr (L % 4 == 0)
            xt_rounds([&](int j, float (&xr)[L]) { global_row_load_wide<L>(Xg + (size_t)j * T, t0, T, xr); });
        } else {
          xt_rounds([&](int j, float (&xr)[L]) { global_row_load_scalar<L>(Xg + (size_t)j * T, t0, T, xr); });
        }
      }
      prof.tick(13);
      // level / slope increments ending at the owned steps need the previous thread's last state
      xlast[tid * D] = lev[L - 1];
      if constexpr (D == 2) xlast[tid * D + 1] = slp[L - 1];
      __syncthreads();
      if constexpr (NEWRED) {
        if constexpr (STREAM && L >= 8) {
          // (two features at a time: the loads of four streamed rows of L float4 do not fit beside the
          //  rest -- same sums, each feature is summed on its own)
          xt_sums_wave<L, 2, STREAM, true>(tgv, Xg, TPAD, P, 4 * wave, false, red, lane, T, xwide);
          __builtin_amdgcn_sched_barrier(0);
          xt_sums_wave<L, 2, STREAM, true>(tgv, Xg, TPAD, P, 4 * wave + 2, wave == NW - 1, red, lane, T, xwide);
        } else {
          xt_sums_wave<L, 4, STREAM>(tgv, STREAM ? Xg : Xs, TPAD, P, 4 * wave, wave == NW - 1, red, lane, T, xwide);
        }
      }
      float ssl = 0.f, sss = 0.f;
      float pl = (tid > 0) ? xlast[(tid - 1) * D] : 0.f;
      float ps = 0.f;
      if constexpr (D == 2) ps = (tid > 0) ? xlast[(tid - 1) * D + 1] : 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int t = t0 + l;
        if (t >= 1 && t < T) {
          float dl = lev[l] - pl;
          if constexpr (D == 2) {
            dl -= ps;
            const float ds = slp[l] - ps;
            sss = fmaf(ds, ds, sss);
          }
          ssl = fmaf(dl, dl, ssl);
        }
        pl = lev[l];
        if constexpr (D == 2) ps = slp[l];
      }
      prof.tick(14);
      if constexpr (NEWRED) {
        const float s1 = wave_prefix_dpp(ssl), s2 = wave_prefix_dpp(sss);
        if (lane == 63) {
          red[RED_INC + 2 * wave] = s1;
          red[RED_INC + 2 * wave + 1] = s2;
        }
      } else {
        const float s0 = wave_prefix_dpp(yty), s1 = wave_prefix_dpp(ssl), s2 = wave_prefix_dpp(sss);
        if (lane == 63) {
          red[wave * RS + RS - 4] = s0;
          red[wave * RS + RS - 3] = s1;
          red[wave * RS + RS - 2] = s2;
        }
      }
    }
    // sigma_obs of iteration it-1's regression draw: the noise scale of its predictive trajectory
    // when there is a regression (read before wave 0 rewrites it in the serial section)
    const float so_prev = scal[SC_OBS_DK];
    __syncthreads();
    prof.tick(0);

    // ---- serial section: scale draws for iteration it-1, regression draw for iteration it
    // (wave 0).  Meanwhile waves 1-3 do everything that does not depend on its results: next
    // iteration's gamma variates, the emission of iteration it-1, their own Durbin-Koopman
    // normals and -- one disturbance type each -- wave 0's.
    float zl[L], zs[L], zo[L];
#pragma unroll
    for (int l = 0; l < L; ++l) zs[l] = 0.f;
    if (wave == 0) {
      serial_section<RPM, PF, NEWRED>(cx, R, red, scal, it, lane, pc, gam + 4 * (it & 1),
                                      gam + 8 + 32 * (it & 1), gam + 72);
    } else {
      if (wave == 1 && it < n_iter) serial_gammas<RPM>(cx, it + 1, lane, gam + 4 * ((it + 1) & 1));
      if constexpr (RPM == 1) {
        if (wave == 2 && it + 1 < n_iter)
          spike_slab_randoms(rng, (uint32_t)(it + 1), P, lane, gam + 8 + 32 * ((it + 1) & 1));
      }
      if constexpr (RPM == 2) {
        if (wave == 2 && it + 1 < n_iter && P > 16)
          block_randoms_store(block_randoms(rng, (uint32_t)(it + 1), P, lane),
                              gam + GAM_BLOCK_PRE + BLOCK_PRE_DOUBLES * ((it + 1) & 1), lane);
      }
      if constexpr (RPM != 0) {
        if (it > a.W) {
          emit(so_prev, nullptr);
          if (wave == 3) {      // wave 0's predictive normals of iteration it-1
            float zp0[L];
            fill_normals<L>(rng, (uint32_t)(it - 1), SITE_PRED, 0, (uint32_t)(lane * L), zp0);
#pragma unroll
            for (int l = 0; l < L; ++l) nz0[(3 * L + l) * 64 + lane] = zp0[l];
          }
        }
      }
      if (wave == 2 && it < n_iter) {     // x+_0 normals of thread 0
        // evaluated by ALL lanes (elements 0 .. 63 of the site), stored by the first D: evaluated
        // under a one-lane mask the compiler moves the whole computation to the scalar unit's
        // registers, and that build of the same source was seen to differ by one ulp from the
        // vector build (round 4) -- values that must agree between kernels stay vector code
        float zi[1];
        fill_normals<1>(rng, (uint32_t)it, SITE_PRIOR_INIT, 0, (uint32_t)lane, zi);
        asm volatile("" : "+v"(zi[0]));      // (keeps the evaluation out of the one-lane block below)
        if (lane < D) nz0[4 * L * 64 + lane] = zi[0];
      }
      if (it < n_iter) {
        dk_normals<D, L>(rng, (uint32_t)it, tid, zl, zs, zo);
        if (D == 2 || wave != 2) {
          const uint32_t site = wave == 1 ? SITE_PRIOR_LEVEL : (wave == 2 ? SITE_PRIOR_SLOPE : SITE_PRIOR_OBS);
          float z0[L];
          fill_normals<L>(rng, (uint32_t)it, site, 0, (uint32_t)(lane * L), z0);
#pragma unroll
          for (int l = 0; l < L; ++l) nz0[((wave - 1) * L + l) * 64 + lane] = z0[l];
        }
      }
    }
    __syncthreads();
    if constexpr (RPM == 2) {
      if (P + 1 > 32 && it < n_iter) {
        // 32-52 columns: the regression draw with its (P+1)^2 sweeps spread over all four waves
        Prof bp;
        bp.start(a.prof, PROF && a.prof != nullptr && blockIdx.x == 0 && tid == 0);
        const double ns = spike_slab_draw_block<false>(R, P, cx->sp, gam[72], gam[73], rng, (uint32_t)it, tid,
                                                       it == 0, PROF ? &bp : nullptr, false, nullptr, 9,
                                                       gam + GAM_BLOCK_PRE + BLOCK_PRE_DOUBLES * (it & 1));
        if (tid == 0) {
          cx->obs_scale = ns;
          scal[SC_OBS_DK] = (float)ns;
        }
        __syncthreads();
      }
    }
    prof.tick(1);

    if (it > a.W) {
      if constexpr (RPM == 0) emit(scal[SC_OBS_EMIT], nullptr);
      else if (wave == 0) emit(so_prev, nz0 + 3 * L * 64);   // waves 1-3 emitted during the serial section
    }
    prof.tick(2);
    if (a.progress != nullptr && it > a.W) {
      const int done = it - a.W;                  // retained draws 0 .. done-1 are in HBM
      if (done % a.progress_every == 0 || done == a.S) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows have reached L2
        __syncthreads();                          // ... and so have every other wave's
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: write back L2
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
    if constexpr (RPM == 1) {
      float wv[16];
      lds_row_load<16>(wls, wv);           // the weights vector is padded to 16 floats
      if constexpr (!STREAM) {
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
        // INCLUDED features (a zero weight adds an exact zero: the same sums, bit for bit) -- a
        // run-time loop as in ci_kernels8.h: the unrolled form let the compiler hoist the loads of all
        // sixteen rows at once (256 registers at L = 16: 6 KB of scratch per lane)
        constexpr int RB = L >= 16 ? 2 : (L >= 8 ? 4 : 8);
        auto stream = [&](auto load_row) {
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
    } else if constexpr (RPM == 2) {
      // 16 features per round (independent row loads, see the X~'targets loop)
      auto xw_rounds = [&](auto load_row, auto wide) {
        // rows in flight: 8 from LDS; from L2 as many as 64 registers hold.  Only the rows of the
        // INCLUDED features are read (a zero weight contributes an exact zero to the sum: same bits)
        constexpr int XB = !decltype(wide)::value ? 8 : (L <= 4 ? 16 : (L == 8 ? 8 : 4));
        unsigned long long todo = __ballot(lane < P && wls[lane < P ? lane : 0] != 0.f);
        while (todo != 0ull) {
          float xr[XB][L], wj[XB];
#pragma unroll
          for (int u = 0; u < XB; ++u) {
            const bool have = todo != 0ull;
            const int j = have ? __ffsll((long long)todo) - 1 : 0;
            todo &= todo - 1ull;
            wj[u] = have ? wls[j] : 0.f;
            load_row(j, xr[u]);
          }
#pragma unroll
          for (int u = 0; u < XB; ++u)
#pragma unroll
            for (int l = 0; l < L; ++l) xw[l] = fmaf(xr[u][l], wj[u], xw[l]);
        }
      };
      if (a.x_in_lds) {
        xw_rounds([&](int j, float (&xr)[L]) { lds_row_load<L>(Xs + j * TPAD + t0, xr); }, std::false_type());
      } else if (xwide) {
        if constexpr (L % 4 == 0)
          xw_rounds([&](int j, float (&xr)[L]) { global_row_load_wide<L>(Xg + (size_t)j * T, t0, T, xr); },
                    std::true_type());
      } else {
        xw_rounds([&](int j, float (&xr)[L]) { global_row_load_scalar<L>(Xg + (size_t)j * T, t0, T, xr); },
                  std::true_type());
      }
    }
    if constexpr (PARK) {
      float yq[L];
      lds_row_load<L>(ybuf + t0, yq);
#pragma unroll
      for (int l = 0; l < L; ++l) resid[l] = yq[l] - xw[l];
      store_targets<L>(xwb, t0, xw);       // read back by the emission of this draw
    } else {
#pragma unroll
      for (int l = 0; l < L; ++l) resid[l] = yv[l] - xw[l];
    }
    DkModel<D> md;
    {
      const float so = scal[SC_OBS_DK];
      md.H = so * so;
      md.sig.v[0] = scal[SC_LEVEL];
      md.a1 = vzero<D>();
      md.a1.v[0] = init_loc;
      md.p1.v[0] = init_var;
      if constexpr (D == 2) {
        md.sig.v[1] = scal[SC_SLOPE];
        md.p1.v[1] = init_svar;
      }
    }
    Vec<D> x[L];
    prof.tick(3);
    if (wave == 0) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        zl[l] = nz0[(0 * L + l) * 64 + lane];
        if constexpr (D == 2) zs[l] = nz0[(1 * L + l) * 64 + lane];
        zo[l] = nz0[(2 * L + l) * 64 + lane];
      }
    }
    dk_draw<D, L>(md, resid, maskbits, rng, (uint32_t)it, tid, lane, wave, slots, x, prof, zl, zs, zo,
                  nz0 + 4 * L * 64);
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
      if constexpr (PARK) {
        if (t < T) pm[t] = (a.S > 0 ? pm[t] : 0.f) * inv;      // (this thread's own sums, written above)
      } else {
        if (t < T) pm[t] = pm_acc[l] * inv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// component test kernel: one Durbin-Koopman draw
// ------------------------------------------------------------------------------------
template <int D, int L>
__global__ __launch_bounds__(NT) void test_dk_kernel(int T, const float* resid_g,
                                                     const uint8_t* mask_g, DkModel<D> md,
                                                     uint32_t k0, uint32_t k1, uint32_t chain,
                                                     uint32_t iter, float* out) {
  __shared__ float slots[3 * NW * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = tid * L;
  float resid[L];
  uint32_t maskbits = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    resid[l] = (t < T) ? resid_g[t] : 0.f;
    if (t >= T || mask_g[t]) maskbits |= 1u << l;
  }
  Rng g{k0, k1, chain};
  Vec<D> x[L];
  Prof prof;
  prof.start(nullptr, false);
  dk_draw<D, L>(md, resid, maskbits, g, iter, tid, lane, wave, slots, x, prof);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    if (t < T)
#pragma unroll
      for (int i = 0; i < D; ++i) out[(size_t)t * D + i] = x[l].v[i];
  }
}

// ------------------------------------------------------------------------------------
// Kalman-filter log-likelihood of the trend + regression model,
// one workgroup per parameter set:  l(theta) = sum_{t observed} log N(v_t; 0, F_t).
// theta[e] = (sigma_obs, sigma_level, sigma_slope, weights[P]).  Same associative filter as
// the sampler; oracle: ci_oracle_kalman_loglik (pinned to the dense MVN log-density).
// ------------------------------------------------------------------------------------
template <int D, int L>
__global__ __launch_bounds__(NT) void loglik_kernel(int T, int P, const float* __restrict__ y,
                                                    const uint8_t* __restrict__ mask,
                                                    const float* __restrict__ Xt,
                                                    const double* __restrict__ theta, float a1,
                                                    float p10, float p11,
                                                    double* __restrict__ out) {
  __shared__ float slots[NW * 16];
  __shared__ float part[NW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* th = theta + (size_t)blockIdx.x * (3 + P);
  const int t0 = tid * L;
  float resid[L];
  uint32_t maskbits = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    float r = 0.f;
    if (t < T && !mask[t]) {
      r = y[t];
      for (int j = 0; j < P; ++j) r = fmaf(-Xt[(size_t)j * T + t], (float)th[3 + j], r);
    } else {
      maskbits |= 1u << l;
    }
    resid[l] = r;
  }
  DkModel<D> md;
  const float so = (float)th[0];
  md.H = so * so;
  md.sig.v[0] = (float)th[1];
  md.a1 = vzero<D>();
  md.a1.v[0] = a1;
  md.p1.v[0] = p10;
  if constexpr (D == 2) {
    md.sig.v[1] = (float)th[2];
    md.p1.v[1] = p11;
  }
  Vec<D> q;
#pragma unroll
  for (int i = 0; i < D; ++i) q.v[i] = md.sig.v[i] * md.sig.v[i];
  Vec<D> ap[L];
  Mat<D> Pp[L];
  Vec<D> kf[L];
  float vf[L], fvar[L];
  Prof prof;
  prof.start(nullptr, false);
  kalman_filter_pass<D, L>(md, md.a1, q, resid, maskbits, tid, lane, wave, slots, ap, Pp, kf, vf,
                           fvar, prof);
  float acc = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    if (((maskbits >> l) & 1u) == 0u)
      acc -= 0.5f * (1.8378770664093453f + __logf(fvar[l]) + vf[l] * vf[l] * fvar[l]);
  }
  const float w = wave_prefix_dpp(acc);
  if (lane == 63) part[wave] = w;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3];
}

// ------------------------------------------------------------------------------------
// Score of the Kalman log-likelihood (Koopman & Shephard 1992):
//   e_t = v_t/F_t - K_t' T' r_t,   dl/dbeta = sum_obs x_t e_t,
//   dl/dH = 1/2 sum_obs (e_t^2 - D_t),  D_t = 1/F_t + K_t' (T' N_t T) K_t,
//   dl/dQ_ii = 1/2 sum_{t<T-1} (r_t[i]^2 - N_t[i][i]),
// with r_{t-1} = (I - K_t Z)' T' r_t + Z' v_t/F_t and N_{t-1} = (T (I - K_t Z))' N_t (T (I - K_t Z))
// + Z'Z/F_t (K_t = 0 and no Z terms at masked steps).  Both recursions are suffix scans
// (affine maps for r, congruence maps N -> L' N L + C for N).  Checked against central
// finite differences of the float64 oracle (tests/test_gpu_components.py).
// ------------------------------------------------------------------------------------
template <int D> struct NElem {
  Mat<D> Lm;
  Mat<D> C;
};
template <int D> __device__ __forceinline__ NElem<D> nelem_identity() {
  NElem<D> e;
  e.Lm = meye<D>();
  e.C = mzero<D>();
  return e;
}
// outer acts after inner:  N -> Lo' (Li' N Li + Ci) Lo + Co
template <int D>
__device__ __forceinline__ NElem<D> nelem_compose(const NElem<D>& o, const NElem<D>& i) {
  NElem<D> r;
  r.Lm = mm(i.Lm, o.Lm);
  r.C = madd(mtm(o.Lm, mm(i.C, o.Lm)), o.C);
  symmetrize(r.C);
  return r;
}

// Both recursions at once: the r map is the transpose of the N map's L (M = L'), so one element
// (L, c, C packed symmetric) carries  r -> L' r + c  and  N -> L' N L + C  -- one suffix scan of
// 4 + 2 + 3 floats (d = 2) instead of two scans of 6 and 8.  outer acts after inner.
template <int D> struct RNElem {
  Mat<D> Lm;
  Vec<D> c;
  float C[D * (D + 1) / 2];
};
template <int D> __device__ __forceinline__ RNElem<D> rnelem_identity() {
  RNElem<D> e;
  e.Lm = meye<D>();
  e.c = vzero<D>();
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) e.C[i] = 0.f;
  return e;
}
template <int D>
__device__ __forceinline__ RNElem<D> rnelem_compose(const RNElem<D>& o, const RNElem<D>& i) {
  RNElem<D> r;
  r.Lm = mm(i.Lm, o.Lm);
  r.c = vadd(mtv(o.Lm, i.c), o.c);
  // Lo' Ci Lo + Co, upper triangle
  Mat<D> t;                                   // Ci Lo
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      float sv = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) sv = fmaf(i.C[symidx<D>(a, k)], o.Lm.m[k][b], sv);
      t.m[a][b] = sv;
    }
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) {
      float sv = o.C[symidx<D>(a, b)];
#pragma unroll
      for (int k = 0; k < D; ++k) sv = fmaf(o.Lm.m[k][a], t.m[k][b], sv);
      r.C[symidx<D>(a, b)] = sv;
    }
  return r;
}

// Log-likelihood and score of ONE parameter set th = (sigma_obs, sigma_level, sigma_slope, beta[P])
// by the whole 256-thread workgroup: ll -> *out_ll, d ll / d th -> out_grad[3 + P].  Contains
// __syncthreads(); slots: 3 * NW * 16 floats, part: NW * (P + 4) floats (LDS).  The results are
// written by threads 0 .. P+3: the caller synchronises before reading them.
// The thread's own observations (0 where masked or beyond T) and their mask bits: callers that
// evaluate many parameter sets load them once (hmc_kernel: a round trip to L2 per leapfrog step).
template <int L>
__device__ __forceinline__ void loglik_load_obs(int T, const float* __restrict__ y,
                                                const uint8_t* __restrict__ mask, int tid,
                                                float (&yv)[L], uint32_t& maskbits) {
  const int t0 = tid * L;
  maskbits = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    const bool obs = t < T && !mask[t];
    yv[l] = obs ? y[t] : 0.f;
    if (!obs) maskbits |= 1u << l;
  }
}

template <int D, int L>
__device__ __forceinline__ void loglik_grad_block_obs(int T, int P, const float (&yv)[L],
                                                      const uint32_t maskbits,
                                                      const float* __restrict__ Xt, const double* th,
                                                      float a1, float p10, float p11, float* slots,
                                                      float* part, double* out_ll, double* out_grad,
                                                      int tid, int lane, int wave, int xstride = 0,
                                                      Prof* pf = nullptr) {
  // Xt: feature-major design, row j at Xt + j * xs (xs = T for the matrix in HBM; callers that
  // evaluate many parameter sets keep a zero-padded copy in LDS and pass its row length)
  const size_t xs = xstride ? (size_t)xstride : (size_t)T;
  const int t0 = tid * L;
  float resid[L];
#pragma unroll
  for (int l = 0; l < L; ++l) resid[l] = yv[l];
  // the caller's zero-padded copy of the design (xstride != 0: rows of a multiple of 4 floats in
  // LDS) is read as whole rows of the L owned steps; the matrix in HBM step by step
  Prof prof;                       // (phase cycles for tools/exp_hmc_phases.py; inert in production)
  prof.p = pf ? pf->p : nullptr;
  prof.t = pf ? pf->t : 0;
  prof.tick(20);
  const bool rows = xstride != 0 && (xstride & 3) == 0;
  // the design streamed from HBM / L2 (xstride == 0): float4 rows when every row is 16-byte aligned.
  // The row source is chosen OUTSIDE the loops over features (see global_row_load_wide) and the
  // rows travel in batches of 4 independent loads: with the choice inside the loop every row
  // cost a full L2 round trip (26 us per leapfrog at P = 36 against 12 at P = 31).
  const bool gwide = xstride == 0 && L % 4 == 0 && (T & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(Xt) & 15) == 0;
  auto with_rows = [&](auto body) {
    if (rows) {
      body([&](int j, float (&xr)[L]) { lds_row_load<L>(Xt + j * xs + t0, xr); });
    } else if (gwide) {
      if constexpr (L % 4 == 0)
        body([&](int j, float (&xr)[L]) { global_row_load_wide<L>(Xt + j * xs, t0, T, xr); });
    } else {
      body([&](int j, float (&xr)[L]) { global_row_load_scalar<L>(Xt + j * xs, t0, T, xr); });
    }
  };
  with_rows([&](auto load_row) {
    // (a zero weight beyond P adds an exact zero; the sums run over j in ascending order either way)
    auto batch = [&](int j0, auto nb) __attribute__((always_inline)) {
      constexpr int NB = decltype(nb)::value;
      float xr[NB][L], bj[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int j = j0 + u < P ? j0 + u : P - 1;
        bj[u] = j0 + u < P ? (float)th[3 + j] : 0.f;
        load_row(j, xr[u]);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int l = 0; l < L; ++l) resid[l] = fmaf(-xr[u][l], bj[u], resid[l]);
    };
    if (L <= 4 && P > 4 && P <= 12) {
      // every row of a small design in flight at once: one LDS / L2 latency instead of three
      batch(0, std::integral_constant<int, 12>());
    } else {
      for (int j0 = 0; j0 < P; j0 += 4) batch(j0, std::integral_constant<int, 4>());
    }
  });
#pragma unroll
  for (int l = 0; l < L; ++l)
    if ((maskbits >> l) & 1u) resid[l] = 0.f;
  DkModel<D> md;
  const float so = (float)th[0];
  md.H = so * so;
  md.sig.v[0] = (float)th[1];
  md.a1 = vzero<D>();
  md.a1.v[0] = a1;
  md.p1.v[0] = p10;
  if constexpr (D == 2) {
    md.sig.v[1] = (float)th[2];
    md.p1.v[1] = p11;
  }
  Vec<D> q;
#pragma unroll
  for (int i = 0; i < D; ++i) q.v[i] = md.sig.v[i] * md.sig.v[i];
  Vec<D> ap[L];
  Mat<D> Pp[L];
  Vec<D> kf[L];
  float vf[L], fvar[L];
  prof.tick(8);
  kalman_filter_pass<D, L>(md, md.a1, q, resid, maskbits, tid, lane, wave, slots, ap, Pp, kf, vf,
                           fvar, prof);
  prof.tick(9);
  float ll = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l)
    if (((maskbits >> l) & 1u) == 0u)
      ll -= 0.5f * (1.8378770664093453f + __logf(fvar[l]) + vf[l] * vf[l] * fvar[l]);

  // per-step maps
  Mat<D> Tt = meye<D>();
  if constexpr (D == 2) Tt.m[1][0] = 1.f;               // T'
  const Mat<D> Tm = trans_mat<D>();
  auto ikz = [&](int l) {                                // I - K Z  (Z = e_0')
    Mat<D> m = meye<D>();
#pragma unroll
    for (int i = 0; i < D; ++i) m.m[i][0] -= kf[l].v[i];
    return m;
  };
  auto rn_map = [&](int l) {
    RNElem<D> e;
    e.Lm = mm(Tm, ikz(l));                   // T (I - K Z);  the r map is its transpose
    e.c = vzero<D>();
    e.c.v[0] = vf[l];
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) e.C[i] = 0.f;
    if (((maskbits >> l) & 1u) == 0u) e.C[0] = __builtin_amdgcn_rcpf(fvar[l]);
    return e;
  };
  RNElem<D> rntot = rn_map(L - 1);
#pragma unroll
  for (int l = L - 2; l >= 0; --l) rntot = rnelem_compose(rn_map(l), rntot);
  prof.tick(10);
  const RNElem<D> rnsuf = block_scan_excl_bwd(
      rntot, [](const RNElem<D>& o, const RNElem<D>& i) { return rnelem_compose(o, i); },
      rnelem_identity<D>(), slots + NW * 16, lane, wave);
  prof.tick(11);

  float gH = 0.f, gQ[D], e_l[L];
#pragma unroll
  for (int i = 0; i < D; ++i) gQ[i] = 0.f;
  {
    Vec<D> r = rnsuf.c;      // r_t, N_t entering this thread's last step
    Mat<D> N;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) N.m[i][j] = rnsuf.C[symidx<D>(i, j)];
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      const int t = t0 + l;
      if (t + 1 < T) {
#pragma unroll
        for (int i = 0; i < D; ++i) gQ[i] += 0.5f * (r.v[i] * r.v[i] - N.m[i][i]);
      }
      const Vec<D> rT = mv(Tt, r);
      const Mat<D> NT = mm(Tt, mm(N, Tm));
      e_l[l] = 0.f;
      if (((maskbits >> l) & 1u) == 0u) {
        float kr = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) kr = fmaf(kf[l].v[i], rT.v[i], kr);
        const float e = vf[l] - kr;
        const Vec<D> nk = mv(NT, kf[l]);
        const float rfv = __builtin_amdgcn_rcpf(fvar[l]);      // the same 1 / F_t the scan's element used
        float dt = rfv;
#pragma unroll
        for (int i = 0; i < D; ++i) dt = fmaf(kf[l].v[i], nk.v[i], dt);
        gH += 0.5f * (e * e - dt);
        e_l[l] = e;
        const Mat<D> m = ikz(l);
        r = mtv(m, rT);
        r.v[0] += vf[l];
        N = mtm(m, mm(NT, m));
        N.m[0][0] += rfv;
        symmetrize(N);
      } else {
        r = rT;
        N = NT;
      }
    }
  }
  prof.tick(12);
  // block sums: ll, dH, dQ[0], dQ[1], then dbeta_j
  const int NS = P + 4;
  // d l / d beta_j = sum_t x_jt e_t: rows in batches of 4 independent loads, 16 wave sums per
  // reduce-scatter (lane l < 16 ends up with the wave total of slot l)
  with_rows([&](auto load_row) {
    auto dots4 = [&](int j0, float* out) {            // out[u] = this thread's share of feature j0 + u
      if (j0 >= P) {                                  // (no design at all, or fewer than j0 columns)
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u] = 0.f;
        return;
      }
      float xr[4][L];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_row(j0 + u < P ? j0 + u : P - 1, xr[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float sv = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) sv = fmaf(xr[u][l], e_l[l], sv);
        out[u] = j0 + u < P ? sv : 0.f;
      }
    };
    {
      // slots 0..3: ll, dH, dQ[0], dQ[1]; slots 4..15: the first 12 features
      float v16[16];
      v16[0] = ll; v16[1] = gH; v16[2] = gQ[0]; v16[3] = D == 2 ? gQ[D - 1] : 0.f;
      dots4(0, v16 + 4);
      dots4(4, v16 + 8);
      dots4(8, v16 + 12);
      prof.tick(22);
      const float tot = wave_reduce_scatter16(v16, lane);
      if (lane < 16 && lane < NS) part[wave * NS + lane] = tot;
    }
    for (int j0 = 12; j0 < P; j0 += 16) {
      float d16[16];
      dots4(j0, d16);
      dots4(j0 + 4, d16 + 4);
      dots4(j0 + 8, d16 + 8);
      dots4(j0 + 12, d16 + 12);
      const float tot = wave_reduce_scatter16(d16, lane);
      if (lane < 16 && j0 + lane < P) part[wave * NS + 4 + j0 + lane] = tot;
    }
  });
  prof.tick(13);
  __syncthreads();
  if (tid < NS) {
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += (double)part[w * NS + tid];
    if (tid == 0) *out_ll = s;
    else {
      double* g = out_grad;
      if (tid == 1) g[0] = 2.0 * th[0] * s;            // d/d sigma_obs   = 2 sigma dl/dH
      else if (tid == 2) g[1] = 2.0 * th[1] * s;       // d/d sigma_level
      else if (tid == 3) g[2] = (D == 2) ? 2.0 * th[2] * s : 0.0;
      else g[3 + (tid - 4)] = s;                       // d/d beta_j
    }
  }
  prof.tick(14);
  if (pf) pf->t = prof.t;
}

template <int D, int L>
__device__ __forceinline__ void loglik_grad_block(int T, int P, const float* __restrict__ y,
                                                  const uint8_t* __restrict__ mask,
                                                  const float* __restrict__ Xt, const double* th,
                                                  float a1, float p10, float p11, float* slots,
                                                  float* part, double* out_ll, double* out_grad,
                                                  int tid, int lane, int wave, int xstride = 0) {
  float yv[L];
  uint32_t maskbits;
  loglik_load_obs<L>(T, y, mask, tid, yv, maskbits);
  loglik_grad_block_obs<D, L>(T, P, yv, maskbits, Xt, th, a1, p10, p11, slots, part, out_ll, out_grad,
                              tid, lane, wave, xstride);
}

template <int D, int L>
__global__ __launch_bounds__(NT) void loglik_grad_kernel(int T, int P, const float* __restrict__ y,
                                                         const uint8_t* __restrict__ mask,
                                                         const float* __restrict__ Xt,
                                                         const double* __restrict__ theta, float a1,
                                                         float p10, float p11,
                                                         double* __restrict__ out_ll,
                                                         double* __restrict__ out_grad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
  float* slots = (float*)smem_g;                 // 3 * NW * 16
  float* part = slots + 3 * NW * 16;             // NW * (P + 4)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  loglik_grad_block<D, L>(T, P, y, mask, Xt, theta + (size_t)blockIdx.x * (3 + P), a1, p10, p11,
                          slots, part, out_ll + blockIdx.x,
                          out_grad + (size_t)blockIdx.x * (3 + P), tid, lane, wave);
}

// ------------------------------------------------------------------------------------
// Latent path + posterior-predictive trajectory for GIVEN parameter draws (one workgroup per
// draw): what one_step_predictive needs after an HMC fit, where the latents are not part of
// the chain state.  theta[e] = (sigma_obs, sigma_level, sigma_slope, weights[P]); RNG stream:
// chain = rng_chain, iteration = e.
// ------------------------------------------------------------------------------------
template <int D, int L>
__global__ __launch_bounds__(NT) void latents_kernel(int T, int P, const float* __restrict__ y,
                                                     const uint8_t* __restrict__ mask,
                                                     const float* __restrict__ Xt,
                                                     const double* __restrict__ theta, float a1,
                                                     float p10, float p11, uint32_t k0, uint32_t k1,
                                                     uint32_t rng_chain, uint32_t iter0,
                                                     int per_chain, int group, int num_rows,
                                                     float* __restrict__ out_level,
                                                     float* __restrict__ out_slope,
                                                     float* __restrict__ out_loc,
                                                     float* __restrict__ out_traj,
                                                     float* __restrict__ out_loc_sum) {
  // Rows (parameter draws) handled by this workgroup.  per_chain > 0: rows are [chain][draw]
  // blocks of per_chain draws (a whole HMC fit in one launch) and the grid is
  // chains x ceil(per_chain / group) workgroups, each running `group` consecutive draws of one
  // chain and leaving the sum of their noise-free predictors in out_loc_sum[workgroup][T] (the
  // per-chain posterior mean is then a short deterministic reduction, with no [draws, T] array
  // written and re-read for it).  per_chain == 0: one row per workgroup (group = 1).
  __shared__ float slots[3 * NW * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = tid * L;
  int row0 = blockIdx.x, row1 = blockIdx.x + 1;
  uint32_t row_chain = 0u;
  if (per_chain > 0) {
    const int ng = (per_chain + group - 1) / group;
    row_chain = blockIdx.x / (uint32_t)ng;
    const int g = blockIdx.x % ng;
    row0 = (int)row_chain * per_chain + g * group;
    row1 = row0 + group;
    const int chain_end = ((int)row_chain + 1) * per_chain;
    if (row1 > chain_end) row1 = chain_end;
  }
  if (row1 > num_rows) row1 = num_rows;
  uint32_t maskbits = 0;
  float yv[L], acc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int t = t0 + l;
    const bool m = (t >= T) || mask[t] != 0;
    if (m) maskbits |= 1u << l;
    yv[l] = m ? 0.f : y[t];
    acc[l] = 0.f;
  }
  Rng g{k0, k1, rng_chain + row_chain};
  for (int row = row0; row < row1; ++row) {
    const double* th = theta + (size_t)row * (3 + P);
    float resid[L], xw[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int t = t0 + l;
      float s = 0.f;
      if (t < T)
        for (int j = 0; j < P; ++j) s = fmaf(Xt[(size_t)j * T + t], (float)th[3 + j], s);
      xw[l] = s;
      resid[l] = ((maskbits >> l) & 1u) ? 0.f : yv[l] - s;
    }
    DkModel<D> md;
    const float so = (float)th[0];
    md.H = so * so;
    md.sig.v[0] = (float)th[1];
    md.a1 = vzero<D>();
    md.a1.v[0] = a1;
    md.p1.v[0] = p10;
    if constexpr (D == 2) {
      md.sig.v[1] = (float)th[2];
      md.p1.v[1] = p11;
    }
    const uint32_t iter = iter0 + (uint32_t)(per_chain > 0 ? row - (int)row_chain * per_chain : row);
    Vec<D> x[L];
    Prof prof;
    prof.start(nullptr, false);
    dk_draw<D, L>(md, resid, maskbits, g, iter, tid, lane, wave, slots, x, prof);
    float zp[L];
    fill_normals<L>(g, iter, SITE_PRED, 0, (uint32_t)t0, zp);
    const size_t base = (size_t)row * T;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int t = t0 + l;
      if (t < T) {
        const float loc = x[l].v[0] + xw[l];
        acc[l] += loc;
        out_level[base + t] = x[l].v[0];
        if constexpr (D == 2) { if (out_slope) out_slope[base + t] = x[l].v[1]; }
        if (out_loc) out_loc[base + t] = loc;
        out_traj[base + t] = fmaf(so, zp[l], loc);
      }
    }
    __syncthreads();     // slots are reused by the next draw
  }
  if (out_loc_sum) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int t = t0 + l;
      if (t < T) out_loc_sum[(size_t)blockIdx.x * T + t] = acc[l];
    }
  }
}

}  // namespace ci
