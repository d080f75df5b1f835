// ci_summary.h -- on-device summarisation of the pooled posterior-predictive draws
// (reference: causalimpact_lib.py:793-837 point / cumulative effect
// trajectories, posterior_processing.py:25-60 per-timestep quantiles, :966-1017 per-draw
// post-period totals).  Once the sampler takes ~20 ms, numpy's T x (C*S) quantiles (250 ms at
// cfg2) are the wall time of fit_causalimpact; these kernels are HBM-bound streaming passes.
//
// Arithmetic is float64 and ordered exactly like the numpy code it replaces (separately rounded
// multiply/add for the scaler, sequential running sums over time), and quantiles come back as
// ORDER STATISTICS (the host applies numpy's own interpolation), so the device path reproduces
// the host frames to round-off, not to "float32 tolerance".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ci {

constexpr int SUMM_MAX_RANKS = 8;

// [N, T] float32 draws (model scale) -> [T, N] float64 on the data scale:
// standardize.py:60-64 `values * stddev + mean` (two roundings, no FMA).
// grid (ceil(T/64), ceil(N/64), B), block (64, 4); series b uses (scale[b], shift[b]).
// TIn: float (the sampler's float32 container) or double (float64 fits pooled on the host, round 5).
template <class TIn>
__global__ __launch_bounds__(256) void summ_transpose_kernel(int N, int T,
                                                             const TIn* __restrict__ traj_all,
                                                             const double* __restrict__ scales,
                                                             const double* __restrict__ shifts,
                                                             double* __restrict__ predT_all) {
  __shared__ double tile[64][65];
  const TIn* traj = traj_all + (size_t)blockIdx.z * N * T;
  double* predT = predT_all + (size_t)blockIdx.z * N * T;
  const double scale = scales[blockIdx.z], shift = shifts[blockIdx.z];
  const int t0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + ty + 4 * i, t = t0 + tx;
    if (n < N && t < T)
      tile[ty + 4 * i][tx] = __dadd_rn(__dmul_rn((double)traj[(size_t)n * T + t], scale), shift);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = t0 + ty + 4 * i, n = n0 + tx;
    if (n < N && t < T) predT[(size_t)t * N + n] = tile[tx][ty + 4 * i];
  }
}

// One thread per draw, sequential over time (coalesced over draws):
//   point = -(pred - obs)                      (:822)   NaN where there is no observation
//   cum_t = running sum of point from the treatment start, NaN steps skipped but reported NaN
//   pred_sum / point_sum over the post-period window (:985-1017; nansum for the effects).
// flags[t]: bit 0 = t >= treatment start, bit 1 = inside the post-period window.
// grid (ceil(N/64), B), one wavefront per workgroup (spread over as many CUs as possible).
__global__ __launch_bounds__(64) void summ_cumsum_kernel(int N, int T,
                                                          const double* __restrict__ predT_all,
                                                          const double* __restrict__ obs_all,
                                                          const uint8_t* __restrict__ flags_all,
                                                          double* __restrict__ cumT_all,
                                                          double* __restrict__ per_draw_all) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const size_t b = blockIdx.y;
  const double* predT = predT_all + b * N * T;
  double* cumT = cumT_all + b * N * T;
  const double* obs = obs_all + b * T;
  const uint8_t* flags = flags_all + b * T;
  double* per_draw = per_draw_all + b * 2 * N;
  double c = 0.0, pred_sum = 0.0, point_sum = 0.0;
  // The sums are strictly sequential in t (numpy's rounding order); the loads are not: 16
  // rows (and their observations / flags) are fetched ahead of the dependent adds.  Rows before
  // the treatment start (flags 0) contribute nothing: their predictions are not even read.
  constexpr int AHEAD = 16;
  for (int t8 = 0; t8 < T; t8 += AHEAD) {
    unsigned f8[AHEAD], fany = 0u;
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      f8[u] = t8 + u < T ? flags[t8 + u] : 0u;
      fany |= f8[u];
    }
    if (fany == 0u) {                                   // uniform: same flags for every draw
#pragma unroll
      for (int u = 0; u < AHEAD; ++u)
        if (t8 + u < T) cumT[(size_t)(t8 + u) * N + n] = c;
      continue;
    }
    double p8[AHEAD], o8[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      p8[u] = (t8 + u < T) ? predT[(size_t)(t8 + u) * N + n] : 0.0;
      o8[u] = obs[t8 + u < T ? t8 + u : T - 1];
    }
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const int t = t8 + u;
      if (t < T) {
        const double p = p8[u];
        const double point = -__dsub_rn(p, o8[u]);
        const unsigned f = f8[u];
        const double base = (f & 1u) ? point : 0.0;
        const bool hole = base != base;
        c = __dadd_rn(c, hole ? 0.0 : base);
        cumT[(size_t)t * N + n] = hole ? base : c;
        if (f & 2u) {
          pred_sum = __dadd_rn(pred_sum, p);
          point_sum = __dadd_rn(point_sum, (point != point) ? 0.0 : point);
        }
      }
    }
  }
  per_draw[n] = pred_sum;
  per_draw[(size_t)N + n] = point_sum;
}

__device__ __forceinline__ unsigned long long summ_key(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double summ_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// Order statistics of every row of M [B*T, N] (float64): out[b, r, t] = ranks[r]-th smallest of
// row b*T + t.  This is the kernel for LONG rows (N > 16384; summ_select_reg_kernel below takes
// the others): one 256-thread workgroup per row; most-significant-digit radix select, up to 8
// passes of 8 bits, all R ranks carried through the same sweeps of the row, swept from L2.
__global__ __launch_bounds__(256) void summ_select_kernel(int N, int T, int R,
                                                          const int* __restrict__ ranks,
                                                          const double* __restrict__ M,
                                                          double* __restrict__ out) {
  __shared__ unsigned hist[SUMM_MAX_RANKS][256];
  __shared__ unsigned long long prefix[SUMM_MAX_RANKS];
  __shared__ unsigned krem[SUMM_MAX_RANKS];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* x = M + (size_t)row * N;
  // The keys of one row share their leading bytes (sign, exponent, often the top mantissa bits
  // of similar doubles): the digit sweeps start at the first byte in which the row's smallest and
  // largest key differ -- typically 2-3 of the 8 sweeps are skipped.
  __shared__ unsigned long long kext[2][4];
  {
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for (int i = tid; i < N; i += 256) {
      const unsigned long long key = summ_key(x[i]);
      kmin = key < kmin ? key : kmin;
      kmax = key > kmax ? key : kmax;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { kext[0][wave] = kmin; kext[1][wave] = kmax; }
    __syncthreads();
  }
  int first_pass = 0;
  unsigned long long common = 0ull;
  {
    unsigned long long kmin = kext[0][0], kmax = kext[1][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      kmin = kext[0][w] < kmin ? kext[0][w] : kmin;
      kmax = kext[1][w] > kmax ? kext[1][w] : kmax;
    }
    const unsigned long long diff = kmin ^ kmax;
    first_pass = diff == 0ull ? 8 : (__clzll((long long)diff) >> 3);
    common = first_pass == 0 ? 0ull : (kmin & (~0ull << (64 - 8 * first_pass)));
  }
  if (tid < R) { prefix[tid] = common; krem[tid] = (unsigned)ranks[tid]; }
  __syncthreads();
  for (int pass = first_pass; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    // wave-aggregated counting pays while the digits are concentrated (the first sweeps after
    // the common prefix); the low mantissa bytes are uniform and take plain atomics
    const int agg_rounds = pass < first_pass + 2 ? 2 : 0;
    const unsigned long long mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int e = tid; e < R * 256; e += 256) (&hist[0][0])[e] = 0u;
    __syncthreads();
    unsigned long long pf[SUMM_MAX_RANKS];
    int grp[SUMM_MAX_RANKS];        // ranks that still share a prefix share one histogram
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) pf[r] = r < R ? prefix[r] : ~0ull;
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
      grp[r] = r;
#pragma unroll
      for (int q = SUMM_MAX_RANKS - 1; q >= 0; --q)
        if (q < r && pf[q] == pf[r]) grp[r] = q;
    }
    for (int i0 = 0; i0 < N; i0 += 256) {
      const int i = i0 + tid;
      const bool in = i < N;
      const unsigned long long key = summ_key(in ? x[i] : 0.0);
      const unsigned bin = (unsigned)(key >> shift) & 255u;
      const unsigned long long hi = key & mask;
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
        if (r >= R || grp[r] != r) continue;                 // uniform
        bool todo = in && hi == pf[r];
        // The leading digits of similar doubles are identical (sign, exponent): almost every lane
        // hits the same bin.  Two rounds of wave-aggregated counting take the dominant bins with
        // one atomic each; whatever is left is spread out and uses plain atomics.
        for (int round = 0; round < agg_rounds; ++round) {
          const unsigned long long act = __ballot(todo);
          if (act == 0ull) break;
          const int leader = __ffsll((long long)act) - 1;
          const unsigned lbin = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
          const unsigned long long same = __ballot(todo && bin == lbin);
          if (lane == leader) atomicAdd(&hist[r][lbin], (unsigned)__popcll(same));
          todo = todo && bin != lbin;
        }
        if (todo) atomicAdd(&hist[r][bin], 1u);
      }
    }
    __syncthreads();
    // wave w resolves ranks w, w+4: lane l owns bins 4l..4l+3
    for (int r = wave; r < R; r += 4) {
      int g = r;
#pragma unroll
      for (int q = 0; q < SUMM_MAX_RANKS; ++q)
        if (q == r) g = grp[q];
      const unsigned h0 = hist[g][4 * lane], h1 = hist[g][4 * lane + 1], h2 = hist[g][4 * lane + 2],
                     h3 = hist[g][4 * lane + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const unsigned excl = incl - mine, k = krem[r];
      if (k >= excl && k < incl) {
        unsigned before = excl, bin = 4 * lane;
        if (k >= before + h0) { before += h0; ++bin;
          if (k >= before + h1) { before += h1; ++bin;
            if (k >= before + h2) { before += h2; ++bin; } } }
        prefix[r] |= (unsigned long long)bin << shift;
        krem[r] = k - before;
      }
    }
    __syncthreads();
  }
  if (tid < R) out[((size_t)(row / T) * R + tid) * T + row % T] = summ_unkey(prefix[tid]);
}



// ---------------------------------------------------------------------------------------------
// Register-resident select (rows of up to SEL_NT * EPT values: the shape fit_causalimpact has,
// N = chains x draws).  One SEL_NT-thread workgroup per row, several rows in flight per CU; the
// row is read from HBM once, into registers (EPT values per thread), and never staged anywhere:
//   1. min / max of the row -> 2048 LINEAR buckets over its range, b(x) = trunc((x - min) * s).
//      b is monotone in x because every rounding involved is, so the buckets are an ordered
//      partition of the row whatever its sign pattern or exponent range (a digit of the bit
//      pattern puts a row that crosses zero, or spans a few binades, into a handful of bins);
//   2. one histogram sweep of the registers and one scan give every rank its bucket;
//   3. the few values sharing a bucket with a rank (N/500 per rank, typically) are compacted into
//      LDS as ordered keys -- per-lane counts, a DPP scan, one atomic per wavefront, then a split
//      into one list per rank -- and
//   4. one wavefront per rank finishes by direct counting inside its list (no more barriers).
// Rows this does not settle (non-finite values, heavy ties or outliers stretching the range:
// candidates that do not fit SEL_CAP keys, or a list longer than SEL_DIRECT_MAX) take the
// generic route instead: most-significant-digit radix select on the ordered keys, 8-bit digits
// aligned to the highest bit in which two keys of the row differ, swept from the registers until
// the candidates fit the lists and from the lists afterwards.
// Two matrices (values, cumulative effects) share one launch: workgroups [0, rows0) select from
// M0, the rest from M1.
// ---------------------------------------------------------------------------------------------
constexpr int SEL_CAP = 1024;
constexpr int SEL_BINS0 = 2048;              // first digit: 11 bits
constexpr int SEL_DIRECT_MAX = 256;          // longest list finished by direct counting

__device__ __forceinline__ unsigned long long sel_uniform(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long sel_lane64(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// inclusive prefix sum over the wavefront through the DPP crossbar (no LDS round trips)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned sel_dpp_add(unsigned v) {
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ unsigned sel_scan(unsigned v) {
  v = sel_dpp_add<0x111, 0xF>(v);   // row_shr:1
  v = sel_dpp_add<0x112, 0xF>(v);   // row_shr:2
  v = sel_dpp_add<0x114, 0xF>(v);   // row_shr:4
  v = sel_dpp_add<0x118, 0xF>(v);   // row_shr:8
  v = sel_dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
  v = sel_dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
  return v;
}
#ifdef CI_SEL_PROF   // tools/bench_select.hip only: cycle stamps of workgroup CI_SEL_PROF
__device__ unsigned long long sel_prof[64];
__device__ int sel_prof_n;
#define SEL_TICK(id) do { if (blockIdx.x == CI_SEL_PROF && threadIdx.x == 0) { \
  const int n_ = sel_prof_n++; if (n_ < 32) { sel_prof[2 * n_] = id; sel_prof[2 * n_ + 1] = __builtin_readcyclecounter(); } } } while (0)
#else
#define SEL_TICK(id) do {} while (0)
#endif

template <int SEL_NT, int EPT>
__global__ __launch_bounds__(SEL_NT) void summ_select_reg_kernel(
    int N, int T, int R, int rows0, const int* __restrict__ ranks, const double* __restrict__ M0,
    const double* __restrict__ M1, double* __restrict__ out0, double* __restrict__ out1) {
  constexpr int NW = SEL_NT / 64;
  static_assert(SEL_BINS0 % SEL_NT == 0 && SUMM_MAX_RANKS * 256 == SEL_BINS0, "histogram layout");
  __shared__ unsigned hist[SUMM_MAX_RANKS][256];         // also the one 2048-bucket first histogram
  __shared__ unsigned long long list[SEL_CAP];           // one list per rank, back to back
  __shared__ unsigned long long comb[SEL_CAP];           // the candidates before the split
  __shared__ unsigned long long prefix[SUMM_MAX_RANKS];
  __shared__ unsigned long long kext[3][NW];
  __shared__ unsigned krem[SUMM_MAX_RANKS], cnt[SUMM_MAX_RANKS], loff[SUMM_MAX_RANKS],
      llen[SUMM_MAX_RANKS], lfill[SUMM_MAX_RANKS], rbin[SUMM_MAX_RANKS], wsum[NW], comb_fill;
  __shared__ unsigned char owner_of_bin[SEL_BINS0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned* const hist0 = &hist[0][0];
  int row = blockIdx.x;
  const double* M = M0;
  double* out = out0;
  if (row >= rows0) { row -= rows0; M = M1; out = out1; }
  const double* x = M + (size_t)row * N;
  const size_t out_base = (size_t)(row / T) * R * T + row % T;
  SEL_TICK(0);

  double xv[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = u * SEL_NT + tid;
    xv[u] = x[i < N ? i : 0];
  }
  for (int e = tid; e < SEL_BINS0; e += SEL_NT) hist0[e] = 0u;
  for (int e = tid; e < SEL_BINS0 / 4; e += SEL_NT)
    reinterpret_cast<unsigned*>(owner_of_bin)[e] = 0xFFFFFFFFu;
  if (tid < SUMM_MAX_RANKS) lfill[tid] = 0u;
  if (tid == 0) comb_fill = 0u;
  const unsigned my_rank = lane < R ? (unsigned)ranks[lane] : 0u;
  // range of the row; `poison` turns NaN as soon as one value is not finite
  double lo = xv[0], hi = xv[0], poison = 0.0;
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    lo = fmin(lo, xv[u]);
    hi = fmax(hi, xv[u]);
    poison = fma(xv[u], 0.0, poison);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, off, 64));
    hi = fmax(hi, __shfl_xor(hi, off, 64));
    poison += __shfl_xor(poison, off, 64);
  }
  if (lane == 0) {
    kext[0][wave] = (unsigned long long)__double_as_longlong(lo);
    kext[1][wave] = (unsigned long long)__double_as_longlong(hi);
    kext[2][wave] = (unsigned long long)__double_as_longlong(poison);
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    lo = fmin(lo, __longlong_as_double((long long)kext[0][w]));
    hi = fmax(hi, __longlong_as_double((long long)kext[1][w]));
    poison += __longlong_as_double((long long)kext[2][w]);
  }
  lo = __longlong_as_double((long long)sel_uniform((unsigned long long)__double_as_longlong(lo)));
  hi = __longlong_as_double((long long)sel_uniform((unsigned long long)__double_as_longlong(hi)));
  const bool finite = sel_uniform((unsigned long long)(poison == poison ? 1 : 0)) != 0ull;
  if (finite && lo == hi && lo != 0.0) {               // a constant row (0: may mix +0 and -0)
    if (tid < R) out[out_base + (size_t)tid * T] = lo;
    return;
  }
#if defined(CI_SEL_STOP_AFTER) && CI_SEL_STOP_AFTER == 1   // tools/bench_select.hip only
  if (tid < R) out[out_base + (size_t)tid * T] = lo;
  return;
#endif
  SEL_TICK(1);
  const double span = hi - lo;
  const double bscale = (SEL_BINS0 - 0.5) / span;      // (hi - lo) * bscale < SEL_BINS0
  bool fast = finite && span > 0.0 && span <= 1.7e308 && bscale <= 1.7e308;
#define SEL_BUCKET(v) min((unsigned)(((v) - lo) * bscale), (unsigned)(SEL_BINS0 - 1))
  if (fast) {
    // ---- 2048 linear buckets: histogram of the registers ----
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      if (u * SEL_NT >= N) continue;                   // uniform
      if (u * SEL_NT + tid < N) atomicAdd(&hist0[SEL_BUCKET(xv[u])], 1u);
    }
    __syncthreads();
    SEL_TICK(2);
    {
      // thread t owns buckets [t * BPT, (t + 1) * BPT): scan, then place every rank
      constexpr int BPT = SEL_BINS0 / SEL_NT;
      unsigned h[BPT], mine = 0u;
#pragma unroll
      for (int b = 0; b < BPT; ++b) { h[b] = hist0[tid * BPT + b]; mine += h[b]; }
      const unsigned incl = sel_scan(mine);
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      unsigned excl = incl - mine;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < wave) excl += wsum[w];
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
        if (r >= R) continue;                          // uniform
        const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)my_rank, r);
        if (k >= excl && k < excl + mine) {
          unsigned before = excl, hb = 0u;
          int bsel = 0;
#pragma unroll
          for (int b = 0; b < BPT; ++b)
            if (k >= before + h[b] && b + 1 < BPT && bsel == b) { before += h[b]; bsel = b + 1; }
#pragma unroll
          for (int b = 0; b < BPT; ++b)
            if (b == bsel) hb = h[b];
          const unsigned bin = (unsigned)(tid * BPT + bsel);
          rbin[r] = bin;
          krem[r] = k - before;
          cnt[r] = hb;
          if (owner_of_bin[bin] == 255) owner_of_bin[bin] = (unsigned char)r;   // the first rank owns it
        }
      }
    }
    __syncthreads();
    SEL_TICK(3);
#if defined(CI_SEL_STOP_AFTER) && CI_SEL_STOP_AFTER == 2
    return;
#endif
    // the state of every rank through one LDS round trip: lane l reads rank l & 7
    const unsigned myb = rbin[lane & 7], myc = cnt[lane & 7], myk = krem[lane & 7];
    unsigned rb[SUMM_MAX_RANKS], cs[SUMM_MAX_RANKS], lo_[SUMM_MAX_RANKS];
    int grp[SUMM_MAX_RANKS];
    unsigned total = 0u, longest = 0u;
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
      rb[r] = r < R ? (unsigned)__builtin_amdgcn_readlane((int)myb, r) : ~0u - r;
      cs[r] = r < R ? (unsigned)__builtin_amdgcn_readlane((int)myc, r) : 0u;
    }
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
      grp[r] = r;
#pragma unroll
      for (int q = SUMM_MAX_RANKS - 1; q >= 0; --q)
        if (q < r && rb[q] == rb[r]) grp[r] = q;
      lo_[r] = total;
      if (r < R && grp[r] == r) { total += cs[r]; longest = cs[r] > longest ? cs[r] : longest; }
    }
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r)
#pragma unroll
      for (int q = 0; q < SUMM_MAX_RANKS; ++q)
        if (q < r && grp[r] == q) lo_[r] = lo_[q];
    fast = total <= (unsigned)SEL_CAP && longest <= (unsigned)SEL_DIRECT_MAX;
    if (fast) {
      // (a) the candidates of all ranks, compacted with one atomic per wavefront
      unsigned candmask = 0u;
      static_assert(EPT <= 32, "one flag bit per value");
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        if (u * SEL_NT >= N) continue;                 // uniform
        const bool f = owner_of_bin[SEL_BUCKET(xv[u])] != 255 && u * SEL_NT + tid < N;
        candmask |= (f ? 1u : 0u) << u;
      }
      const unsigned mycount = (unsigned)__popc(candmask);
      const unsigned incl = sel_scan(mycount);
      unsigned base = 0u;
      if (lane == 63 && incl) base = atomicAdd(&comb_fill, incl);
      unsigned pos = (unsigned)__builtin_amdgcn_readlane((int)base, 63) + incl - mycount;
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        if (u * SEL_NT >= N) continue;                 // uniform
        if ((candmask >> u) & 1u) comb[pos++] = summ_key(xv[u]);
      }
      if (tid < R) {
        unsigned mylo = 0u;
#pragma unroll
        for (int r = 0; r < SUMM_MAX_RANKS; ++r)
          if (tid == r) mylo = lo_[r];
        loff[tid] = mylo;
      }
      __syncthreads();
      // (b) split into one list per rank (ranks sharing the bucket share the list)
      for (unsigned j = tid; j < total; j += SEL_NT) {
        const unsigned long long k = comb[j];
        const unsigned o = owner_of_bin[SEL_BUCKET(summ_unkey(k))];
        list[loff[o] + atomicAdd(&lfill[o], 1u)] = k;
      }
      __syncthreads();
      SEL_TICK(4);
#if defined(CI_SEL_STOP_AFTER) && CI_SEL_STOP_AFTER == 3
      return;
#endif
      // (c) one wavefront per rank: the candidate with `krem` smaller keys in its list
      for (int r = wave; r < R; r += NW) {
        unsigned l0 = 0u, len = 0u;
#pragma unroll
        for (int q = 0; q < SUMM_MAX_RANKS; ++q)
          if (q == r) { l0 = lo_[q]; len = cs[q]; }
        const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)myk, r);
        for (unsigned c0 = 0; c0 < len; c0 += 64) {
          const bool valid = c0 + lane < len;
          const unsigned long long c = valid ? list[l0 + c0 + lane] : ~0ull;
          unsigned lt = 0u, le = 0u;
          for (unsigned i0 = 0; i0 < len; i0 += 64) {
            // 64 list entries spread over the lanes, broadcast one at a time
            const unsigned long long vv = i0 + lane < len ? list[l0 + i0 + lane] : ~0ull;
            const int ni = len - i0 < 64u ? (int)(len - i0) : 64;
            for (int i = 0; i < ni; ++i) {
              const unsigned long long v = sel_lane64(vv, i);
              lt += v < c ? 1u : 0u;
              le += v <= c ? 1u : 0u;
            }
          }
          if (valid && lt <= k && k < le) out[out_base + (size_t)r * T] = summ_unkey(c);
        }
      }
      SEL_TICK(5);
      return;
    }
  }
#undef SEL_BUCKET

  // ---- generic route: radix select on the ordered keys ----
  // the bits in which the keys do not all agree: (OR of the keys) ^ (AND of the keys)
  unsigned long long kor = 0ull, kand = ~0ull;
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const unsigned long long k = summ_key(xv[u]);
    kor |= k;
    kand &= k;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    kor |= __shfl_xor(kor, off, 64);
    kand &= __shfl_xor(kand, off, 64);
  }
  __syncthreads();                                     // (kext, hist are being reused)
  if (lane == 0) { kext[0][wave] = kor; kext[1][wave] = kand; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) { kor |= kext[0][w]; kand &= kext[1][w]; }
  kor = sel_uniform(kor);
  kand = sel_uniform(kand);
  const unsigned long long diff = kor ^ kand;
  if (diff == 0ull) {                                  // all keys equal (all-NaN rows included)
    if (tid < R) out[out_base + (size_t)tid * T] = summ_unkey(kor);
    return;
  }
  int shift = 64 - __clzll((long long)diff), width = 0;  // first digit: the 8 bits from the top one
  if (tid < R) {
    // `kand` carries the common bits above the highest differing one
    prefix[tid] = shift >= 64 ? 0ull : (kand & (~0ull << shift));
    krem[tid] = (unsigned)ranks[tid];
  }
  bool compacted = false;
  while (shift > 0) {
    width = shift < 8 ? shift : 8;
    shift -= width;
    const unsigned long long himask = shift + width >= 64 ? 0ull : (~0ull << (shift + width));
    const unsigned bmask = (1u << width) - 1u;
    for (int e = tid; e < SUMM_MAX_RANKS * 256; e += SEL_NT) hist0[e] = 0u;
    __syncthreads();
    unsigned long long pf[SUMM_MAX_RANKS];
    int grp[SUMM_MAX_RANKS];                 // ranks that share a prefix share a histogram
    {
      const unsigned long long myp = prefix[lane & 7];
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) pf[r] = r < R ? sel_lane64(myp, r) : ~0ull;
    }
    bool single = true;
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
      grp[r] = r;
#pragma unroll
      for (int q = SUMM_MAX_RANKS - 1; q >= 0; --q)
        if (q < r && pf[q] == pf[r]) grp[r] = q;
      if (r < R && grp[r] != 0) single = false;
    }
    const bool copies = !compacted && single;          // 8 histogram copies of the one group
    if (!compacted) {
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        if (u * SEL_NT >= N) continue;                 // uniform
        const bool in = u * SEL_NT + tid < N;
        const unsigned long long key = summ_key(xv[u]);
        const unsigned bin = (unsigned)(key >> shift) & bmask;
        const unsigned long long hi = key & himask;
        if (copies) {
          if (in && hi == pf[0]) atomicAdd(&hist[wave & 7][bin], 1u);
        } else {
#pragma unroll
          for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
            if (r >= R || grp[r] != r) continue;       // uniform
            if (in && hi == pf[r]) atomicAdd(&hist[r][bin], 1u);
          }
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
        if (r >= R || grp[r] != r) continue;           // uniform
        const unsigned l0 = loff[r], len = llen[r];
        for (unsigned j = tid; j < len; j += SEL_NT) {
          const unsigned long long k = list[l0 + j];
          if ((k & himask) == pf[r]) atomicAdd(&hist[r][(unsigned)(k >> shift) & bmask], 1u);
        }
      }
    }
    __syncthreads();
    // wave w resolves ranks w, w + #waves, ...: lane l owns bins 4l..4l+3
    for (int r = wave; r < R; r += NW) {
      int g = r;
#pragma unroll
      for (int q = 0; q < SUMM_MAX_RANKS; ++q)
        if (q == r) g = grp[q];
      unsigned h0, h1, h2, h3;
      if (copies) {
        h0 = h1 = h2 = h3 = 0u;
#pragma unroll
        for (int c = 0; c < (NW < 8 ? NW : 8); ++c) {
          h0 += hist[c][4 * lane]; h1 += hist[c][4 * lane + 1];
          h2 += hist[c][4 * lane + 2]; h3 += hist[c][4 * lane + 3];
        }
      } else {
        h0 = hist[g][4 * lane]; h1 = hist[g][4 * lane + 1];
        h2 = hist[g][4 * lane + 2]; h3 = hist[g][4 * lane + 3];
      }
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const unsigned excl = incl - mine, k = krem[r];
      if (k >= excl && k < incl) {
        unsigned before = excl, bin = 4 * lane, hb = h0;
        if (k >= before + h0) { before += h0; ++bin; hb = h1;
          if (k >= before + h1) { before += h1; ++bin; hb = h2;
            if (k >= before + h2) { before += h2; ++bin; hb = h3; } } }
        prefix[r] |= (unsigned long long)bin << shift;
        krem[r] = k - before;
        cnt[r] = hb;
      }
    }
    __syncthreads();
    if (shift == 0 || compacted) continue;
    // do the candidates fit the lists now?
    unsigned total = 0u;
    unsigned cs[SUMM_MAX_RANKS], lo[SUMM_MAX_RANKS];
    {
      const unsigned long long myp = prefix[lane & 7];
      const unsigned myc = cnt[lane & 7];
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
        pf[r] = r < R ? sel_lane64(myp, r) : ~0ull;
        cs[r] = r < R ? (unsigned)__builtin_amdgcn_readlane((int)myc, r) : 0u;
      }
    }
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
      grp[r] = r;
#pragma unroll
      for (int q = SUMM_MAX_RANKS - 1; q >= 0; --q)
        if (q < r && pf[q] == pf[r]) grp[r] = q;
      lo[r] = total;
      if (r < R && grp[r] == r) total += cs[r];
    }
    if (total > (unsigned)SEL_CAP) continue;
#pragma unroll
    for (int r = 0; r < SUMM_MAX_RANKS; ++r)
#pragma unroll
      for (int q = 0; q < SUMM_MAX_RANKS; ++q)
        if (q < r && grp[r] == q) lo[r] = lo[q];
    if (tid < R) {
      unsigned mylo = 0u, mylen = 0u;
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r)
        if (tid == r) { mylo = lo[r]; mylen = cs[r]; }
      loff[tid] = mylo; llen[tid] = mylen; lfill[tid] = 0u;
    }
    __syncthreads();
    const unsigned long long himask2 = ~0ull << shift;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      if (u * SEL_NT >= N) continue;                   // uniform
      const bool in = u * SEL_NT + tid < N;
      const unsigned long long key = summ_key(xv[u]);
      const unsigned long long hi = key & himask2;
#pragma unroll
      for (int r = 0; r < SUMM_MAX_RANKS; ++r) {
        if (r >= R || grp[r] != r) continue;           // uniform
        const bool m = in && hi == pf[r];
        const unsigned long long b = __ballot(m);
        if (b == 0ull) continue;                       // uniform
        const int leader = __ffsll((long long)b) - 1;
        unsigned base = 0u;
        if (lane == leader) base = atomicAdd(&lfill[r], (unsigned)__popcll(b));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
        if (m) list[lo[r] + base + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = key;
      }
    }
    compacted = true;
    __syncthreads();
  }
  if (tid < R) out[out_base + (size_t)tid * T] = summ_unkey(prefix[tid]);
}

}  // namespace ci
