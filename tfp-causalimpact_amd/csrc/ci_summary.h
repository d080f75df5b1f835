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
constexpr int SUMM_STAGE_MAX_N = 16384;   // 128 KiB of LDS per row

// [N, T] float32 draws (model scale) -> [T, N] float64 on the data scale:
// standardize.py:60-64 `values * stddev + mean` (two roundings, no FMA).
// grid (ceil(T/64), ceil(N/64), B), block (64, 4); series b uses (scale[b], shift[b]).
__global__ __launch_bounds__(256) void summ_transpose_kernel(int N, int T,
                                                             const float* __restrict__ traj_all,
                                                             const double* __restrict__ scales,
                                                             const double* __restrict__ shifts,
                                                             double* __restrict__ predT_all) {
  __shared__ double tile[64][65];
  const float* traj = traj_all + (size_t)blockIdx.z * N * T;
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
// grid (ceil(N/256), B).
__global__ __launch_bounds__(256) void summ_cumsum_kernel(int N, int T,
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
  // The sums are strictly sequential in t (numpy's rounding order); the loads are not: 8
  // rows (and their observations / flags) are fetched ahead of the dependent adds.
  constexpr int AHEAD = 8;
  for (int t8 = 0; t8 < T; t8 += AHEAD) {
    double p8[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) p8[u] = (t8 + u < T) ? predT[(size_t)(t8 + u) * N + n] : 0.0;
    double o8[AHEAD];
    unsigned f8[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const int tc = t8 + u < T ? t8 + u : T - 1;
      o8[u] = obs[tc];
      f8[u] = flags[tc];
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
// row b*T + t.
// One 256-thread workgroup per row; most-significant-digit radix select, 8 passes of 8 bits, all
// R ranks carried through the same sweeps of the row.  When the row fits LDS (stage_row: N * 8
// bytes of dynamic shared memory, N <= 16384) it is staged there once, so the matrix is read from
// HBM exactly once; longer rows are swept from L2.
__global__ __launch_bounds__(256) void summ_select_kernel(int N, int T, int R,
                                                          const int* __restrict__ ranks,
                                                          const double* __restrict__ M,
                                                          double* __restrict__ out, int stage_row) {
  __shared__ unsigned hist[SUMM_MAX_RANKS][256];
  __shared__ unsigned long long prefix[SUMM_MAX_RANKS];
  __shared__ unsigned krem[SUMM_MAX_RANKS];
  extern __shared__ __attribute__((aligned(16))) unsigned char summ_dyn[];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* x = M + (size_t)row * N;
  if (stage_row) {
    // the row is read from HBM exactly once (coalesced) and the 8 digit passes sweep LDS
    double* buf = reinterpret_cast<double*>(summ_dyn);
    for (int i = tid; i < N; i += 256) buf[i] = x[i];
    x = buf;
  }
  // The keys of one row share their leading bytes (sign, exponent, often the top mantissa bits
  // of similar doubles): the digit sweeps start at the first byte in which the row's smallest and
  // largest key differ -- typically 2-3 of the 8 sweeps are skipped.
  __shared__ unsigned long long kext[2][4];
  {
    if (stage_row) __syncthreads();
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

}  // namespace ci
