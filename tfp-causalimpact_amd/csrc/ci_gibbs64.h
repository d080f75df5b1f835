// ci_gibbs64.h -- the Gibbs sampler computed in FLOAT64 (DataOptions.dtype = float64).
//
// The reference runs its whole sampler in the requested dtype (causalimpact_lib.py:159) and its
// numeric pin runs float32 AND float64 (causalimpact_lib_test.py:655-662).  The register-resident
// and time-parallel kernels are float32 designs; this is the float64 build of the SEQUENTIAL
// one-wavefront kernel (ci_seasonal.h -- same passes, same lane layout, same random stream, any
// model: trend, optional slope, any list of seasonal blocks, any number of covariates, any length),
// with every quantity float64:
//   * arrays over time and the regression block's O(P^2) arrays in a per-chain HBM workspace;
//   * Box-Muller normals, Marsaglia-Tsang gammas, logs and square roots in float64 (the float32
//     kernels use hardware float32 transcendentals in places where 1e-7 does not matter);
//   * the regression draw of spike_slab_draw_big (dense float64 sweeps) for every P.
// Consequence: the device and the float64 oracle (oracle/ci_oracle.c) run the same arithmetic up
// to summation order, so they agree draw for draw to ~1e-9 over whole fits
// (tests/test_gpu_float64.py) -- a far tighter pin of the device algorithm than the float32
// kernels' 5e-3.  It is the precision option, not the fast one: sequential in time for models
// with seasonal blocks; trend-only models (the reference's default) are time-parallel over the
// 64 lanes of the chain's wavefront (Drift2 / Kf2 / Aff2 below).
#pragma once
#include "ci_seasonal.h"
#include "ci_hmc.h"       // wave_sum_d

namespace ci {

struct K64 {
  int T, P, W, S, C, B, chain_offset, series_stream_base;
  uint32_t seed0, seed1;
  const double* y;          // [B,T]   0 where masked
  const uint8_t* mask;      // [B,T]
  const double* Xt;         // [B,P,T] feature-major
  const double* xtx;        // [B,P,P]
  const double* omega;      // [B,P,P]
  const DevSeriesParams* sp;
  double *out_obs, *out_level_scale, *out_slope_scale, *out_weights, *out_level, *out_slope,
      *out_pred_mean, *out_traj;
  long long* prof;
};
struct G64Args {
  K64 k;
  int K, has_slope, dred;
  int nseas[SMAXK];
  const uint8_t* season_change;      // [K,T]
  const DevSeasonalParams* ssp;      // [B]
  const double* p1_chol;             // [B,dred,dred]
  double* out_drift;                 // [B,C,S,K]
  double* out_seasonal;              // [B,C,S,T,K]
  unsigned char* ws;                 // [B*C, ws_stride] bytes
  size_t ws_stride;
  const double* lat_theta;           // latents-only mode (see SArgs); NULL = sample
  int lat_S;
  int reg_lds;                       // regression block in LDS (small P)
};

struct Layout64 {
  // per-chain HBM workspace (arrays over time), bytes
  size_t yv, lev, slp, xw, ytil, vf, zl, zs, zo, seas, zk, gd, kf, rs, mask, cbits, t_total;
  // LDS
  size_t Pa, Pb, pzv, zi, x0r, egg, emeta, d2, bvec, w, reg, total;
};
// global_ws = 1: the arrays over time in the per-chain HBM workspace (any T); 0: in LDS behind the
// fixed part (short series: every per-step access then is an LDS access instead of an L2 round trip)
// reg_lds = 1: the regression block's O(P^2) arrays in LDS too (small P): LDS latency instead of
// L2 round trips in every sweep.
__host__ __device__ inline Layout64 make_layout64(int T, int P, int K, int D, int dred, int has_slope,
                                                  int global_ws = 1, int reg_lds = 0) {
  Layout64 l;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 31) & ~(size_t)31; return r; };
  const size_t TS = (size_t)((T + 3) & ~3);
  const size_t Td = sizeof(double) * TS;
  const int Pp = P > 0 ? P : 1, Kp = K > 0 ? K : 1;
  l.Pa = take(sizeof(double) * D * D); l.Pb = take(sizeof(double) * D * D);
  l.pzv = take(sizeof(double) * D); l.zi = take(sizeof(double) * (dred + 1));
  l.x0r = take(sizeof(double) * (dred + 1));
  l.egg = take(sizeof(double) * D * D);
  l.emeta = take(sizeof(uint32_t) * D * D);
  l.d2 = take(sizeof(double) * SMAXK);
  l.bvec = take(sizeof(double) * (Pp + 4));
  l.w = take(sizeof(double) * (Pp > 16 ? Pp : 16));
  l.reg = take(reg_lds ? bigp_workspace_bytes(Pp) : 32);
  const size_t lds_fixed = o;
  if (global_ws) o = 0;
  l.yv = take(Td); l.lev = take(Td); l.slp = take(has_slope ? Td : 32); l.xw = take(Td);
  l.ytil = take(Td); l.vf = take(Td); l.zl = take(Td); l.zs = take(has_slope ? Td : 32);
  l.zo = take(Td);
  const size_t Tk = K > 0 ? Td * Kp : 32;          // (no seasonal arrays without blocks)
  l.seas = take(Tk); l.zk = take(Tk); l.gd = take(Tk);
  l.kf = take(sizeof(double) * (size_t)T * D); l.rs = take(sizeof(double) * (size_t)T * D);
  l.mask = take(TS); l.cbits = take(TS);
  if (global_ws) { l.t_total = o; l.total = lds_fixed; }
  else { l.t_total = 0; l.total = o; }
  return l;
}
__host__ __device__ inline size_t gibbs64_ws_bytes(int T, int P, int K, int D, int dred, int has_slope,
                                                  int global_ws, int reg_lds) {
  const Layout64 l = make_layout64(T, P, K, D, dred, has_slope, global_ws, reg_lds);
  return ((l.t_total + 255) & ~(size_t)255) + (reg_lds ? 256 : bigp_workspace_bytes(P > 0 ? P : 1));
}

// ---- float64 random variates of the specified stream (the oracle's formulas)
__device__ __forceinline__ void normals4(const U4& r, double z[4]) {
  box_muller_d(r.x, r.y, z[0], z[1]);
  box_muller_d(r.z, r.w, z[2], z[3]);
}
// Gamma(alpha, 1), Marsaglia-Tsang, attempt k = Philox call k of the site (ci_oracle_gamma): the 64
// lanes evaluate attempts 0..63 at once, the first accepted one wins.
static __device__ __noinline__ double gamma_wave_d(double alpha, const Rng& g, uint32_t iter,
                                                   uint32_t site, uint32_t sub, int lane) {
  const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  const U4 r = site_call(g, iter, site, sub, (uint32_t)lane);
  double x, unused;
  box_muller_d(r.x, r.y, x, unused);
  const double t = 1.0 + c * x;
  const double v = t * t * t;
  bool ok = false;
  double gval = d;
  if (v > 0.0) {
    ok = log(u01d(r.z)) < 0.5 * x * x + d - d * v + d * log(v);
    gval = d * v;
    if (alpha < 1.0) gval *= pow(u01d(r.w), 1.0 / alpha);
  }
  const unsigned long long m = __ballot(ok);
  if (m == 0ull) return d;
  return readlane_d(gval, __ffsll((long long)m) - 1);
}
__device__ __forceinline__ double scale_draw_d(double conc, double scale, double ub, double n,
                                               double ss, const Rng& rng, uint32_t iter,
                                               uint32_t site, int lane) {
  const double g = gamma_wave_d(conc + 0.5 * n, rng, iter, site, 0, lane);
  const double s = sqrt((scale + 0.5 * ss) / g);
  return s < ub ? s : ub;
}

#ifndef CI_SEASONAL_DECL_ONLY
// ---- time-parallel pieces of the trend-only path (one wavefront, lane j owns Lc consecutive
// steps): every pass over time is  (i) summarise the own chunk as an element of a semigroup,
// (ii) scan the 64 elements across the lanes, (iii) replay the own chunk from the true incoming
// state.  Float64 throughout; the elements are those of ci_kernels.h's float32 scans for d = 2.
__device__ __forceinline__ double shfl_up_d(double v, int off) { return __shfl_up(v, (unsigned)off, 64); }
__device__ __forceinline__ double shfl_down_d(double v, int off) { return __shfl_down(v, (unsigned)off, 64); }

// x -> T^n x + c with T = [[1, 1], [0, 1]] (level += slope): the prior simulation and the
// reconstruction recursions.  then(a, b) = b after a.
struct Drift2 { double n, c0, c1; };
__device__ __forceinline__ Drift2 drift2_then(const Drift2& a, const Drift2& b) {
  Drift2 r;
  r.n = a.n + b.n;
  r.c0 = fma(b.n, a.c1, a.c0) + b.c0;
  r.c1 = a.c1 + b.c1;
  return r;
}
// exclusive forward scan: lane j receives e_0 then ... then e_{j-1} (lane 0: the identity)
__device__ __forceinline__ Drift2 drift2_scan_excl(Drift2 e, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    Drift2 o;
    o.n = shfl_up_d(e.n, off); o.c0 = shfl_up_d(e.c0, off); o.c1 = shfl_up_d(e.c1, off);
    if (lane >= off) e = drift2_then(o, e);
  }
  Drift2 x;
  x.n = shfl_up_d(e.n, 1); x.c0 = shfl_up_d(e.c0, 1); x.c1 = shfl_up_d(e.c1, 1);
  if (lane == 0) { x.n = 0.0; x.c0 = 0.0; x.c1 = 0.0; }
  return x;
}

// x -> M x + c, general 2 x 2 (the backward recursion r_{t-1} = L_t' r_t + z v_t / F_t)
struct Aff2 { double m00, m01, m10, m11, c0, c1; };
__device__ __forceinline__ Aff2 aff2_then(const Aff2& a, const Aff2& b) {
  Aff2 r;
  r.m00 = fma(b.m00, a.m00, b.m01 * a.m10); r.m01 = fma(b.m00, a.m01, b.m01 * a.m11);
  r.m10 = fma(b.m10, a.m00, b.m11 * a.m10); r.m11 = fma(b.m10, a.m01, b.m11 * a.m11);
  r.c0 = fma(b.m00, a.c0, fma(b.m01, a.c1, b.c0));
  r.c1 = fma(b.m10, a.c0, fma(b.m11, a.c1, b.c1));
  return r;
}
// exclusive BACKWARD scan: lane j receives e_63 then ... then e_{j+1} (lane 63: the identity)
__device__ __forceinline__ Aff2 aff2_scan_excl_bwd(Aff2 e, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    Aff2 o;
    o.m00 = shfl_down_d(e.m00, off); o.m01 = shfl_down_d(e.m01, off);
    o.m10 = shfl_down_d(e.m10, off); o.m11 = shfl_down_d(e.m11, off);
    o.c0 = shfl_down_d(e.c0, off); o.c1 = shfl_down_d(e.c1, off);
    if (lane + off < 64) e = aff2_then(o, e);
  }
  Aff2 x;
  x.m00 = shfl_down_d(e.m00, 1); x.m01 = shfl_down_d(e.m01, 1);
  x.m10 = shfl_down_d(e.m10, 1); x.m11 = shfl_down_d(e.m11, 1);
  x.c0 = shfl_down_d(e.c0, 1); x.c1 = shfl_down_d(e.c1, 1);
  if (lane == 63) { x.m00 = 1.0; x.m01 = 0.0; x.m10 = 0.0; x.m11 = 1.0; x.c0 = 0.0; x.c1 = 0.0; }
  return x;
}

// The Kalman filter over a stretch of steps as a map of the predicted moments (a, P):
//   a -> A (I + P J)^-1 (a + P eta) + b,   P -> A (I + P J)^-1 P A' + C      (C, J symmetric)
// (Sarkka & Garcia-Fernandez 2021, the element of ci_kernels.h's filter scan), d = 2.
struct Kf2 { double a00, a01, a10, a11, b0, b1, c00, c01, c11, e0, e1, j00, j01, j11; };
__device__ __forceinline__ Kf2 kf2_then(const Kf2& x, const Kf2& y) {
  // G = I + C1 J2,  W = G^-1
  const double g00 = 1.0 + fma(x.c00, y.j00, x.c01 * y.j01), g01 = fma(x.c00, y.j01, x.c01 * y.j11);
  const double g10 = fma(x.c01, y.j00, x.c11 * y.j01), g11 = 1.0 + fma(x.c01, y.j01, x.c11 * y.j11);
  const double rdet = 1.0 / fma(g00, g11, -(g01 * g10));
  const double w00 = g11 * rdet, w01 = -g01 * rdet, w10 = -g10 * rdet, w11 = g00 * rdet;
  // U = A2 W
  const double u00 = fma(y.a00, w00, y.a01 * w10), u01 = fma(y.a00, w01, y.a01 * w11);
  const double u10 = fma(y.a10, w00, y.a11 * w10), u11 = fma(y.a10, w01, y.a11 * w11);
  Kf2 r;
  r.a00 = fma(u00, x.a00, u01 * x.a10); r.a01 = fma(u00, x.a01, u01 * x.a11);
  r.a10 = fma(u10, x.a00, u11 * x.a10); r.a11 = fma(u10, x.a01, u11 * x.a11);
  // b = U (b1 + C1 eta2) + b2
  const double t0 = x.b0 + fma(x.c00, y.e0, x.c01 * y.e1), t1 = x.b1 + fma(x.c01, y.e0, x.c11 * y.e1);
  r.b0 = fma(u00, t0, fma(u01, t1, y.b0));
  r.b1 = fma(u10, t0, fma(u11, t1, y.b1));
  // S = W C1 (symmetric), C = A2 S A2' + C2
  const double s00 = fma(w00, x.c00, w01 * x.c01), s01 = fma(w00, x.c01, w01 * x.c11);
  const double s11 = fma(w10, x.c01, w11 * x.c11);
  const double q00 = fma(y.a00, s00, y.a01 * s01), q01 = fma(y.a00, s01, y.a01 * s11);   // A2 S
  const double q10 = fma(y.a10, s00, y.a11 * s01), q11 = fma(y.a10, s01, y.a11 * s11);
  r.c00 = fma(q00, y.a00, fma(q01, y.a01, y.c00));
  r.c01 = fma(q00, y.a10, fma(q01, y.a11, y.c01));
  r.c11 = fma(q10, y.a10, fma(q11, y.a11, y.c11));
  // eta = A1' W' (eta2 - J2 b1) + eta1
  const double d0 = y.e0 - fma(y.j00, x.b0, y.j01 * x.b1), d1 = y.e1 - fma(y.j01, x.b0, y.j11 * x.b1);
  const double v0 = fma(w00, d0, w10 * d1), v1 = fma(w01, d0, w11 * d1);                 // W' d
  r.e0 = fma(x.a00, v0, fma(x.a10, v1, x.e0));
  r.e1 = fma(x.a01, v0, fma(x.a11, v1, x.e1));
  // R = W' J2 (symmetric), J = A1' R A1 + J1
  const double r00 = fma(w00, y.j00, w10 * y.j01), r01 = fma(w00, y.j01, w10 * y.j11);
  const double r11 = fma(w01, y.j01, w11 * y.j11);
  const double p00 = fma(r00, x.a00, r01 * x.a10), p01 = fma(r00, x.a01, r01 * x.a11);   // R A1
  const double p10 = fma(r01, x.a00, r11 * x.a10), p11 = fma(r01, x.a01, r11 * x.a11);
  r.j00 = fma(x.a00, p00, fma(x.a10, p10, x.j00));
  r.j01 = fma(x.a00, p01, fma(x.a10, p11, x.j01));
  r.j11 = fma(x.a01, p01, fma(x.a11, p11, x.j11));
  return r;
}
// inclusive forward scan of the filter elements; only (b, C) of the result is used (lane 0's
// element starts from the prior, so every prefix has A = 0 and (b, C) = the predicted moments)
__device__ __forceinline__ Kf2 kf2_scan_incl(Kf2 e, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    Kf2 o;
    o.a00 = shfl_up_d(e.a00, off); o.a01 = shfl_up_d(e.a01, off); o.a10 = shfl_up_d(e.a10, off);
    o.a11 = shfl_up_d(e.a11, off); o.b0 = shfl_up_d(e.b0, off); o.b1 = shfl_up_d(e.b1, off);
    o.c00 = shfl_up_d(e.c00, off); o.c01 = shfl_up_d(e.c01, off); o.c11 = shfl_up_d(e.c11, off);
    o.e0 = shfl_up_d(e.e0, off); o.e1 = shfl_up_d(e.e1, off);
    o.j00 = shfl_up_d(e.j00, off); o.j01 = shfl_up_d(e.j01, off); o.j11 = shfl_up_d(e.j11, off);
    if (lane >= off) e = kf2_then(o, e);
  }
  return e;
}

template <bool GWS>
__global__ __launch_bounds__(64) void gibbs64_kernel(G64Args a) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smem64[];
  unsigned char* smem = smem64;
  const int lane = threadIdx.x;
  const K64& g = a.k;
  const int T = g.T, P = g.P, K = a.K;
  const int series = blockIdx.x / g.C, chain = blockIdx.x % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  const int trend = a.has_slope ? 2 : 1;
  // block geometry in registers: every loop over blocks is fully unrolled with static indices
  int off[SMAXK], nsz[SMAXK], roff[SMAXK];
  int D = trend;
  {
    int rr = trend;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) {
      off[k] = D; roff[k] = rr; nsz[k] = (k < K) ? a.nseas[k] : 0;
      if (k < K) { D += nsz[k]; rr += nsz[k] - 1; }
    }
  }
  const Layout64 L = make_layout64(T, P, K, D, a.dred, a.has_slope, GWS ? 1 : 0, a.reg_lds);
  // the arrays over time: LDS, or (GWS) this chain's slice of the HBM workspace -- every access
  // below is either one 16-byte row per 4 steps or lane-contiguous, and a chain only ever reads
  // what it wrote, so the slice stays in this XCD's L2
  unsigned char* wsc = a.ws + chain_lin * a.ws_stride;        // this chain's slice of the workspace
  unsigned char* tb_ = GWS ? wsc : smem;                       // arrays over time: workspace or LDS
  double* yv = (double*)(tb_ + L.yv); double* lev = (double*)(tb_ + L.lev);
  double* slp = (double*)(tb_ + L.slp); double* xw = (double*)(tb_ + L.xw);
  double* ytil = (double*)(tb_ + L.ytil); double* vf = (double*)(tb_ + L.vf);
  double* zl = (double*)(tb_ + L.zl); double* zs = (double*)(tb_ + L.zs);
  double* zo = (double*)(tb_ + L.zo); double* seas = (double*)(tb_ + L.seas);
  double* zk = (double*)(tb_ + L.zk); double* gd = (double*)(tb_ + L.gd);
  double* kf = (double*)(tb_ + L.kf);
  double* rs = (double*)(tb_ + L.rs); double* Pcur = (double*)(smem + L.Pa);
  double* Pnxt = (double*)(smem + L.Pb); double* pzv = (double*)(smem + L.pzv);
  double* zi = (double*)(smem + L.zi); double* x0r = (double*)(smem + L.x0r);
  double* egg = (double*)(smem + L.egg); double* d2 = (double*)(smem + L.d2);
  uint32_t* emeta = (uint32_t*)(smem + L.emeta);   // i | si<<6 | j<<12 | sj<<18 | bi<<24 | bj<<28
  uint8_t* msk = tb_ + L.mask; uint8_t* cbv = tb_ + L.cbits;
  const int TS = (T + 3) & ~3;       // padded length of every T-array (4-step blocks)
  RegLds R;
  R.bvec = (double*)(smem + L.bvec);
  R.w = nullptr;                        // (float weights of the float32 kernels: unused here)
  double* wv = (double*)(smem + L.w);   // the weights, float64
  R.xtx = const_cast<double*>(g.xtx) + (size_t)series * P * P;
  R.omega = const_cast<double*>(g.omega) + (size_t)series * P * P;
  bigp_point(R, a.reg_lds ? smem + L.reg : wsc + ((L.t_total + 255) & ~(size_t)255), P > 0 ? P : 1);

  const DevSeriesParams sp = g.sp[series];
  const DevSeasonalParams ss = a.ssp[series];
  Rng rng{stream_key0(g.seed0, g.series_stream_base, series), stream_key1(g.seed1, g.series_stream_base, series), (uint32_t)(g.chain_offset + chain)};
  const bool lat = a.lat_theta != nullptr;
  uint32_t itb = 0u;                          // iteration offset of the random stream
  const double* lth = nullptr;
  if (lat) {
    rng.chain = (uint32_t)(g.chain_offset + chain / a.lat_S);
    itb = (uint32_t)(chain % a.lat_S);
    lth = a.lat_theta + (size_t)chain * (3 + K + P);
  }
  const double* Xg = g.Xt + (size_t)series * P * T;
  const double* chol1 = a.p1_chol + (size_t)series * a.dred * a.dred;

  // ---- lane roles: component `lane` of the state, block membership, shift partners
  int blk = -1, pos = 0, nb = 1, boff = 0, rbase = 0;
#pragma unroll
  for (int k = 0; k < SMAXK; ++k)
    if (k < K && lane >= off[k] && lane < off[k] + nsz[k]) {
      blk = k; pos = lane - off[k]; nb = nsz[k]; boff = off[k]; rbase = roff[k];
    }
  const bool comp = lane < D;
  const bool isz = comp && (lane == 0 || (blk >= 0 && pos == 0));   // rows of Z
  const int fwd_src = blk >= 0 ? boff + (pos + 1 == nb ? 0 : pos + 1) : lane;   // x'_p = x_{p+1}
  const int bwd_src = blk >= 0 ? boff + (pos == 0 ? nb - 1 : pos - 1) : lane;   // (T'r)_p = r_{p-1}
  const double gpos = blk >= 0 ? ((pos == nb - 1) ? 1.f - 1.f / (double)nb : -1.f / (double)nb) : 0.f;
  const int blk0 = blk >= 0 ? blk : 0;

  // ---- stage constants
  for (int t = lane; t < TS; t += 64) {
    const bool in = t < T;
    const bool m = in ? g.mask[(size_t)series * T + t] != 0 : true;
    msk[t] = m ? 1 : 0;
    yv[t] = m ? 0.f : g.y[(size_t)series * T + t];
    lev[t] = 0.f; xw[t] = 0.f; ytil[t] = 0.f; vf[t] = 0.f; zl[t] = 0.f; zo[t] = 0.f;
    if (a.has_slope) { slp[t] = 0.f; zs[t] = 0.f; }
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        seas[k * TS + t] = 0.f; zk[k * TS + t] = 0.f; gd[k * TS + t] = 0.f;
        if (in && a.season_change[(size_t)k * T + t]) bits |= 1u << k;
      }
    cbv[t] = (uint8_t)bits;
  }
  // per-entry tables of the covariance time update P <- T P T' + Q
  for (int e = lane; e < D * D; e += 64) {
    const int i = e / D, j = e - i * D;
    int bi = 15, bj = 15, si = i, sj = j;
    double gi = 0.f, gj = 0.f;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        if (i >= off[k] && i < off[k] + nsz[k]) {
          bi = k; const int p = i - off[k]; si = off[k] + (p + 1 == nsz[k] ? 0 : p + 1);
          gi = (p == nsz[k] - 1) ? 1.f - 1.f / (double)nsz[k] : -1.f / (double)nsz[k];
        }
        if (j >= off[k] && j < off[k] + nsz[k]) {
          bj = k; const int p = j - off[k]; sj = off[k] + (p + 1 == nsz[k] ? 0 : p + 1);
          gj = (p == nsz[k] - 1) ? 1.f - 1.f / (double)nsz[k] : -1.f / (double)nsz[k];
        }
      }
    egg[e] = (bi == bj && bi != 15) ? gi * gj : 0.f;
    emeta[e] = (uint32_t)i | ((uint32_t)si << 6) | ((uint32_t)j << 12) | ((uint32_t)sj << 18) |
               ((uint32_t)bi << 24) | ((uint32_t)bj << 28);
  }
  for (int j = lane; j < (P > 16 ? P : 16); j += 64) wv[j] = 0.f;
  wave_sync();
  double n_changes[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) {
    n_changes[k] = 0.0;
    if (k < K) {
      double c = 0.f;
      for (int t = lane; t + 1 < T; t += 64) c += ((cbv[t] >> k) & 1) ? 1.f : 0.f;
      n_changes[k] = (double)wave_sum_d(c);
    }
  }

  double obs_scale = sp.obs_scale0, level_scale = sp.level_scale0, slope_scale = sp.slope_scale0;
  double drift[SMAXK];
#pragma unroll
  for (int k = 0; k < SMAXK; ++k) drift[k] = (k < K) ? ss.drift_scale0[k] : 0.0;
  double ssl = 0.f, sss = 0.f, ssd = 0.f;   // lane 0 / lane 1 / lane off[k] accumulate
  const double p1l = (double)(sp.init_level_scale * sp.init_level_scale);
  const double p1s = (double)(sp.init_slope_scale * sp.init_slope_scale);
  const double p1e = (double)(ss.init_seasonal_scale * ss.init_seasonal_scale);
  Prof prof;
  prof.start(g.prof, g.prof != nullptr && blockIdx.x == 0 && lane == 0);

  auto zsum = [&](double x) -> double {   // Z x for a lane-distributed vector
    double s = readlane_d(x, 0);
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) s += readlane_d(x, off[k]);
    return s;
  };
  // x <- T_t x : cyclic shift of the blocks that change season at t (+ level += slope)
  auto transition = [&](double x, unsigned cb) -> double {
    const double sh = __shfl(x, fwd_src, 64);
    double r = (blk >= 0 && ((cb >> blk) & 1u)) ? sh : x;
    if (a.has_slope) {
      const double s1 = readlane_d(x, 1);
      if (lane == 0) r += s1;
    }
    return r;
  };
  auto transition_T = [&](double x, unsigned cb) -> double {   // x <- T_t' x
    const double sh = __shfl(x, bwd_src, 64);
    double r = (blk >= 0 && ((cb >> blk) & 1u)) ? sh : x;
    if (a.has_slope) {
      const double r0 = readlane_d(x, 0);
      if (lane == 1) r += r0;
    }
    return r;
  };
  auto ld4 = [](const double* p) { return *reinterpret_cast<const double4*>(p); };
  auto ldb4 = [](const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); };
  auto at4 = [](const double4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };

  const int n_iter = g.W + g.S;
  if (lat) {
    obs_scale = lth[0]; level_scale = lth[1]; slope_scale = lth[2];
#pragma unroll
    for (int k = 0; k < SMAXK; ++k) if (k < K) drift[k] = lth[3 + k];
    for (int j = lane; j < P; j += 64) wv[j] = (double)lth[3 + K + j];
    wave_sync();
  }
  for (int it = 0; it <= n_iter; ++it) {
    // ---- (1) X~'targets, y'y from the current latents
    if (!lat) {
      double yty = 0.f;
      for (int t = lane; t < T; t += 64) {
        double tg = 0.f;
        if (!msk[t]) {
          tg = yv[t] - lev[t];
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K) tg -= seas[k * TS + t];
        }
        ytil[t] = tg;            // reused as the targets buffer here
        yty = fma(tg, tg, yty);
      }
      wave_sync();
      for (int j = 0; j < P; ++j) {
        double pj = 0.f;
        for (int t = lane; t < T; t += 64) pj = fma(Xg[(size_t)j * T + t], ytil[t], pj);
        const double s = wave_sum_d(pj);
        if (lane == 0) R.bvec[j] = (double)s;
      }
      const double s0 = wave_sum_d(yty);
      if (lane == 0) R.bvec[P] = (double)s0;
      wave_sync();
    }
    prof.tick(20);
    // ---- (2) scale draws of iteration it-1, regression draw of iteration it
    double emit_obs = obs_scale;
    if (it > 0) {
      const uint32_t pit = (uint32_t)(it - 1) + itb;
      if (!lat) {
      const double v_l = (double)readlane_d(ssl, 0);
      level_scale = scale_draw_d(sp.level_conc, sp.level_scale, sp.level_ub, (double)(T - 1), v_l,
                               rng, pit, SITE_LEVEL_SCALE, lane);
      if (a.has_slope) {
        const double v_s = (double)readlane_d(sss, 1);
        slope_scale = scale_draw_d(sp.slope_conc, sp.slope_scale, sp.slope_ub, (double)(T - 1), v_s,
                                 rng, pit, SITE_SLOPE_SCALE, lane);
      }
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          const double v_d = (double)readlane_d(ssd, off[k]);
          const double gk = gamma_wave_d(ss.drift_conc + 0.5 * n_changes[k], rng, pit,
                                       SITE_DRIFT_SCALE, (uint32_t)k, lane);
          const double sd = (double)sqrt(((ss.drift_scale + 0.5 * v_d) * (1.0 / gk)));
          drift[k] = sd < ss.drift_ub ? sd : ss.drift_ub;
        }
      if (P == 0)
        obs_scale = scale_draw_d(sp.obs_conc, sp.obs_scale, sp.obs_ub, sp.n_obs, R.bvec[P], rng, pit,
                               SITE_OBS_SCALE, lane);
      }
      emit_obs = obs_scale;
      const int s = it - 1 - g.W;
      if (s >= 0) {
        const size_t o = chain_lin * g.S + s;
        if (lane == 0) {
          if (g.out_obs) g.out_obs[o] = (double)obs_scale;
          if (g.out_level_scale) g.out_level_scale[o] = (double)level_scale;
          if (g.out_slope_scale) g.out_slope_scale[o] = (double)(a.has_slope ? slope_scale : 0.0);
        }
        if (a.out_drift) {
#pragma unroll
          for (int k = 0; k < SMAXK; ++k)
            if (k < K && lane == k) a.out_drift[o * K + k] = (double)drift[k];
        }
        if (g.out_weights)
          for (int j = lane; j < P; j += 64) g.out_weights[o * P + j] = wv[j];
        // level / seasonal contributions / posterior-predictive trajectory of iteration it-1
        const double so = (double)emit_obs;
        const size_t row = o * T;
        for (int c = lane; c < (T + 3) / 4; c += 64) {
          double zp[4];
          normals4(site_call(rng, pit, SITE_PRED, 0, (uint32_t)c), zp);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * c + q;
            if (t < T) {
              double loc = lev[t] + xw[t];
#pragma unroll
              for (int k = 0; k < SMAXK; ++k)
                if (k < K) {
                  const double sv = seas[k * TS + t];
                  loc += sv;
                  if (a.out_seasonal) a.out_seasonal[(row + t) * K + k] = sv;
                }
              if (g.out_level) g.out_level[row + t] = lev[t];
              if (g.out_slope && a.has_slope) g.out_slope[row + t] = slp[t];
              if (g.out_traj) g.out_traj[row + t] = fma(so, zp[q], loc);
              if (g.out_pred_mean) {
                double* pm = g.out_pred_mean + chain_lin * T + t;   // running sum, scaled at the end
                *pm = (s == 0 ? 0.f : *pm) + loc;
              }
            }
          }
        }
      }
    }
    if (it == n_iter) break;
    if (P > 0 && !lat) {
      const double g_obs = gamma_wave_d(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
      obs_scale = spike_slab_draw_big(R, wv, P, sp, obs_scale, g_obs, rng, (uint32_t)it, lane, it == 0);
    }
    wave_sync();
    prof.tick(21);

    // ---- (3) residual, normals of this iteration
    for (int t = lane; t < T; t += 64) {
      double s = 0.f;
      for (int j = 0; j < P; ++j) s = fma(Xg[(size_t)j * T + t], wv[j], s);
      xw[t] = s;
    }
    for (int c = lane; c < (T + 3) / 4; c += 64) {
      double z4[4];
      normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_LEVEL, 0, (uint32_t)c), z4);
      *reinterpret_cast<double4*>(zl + 4 * c) = make_double4(z4[0], z4[1], z4[2], z4[3]);
      normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_OBS, 0, (uint32_t)c), z4);
      *reinterpret_cast<double4*>(zo + 4 * c) = make_double4(z4[0], z4[1], z4[2], z4[3]);
      if (a.has_slope) {
        normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_SLOPE, 0, (uint32_t)c), z4);
        *reinterpret_cast<double4*>(zs + 4 * c) = make_double4(z4[0], z4[1], z4[2], z4[3]);
      }
#pragma unroll
      for (int k = 0; k < SMAXK; ++k)
        if (k < K) {
          normals4(site_call(rng, (uint32_t)it + itb, SITE_PRIOR_SEAS, (uint32_t)k, (uint32_t)c), z4);
          *reinterpret_cast<double4*>(zk + k * TS + 4 * c) = make_double4(z4[0], z4[1], z4[2], z4[3]);
        }
    }
    if (lane < a.dred) {
      zi[lane] = normal_d(rng, (uint32_t)it + itb, SITE_PRIOR_INIT, 0, (uint32_t)lane);
    }
    double mydrift = 0.f;
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        if (blk == k) mydrift = (double)drift[k];
        if (lane == 0) d2[k] = (double)(drift[k] * drift[k]);
      }
    wave_sync();
    // x+_0 = chol(P_1) z in the oracle's reduced coordinates, folded into the prior mean
    if (lane < a.dred) {
      double s = 0.f;
      for (int j = 0; j <= lane; ++j) s = fma(chol1[lane * a.dred + j], zi[j], s);
      x0r[lane] = s;
    }
    wave_sync();
    double a1e = 0.f;
    if (lane == 0) a1e = (double)sp.init_level_loc + x0r[0];
    if (a.has_slope && lane == 1) a1e = x0r[1];
    if (blk >= 0) {
      if (pos < nb - 1) a1e = x0r[rbase + pos];
      else { double s = 0.f; for (int q = 0; q < nb - 1; ++q) s += x0r[rbase + q]; a1e = -s; }
    }
    const double so = (double)obs_scale, sl = (double)level_scale, ssc = (double)slope_scale;
    const double H = so * so, ql = sl * sl, qs = ssc * ssc;
    const double* zkb = zk + blk0 * TS;
    const double dg = mydrift * gpos;            // this lane's share of a unit drift shock
    prof.tick(22);

    if (K == 0) {
      // ---- trend-only models (the reference's default: LocalLevel, D = 1; LocalLinearTrend, D = 2),
      // TIME-PARALLEL within the wavefront: lane j owns the steps [t0, t1) (an ODD number of
      // steps per lane, so that the lanes' strided accesses spread over the LDS banks) and every
      // pass is chunk summary -> scan across lanes -> replay (helpers above).  LocalLevel runs
      // the same 2 x 2 code with an identically-zero slope (0 variance, 0 noise: exact).
      const bool s2 = a.has_slope != 0;
      const double a0 = readlane_d(a1e, 0), a1s = s2 ? readlane_d(a1e, 1) : 0.0;
      const double qs2 = s2 ? qs : 0.0, ss2 = s2 ? ssc : 0.0, p1s2 = s2 ? p1s : 0.0;
      const int Lc = ((T + 63) / 64) | 1;
      const int t0 = lane * Lc < T ? lane * Lc : T;
      const int t1 = t0 + Lc < T ? t0 + Lc : T;
      double xin0, xin1;                       // x+ at t0 (kept for pass 3)
      {   // pass 0: x+ and y~ = resid - y+
        Drift2 e; e.n = 0.0; e.c0 = 0.0; e.c1 = 0.0;
        for (int t = t0; t < t1; ++t)
          if (t + 1 < T) {
            e.n += 1.0;
            e.c0 = fma(sl, zl[t], e.c0 + e.c1);
            if (s2) e.c1 = fma(ss2, zs[t], e.c1);
          }
        const Drift2 in = drift2_scan_excl(e, lane);
        xin0 = in.c0; xin1 = in.c1;            // x+_0 = 0
        double x0 = xin0, x1 = xin1;
        for (int t = t0; t < t1; ++t) {
          ytil[t] = (yv[t] - xw[t]) - fma(so, zo[t], x0);
          if (t + 1 < T) {
            x0 = fma(sl, zl[t], x0 + x1);
            if (s2) x1 = fma(ss2, zs[t], x1);
          }
        }
      }
      {   // pass 1: Kalman filter, gains and scaled innovations
        Kf2 e;
        e.a00 = 1.0; e.a01 = 0.0; e.a10 = 0.0; e.a11 = 1.0; e.b0 = 0.0; e.b1 = 0.0;
        e.c00 = 0.0; e.c01 = 0.0; e.c11 = 0.0; e.e0 = 0.0; e.e1 = 0.0; e.j00 = 0.0; e.j01 = 0.0; e.j11 = 0.0;
        if (lane == 0) { e.a00 = 0.0; e.a11 = 0.0; e.b0 = a0; e.b1 = a1s; e.c00 = p1l; e.c11 = p1s2; }
        // the own chunk: a Kalman sweep on (b, C) that also carries A, eta, J -- appending one
        // step (J2 = z z'/H, eta2 = z y/H, A2 = T, C2 = Q) is a rank-one update of each
        for (int t = t0; t < t1; ++t) {
          if (msk[t] == 0) {
            const double rf = 1.0 / (e.c00 + H);
            const double v = ytil[t] - e.b0;
            const double k0 = e.c00 * rf, k1 = e.c01 * rf;
            const double vr = v * rf;
            e.e0 = fma(e.a00, vr, e.e0); e.e1 = fma(e.a01, vr, e.e1);          // eta += A'z v / f
            e.j00 = fma(e.a00 * rf, e.a00, e.j00); e.j01 = fma(e.a00 * rf, e.a01, e.j01);
            e.j11 = fma(e.a01 * rf, e.a01, e.j11);                              // J += A'z z'A / f
            e.b0 = fma(k0, v, e.b0); e.b1 = fma(k1, v, e.b1);
            const double r00 = e.a00, r01 = e.a01;                              // A -= k (z'A)
            e.a10 = fma(-k1, r00, e.a10); e.a11 = fma(-k1, r01, e.a11);
            e.a00 = fma(-k0, r00, e.a00); e.a01 = fma(-k0, r01, e.a01);
            const double c00 = e.c00, c01 = e.c01;
            e.c00 -= c00 * c00 * rf; e.c01 -= c00 * c01 * rf; e.c11 -= c01 * c01 * rf;
          }
          // time update with T = [[1, 1], [0, 1]], Q = diag(ql, qs)
          e.a00 += e.a10; e.a01 += e.a11;
          e.b0 += e.b1;
          e.c00 = e.c00 + 2.0 * e.c01 + e.c11 + ql;
          e.c01 = e.c01 + e.c11;
          e.c11 = e.c11 + qs2;
        }
        const Kf2 pre = kf2_scan_incl(e, lane);
        double m0 = shfl_up_d(pre.b0, 1), m1 = shfl_up_d(pre.b1, 1);
        double p00 = shfl_up_d(pre.c00, 1), p01 = shfl_up_d(pre.c01, 1), p11 = shfl_up_d(pre.c11, 1);
        if (lane == 0) { m0 = a0; m1 = a1s; p00 = p1l; p01 = 0.0; p11 = p1s2; }
        for (int t = t0; t < t1; ++t) {
          double k0 = 0.0, k1 = 0.0, vfq = 0.0;
          if (msk[t] == 0) {
            const double rF = 1.0 / (p00 + H);
            const double v = ytil[t] - m0;
            k0 = p00 * rF; k1 = p01 * rF;
            vfq = v * rF;
            m0 = fma(k0, v, m0); m1 = fma(k1, v, m1);
            const double q00 = p00, q01 = p01;
            p00 -= q00 * q00 * rF; p01 -= q00 * q01 * rF; p11 -= q01 * q01 * rF;
          }
          kf[(size_t)t * D] = k0;
          if (s2) kf[(size_t)t * D + 1] = k1;
          vf[t] = vfq;
          if (t + 1 < T) {
            m0 += m1;
            p00 = p00 + 2.0 * p01 + p11 + ql;
            p01 = p01 + p11;
            p11 = p11 + qs2;
          }
        }
      }
      {   // pass 2: backward recursion, rs[t] = r_{t-1}
        auto step = [&](int t, double& r0, double& r1, double k0, double k1, double vft, bool obs) {
          if (t + 1 < T) r1 += r0;                      // r <- T' r
          if (obs) r0 += vft - fma(k0, r0, k1 * r1);    // + z (v/F - K'r)
        };
        Aff2 e; e.m00 = 1.0; e.m01 = 0.0; e.m10 = 0.0; e.m11 = 1.0; e.c0 = 0.0; e.c1 = 0.0;
        for (int t = t1 - 1; t >= t0; --t) {
          const bool obs = msk[t] == 0;
          const double k0 = kf[(size_t)t * D], k1 = s2 ? kf[(size_t)t * D + 1] : 0.0, vft = vf[t];
          // the step applied to the three columns (M e_0, M e_1, c) of the element so far
          double c0 = e.c0, c1 = e.c1, x0 = e.m00, x1 = e.m10, y0 = e.m01, y1 = e.m11;
          step(t, c0, c1, k0, k1, vft, obs);
          step(t, x0, x1, k0, k1, 0.0, obs);
          step(t, y0, y1, k0, k1, 0.0, obs);
          e.c0 = c0; e.c1 = c1; e.m00 = x0; e.m10 = x1; e.m01 = y0; e.m11 = y1;
        }
        const Aff2 in = aff2_scan_excl_bwd(e, lane);
        double r0 = in.c0, r1 = in.c1;                  // r_{T-1} = 0 at the right end
        for (int t = t1 - 1; t >= t0; --t) {
          const double k0 = kf[(size_t)t * D], k1 = s2 ? kf[(size_t)t * D + 1] : 0.0;
          step(t, r0, r1, k0, k1, vf[t], msk[t] == 0);
          rs[(size_t)t * D] = r0;
          if (s2) rs[(size_t)t * D + 1] = r1;
        }
      }
      wave_sync();
      {   // pass 3: x^ forward, x+ again, the draw and its increment statistics
        const double hi0 = fma(p1l, rs[0], a0), hi1 = s2 ? fma(p1s2, rs[1], a1s) : 0.0;
        Drift2 e; e.n = 0.0; e.c0 = 0.0; e.c1 = 0.0;
        for (int t = t0; t < t1; ++t)
          if (t + 1 < T) {
            e.n += 1.0;
            e.c0 = fma(ql, rs[(size_t)(t + 1) * D], e.c0 + e.c1);
            if (s2) e.c1 = fma(qs2, rs[(size_t)(t + 1) * D + 1], e.c1);
          }
        const Drift2 in = drift2_scan_excl(e, lane);
        double h0 = fma(in.n, hi1, hi0) + in.c0, h1 = hi1 + in.c1;
        double x0 = xin0, x1 = xin1, al = 0.0, as = 0.0;
        for (int t = t0; t < t1; ++t) {
          const double xt0 = h0 + x0, xt1 = h1 + x1;
          lev[t] = xt0;
          if (s2) slp[t] = xt1;
          if (t + 1 < T) {
            h0 = fma(ql, rs[(size_t)(t + 1) * D], h0 + h1);
            x0 = fma(sl, zl[t], x0 + x1);
            if (s2) {
              h1 = fma(qs2, rs[(size_t)(t + 1) * D + 1], h1);
              x1 = fma(ss2, zs[t], x1);
            }
            const double nx0 = h0 + x0, nx1 = h1 + x1;       // the increment t -> t + 1
            const double dl = (nx0 - xt0) - xt1;
            al = fma(dl, dl, al);
            if (s2) { const double ds = nx1 - xt1; as = fma(ds, ds, as); }
          }
        }
        ssl = wave_sum_d(al); sss = wave_sum_d(as); ssd = 0.0;
      }
      wave_sync();
      prof.tick(27);
      continue;
    }

    // ---- (4) pass 0: simulate x+ (zero initial state) and form y~ = resid - y+.
    // Every pass walks time in blocks of 4 steps so that the per-step scalars arrive as one
    // batch of 16-byte LDS loads instead of one exposed round trip each.
    {
      double xp = 0.f;
      for (int t4 = 0; t4 < T; t4 += 4) {
        const double4 zo4 = ld4(zo + t4), zl4 = ld4(zl + t4), zk4 = ld4(zkb + t4);
        const double4 yv4 = ld4(yv + t4), xw4 = ld4(xw + t4);
        double4 zs4 = make_double4(0.f, 0.f, 0.f, 0.f);
        if (a.has_slope) zs4 = ld4(zs + t4);
        const uint32_t cb4 = ldb4(cbv + t4);
        double yt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          const double zx = zsum(xp);
          yt[q] = (at4(yv4, q) - at4(xw4, q)) - (zx + so * at4(zo4, q));
          if (t + 1 < T) {
            const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
            double r = transition(xp, cb);
            if (lane == 0) r = fma(sl, at4(zl4, q), r);
            if (a.has_slope && lane == 1) r = fma(ssc, at4(zs4, q), r);
            if (blk >= 0 && ((cb >> blk) & 1u)) r = fma(dg, at4(zk4, q), r);
            xp = r;
          }
        }
        if (lane == 0) *reinterpret_cast<double4*>(ytil + t4) = make_double4(yt[0], yt[1], yt[2], yt[3]);
      }
    }
    prof.tick(23);
    // prior covariance of x_0 in full-effect form: sd^2 (I - 11'/n) per block
    for (int e = lane; e < D * D; e += 64) {
      const uint32_t mt = emeta[e];
      const int i = mt & 63u, j = (mt >> 12) & 63u;
      const unsigned bi = (mt >> 24) & 15u, bj = mt >> 28;
      double v = 0.f;
      if (e == 0) v = p1l;
      else if (a.has_slope && i == 1 && j == 1) v = p1s;
      else if (bi != 15u && bi == bj) {
        int nn = 1;
#pragma unroll
        for (int k = 0; k < SMAXK; ++k) if (k < K && bi == (unsigned)k) nn = nsz[k];
        v = p1e * ((i == j ? 1.f : 0.f) - 1.f / (double)nn);
      }
      Pcur[e] = v;
    }
    wave_sync();
    prof.tick(24);

    // ---- (5) pass 1: Kalman filter, storing K_t and v_t / F_t.  The measurement update and the
    // time update of the covariance are ONE sweep over the D x D entries:
    //   P'[i][j] = P[si][sj] - pz[si] pz[sj] / F + Q[i][j]      (si, sj: sources under the shifts)
    // with the per-entry table lookups issued before the dependent loads.
    {
      double am = a1e;
      const int DD = D * D;
      uint32_t mt0[4];          // the first 256 entries' tables stay in registers
      double gq0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        mt0[u] = (e < DD) ? emeta[e] : 0u;
        gq0[u] = (e < DD) ? egg[e] : 0.f;
      }
      for (int t4 = 0; t4 < T; t4 += 4) {
        const double4 yt4 = ld4(ytil + t4);
        const uint32_t cb4 = ldb4(cbv + t4), mk4 = ldb4(msk + t4);
        double vfq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          vfq[q] = 0.f;
          if (t >= T) continue;
          const bool obs = ((mk4 >> (8 * q)) & 0xFFu) == 0u;
          const unsigned cb = (t + 1 < T) ? ((cb4 >> (8 * q)) & 0xFFu) : 0u;
          double kfi = 0.f, rF = 0.f;
          if (obs) {
            double pz = 0.f;
            if (comp) {
              pz = Pcur[lane * D];
#pragma unroll
              for (int k = 0; k < SMAXK; ++k)
                if (k < K) pz += Pcur[lane * D + off[k]];
              pzv[lane] = pz;
            }
            const double F = zsum(pz) + H;
            rF = 1.0 / F;
            const double v = at4(yt4, q) - zsum(am);
            kfi = pz * rF;
            vfq[q] = v * rF;
            am = fma(kfi, v, am);
          } else if (comp) {
            pzv[lane] = 0.f;
          }
          if (comp) kf[(size_t)t * D + lane] = kfi;
          if (t + 1 == T) continue;
          am = transition(am, cb);
          wave_sync();
          if (!obs && cb == 0u && !a.has_slope) {
            if (lane == 0) Pcur[0] += ql;
            wave_sync();
            continue;
          }
          for (int e0 = lane; e0 < DD; e0 += 64 * 4) {
            uint32_t mt[4];
            double gq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int e = e0 + 64 * u;
              if (e0 == lane) { mt[u] = mt0[u]; gq[u] = gq0[u]; }
              else {
                mt[u] = (e < DD) ? emeta[e] : 0u;
                gq[u] = (e < DD) ? egg[e] : 0.f;
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int e = e0 + 64 * u;
              if (e < DD) {
                const unsigned bi = (mt[u] >> 24) & 15u, bj = mt[u] >> 28;
                const bool ci = bi != 15u && ((cb >> bi) & 1u), cj = bj != 15u && ((cb >> bj) & 1u);
                const int i = mt[u] & 63u, j = (mt[u] >> 12) & 63u;
                const int si = ci ? (int)((mt[u] >> 6) & 63u) : i;
                const int sj = cj ? (int)((mt[u] >> 18) & 63u) : j;
                double v = Pcur[__mul24(si, D) + sj] - pzv[si] * pzv[sj] * rF;
                if (a.has_slope) {       // level <- level + slope
                  if (i == 0) v += Pcur[D + sj] - pzv[1] * pzv[sj] * rF;
                  if (j == 0) v += Pcur[__mul24(si, D) + 1] - pzv[si] * pzv[1] * rF;
                  if (i == 0 && j == 0) v += Pcur[D + 1] - pzv[1] * pzv[1] * rF;
                  if (i == 1 && j == 1) v += qs;
                }
                if (e == 0) v += ql;
                if (ci && bi == bj) v = fma(d2[bi], gq[u], v);
                Pnxt[e] = v;
              }
            }
          }
          wave_sync();
          double* tmp = Pcur; Pcur = Pnxt; Pnxt = tmp;
        }
        if (lane == 0) *reinterpret_cast<double4*>(vf + t4) = make_double4(vfq[0], vfq[1], vfq[2], vfq[3]);
      }
    }
    wave_sync();
    prof.tick(25);
    // ---- (6) pass 2: backward recursion, rs[t] = r_{t-1}; then gd[k][t] = g . r_{t-1} per block
    // (the projection the forward reconstruction needs at season changes)
    {
      double r = 0.f;
      for (int t4 = ((T - 1) & ~3); t4 >= 0; t4 -= 4) {
        const double4 vf4 = ld4(vf + t4);
        const uint32_t cb4 = ldb4(cbv + t4), mk4 = ldb4(msk + t4);
        double kfq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kfq[q] = (comp && t4 + q < T) ? kf[(size_t)(t4 + q) * D + lane] : 0.f;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
          const int t = t4 + q;
          if (t >= T) continue;
          r = (t + 1 < T) ? transition_T(r, (cb4 >> (8 * q)) & 0xFFu) : 0.f;
          if (((mk4 >> (8 * q)) & 0xFFu) == 0u) {
            const double kr = wave_sum_d(kfq[q] * r);
            if (isz) r += at4(vf4, q) - kr;
          }
          if (comp) rs[(size_t)t * D + lane] = r;
        }
      }
    }
    wave_sync();
    // g . r_{t-1} per block, time-parallel: g = e_last - 1/n
#pragma unroll
    for (int k = 0; k < SMAXK; ++k)
      if (k < K) {
        const double rn = 1.0 / (double)nsz[k];
        for (int t = lane; t < T; t += 64) {
          const double* rr = rs + (size_t)t * D + off[k];
          double sb = 0.f;
          for (int q = 0; q < nsz[k]; ++q) sb += rr[q];
          gd[k * TS + t] = rr[nsz[k] - 1] - sb * rn;
        }
      }
    wave_sync();
    prof.tick(26);
    // ---- (7) pass 3: reconstruct x^ forward, re-simulate x+, write the draw, gather statistics
    {
      double xh = a1e;
      {   // x^_0 = a_1 + P_1 r_{-1}
        const double r0 = comp ? rs[lane] : 0.f;
        if (lane == 0) xh += p1l * r0;
        if (a.has_slope && lane == 1) xh += p1s * r0;
        if (blk >= 0) {
          double sb = 0.f;
          for (int q = 0; q < nb; ++q) sb += rs[boff + q];
          xh += p1e * (r0 - sb / (double)nb);
        }
      }
      double xp = 0.f, prev = 0.f, prev_next = 0.f;
      ssl = 0.f; sss = 0.f; ssd = 0.f;
      unsigned cb_prev = 0u;
      const double* gdb = gd + blk0 * TS;
      const double dgd = mydrift * dg;            // sigma_d^2 g_i
      for (int t4 = 0; t4 < T; t4 += 4) {
        const double4 zl4 = ld4(zl + t4), zk4 = ld4(zkb + t4);
        double4 zs4 = make_double4(0.f, 0.f, 0.f, 0.f);
        if (a.has_slope) zs4 = ld4(zs + t4);
        const uint32_t cb4 = ldb4(cbv + t4);
        double rnq[4], gdq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // r_t = rs[t + 1] and g . r_t of this lane's block
          const int t1 = t4 + q + 1;
          rnq[q] = (comp && t1 < T) ? rs[(size_t)t1 * D + lane] : 0.f;
          gdq[q] = (t1 < T) ? gdb[t1] : 0.f;
        }
        double xo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          xo[q] = 0.f;
          if (t >= T) continue;
          const double xt = xh + xp;
          xo[q] = xt;
          if (t > 0) {
            if (lane == 0) {
              double dl = xt - prev;
              if (a.has_slope) dl -= prev_next;       // slope_{t-1} is lane 1 = "next" of lane 0
              ssl = fma(dl, dl, ssl);
            }
            if (a.has_slope && lane == 1) { const double ds = xt - prev; sss = fma(ds, ds, sss); }
            if (blk >= 0 && pos == 0 && ((cb_prev >> blk) & 1u)) {
              const double w = (double)nb * (prev_next - xt);   // n (e_{t-1,1} - e_{t,0})
              ssd = fma(w, w, ssd);
            }
          }
          prev = xt;
          prev_next = __shfl_down(xt, 1, 64);
          if (t + 1 < T) {
            const unsigned cb = (cb4 >> (8 * q)) & 0xFFu;
            const bool mych = blk >= 0 && ((cb >> blk) & 1u);
            double h = transition(xh, cb);
            if (lane == 0) h = fma(ql, rnq[q], h);
            if (a.has_slope && lane == 1) h = fma(qs, rnq[q], h);
            if (mych) h = fma(dgd, gdq[q], h);
            xh = h;
            double r = transition(xp, cb);
            if (lane == 0) r = fma(sl, at4(zl4, q), r);
            if (a.has_slope && lane == 1) r = fma(ssc, at4(zs4, q), r);
            if (mych) r = fma(dg, at4(zk4, q), r);
            xp = r;
            cb_prev = cb;
          }
        }
        // the observed components of the draw, 4 steps at a time
        if (lane == 0) *reinterpret_cast<double4*>(lev + t4) = make_double4(xo[0], xo[1], xo[2], xo[3]);
        if (a.has_slope && lane == 1)
          *reinterpret_cast<double4*>(slp + t4) = make_double4(xo[0], xo[1], xo[2], xo[3]);
        if (blk >= 0 && pos == 0)
          *reinterpret_cast<double4*>(seas + blk * TS + t4) = make_double4(xo[0], xo[1], xo[2], xo[3]);
      }
    }
    wave_sync();
    prof.tick(27);
  }
  if (g.out_pred_mean) {
    const double inv = 1.0 / (double)(g.S > 0 ? g.S : 1);
    for (int t = lane; t < T; t += 64) g.out_pred_mean[chain_lin * T + t] *= inv;
  }
}

// ------------------------------------------------------------------------------------
// Trend-only models (LocalLevel / LocalLinearTrend + regression: the reference's default) in
// float64 on EIGHT wavefronts per chain.  gibbs64_kernel runs a chain on one wavefront, i.e. on
// one of the CU's four SIMDs, and float64 transcendentals (Box-Muller, Marsaglia-Tsang) make it
// instruction-issue bound (119 us per iteration at T = 1000).  Here the same arithmetic is spread
// over 512 threads:
//   * X~'targets: one feature per wave over the whole series (the summation order of the
//     one-wave kernel: the sums are bit-identical to its sums);
//   * wave 0 draws (sigma^2_obs, weights) -- spike_slab_draw_big, unchanged -- WHILE wave 1 and
//     wave 2 draw the level / slope scales and waves 1-7 generate the float64 normals of this
//     iteration's Durbin-Koopman draw and emit the previous iteration's trajectory;
//   * every pass over time is chunk summary -> scan -> replay as in the one-wave kernel, the
//     scan now wave scan + a short carry across the eight wave totals ("apply" form: only the
//     moments / the adjoint vector cross waves).
// Same random stream, same layout (make_layout64 with K = 0), same outputs; sums over time are
// taken in a different order, so results equal the one-wave kernel's and the oracle's to round-off
// (tests/test_gpu_float64.py: 1e-8).
// ------------------------------------------------------------------------------------
constexpr int NW64 = 8, NT64 = NW64 * 64;

// predicted moments (m, P) pushed through a Kalman element: m -> A (I + P J)^-1 (m + P eta) + b,
// P -> A (I + P J)^-1 P A' + C
__device__ __forceinline__ void kf2_apply(const Kf2& y, double& m0, double& m1, double& p00, double& p01,
                                          double& p11) {
  const double g00 = 1.0 + fma(p00, y.j00, p01 * y.j01), g01 = fma(p00, y.j01, p01 * y.j11);
  const double g10 = fma(p01, y.j00, p11 * y.j01), g11 = 1.0 + fma(p01, y.j01, p11 * y.j11);
  const double rdet = 1.0 / fma(g00, g11, -(g01 * g10));
  const double w00 = g11 * rdet, w01 = -g01 * rdet, w10 = -g10 * rdet, w11 = g00 * rdet;
  const double u00 = fma(y.a00, w00, y.a01 * w10), u01 = fma(y.a00, w01, y.a01 * w11);
  const double u10 = fma(y.a10, w00, y.a11 * w10), u11 = fma(y.a10, w01, y.a11 * w11);
  const double t0 = m0 + fma(p00, y.e0, p01 * y.e1), t1 = m1 + fma(p01, y.e0, p11 * y.e1);
  const double s00 = fma(w00, p00, w01 * p01), s01 = fma(w00, p01, w01 * p11);
  const double s11 = fma(w10, p01, w11 * p11);
  const double q00 = fma(y.a00, s00, y.a01 * s01), q01 = fma(y.a00, s01, y.a01 * s11);
  const double q10 = fma(y.a10, s00, y.a11 * s01), q11 = fma(y.a10, s01, y.a11 * s11);
  m0 = fma(u00, t0, fma(u01, t1, y.b0));
  m1 = fma(u10, t0, fma(u11, t1, y.b1));
  p00 = fma(q00, y.a00, fma(q01, y.a01, y.c00));
  p01 = fma(q00, y.a10, fma(q01, y.a11, y.c01));
  p11 = fma(q10, y.a10, fma(q11, y.a11, y.c11));
}

template <bool GWS>
__global__ __launch_bounds__(NT64) void gibbs64_trend_kernel(G64Args a) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smem64t[];
  __shared__ double xs[NW64 * 16];      // wave totals of the scans / sums
  __shared__ double sc[8];              // scalars handed between waves
  unsigned char* smem = smem64t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const K64& g = a.k;
  const int T = g.T, P = g.P;
  const int series = blockIdx.x / g.C, chain = blockIdx.x % g.C;
  const size_t chain_lin = (size_t)series * g.C + chain;
  const bool s2 = a.has_slope != 0;
  const int D = s2 ? 2 : 1;
  const Layout64 L = make_layout64(T, P, 0, D, a.dred, a.has_slope, GWS ? 1 : 0, a.reg_lds);
  unsigned char* wsc = a.ws + chain_lin * a.ws_stride;
  unsigned char* tb_ = GWS ? wsc : smem;
  double* yv = (double*)(tb_ + L.yv); double* lev = (double*)(tb_ + L.lev);
  double* slp = (double*)(tb_ + L.slp); double* xw = (double*)(tb_ + L.xw);
  double* ytil = (double*)(tb_ + L.ytil); double* vf = (double*)(tb_ + L.vf);
  double* zl = (double*)(tb_ + L.zl); double* zs = (double*)(tb_ + L.zs);
  double* zo = (double*)(tb_ + L.zo);
  double* kf = (double*)(tb_ + L.kf); double* rs = (double*)(tb_ + L.rs);
  double* zi = (double*)(smem + L.zi);
  uint8_t* msk = tb_ + L.mask;
  const int TS = (T + 3) & ~3;
  RegLds R;
  R.bvec = (double*)(smem + L.bvec);
  R.w = nullptr;
  double* wv = (double*)(smem + L.w);
  R.xtx = const_cast<double*>(g.xtx) + (size_t)series * P * P;
  R.omega = const_cast<double*>(g.omega) + (size_t)series * P * P;
  bigp_point(R, a.reg_lds ? smem + L.reg : wsc + ((L.t_total + 255) & ~(size_t)255), P > 0 ? P : 1);

  const DevSeriesParams sp = g.sp[series];
  const Rng rng{stream_key0(g.seed0, g.series_stream_base, series), stream_key1(g.seed1, g.series_stream_base, series), (uint32_t)(g.chain_offset + chain)};
  const double* Xg = g.Xt + (size_t)series * P * T;
  const double* chol1 = a.p1_chol + (size_t)series * a.dred * a.dred;

  for (int t = tid; t < TS; t += NT64) {
    const bool in = t < T;
    const bool m = in ? g.mask[(size_t)series * T + t] != 0 : true;
    msk[t] = m ? 1 : 0;
    yv[t] = m ? 0.0 : g.y[(size_t)series * T + t];
    lev[t] = 0.0; xw[t] = 0.0; ytil[t] = 0.0; vf[t] = 0.0; zl[t] = 0.0; zo[t] = 0.0;
    if (s2) { slp[t] = 0.0; zs[t] = 0.0; }
  }
  for (int j = tid; j < (P > 16 ? P : 16); j += NT64) wv[j] = 0.0;
  if (a.reg_lds && P > 0 && P <= 16) {
    // the register-resident draw re-reads X'X and Omega every iteration: keep them in the LDS
    // block the dense draw would have used
    double* lx = (double*)(smem + L.reg);
    for (int e = tid; e < P * P; e += NT64) { lx[e] = R.xtx[e]; lx[P * P + e] = R.omega[e]; }
    R.xtx = lx; R.omega = lx + P * P;
  }
  if (tid == 0) {
    sc[0] = sp.obs_scale0; sc[1] = sp.level_scale0; sc[2] = sp.slope_scale0;
    sc[3] = 0.0; sc[4] = 0.0;            // sums of squared level / slope increments of the last draw
  }
  __syncthreads();

  const double p1l = (double)(sp.init_level_scale * sp.init_level_scale);
  const double p1s = (double)(sp.init_slope_scale * sp.init_slope_scale);
  const int C4 = (T + 3) / 4;                      // chunks of four steps (one Philox call each)
  const int n_iter = g.W + g.S;
  // The passes over time run on the first NWP waves -- one per SIMD: a second wave on a SIMD would
  // only share its issue slots -- thread j < 64 NWP owning an odd number Lc of consecutive steps.
  constexpr int NWP = 4;
  const int Lc = ((T + 64 * NWP - 1) / (64 * NWP)) | 1;
  const int t0 = tid * Lc < T ? tid * Lc : T;
  const int t1 = t0 + Lc < T ? t0 + Lc : T;
  const bool pw = wave < NWP;             // this wave takes part in the passes

  PriorCarry pc;                          // wave 0: the prior block swept on the current model
  pc.valid = 0; pc.S = 0ull; pc.pdiag = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) pc.p[r] = 0.0;
  Prof prof;
  prof.start(g.prof, g.prof != nullptr && blockIdx.x == 0 && tid == 0);
  for (int it = 0; it <= n_iter; ++it) {
    // ---- (1) targets, X~'targets (one feature per wave over the whole series), y'y
    for (int t = tid; t < T; t += NT64) ytil[t] = msk[t] ? 0.0 : yv[t] - lev[t];
    __syncthreads();
    prof.tick(8);
    for (int j = wave; j <= P; j += 2 * NW64) {
      // two features per wave and round: their rows of X come from L2 in ONE round trip (sixteen
      // loads each); the sums are those of the one-wave kernel, term for term
      const int j2 = j + NW64;
      const bool f1 = j < P, f2 = j2 < P;
      double pj = 0.0, pj2 = 0.0;
      const double* xr1 = Xg + (size_t)(f1 ? j : 0) * T;
      const double* xr2 = Xg + (size_t)(f2 ? j2 : 0) * T;
      for (int tb = lane; tb < T; tb += 64 * 16) {
        double xv[16], xv2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int t = tb + 64 * u, tc = t < T ? t : T - 1;
          xv[u] = f1 ? xr1[tc] : 0.0;
          xv2[u] = f2 ? xr2[tc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int t = tb + 64 * u;
          if (t < T) {
            const double tg = ytil[t];
            pj = fma(f1 ? xv[u] : tg, tg, pj);          // (j == P: y'y)
            pj2 = fma(f2 ? xv2[u] : tg, tg, pj2);
          }
        }
      }
      const double sj = wave_sum_d(pj);
      if (lane == 0) R.bvec[j] = sj;
      if (j2 <= P) {
        const double sj2 = wave_sum_d(pj2);
        if (lane == 0) R.bvec[j2] = sj2;
      }
    }
    prof.tick(9);
    __syncthreads();
    prof.tick(0);
    const double obs_prev = sc[0];
    double emit_obs = obs_prev;
    const int s_out = it - 1 - g.W;                // retained index of iteration it - 1
    const size_t o_out = chain_lin * g.S + (s_out >= 0 ? s_out : 0);
    // ---- (2) in parallel: wave 0 the regression draw of iteration `it` (or, without regression,
    // the observation scale of it - 1); wave 1 / wave 2 the level / slope scale of it - 1; waves
    // 1-7 this iteration's normals and (with regression) the emission of iteration it - 1
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);          // the critical path: ahead of the wave sharing its SIMD
      if (it > 0 && s_out >= 0 && g.out_weights)
        for (int j = lane; j < P; j += 64) g.out_weights[o_out * P + j] = wv[j];
      if (P == 0) {
        if (it > 0) {
          const double so = scale_draw_d(sp.obs_conc, sp.obs_scale, sp.obs_ub, sp.n_obs, R.bvec[P], rng,
                                         (uint32_t)(it - 1), SITE_OBS_SCALE, lane);
          if (lane == 0) sc[0] = so;
        }
      } else if (it < n_iter) {
        const double g_obs = gamma_wave_d(sp.obs_conc + 0.5 * sp.n_obs, rng, (uint32_t)it, SITE_OBSVAR, 0, lane);
        double so;
        if (P <= 16) {      // the register-resident block of the float32 kernels with float64 transcendentals
          NoProf np;
          so = spike_slab_draw_regs<NoProf, NoPublish, true>(R, P, sp, obs_prev, g_obs, rng, (uint32_t)it, lane, np,
                                                             pc, nullptr, NoPublish(), wv);
        } else {
          so = spike_slab_draw_big(R, wv, P, sp, obs_prev, g_obs, rng, (uint32_t)it, lane, it == 0);
        }
        if (lane == 0) sc[5] = so;        // (sc[0] keeps the scale of it - 1 until everyone has read it)
      }
      __builtin_amdgcn_s_setprio(0);
      prof.tick(1);
    } else {
      if (it > 0 && wave == 1) {
        const double v = scale_draw_d(sp.level_conc, sp.level_scale, sp.level_ub, (double)(T - 1), sc[3], rng,
                                      (uint32_t)(it - 1), SITE_LEVEL_SCALE, lane);
        if (lane == 0) sc[1] = v;
      }
      if (it > 0 && wave == 2 && s2) {
        const double v = scale_draw_d(sp.slope_conc, sp.slope_scale, sp.slope_ub, (double)(T - 1), sc[4], rng,
                                      (uint32_t)(it - 1), SITE_SLOPE_SCALE, lane);
        if (lane == 0) sc[2] = v;
      }
      if (it < n_iter) {
        const int nsite = s2 ? 3 : 2;
        for (int q = tid - 64; q < nsite * C4; q += NT64 - 64) {
          const int si = q / C4, c = q - si * C4;
          const uint32_t site = si == 0 ? SITE_PRIOR_LEVEL : (si == 1 ? SITE_PRIOR_OBS : SITE_PRIOR_SLOPE);
          double* dst = si == 0 ? zl : (si == 1 ? zo : zs);
          double z4[4];
          normals4(site_call(rng, (uint32_t)it, site, 0, (uint32_t)c), z4);
          *reinterpret_cast<double4*>(dst + 4 * c) = make_double4(z4[0], z4[1], z4[2], z4[3]);
        }
        if (tid - 64 < a.dred) zi[tid - 64] = normal_d(rng, (uint32_t)it, SITE_PRIOR_INIT, 0, (uint32_t)(tid - 64));
      }
    }
    auto emit = [&](int c_first, int c_step, double so) {
      const size_t row = o_out * T;
      const uint32_t pit = (uint32_t)(it - 1);
      for (int c = c_first; c < C4; c += c_step) {
        double zp[4];
        normals4(site_call(rng, pit, SITE_PRED, 0, (uint32_t)c), zp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = 4 * c + q;
          if (t < T) {
            const double loc = lev[t] + xw[t];
            if (g.out_level) g.out_level[row + t] = lev[t];
            if (g.out_slope && s2) g.out_slope[row + t] = slp[t];
            if (g.out_traj) g.out_traj[row + t] = fma(so, zp[q], loc);
            if (g.out_pred_mean) {
              double* pm = g.out_pred_mean + chain_lin * T + t;   // running sum, scaled at the end
              *pm = (s_out == 0 ? 0.0 : *pm) + loc;
            }
          }
        }
      }
    };
    if (P > 0 && s_out >= 0 && wave > 0) emit(tid - 64, NT64 - 64, emit_obs);
    __syncthreads();
    prof.tick(2);
    if (P == 0) {
      emit_obs = sc[0];
      if (s_out >= 0) emit(tid, NT64, emit_obs);
    }
    const double level_scale = sc[1], slope_scale = sc[2];
    if (s_out >= 0 && tid == 0) {
      if (g.out_obs) g.out_obs[o_out] = emit_obs;
      if (g.out_level_scale) g.out_level_scale[o_out] = level_scale;
      if (g.out_slope_scale) g.out_slope_scale[o_out] = s2 ? slope_scale : 0.0;
    }
    if (it == n_iter) break;
    __syncthreads();                        // everyone has read sc[0] (the scale of it - 1)
    if (P > 0 && tid == 0) sc[0] = sc[5];
    const double obs_scale = P > 0 ? sc[5] : sc[0];

    // ---- (3) X w
    {
      // only the rows of the INCLUDED features are read (a zero weight adds an exact zero: the same
      // sums); two time steps per thread, up to sixteen rows each, in one L2 round trip
      unsigned long long inc = 0ull;
      for (int j0 = 0; j0 < P; j0 += 64) {
        const unsigned long long m = __ballot(j0 + lane < P && wv[j0 + lane < P ? j0 + lane : 0] != 0.0);
        if (j0 == 0) inc = m;
      }
      const bool dense = P > 64;                 // (the mask covers 64 columns)
      for (int tb = tid; tb < T; tb += 2 * NT64) {
        const int tb2 = tb + NT64;
        const bool h2 = tb2 < T;
        double s = 0.0, sb = 0.0;
        if (!dense) {
          for (unsigned long long todo = inc; todo != 0ull;) {
            double xv[16], xv2[16], wj[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const bool have = todo != 0ull;
              const int j = have ? __ffsll((long long)todo) - 1 : 0;
              todo &= todo - 1ull;
              wj[u] = have ? wv[j] : 0.0;
              xv[u] = Xg[(size_t)j * T + tb];
              xv2[u] = h2 ? Xg[(size_t)j * T + tb2] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { s = fma(xv[u], wj[u], s); sb = fma(xv2[u], wj[u], sb); }
          }
        } else {
          for (int j = 0; j < P; ++j) {
            const double w = wv[j];
            s = fma(Xg[(size_t)j * T + tb], w, s);
            if (h2) sb = fma(Xg[(size_t)j * T + tb2], w, sb);
          }
        }
        xw[tb] = s;
        if (h2) xw[tb2] = sb;
      }
    }
    // x+_0 = chol(P_1) z folded into the prior mean (reduced coordinates = full ones: no blocks)
    const double x00 = chol1[0] * zi[0];
    const double a0 = (double)sp.init_level_loc + x00;
    double a1s = 0.0;
    if (s2) a1s = fma(chol1[a.dred + 1], zi[1], chol1[a.dred] * zi[0]);
    const double so = obs_scale, sl = level_scale, ss2 = s2 ? slope_scale : 0.0;
    const double H = so * so, ql = sl * sl, qs2 = ss2 * ss2, p1s2 = s2 ? p1s : 0.0;
    __syncthreads();
    prof.tick(3);

    double xin0 = 0.0, xin1 = 0.0;           // x+ at t0 (kept for pass 3)
    if (pw) {   // pass 0: x+ and y~ = resid - y+
      Drift2 e; e.n = 0.0; e.c0 = 0.0; e.c1 = 0.0;
      for (int t = t0; t < t1; ++t)
        if (t + 1 < T) {
          e.n += 1.0;
          e.c0 = fma(sl, zl[t], e.c0 + e.c1);
          if (s2) e.c1 = fma(ss2, zs[t], e.c1);
        }
      // (wave scans on the DPP crossbar -- ci_kernels.h's wave_scan_incl_* move any element as
      //  32-bit words -- instead of ds_bpermute shuffles: no LDS round trip per level)
      const Drift2 tot = wave_scan_incl_fwd(e, [](const Drift2& x, const Drift2& y) { return drift2_then(x, y); }, lane);
      Drift2 in = dpp_move_e<0x138, 0xF>(tot);                 // wave_shr:1: exclusive
      if (lane == 0) { in.n = 0.0; in.c0 = 0.0; in.c1 = 0.0; }
      {   // across waves: the totals of the waves in front
        if (lane == 63) { xs[wave * 16] = tot.n; xs[wave * 16 + 1] = tot.c0; xs[wave * 16 + 2] = tot.c1; }
        __syncthreads();      // (the waves outside the passes meet every barrier of this section below)
        Drift2 pre; pre.n = 0.0; pre.c0 = 0.0; pre.c1 = 0.0;
        for (int w = 0; w < wave; ++w) {
          Drift2 y; y.n = xs[w * 16]; y.c0 = xs[w * 16 + 1]; y.c1 = xs[w * 16 + 2];
          pre = drift2_then(pre, y);
        }
        in = drift2_then(pre, in);
      }
      xin0 = in.c0; xin1 = in.c1;            // x+_0 = 0
      double x0 = xin0, x1 = xin1;
      for (int t = t0; t < t1; ++t) {
        ytil[t] = (yv[t] - xw[t]) - fma(so, zo[t], x0);
        if (t + 1 < T) {
          x0 = fma(sl, zl[t], x0 + x1);
          if (s2) x1 = fma(ss2, zs[t], x1);
        }
      }
    } else {
      __syncthreads();
    }
    __syncthreads();                         // (xs is reused; y~ of a chunk is read by its owner only)
    prof.tick(4);
    if (pw) {   // pass 1: Kalman filter, gains and scaled innovations
      Kf2 e;
      e.a00 = 1.0; e.a01 = 0.0; e.a10 = 0.0; e.a11 = 1.0; e.b0 = 0.0; e.b1 = 0.0;
      e.c00 = 0.0; e.c01 = 0.0; e.c11 = 0.0; e.e0 = 0.0; e.e1 = 0.0; e.j00 = 0.0; e.j01 = 0.0; e.j11 = 0.0;
      if (tid == 0) { e.a00 = 0.0; e.a11 = 0.0; e.b0 = a0; e.b1 = a1s; e.c00 = p1l; e.c11 = p1s2; }
      for (int t = t0; t < t1; ++t) {
        if (msk[t] == 0) {
          const double rf = 1.0 / (e.c00 + H);
          const double v = ytil[t] - e.b0;
          const double k0 = e.c00 * rf, k1 = e.c01 * rf;
          const double vr = v * rf;
          e.e0 = fma(e.a00, vr, e.e0); e.e1 = fma(e.a01, vr, e.e1);
          e.j00 = fma(e.a00 * rf, e.a00, e.j00); e.j01 = fma(e.a00 * rf, e.a01, e.j01);
          e.j11 = fma(e.a01 * rf, e.a01, e.j11);
          e.b0 = fma(k0, v, e.b0); e.b1 = fma(k1, v, e.b1);
          const double r00 = e.a00, r01 = e.a01;
          e.a10 = fma(-k1, r00, e.a10); e.a11 = fma(-k1, r01, e.a11);
          e.a00 = fma(-k0, r00, e.a00); e.a01 = fma(-k0, r01, e.a01);
          const double c00 = e.c00, c01 = e.c01;
          e.c00 -= c00 * c00 * rf; e.c01 -= c00 * c01 * rf; e.c11 -= c01 * c01 * rf;
        }
        e.a00 += e.a10; e.a01 += e.a11;
        e.b0 += e.b1;
        e.c00 = e.c00 + 2.0 * e.c01 + e.c11 + ql;
        e.c01 = e.c01 + e.c11;
        e.c11 = e.c11 + qs2;
      }
      const Kf2 inc = wave_scan_incl_fwd(e, [](const Kf2& x, const Kf2& y) { return kf2_then(x, y); }, lane);
      if (lane == 63) {
        double* d = xs + wave * 16;
        d[0] = inc.a00; d[1] = inc.a01; d[2] = inc.a10; d[3] = inc.a11; d[4] = inc.b0; d[5] = inc.b1;
        d[6] = inc.c00; d[7] = inc.c01; d[8] = inc.c11; d[9] = inc.e0; d[10] = inc.e1;
        d[11] = inc.j00; d[12] = inc.j01; d[13] = inc.j11;
      }
      __syncthreads();
      // predicted moments entering this wave: wave 0's total carries the prior (A = 0: its (b, C)
      // ARE the moments after it), each further wave's total is applied to them
      double m0 = a0, m1 = a1s, p00 = p1l, p01 = 0.0, p11 = p1s2;
      if (wave > 0) {
        m0 = xs[4]; m1 = xs[5]; p00 = xs[6]; p01 = xs[7]; p11 = xs[8];
        for (int w = 1; w < wave; ++w) {
          const double* d = xs + w * 16;
          Kf2 y;
          y.a00 = d[0]; y.a01 = d[1]; y.a10 = d[2]; y.a11 = d[3]; y.b0 = d[4]; y.b1 = d[5];
          y.c00 = d[6]; y.c01 = d[7]; y.c11 = d[8]; y.e0 = d[9]; y.e1 = d[10];
          y.j00 = d[11]; y.j01 = d[12]; y.j11 = d[13];
          kf2_apply(y, m0, m1, p00, p01, p11);
        }
      }
      {   // ... and the in-wave prefix up to the previous lane on top
        const Kf2 pl = dpp_move_e<0x138, 0xF>(inc);       // wave_shr:1
        if (lane > 0) kf2_apply(pl, m0, m1, p00, p01, p11);
      }
      for (int t = t0; t < t1; ++t) {
        double k0 = 0.0, k1 = 0.0, vfq = 0.0;
        if (msk[t] == 0) {
          const double rF = 1.0 / (p00 + H);
          const double v = ytil[t] - m0;
          k0 = p00 * rF; k1 = p01 * rF;
          vfq = v * rF;
          m0 = fma(k0, v, m0); m1 = fma(k1, v, m1);
          const double q00 = p00, q01 = p01;
          p00 -= q00 * q00 * rF; p01 -= q00 * q01 * rF; p11 -= q01 * q01 * rF;
        }
        kf[(size_t)t * D] = k0;
        if (s2) kf[(size_t)t * D + 1] = k1;
        vf[t] = vfq;
        if (t + 1 < T) {
          m0 += m1;
          p00 = p00 + 2.0 * p01 + p11 + ql;
          p01 = p01 + p11;
          p11 = p11 + qs2;
        }
      }
    } else {
      __syncthreads();
    }
    __syncthreads();
    prof.tick(5);
    if (pw) {   // pass 2: backward recursion, rs[t] = r_{t-1}
      auto step = [&](int t, double& r0, double& r1, double k0, double k1, double vft, bool obs) {
        if (t + 1 < T) r1 += r0;                      // r <- T' r
        if (obs) r0 += vft - fma(k0, r0, k1 * r1);    // + z (v/F - K'r)
      };
      Aff2 e; e.m00 = 1.0; e.m01 = 0.0; e.m10 = 0.0; e.m11 = 1.0; e.c0 = 0.0; e.c1 = 0.0;
      for (int t = t1 - 1; t >= t0; --t) {
        const bool obs = msk[t] == 0;
        const double k0 = kf[(size_t)t * D], k1 = s2 ? kf[(size_t)t * D + 1] : 0.0, vft = vf[t];
        double c0 = e.c0, c1 = e.c1, x0 = e.m00, x1 = e.m10, y0 = e.m01, y1 = e.m11;
        step(t, c0, c1, k0, k1, vft, obs);
        step(t, x0, x1, k0, k1, 0.0, obs);
        step(t, y0, y1, k0, k1, 0.0, obs);
        e.c0 = c0; e.c1 = c1; e.m00 = x0; e.m10 = x1; e.m01 = y0; e.m11 = y1;
      }
      // suffix scan: op(outer, inner) with the lanes to the right acting first
      const Aff2 tot = wave_scan_incl_bwd(e, [](const Aff2& outer, const Aff2& inner) { return aff2_then(inner, outer); }, lane);
      Aff2 in = dpp_move_e<0x130, 0xF>(tot);             // wave_shl:1: the lanes to the right only
      if (lane == 63) { in.m00 = 1.0; in.m01 = 0.0; in.m10 = 0.0; in.m11 = 1.0; in.c0 = 0.0; in.c1 = 0.0; }
      {
        if (lane == 0) {
          double* d = xs + wave * 16;
          d[0] = tot.m00; d[1] = tot.m01; d[2] = tot.m10; d[3] = tot.m11; d[4] = tot.c0; d[5] = tot.c1;
        }
      }
      __syncthreads();
      double rw0 = 0.0, rw1 = 0.0;                        // r entering this wave from the right (0 at the end)
      for (int w = NWP - 1; w > wave; --w) {
        const double* d = xs + w * 16;
        const double n0 = fma(d[0], rw0, fma(d[1], rw1, d[4]));
        const double n1 = fma(d[2], rw0, fma(d[3], rw1, d[5]));
        rw0 = n0; rw1 = n1;
      }
      double r0 = fma(in.m00, rw0, fma(in.m01, rw1, in.c0));
      double r1 = fma(in.m10, rw0, fma(in.m11, rw1, in.c1));
      for (int t = t1 - 1; t >= t0; --t) {
        const double k0 = kf[(size_t)t * D], k1 = s2 ? kf[(size_t)t * D + 1] : 0.0;
        step(t, r0, r1, k0, k1, vf[t], msk[t] == 0);
        rs[(size_t)t * D] = r0;
        if (s2) rs[(size_t)t * D + 1] = r1;
      }
    } else {
      __syncthreads();
    }
    __syncthreads();
    prof.tick(6);
    double al = 0.0, as = 0.0;
    if (pw) {   // pass 3: x^ forward, x+ again, the draw and its increment statistics
      const double hi0 = fma(p1l, rs[0], a0), hi1 = s2 ? fma(p1s2, rs[1], a1s) : 0.0;
      Drift2 e; e.n = 0.0; e.c0 = 0.0; e.c1 = 0.0;
      for (int t = t0; t < t1; ++t)
        if (t + 1 < T) {
          e.n += 1.0;
          e.c0 = fma(ql, rs[(size_t)(t + 1) * D], e.c0 + e.c1);
          if (s2) e.c1 = fma(qs2, rs[(size_t)(t + 1) * D + 1], e.c1);
        }
      const Drift2 tot = wave_scan_incl_fwd(e, [](const Drift2& x, const Drift2& y) { return drift2_then(x, y); }, lane);
      Drift2 in = dpp_move_e<0x138, 0xF>(tot);
      if (lane == 0) { in.n = 0.0; in.c0 = 0.0; in.c1 = 0.0; }
      {
        if (lane == 63) { xs[wave * 16] = tot.n; xs[wave * 16 + 1] = tot.c0; xs[wave * 16 + 2] = tot.c1; }
        __syncthreads();
        Drift2 pre; pre.n = 0.0; pre.c0 = 0.0; pre.c1 = 0.0;
        for (int w = 0; w < wave; ++w) {
          Drift2 y; y.n = xs[w * 16]; y.c0 = xs[w * 16 + 1]; y.c1 = xs[w * 16 + 2];
          pre = drift2_then(pre, y);
        }
        in = drift2_then(pre, in);
      }
      double h0 = fma(in.n, hi1, hi0) + in.c0, h1 = hi1 + in.c1;
      double x0 = xin0, x1 = xin1;
      for (int t = t0; t < t1; ++t) {
        const double xt0 = h0 + x0, xt1 = h1 + x1;
        lev[t] = xt0;
        if (s2) slp[t] = xt1;
        if (t + 1 < T) {
          h0 = fma(ql, rs[(size_t)(t + 1) * D], h0 + h1);
          x0 = fma(sl, zl[t], x0 + x1);
          if (s2) {
            h1 = fma(qs2, rs[(size_t)(t + 1) * D + 1], h1);
            x1 = fma(ss2, zs[t], x1);
          }
          const double nx0 = h0 + x0, nx1 = h1 + x1;       // the increment t -> t + 1
          const double dl = (nx0 - xt0) - xt1;
          al = fma(dl, dl, al);
          if (s2) { const double ds = nx1 - xt1; as = fma(ds, ds, as); }
        }
      }
    } else {
      __syncthreads();
    }
    {
      double wl = 0.0, wsl = 0.0;
      if (pw) { wl = wave_sum_d(al); wsl = wave_sum_d(as); }
      if (pw && lane == 0) { xs[wave * 16 + 14] = wl; xs[wave * 16 + 15] = wsl; }   // (slots the scans do not use)
      __syncthreads();
      if (tid == 0) {
        double tl = 0.0, tsl = 0.0;
        for (int w = 0; w < NWP; ++w) { tl += xs[w * 16 + 14]; tsl += xs[w * 16 + 15]; }
        sc[3] = tl; sc[4] = tsl;
      }
    }
    __syncthreads();
    prof.tick(7);
  }
  __syncthreads();
  if (g.out_pred_mean) {
    const double inv = 1.0 / (double)(g.S > 0 ? g.S : 1);
    for (int t = tid; t < T; t += NT64) g.out_pred_mean[chain_lin * T + t] *= inv;
  }
}
#endif  // CI_SEASONAL_DECL_ONLY

}  // namespace ci
