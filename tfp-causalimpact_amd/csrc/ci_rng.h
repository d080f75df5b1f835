// ci_rng.h -- the specified counter-based random stream, device side (gfx950).
//
// Same stream as oracle/ci_oracle.c:  Philox4x32-10,
//   key = (seed0, seed1), counter = (call, site | sub << 8, iteration, chain).
// One call = 4 uniforms or 4 Box-Muller normals (components 0..3); element idx of a
// site lives in call idx >> 2, component idx & 3.  Replaces the stateless seeds TFP
// threads through gibbs_sampler (tfp.random.split_seed; reference call sites
// causalimpact_lib.py:364, :543).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ci {

enum Site : uint32_t {
  SITE_PERM = 1, SITE_FLIP = 2, SITE_OBSVAR = 3, SITE_WEIGHTS = 4, SITE_PRIOR_INIT = 5,
  SITE_PRIOR_LEVEL = 6, SITE_PRIOR_SLOPE = 7, SITE_PRIOR_OBS = 8, SITE_PRIOR_SEAS = 9,
  SITE_LEVEL_SCALE = 10, SITE_SLOPE_SCALE = 11, SITE_DRIFT_SCALE = 12, SITE_OBS_SCALE = 13,
  SITE_PRED = 14,
  // HMC extension (ci_hmc.h): momentum, accept uniform, initial jitter
  SITE_HMC_MOMENTUM = 15, SITE_HMC_ACCEPT = 16, SITE_HMC_INIT = 17
};

struct U4 { uint32_t x, y, z, w; };

struct Rng {
  uint32_t k0, k1, chain;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply (v_mad_u64_u32) per product instead of mul_hi + mul_lo
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ U4 site_call(const Rng& g, uint32_t iter, uint32_t site, uint32_t sub,
                                        uint32_t call) {
  return philox4x32_10(call, site | (sub << 8), iter, g.chain, g.k0, g.k1);
}

__device__ __forceinline__ float u01f(uint32_t r) { return ((float)r + 0.5f) * 2.3283064365386963e-10f; }
__device__ __forceinline__ double u01d(uint32_t r) { return ((double)r + 0.5) * (1.0 / 4294967296.0); }

// Box-Muller on one (ra, rb) pair; the angle is taken in revolutions so it maps onto
// v_sin_f32 / v_cos_f32 directly.
__device__ __forceinline__ void box_muller_f(uint32_t ra, uint32_t rb, float& z0, float& z1) {
  const float u1 = fminf(u01f(ra), 1.0f);
  const float rev = (float)rb * 2.3283064365386963e-10f;
  const float rad = __fsqrt_rn(-2.0f * __logf(u1));
  z0 = rad * __builtin_amdgcn_cosf(rev);
  z1 = rad * __builtin_amdgcn_sinf(rev);
}

__device__ __forceinline__ void box_muller_d(uint32_t ra, uint32_t rb, double& z0, double& z1) {
  const double u1 = u01d(ra);
  const double rev = (double)rb * (1.0 / 4294967296.0);
  const double rad = sqrt(-2.0 * log(u1));
  // the oracle takes cos / sin of 2 pi rev; sincospi of 2 rev is that angle without the rounding of
  // the product (a 4e-16 difference in the angle) and one argument reduction instead of two
  double sn, cs;
  sincospi(2.0 * rev, &sn, &cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

__device__ __forceinline__ void normals4(const U4& r, float z[4]) {
  box_muller_f(r.x, r.y, z[0], z[1]);
  box_muller_f(r.z, r.w, z[2], z[3]);
}

// Normals for the L consecutive elements [t0, t0 + L) of a site (t0 a multiple of L,
// L in {1, 2, 4, 8, 16}).
template <int L>
__device__ __forceinline__ void fill_normals(const Rng& g, uint32_t iter, uint32_t site,
                                             uint32_t sub, uint32_t t0, float (&z)[L]) {
  if constexpr (L % 4 == 0) {
#pragma unroll
    for (int q = 0; q < L / 4; ++q) {
      const U4 r = site_call(g, iter, site, sub, (t0 >> 2) + q);
      float zz[4];
      normals4(r, zz);
#pragma unroll
      for (int i = 0; i < 4; ++i) z[4 * q + i] = zz[i];
    }
  } else if constexpr (L == 2) {
    const U4 r = site_call(g, iter, site, sub, t0 >> 2);
    const bool hi = (t0 & 2u) != 0u;
    box_muller_f(hi ? r.z : r.x, hi ? r.w : r.y, z[0], z[1]);
  } else {
    static_assert(L == 1, "L must be 1, 2 or a multiple of 4");
    const U4 r = site_call(g, iter, site, sub, t0 >> 2);
    const bool hi = (t0 & 2u) != 0u;
    float a, b;
    box_muller_f(hi ? r.z : r.x, hi ? r.w : r.y, a, b);
    z[0] = (t0 & 1u) ? b : a;
  }
}

__device__ __forceinline__ double normal_d(const Rng& g, uint32_t iter, uint32_t site,
                                           uint32_t sub, uint32_t idx) {
  const U4 r = site_call(g, iter, site, sub, idx >> 2);
  const bool hi = (idx & 2u) != 0u;
  double a, b;
  box_muller_d(hi ? r.z : r.x, hi ? r.w : r.y, a, b);
  return (idx & 1u) ? b : a;
}

__device__ __forceinline__ double uniform_d(const Rng& g, uint32_t iter, uint32_t site,
                                            uint32_t sub, uint32_t idx) {
  const U4 r = site_call(g, iter, site, sub, idx >> 2);
  const uint32_t c = idx & 3u;
  return u01d(c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w);
}

// ---- cheap float64 helpers for the serial (wave 0) section.  IEEE f64 division / sqrt /
// log expand to long ocml sequences; a float32 hardware seed plus Newton steps is ~10 ops.
// (v_rcp_f32 / v_rsq_f32 are accurate to ~1 ulp = 6e-8; one Newton step squares that to
//  ~1e-14, a second one reaches the float64 rounding floor.  Each step is 2-3 DEPENDENT f64
//  ops at ~10+ cycles with one wave per SIMD, so the second step is kept only for rcp.)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = (double)__builtin_amdgcn_rcpf((float)d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double d) {
  double r = (double)__builtin_amdgcn_rsqf((float)d);
  const double h = 0.5 * d;
  r = fma(r, fma(-h * r, r, 0.5), r);
  r = fma(r, fma(-h * r, r, 0.5), r);
  return r;
}
// log(1 + x): atanh series (|x| < 0.3, ~1e-15) else float32 hardware log (|log| is large
// there, so its 1e-7 relative error is irrelevant to an accept/flip decision).
__device__ __forceinline__ double fast_log1p(double x) {
  if (fabs(x) < 0.3) {
    const double w = x * fast_rcp(2.0 + x);
    const double w2 = w * w;
    double p = 1.0 / 19.0;
    p = fma(p, w2, 1.0 / 17.0);
    p = fma(p, w2, 1.0 / 15.0);
    p = fma(p, w2, 1.0 / 13.0);
    p = fma(p, w2, 1.0 / 11.0);
    p = fma(p, w2, 1.0 / 9.0);
    p = fma(p, w2, 1.0 / 7.0);
    p = fma(p, w2, 1.0 / 5.0);
    p = fma(p, w2, 1.0 / 3.0);
    p = fma(p, w2, 1.0);
    return 2.0 * w * p;
  }
  return (double)__logf((float)(1.0 + x));
}
__device__ __forceinline__ double readlane_d(double v, int l) {   // l must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Marsaglia-Tsang acceptance bound  0.5 x^2 + d (1 - v + log v),  v = (1 + z)^3, z = c x.
// With 9 d c^2 = 1 the quadratic terms cancel EXACTLY and what is left is
//   3 d * sum_{k>=4} (-1)^(k+1) z^k / k  =  3 d z^4 (-1/4 + z/5 - z^2/6 + ...),
// free of cancellation, so float32 suffices for |z| < 0.4 (remainder < 2e-7 relative); the
// float64 closed form is kept for large |z| (small shapes).
__device__ __forceinline__ double mt_accept_bound(double x, double d, double z) {
  const float zf = (float)z;
  if (fabsf(zf) < 0.4f) {
    float q = 1.0f / 18.0f;
    q = fmaf(q, -zf, 1.0f / 17.0f);
    q = fmaf(q, -zf, 1.0f / 16.0f);
    q = fmaf(q, -zf, 1.0f / 15.0f);
    q = fmaf(q, -zf, 1.0f / 14.0f);
    q = fmaf(q, -zf, 1.0f / 13.0f);
    q = fmaf(q, -zf, 1.0f / 12.0f);
    q = fmaf(q, -zf, 1.0f / 11.0f);
    q = fmaf(q, -zf, 1.0f / 10.0f);
    q = fmaf(q, -zf, 1.0f / 9.0f);
    q = fmaf(q, -zf, 1.0f / 8.0f);
    q = fmaf(q, -zf, 1.0f / 7.0f);
    q = fmaf(q, -zf, 1.0f / 6.0f);
    q = fmaf(q, -zf, 1.0f / 5.0f);
    q = fmaf(q, -zf, 1.0f / 4.0f);      // q = 1/4 - z/5 + z^2/6 - ...
    const float z2 = zf * zf;
    return -3.0 * d * (double)(z2 * z2 * q);
  }
  const double t = 1.0 + z;
  return 0.5 * x * x + d * (1.0 - t * t * t + 3.0 * fast_log1p(z));
}

// Gamma(alpha, 1), Marsaglia-Tsang.  The 64 lanes of the calling wavefront evaluate
// attempts 0..63 at once; the first accepted attempt (lowest index) wins, which is the
// sequential oracle's answer.  Must be called by a full, converged wave.  The proposal
// normal is the float32 Box-Muller of the stream (the draw inherits ~c * 1e-6 relative
// error, c = 1/sqrt(9 d) << 1); the acceptance test is evaluated in float64.
static __device__ __noinline__ double gamma_wave(double alpha, const Rng& g, uint32_t iter,
                                             uint32_t site, uint32_t sub, int lane) {
  const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  const double d = a - 1.0 / 3.0;
  const double c = fast_rsqrt(9.0 * d);
  const U4 r = site_call(g, iter, site, sub, (uint32_t)lane);
  float xf, unused;
  box_muller_f(r.x, r.y, xf, unused);
  const double x = (double)xf;
  const double cx = c * x;
  const double t = 1.0 + cx;
  const double v = t * t * t;
  bool ok = false;
  double gval = d;
  if (v > 0.0) {
    const double lhs = (double)__logf(u01f(r.z));
    ok = lhs < mt_accept_bound(x, d, cx);
    gval = d * v;
    if (alpha < 1.0) gval *= pow(u01d(r.w), 1.0 / alpha);
  }
  const unsigned long long m = __ballot(ok);
  if (m == 0ull) return d;
  const int first = __ffsll((long long)m) - 1;
  return readlane_d(gval, first);
}

// Three independent Gamma(alpha_q, 1) draws in one pass: quadrant q = lane >> 4 (q < 3)
// evaluates attempts of draw q (attempt index = 16 * round + (lane & 15)); identical results to
// three gamma_wave() calls because the first accepted attempt index wins either way.  Requests
// are plain scalars (an array indexed by the lane's quadrant would live in scratch memory).
struct GammaReq {
  double alpha;
  uint32_t iter, site, sub;
};
__device__ __forceinline__ void gamma_wave3(const GammaReq& r0, const GammaReq& r1,
                                            const GammaReq& r2, unsigned active, double& g0,
                                            double& g1, double& g2, const Rng& g, int lane) {
  const int q = lane >> 4, at = lane & 15;
  const double alpha = q == 1 ? r1.alpha : (q == 2 ? r2.alpha : r0.alpha);
  const uint32_t iter = q == 1 ? r1.iter : (q == 2 ? r2.iter : r0.iter);
  const uint32_t site = q == 1 ? r1.site : (q == 2 ? r2.site : r0.site);
  const uint32_t sub = q == 1 ? r1.sub : (q == 2 ? r2.sub : r0.sub);
  const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  const double d = a - 1.0 / 3.0;
  const double c = fast_rsqrt(9.0 * d);
  const bool mine_active = q < 3 && ((active >> q) & 1u) != 0u;
  unsigned pending = active & 7u;
#pragma unroll 1
  for (int round = 0; round < 4 && pending != 0u; ++round) {
    const U4 r = site_call(g, iter, site, sub, (uint32_t)(16 * round + at));
    float xf, unused;
    box_muller_f(r.x, r.y, xf, unused);
    const double x = (double)xf;
    const double cx = c * x;
    const double t = 1.0 + cx;
    const double v = t * t * t;
    bool ok = false;
    double gval = d;
    if (v > 0.0) {
      const double lhs = (double)__logf(u01f(r.z));
      ok = lhs < mt_accept_bound(x, d, cx);
      gval = d * v;
      if (alpha < 1.0) gval *= pow(u01d(r.w), 1.0 / alpha);
    }
    const unsigned long long m = __ballot(ok && mine_active);
    const unsigned b0 = (unsigned)(m & 0xFFFFull), b1 = (unsigned)((m >> 16) & 0xFFFFull),
                   b2 = (unsigned)((m >> 32) & 0xFFFFull);
    if ((pending & 1u) && b0) { g0 = readlane_d(gval, __ffs((int)b0) - 1); pending &= ~1u; }
    if ((pending & 2u) && b1) { g1 = readlane_d(gval, 16 + __ffs((int)b1) - 1); pending &= ~2u; }
    if ((pending & 4u) && b2) { g2 = readlane_d(gval, 32 + __ffs((int)b2) - 1); pending &= ~4u; }
  }
  // 64 rejections in a row: unreachable in practice; same fallback as the oracle
  if (pending & 1u) g0 = (r0.alpha < 1.0 ? r0.alpha + 1.0 : r0.alpha) - 1.0 / 3.0;
  if (pending & 2u) g1 = (r1.alpha < 1.0 ? r1.alpha + 1.0 : r1.alpha) - 1.0 / 3.0;
  if (pending & 4u) g2 = (r2.alpha < 1.0 ? r2.alpha + 1.0 : r2.alpha) - 1.0 / 3.0;
}

}  // namespace ci
