// ci_api.hip -- C-ABI (include/causalimpact_amd.h) over the HIP kernels in ci_kernels.h.
// Plain HIP runtime: no torch, no TensorFlow.  One ci_session == one device-resident fit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/causalimpact_amd.h"
#include "ci_kernels.h"
#include "ci_kernels8.h"
#define CI_SEASONAL_DECL_ONLY
#include "ci_seasonal.h"
#include "ci_wide.h"
#include "ci_seasonal_tp.h"
#include "ci_summary.h"
#include "ci_hmc.h"
#include "ci_score_seq.h"
#include "ci_gibbs64.h"
#include "ci_wide_score.h"

extern "C" void* ci_gibbs_seasonal_fn(int);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq2(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq3(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq4(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq5(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq6(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq7(void);
extern "C" void* ci_gibbs_seasonal_tp_fn_nq8(void);
extern "C" void ci_launch_seq_score(const ci::SeqScoreArgs*, int, hipStream_t);
extern "C" void ci_launch_hmc_seq(const ci::HmcSeqArgs*, int, hipStream_t);
extern "C" void ci_launch_gibbs64(const ci::G64Args*, int, size_t, int, hipStream_t);
#define CI_WIDEBIG_DECL(NS)                                    \
  extern "C" void* ci_gibbs_wide_bigp_fn_tr1_ns##NS(void);     \
  extern "C" void* ci_gibbs_wide_bigp_fn_tr2_ns##NS(void);
CI_WIDEBIG_DECL(2) CI_WIDEBIG_DECL(3) CI_WIDEBIG_DECL(4) CI_WIDEBIG_DECL(5) CI_WIDEBIG_DECL(6) CI_WIDEBIG_DECL(7)
#undef CI_WIDEBIG_DECL
#define CI_WIDE_DECL(NS)                                  \
  extern "C" void* ci_gibbs_wide_fn_tr1_ns##NS(void);     \
  extern "C" void* ci_gibbs_wide_fn_tr2_ns##NS(void);     \
  extern "C" void ci_launch_wide_score_tr1_ns##NS(const ci::WideScoreArgs*, hipStream_t);  \
  extern "C" void ci_launch_wide_score_tr2_ns##NS(const ci::WideScoreArgs*, hipStream_t);  \
  extern "C" void ci_launch_hmc_wide_tr1_ns##NS(const ci::HmcWideArgs*, hipStream_t);      \
  extern "C" void ci_launch_hmc_wide_tr2_ns##NS(const ci::HmcWideArgs*, hipStream_t);
CI_WIDE_DECL(2) CI_WIDE_DECL(3) CI_WIDE_DECL(4) CI_WIDE_DECL(5) CI_WIDE_DECL(6) CI_WIDE_DECL(7)
#undef CI_WIDE_DECL

// One object file per (D, L) instantiation (ci_inst.hip).
#define CI_DECL(D, L)                                                                          \
  extern "C" void* ci_gibbs_fn_d##D##_l##L(int);                                              \
  extern "C" void* ci_gibbs8_fn_d##D##_l##L(int, int, size_t*);                               \
  extern "C" void ci_launch_dk_d##D##_l##L(int, const float*, const uint8_t*, float, float,    \
                                           float, float, float, float, uint32_t, uint32_t,     \
                                           uint32_t, uint32_t, float*);                           \
  extern "C" void ci_launch_loglik_d##D##_l##L(int, int, int, const float*, const uint8_t*,       \
                                               const float*, const double*, float, float, float,  \
                                               double*, hipStream_t);                             \
  extern "C" void ci_launch_llgrad_d##D##_l##L(int, int, int, const float*, const uint8_t*,       \
                                               const float*, const double*, float, float, float,  \
                                               double*, double*, hipStream_t);                    \
  extern "C" void ci_launch_latents_d##D##_l##L(int, int, int, const float*, const uint8_t*,      \
                                                const float*, const double*, float, float, float, \
                                                uint32_t, uint32_t, uint32_t, uint32_t, int, int, \
                                                float*, float*, float*, float*, float*,           \
                                                hipStream_t);                                     \
  extern "C" void ci_launch_hmc_d##D##_l##L(const ci::HmcArgs*, hipStream_t);
CI_DECL(1, 1) CI_DECL(1, 2) CI_DECL(1, 4) CI_DECL(1, 8) CI_DECL(1, 16)
CI_DECL(2, 1) CI_DECL(2, 2) CI_DECL(2, 4) CI_DECL(2, 8) CI_DECL(2, 16)
#undef CI_DECL

namespace ci {
// ------------------------------------------------------------------------------------
// setup: X~'X~ (observed rows) and the weights-prior precision (all rows), float64.
// causalimpact_lib.py:451-453; SpikeSlabSampler.__init__ (design rows at missing steps = 0).
// One workgroup per series; thread (i, j) streams over T (coalesced over the
// feature-major copy).
// ------------------------------------------------------------------------------------
template <class XT>
static __global__ void setup_regression_kernel(int T, int P, const XT* Xt, const uint8_t* mask,
                                        const double* __restrict__ prior_scale, double* xtx,
                                        double* omega) {
  // one wavefront per (series, i, j): lanes stride over time (both rows coalesced), float64 sums
  const int e = blockIdx.x % (P * P), series = blockIdx.x / (P * P);
  const int i = e / P, j = e % P, lane = threadIdx.x;
  const XT* xi = Xt + ((size_t)series * P + i) * T;
  const XT* xj = Xt + ((size_t)series * P + j) * T;
  const uint8_t* m = mask + (size_t)series * T;
  double so = 0.0, sa = 0.0;
  for (int t = lane; t < T; t += 64) {
    const double v = (double)xi[t] * (double)xj[t];
    sa += v;
    if (!m[t]) so += v;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    so += __shfl_xor(so, off, 64);
    sa += __shfl_xor(sa, off, 64);
  }
  if (lane == 0) {
    xtx[(size_t)series * P * P + e] = so;
    omega[(size_t)series * P * P + e] =
        0.01 * (i == j ? sa : 0.5 * sa) / (double)T * prior_scale[series];
  }
}

// ------------------------------------------------------------------------------------
// component test kernels
// ------------------------------------------------------------------------------------
static __global__ void test_rng_kernel(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t iter,
                                uint32_t site, uint32_t sub, int n, float* uni, float* nor,
                                double alpha, double* gam) {
  Rng g{k0, k1, chain};
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += blockDim.x) {
    uni[i] = (float)uniform_d(g, iter, site, sub, (uint32_t)i);
    float z[1];
    fill_normals<1>(g, iter, site, sub, (uint32_t)i, z);
    nor[i] = z[0];
  }
  if (tid < 64) {
    const double v = gamma_wave(alpha, g, iter, site, sub, tid);
    if (tid == 0) gam[0] = v;
    // also exercise the 4-wide path: normals via fill_normals<4> must agree with <1>
    float z4[4];
    fill_normals<4>(g, iter, site, sub, (uint32_t)(tid * 4), z4);
    for (int q = 0; q < 4; ++q)
      if (tid * 4 + q < n) nor[n + tid * 4 + q] = z4[q];
  }
}

// Per-chain mean over the S retained draws of the noise-free predictor (causalimpact_lib.py:627)
// from the per-group sums the latents pass leaves: part [C, NG, T] -> pm [C, T].  One thread per
// (chain, t), coalesced over t; the order of the sums is fixed, so chain c's mean does not depend
// on how chains are split over launches.
static __global__ void hmc_mean_kernel(int C, int NG, int S, int T, const float* __restrict__ part,
                                       float* __restrict__ pm) {
  // part [C, NG, T]: sums of the predictor over groups of consecutive draws (latents_kernel)
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t >= T || c >= C) return;
  const float* p = part + (size_t)c * NG * T + t;
  float acc = 0.f;
  for (int g = 0; g < NG; ++g) acc += p[(size_t)g * T];
  pm[(size_t)c * T + t] = acc / (float)S;
}

// (sigma_obs, sigma_level, sigma_slope, beta) rows in float64 -> the float32 sample container.
static __global__ void hmc_unpack_kernel(int N, int P, const double* __restrict__ draws,
                                         float* __restrict__ obs, float* __restrict__ lscale,
                                         float* __restrict__ sscale, float* __restrict__ w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double* r = draws + (size_t)n * (3 + P);
  obs[n] = (float)r[0];
  lscale[n] = (float)r[1];
  sscale[n] = (float)r[2];
  for (int j = 0; j < P; ++j) w[(size_t)n * P + j] = (float)r[3 + j];
}


}  // namespace ci

namespace {
using KernelFn = void (*)(ci::KArgs);
KernelFn pick_kernel8(int D, int L, int profiled, int xg, size_t* lds_base) {
#define CI_CASE5(DD, LL) if (D == DD && L == LL) return (KernelFn)ci_gibbs8_fn_d##DD##_l##LL(profiled, xg, lds_base);
  CI_CASE5(1, 1) CI_CASE5(1, 2) CI_CASE5(1, 4) CI_CASE5(1, 8) CI_CASE5(1, 16)
  CI_CASE5(2, 1) CI_CASE5(2, 2) CI_CASE5(2, 4) CI_CASE5(2, 8) CI_CASE5(2, 16)
#undef CI_CASE5
  return nullptr;
}
KernelFn pick_kernel(int D, int L, int pm) {
#define CI_CASE(DD, LL) if (D == DD && L == LL) return (KernelFn)ci_gibbs_fn_d##DD##_l##LL(pm);
  CI_CASE(1, 1) CI_CASE(1, 2) CI_CASE(1, 4) CI_CASE(1, 8) CI_CASE(1, 16)
  CI_CASE(2, 1) CI_CASE(2, 2) CI_CASE(2, 4) CI_CASE(2, 8) CI_CASE(2, 16)
#undef CI_CASE
  return nullptr;
}
}  // namespace

namespace {

thread_local std::string g_err;
thread_local float g_f64_kernel_ms = 0.f;   // duration of the last ci_fit_gibbs_f64 kernel on this thread

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Device allocations are recycled through a small per-process pool (exact-size match per
// device, at most POOL_CAP bytes parked; ci_pool_trim returns them): a fit allocates ~10 buffers, 100+ MB of them outputs, and
// hipMalloc / hipFree of those cost several milliseconds per fit_causalimpact() call -- comparable
// to the 12 ms the sampler itself takes.  The pool holds no caller data and no pointers escape.
struct PoolEntry { void* p; size_t bytes; int device; };
std::mutex g_pool_mu;
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
// 32 GiB of 288: a 512-series batch parks 6 GB (1 GB each of level / trajectories, 2 GB each of the
// float64 summary matrices); with the 2 GiB cap of rounds 1-2 every batch call re-allocated them.
constexpr size_t POOL_CAP = (size_t)32 << 30;
constexpr size_t HOST_POOL_CAP = (size_t)2 << 30;   // pinned host memory is the scarcer resource

hipError_t pool_alloc(void** out, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].bytes == bytes && g_pool[i].device == dev) {
        *out = g_pool[i].p;
        g_pool_bytes -= bytes;
        g_pool[i] = g_pool.back();
        g_pool.pop_back();
        return hipSuccess;
      }
  }
  e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    // out of memory with buffers parked: give them back and retry once
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& pe : g_pool) { (void)hipSetDevice(pe.device); (void)hipFree(pe.p); }
    g_pool.clear();
    g_pool_bytes = 0;
    (void)hipSetDevice(dev);
    e = hipMalloc(out, bytes);
  }
  return e;
}

void pool_free(void* p, size_t bytes, int dev) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_bytes + bytes <= POOL_CAP && g_pool.size() < 256) {
      g_pool.push_back({p, bytes, dev});
      g_pool_bytes += bytes;
      return;
    }
  }
  (void)hipFree(p);
}

// Streams and events are recycled too: hipStreamCreate / hipStreamDestroy cost about a
// millisecond each on this runtime, four of them per fit.  A parked stream is idle (it is
// synchronised before it is parked) and carries no state of the session that used it.
struct StreamEntry { hipStream_t s; int device; };
struct EventEntry { hipEvent_t e; int device; };
std::vector<StreamEntry> g_stream_pool;
std::vector<EventEntry> g_event_pool;

hipError_t pool_stream_get(hipStream_t* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
      if (g_stream_pool[i].device == dev) {
        *out = g_stream_pool[i].s;
        g_stream_pool[i] = g_stream_pool.back();
        g_stream_pool.pop_back();
        return hipSuccess;
      }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void pool_stream_put(hipStream_t st, int dev) {
  if (!st) return;
  if (hipStreamSynchronize(st) == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_stream_pool.size() < 64) { g_stream_pool.push_back({st, dev}); return; }
  }
  (void)hipStreamDestroy(st);
}

hipError_t pool_event_get(hipEvent_t* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_event_pool.size(); ++i)
      if (g_event_pool[i].device == dev) {
        *out = g_event_pool[i].e;
        g_event_pool[i] = g_event_pool.back();
        g_event_pool.pop_back();
        return hipSuccess;
      }
  }
  return hipEventCreate(out);
}

void pool_event_put(hipEvent_t ev, int dev) {
  if (!ev) return;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_event_pool.size() < 128) { g_event_pool.push_back({ev, dev}); return; }
  }
  (void)hipEventDestroy(ev);
}

// Pinned host buffers (ci_host_alloc) are recycled the same way: pinning 100 MB costs tens of
// milliseconds, several fits' worth.
struct HostEntry { void* p; size_t bytes; };
std::mutex g_host_mu;
std::vector<HostEntry> g_host_pool;       // parked (free) buffers
std::vector<HostEntry> g_host_live;       // handed out
size_t g_host_pool_bytes = 0;

template <class T> struct DevBuf;
// Releases the listed buffers when an entry point leaves through any path (HIP_TRY returns early).
template <class... B> struct BufGuard {
  std::tuple<B*...> bufs;
  explicit BufGuard(B*... b) : bufs(b...) {}
  ~BufGuard() { std::apply([](auto*... b) { (b->release(), ...); }, bufs); }
};

template <class T> struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int device = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) return hipSuccess;
    (void)hipGetDevice(&device);
    return pool_alloc((void**)&p, count * sizeof(T));
  }
  void release() {
    if (p) pool_free((void*)p, n * sizeof(T), device);
    p = nullptr;
  }
};




// Time-parallel trend + seasonal kernel (ci_wide.h): which instantiations exist.
void* pick_wide_kernel(int has_slope, int num_seasons) {
#define CI_WIDE_CASE(NS) \
  if (num_seasons == NS) return has_slope ? ci_gibbs_wide_fn_tr2_ns##NS() : ci_gibbs_wide_fn_tr1_ns##NS();
  CI_WIDE_CASE(2) CI_WIDE_CASE(3) CI_WIDE_CASE(4) CI_WIDE_CASE(5) CI_WIDE_CASE(6) CI_WIDE_CASE(7)
#undef CI_WIDE_CASE
  return nullptr;
}
// ... and its BIGP builds (53+ design columns): trend-only models (through the inert 2-season block)
// and trend + one block of 2-7 seasons, while the packed regression matrix and its index table fit in LDS
// (P <= ~150); wider designs, other block lists and very short series keep the general routes.
void* pick_wide_bigp_kernel(int has_slope, int num_seasons) {
#define CI_WIDEBIG_CASE(NS) \
  if (num_seasons == NS) return has_slope ? ci_gibbs_wide_bigp_fn_tr2_ns##NS() : ci_gibbs_wide_bigp_fn_tr1_ns##NS();
  CI_WIDEBIG_CASE(2) CI_WIDEBIG_CASE(3) CI_WIDEBIG_CASE(4) CI_WIDEBIG_CASE(5) CI_WIDEBIG_CASE(6) CI_WIDEBIG_CASE(7)
#undef CI_WIDEBIG_CASE
  return nullptr;
}
bool wide_bigp_ok(const ci_problem* pb) {
  if (pb->P <= ci::MAXP || pb->T < 64) return false;
  if (pb->flags & (CI_FLAG_SEQUENTIAL_SEASONAL | CI_FLAG_CLUSTER_SEASONAL | CI_FLAG_SEASONAL_WORKSPACE)) return false;
  if (!(pb->num_blocks == 0 || (pb->num_blocks == 1 && pb->num_seasons[0] >= 2 && pb->num_seasons[0] <= 7)))
    return false;
  const int d = (pb->has_slope ? 2 : 1) + (pb->num_blocks == 1 ? pb->num_seasons[0] - 1 : 1);
  return ci::make_wlayout(pb->P, d).total <= 160 * 1024 - 512;
}
bool use_wide(const ci_problem* pb) {
  if (pb->num_blocks == 1 && pb->P > ci::MAXP) return wide_bigp_ok(pb);
  return pb->num_blocks == 1 && pb->P <= ci::MAXP && !(pb->flags & CI_FLAG_SEQUENTIAL_SEASONAL) &&
         !(pb->flags & CI_FLAG_CLUSTER_SEASONAL) &&
         pick_wide_kernel(pb->has_slope, pb->num_seasons[0]) != nullptr;
}
int wide_steps_per_thread(int T) {
  int lc = (T + ci::NT - 1) / ci::NT;
  return (lc + 3) & ~3;
}

int steps_per_thread(int T) {
  for (int L = 1; L <= 16; L *= 2)
    if (ci::NT * L >= T) return L;
  return 0;
}

}  // namespace

namespace {
// Lower Cholesky factor of the prior covariance of x_0 in the (n-1)-effect coordinates:
// diag(level, [slope]) (+) sd^2 (I - 11'/n) per block   (SURVEY.md Appendix F); row-major [dr, dr].
std::vector<double> prior_chol_reduced_d(const ci_problem* pb, const ci_series_params& q, int dr,
                                         bool inert_blocks) {
  const int K = pb->num_blocks;
  std::vector<double> A((size_t)dr * dr, 0.0);
  A[0] = q.init_level_scale * q.init_level_scale;
  int o = 1;
  if (pb->has_slope) { A[(size_t)1 * dr + 1] = q.init_slope_scale * q.init_slope_scale; o = 2; }
  for (int k = 0; k < K; ++k) {
    const int n = pb->num_seasons[k];
    const double v = inert_blocks ? 0.0 : q.init_seasonal_scale * q.init_seasonal_scale;
    for (int i = 0; i < n - 1; ++i)
      for (int j = 0; j < n - 1; ++j)
        A[(size_t)(o + i) * dr + o + j] = v * ((i == j ? 1.0 : 0.0) - 1.0 / n);
    o += n - 1;
  }
  for (int j = 0; j < dr; ++j) {
    double sdiag = A[(size_t)j * dr + j];
    for (int k2 = 0; k2 < j; ++k2) sdiag -= A[(size_t)j * dr + k2] * A[(size_t)j * dr + k2];
    const double ljj = sdiag > 0.0 ? std::sqrt(sdiag) : 0.0;
    A[(size_t)j * dr + j] = ljj;
    for (int i = j + 1; i < dr; ++i) {
      double t2 = A[(size_t)i * dr + j];
      for (int k2 = 0; k2 < j; ++k2) t2 -= A[(size_t)i * dr + k2] * A[(size_t)j * dr + k2];
      A[(size_t)i * dr + j] = ljj > 0.0 ? t2 / ljj : 0.0;
    }
    for (int i = 0; i < j; ++i) A[(size_t)i * dr + j] = 0.0;
  }
  return A;
}
std::vector<float> prior_chol_reduced(const ci_problem* pb, const ci_series_params& q, int dr,
                                      bool inert_blocks) {
  const std::vector<double> A = prior_chol_reduced_d(pb, q, dr, inert_blocks);
  std::vector<float> out(A.size());
  for (size_t e = 0; e < A.size(); ++e) out[e] = (float)A[e];
  return out;
}

ci::DevSeriesParams dev_series_params(const ci_series_params& q, double n_obs) {
  ci::DevSeriesParams d;
  d.level_conc = q.level_conc; d.level_scale = q.level_scale; d.level_ub = q.level_ub;
  d.slope_conc = q.slope_conc; d.slope_scale = q.slope_scale; d.slope_ub = q.slope_ub;
  d.obs_conc = q.obs_conc; d.obs_scale = q.obs_scale; d.obs_ub = q.obs_ub;
  d.nonzero_prob = q.nonzero_prob;
  d.init_level_loc = q.init_level_loc; d.init_level_scale = q.init_level_scale;
  d.init_slope_scale = q.init_slope_scale;
  d.obs_scale0 = q.obs_scale0; d.level_scale0 = q.level_scale0; d.slope_scale0 = q.slope_scale0;
  d.n_obs = n_obs;
  return d;
}
}  // namespace

struct ci_session {
  ci_problem pb;
  int L = 0, x_in_lds = 0;
  bool eight_waves = false;    // dispatching to the eight-wave latency kernel (ci_kernels8.h)
  int sched_word = 0;          // $CI_SCHED_WORD, read and validated once at session creation (0: the kernel's default)
  size_t lds_bytes = 0;
  KernelFn fn = nullptr, fn_prof = nullptr, fn_prof8 = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf<float> y, Xt, o_obs, o_lscale, o_sscale, o_w, o_level, o_slope, o_pm, o_traj;
  DevBuf<uint8_t> mask;
  DevBuf<double> xtx, omega, wps;      // wps [B]: ci_series_params.weights_prior_scale
  DevBuf<ci::DevSeriesParams> sp;
  DevBuf<long long> prof;
  bool profile = false;
  // seasonal models
  int D_full = 0, dred = 0;
  DevBuf<uint8_t> season_change;
  DevBuf<ci::DevSeasonalParams> ssp;
  DevBuf<float> p1_chol, o_drift, o_seasonal;
  // time-parallel seasonal kernel
  bool wide = false;
  bool seasonal_gws = false;           // sequential seasonal kernel with its arrays over time in HBM
  size_t seasonal_ws_bytes = 0;
  int Lc = 0;
  DevBuf<float> ws;
  int cluster = 1;            // time-parallel seasonal kernel: workgroups per chain
  int dk_lds = 0;             //   its DK workers keep the draw's per-step rows in LDS (clusters of 16)
  bool tp = false;            // general seasonal models / trend + P > MAXP on ci_seasonal_tp.h
  size_t tp_ws_bytes = 0;     //   its per-chain HBM workspace
  DevBuf<int> csync;
  DevBuf<float> cpart, cw;
  DevBuf<double> cv;
  // on-device summarisation (ci_summary.h)
  DevBuf<double> s_value, s_cum, s_obs, s_order, s_draw;
  DevBuf<uint8_t> s_flags;
  DevBuf<int> s_ranks;
  bool ran = false;
  ci_problem kpb;          // what the kernel runs (== pb except for long trend-only series)
  bool inert_block = false;
  std::string kernel_name; // the Gibbs kernel this session dispatches to (as rocprofv3 names it)
  // streamed fetch (ci_session_run_streamed)
  hipStream_t copy_stream = nullptr;
  unsigned int* progress = nullptr;   // host-coherent pinned [B * C]
  int progress_every = 0;             // != 0 only while a streamed run is in flight
};

extern "C" {

const char* ci_last_error(void) { return g_err.c_str(); }

int ci_abi_version(void) { return CI_ABI_VERSION; }

void ci_series_stream_key(const uint32_t seed[2], int32_t series_id, uint32_t key[2]) {
  key[0] = ci::stream_key0(seed[0], 0, series_id);
  key[1] = ci::stream_key1(seed[1], 0, series_id);
}

int ci_device_count(int* count) {
  if (!count) return fail("count is NULL");
  HIP_TRY(hipGetDeviceCount(count));
  return 0;
}

int ci_device_synchronize(int device) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

int ci_host_alloc(void** ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail("ci_host_alloc: NULL pointer or zero size");
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (size_t i = 0; i < g_host_pool.size(); ++i)
      if (g_host_pool[i].bytes == bytes) {
        *ptr = g_host_pool[i].p;
        g_host_live.push_back(g_host_pool[i]);
        g_host_pool_bytes -= bytes;
        g_host_pool[i] = g_host_pool.back();
        g_host_pool.pop_back();
        return 0;
      }
  }
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  std::lock_guard<std::mutex> lk(g_host_mu);
  g_host_live.push_back({p, bytes});
  *ptr = p;
  return 0;
}

int ci_host_free(void* ptr) {
  if (!ptr) return 0;
  HostEntry e{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (size_t i = 0; i < g_host_live.size(); ++i)
      if (g_host_live[i].p == ptr) {
        e = g_host_live[i];
        g_host_live[i] = g_host_live.back();
        g_host_live.pop_back();
        break;
      }
    if (!e.p) return fail("ci_host_free: pointer was not allocated by ci_host_alloc");
    if (g_host_pool_bytes + e.bytes <= HOST_POOL_CAP && g_host_pool.size() < 64) {
      g_host_pool.push_back(e);
      g_host_pool_bytes += e.bytes;
      return 0;
    }
  }
  HIP_TRY(hipHostFree(e.p));
  return 0;
}

int ci_pool_trim(void) {
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (auto& he : g_host_pool) (void)hipHostFree(he.p);
    g_host_pool.clear();
    g_host_pool_bytes = 0;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  for (auto& pe : g_pool) { (void)hipSetDevice(pe.device); (void)hipFree(pe.p); }
  g_pool.clear();
  g_pool_bytes = 0;
  for (auto& se : g_stream_pool) { (void)hipSetDevice(se.device); (void)hipStreamDestroy(se.s); }
  g_stream_pool.clear();
  for (auto& ee : g_event_pool) { (void)hipSetDevice(ee.device); (void)hipEventDestroy(ee.e); }
  g_event_pool.clear();
  (void)hipSetDevice(dev);
  return 0;
}

static int validate(const ci_problem* pb) {
  if (!pb) return fail("problem is NULL");
  if (pb->abi_version != CI_ABI_VERSION)
    return fail("ABI mismatch: caller %d, library %d", pb->abi_version, CI_ABI_VERSION);
  if (pb->T < 3) return fail("T must be >= 3, got %d", pb->T);
  if (pb->P < 0 || pb->P > ci::MAXP_BIG)
    return fail("P must be in [0, %d], got %d", ci::MAXP_BIG, pb->P);
  if (pb->num_blocks < 0 || pb->num_blocks > CI_MAX_BLOCKS)
    return fail("num_blocks must be in [0, %d], got %d", CI_MAX_BLOCKS, pb->num_blocks);
  if (pb->num_blocks > 0) {
    int dfull = pb->has_slope ? 2 : 1;
    for (int k = 0; k < pb->num_blocks; ++k) {
      if (pb->num_seasons[k] < 2) return fail("num_seasons[%d] must be >= 2", k);
      dfull += pb->num_seasons[k];
    }
    if (use_wide(pb)) {
      if (wide_steps_per_thread(pb->T) > ci::WIDE_MAX_LC)
        return fail("T=%d exceeds the time-parallel seasonal path (max %d)", pb->T, ci::NT * ci::WIDE_MAX_LC);
    } else {
      if (dfull > 64) return fail("seasonal state too wide for one wavefront: %d > 64", dfull);
    }
  }
  if (pb->num_warmup < 0 || pb->num_results < 1) return fail("need num_warmup >= 0, num_results >= 1");
  if (pb->num_chains < 1 || pb->num_series < 1) return fail("need num_chains >= 1, num_series >= 1");
  if (pb->series_offset < 0 || pb->chain_offset < 0) return fail("series_offset and chain_offset must be >= 0");
  // (series ids enter the Philox key, chain ids the counter: no packing limit on either)
  if (pb->num_blocks == 0 && steps_per_thread(pb->T) == 0 &&
      wide_steps_per_thread(pb->T) > ci::WIDE_MAX_LC)
    return fail("T=%d exceeds the longest supported series (%d)", pb->T, ci::NT * ci::WIDE_MAX_LC);
  return 0;
}

int ci_session_destroy(ci_session* s);

// Releases a half-built session when ci_session_create leaves through an error path.
struct SessionGuard {
  ci_session* s;
  ~SessionGuard() { if (s) ci_session_destroy(s); }
};

int ci_session_create(const ci_problem* pb, const float* y, const uint8_t* mask, const float* X,
                      const uint8_t* season_change, const ci_series_params* params,
                      ci_session** out) {
  if (validate(pb)) return 1;
  if (!y || !mask || !params || !out) return fail("NULL argument");
  if (pb->num_blocks > 0 && !season_change) return fail("season_change is NULL but num_blocks > 0");
  if (pb->P > 0 && !X) return fail("X is NULL but P=%d", pb->P);
  HIP_TRY(hipSetDevice(pb->device));
  ci_session* s = new ci_session();
  SessionGuard guard{s};
  s->pb = *pb;
  const ci_problem* caller_pb = pb;
  // Trend-only series longer than the register-resident kernel holds (T > 4096) run on the
  // time-parallel kernel (ci_wide.h) with one INERT seasonal block: 2 seasons, zero initial
  // variance, no season changes => its effect is identically 0, the model and every random
  // number of the trend / regression draws are unchanged (the block's own drift-scale draw uses
  // its own Philox site).
  s->kpb = *pb;
  // More than MAXP design columns: every model runs on the sequential one-wavefront kernel, whose
  // regression block then keeps its O(P^2) arrays in a per-chain HBM workspace (any T as well).
  const bool bigp = pb->P > ci::MAXP;
  // (53+ columns, trend only: the BIGP build of that kernel, at any length -- wide_bigp_ok)
  const bool long_trend = pb->num_blocks == 0 && ((steps_per_thread(pb->T) == 0 && !bigp) || wide_bigp_ok(pb));
  std::vector<uint8_t> no_changes;
  if (long_trend) {
    s->kpb.num_blocks = 1;
    s->kpb.num_seasons[0] = 2;
    s->kpb.flags &= ~CI_FLAG_SEQUENTIAL_SEASONAL;
    no_changes.assign((size_t)pb->T, 0);
    season_change = no_changes.data();
  }
  pb = &s->kpb;
  const int T = pb->T, P = pb->P, B = pb->num_series, C = pb->num_chains, S = pb->num_results;
  const int D = pb->has_slope ? 2 : 1;
  const int K = pb->num_blocks;
  (void)caller_pb;
  if (K == 0 && !bigp) {
    s->L = steps_per_thread(T);
    // X lives in LDS when the whole layout fits in 160 KiB (leave room for a second block).
    const ci::LdsLayout with_x = ci::make_layout(P, D, ci::NT * s->L, 1);
    s->x_in_lds = (P > 0 && with_x.total <= 150 * 1024) ? 1 : 0;
    s->lds_bytes = ci::make_layout(P, D, ci::NT * s->L, s->x_in_lds).total;
    // P <= 16: the register-resident regression block, with the design in LDS or -- long series --
    // streamed from L2; beyond 16 columns the LDS block drawn by the whole workgroup
    const int pm = (P == 0) ? 0 : (P <= 16 ? (s->x_in_lds ? 1 : 3) : 2);
    s->fn = pick_kernel(D, s->L, pm);
    s->fn_prof = pick_kernel(D, s->L, pm + 8);       // instrumented variant (ci_session_profile)
    if (!s->fn || !s->fn_prof) return fail("no kernel for L=%d", s->L);
    char nm[96];
    snprintf(nm, sizeof(nm), "ci::gibbs_kernel<%d,%d,%d,false>", D, s->L, pm);
    s->kernel_name = nm;
    // latency build (ci_kernels8.h): eight wavefronts per chain -- four time waves, a regression
    // wave that owns the serial section and sweeps the next iteration's matrix during the draw,
    // three randomness waves.  Same draws, bit for bit (shared functions for everything that
    // rounds, -ffp-contract=on; tested across the CU-count boundary), so choosing by launch size
    // never changes a result.
    // ... when every chain has a compute unit to itself: eight 256-register wavefronts fill a CU,
    // the four-wave kernel leaves room for two workgroups, so launches with more workgroups than
    // CUs (batches of series: throughput, not latency) keep the four-wave kernel
    int num_cus = 256;
    (void)hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, pb->device);
    const bool latency_regime = (long long)B * C <= (long long)num_cus;
    if ((pm == 1 || pm == 3) && latency_regime && !(pb->flags & CI_FLAG_FOUR_WAVES)) {
      // the design in LDS when it fits beside the randomness buffers, else read from L2 (the
      // same sums in the same order: the same bits)
      size_t base8 = 0;
      KernelFn f8 = pick_kernel8(D, s->L, 0, 0, &base8);
      size_t lds8 = base8 + (size_t)P * ci::NT * s->L * sizeof(float);
      int xg = 0;
      if (!f8 || lds8 > 160 * 1024) {
        f8 = pick_kernel8(D, s->L, 0, 1, &base8);
        lds8 = base8;
        xg = 1;
      }
      if (f8 && lds8 <= 160 * 1024) {
        s->fn = f8;
        s->fn_prof8 = xg ? nullptr : pick_kernel8(D, s->L, 1, 0, nullptr);
        if (s->fn_prof8)
          HIP_TRY(hipFuncSetAttribute((const void*)s->fn_prof8,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
        s->eight_waves = true;
        // Experiment / test knob (tools/exp_sched.py, the schedule-independence test): a
        // replacement for the helper waves' schedule word.  Read ONCE here, never on the launch
        // path, and only a well-formed word is accepted -- a stray variable cannot silently change
        // the production schedule.
        if (const char* e_ = getenv("CI_SCHED_WORD")) {
          char* end_ = nullptr;
          const long v_ = strtol(e_, &end_, 0);
          if (end_ == e_ || *end_ != 0 || v_ < 0 || v_ > 0xFFFF)
            return fail("CI_SCHED_WORD must be an integer in [0, 65535], got '%s'", e_);
          s->sched_word = (int)v_;
        }
        s->lds_bytes = lds8;
        snprintf(nm, sizeof(nm), xg ? "ci::gibbs_kernel8<%d,%d,L2>" : "ci::gibbs_kernel8<%d,%d>", D, s->L);
        s->kernel_name = nm;
      }
    }
  } else {
    s->D_full = D;
    s->dred = D;
    for (int k = 0; k < K; ++k) { s->D_full += pb->num_seasons[k]; s->dred += pb->num_seasons[k] - 1; }
    s->wide = use_wide(pb);
    if (s->wide) {
      s->Lc = ci::wide_quad_steps(T);
      s->lds_bytes = ci::make_wlayout(P, s->dred).total;
      s->fn = (KernelFn)(bigp ? pick_wide_bigp_kernel(pb->has_slope, pb->num_seasons[0])
                              : pick_wide_kernel(pb->has_slope, pb->num_seasons[0]));
    } else {
      // arrays over time in LDS when the whole layout fits (fastest), else in a per-chain HBM
      // workspace: no bound on the series length, and room in LDS for the P > 16 regression block
      const ci::SLayout in_lds = ci::make_slayout(T, P, K, s->D_full, s->dred, pb->has_slope, 0);
      s->seasonal_gws = in_lds.total > 150 * 1024 || (pb->flags & CI_FLAG_SEASONAL_WORKSPACE) != 0;
      const ci::SLayout lay = ci::make_slayout(T, P, K, s->D_full, s->dred, pb->has_slope,
                                               s->seasonal_gws ? 1 : 0);
      s->lds_bytes = lay.total;
      s->seasonal_ws_bytes = ((lay.t_total + 255) & ~(size_t)255) + (bigp ? ci::bigp_workspace_bytes(P) : 0);
    }
    // Any other block list -- and trend models with more than MAXP design columns -- with a state
    // of at most 32 components: the TIME-PARALLEL kernel of ci_seasonal_tp.h, a cluster of up to 32
    // workgroups of 4 wavefronts per chain (one chunk of the series per wavefront).  The ROUTE is a
    // function of the model and the series alone (T, D, P) -- never of the launch size or the device's
    // CU count: the two seasonal kernels agree only up to float summation order, and a series of a
    // batch must reproduce fit_causalimpact on that series alone bit for bit, on any device.  The
    // launch size only decides how many real workgroups share a chain's chunks (same bits).
    // (measured, round 5: below ~110 steps the one-wavefront kernel is as fast or faster -- 158 us
    // against 171 us at T = 96 on the 4+7+6 model, 202 us against 174 us at T = 128)
    const int tp_min_t = P > ci::MAXP ? 64 : 112;
    if (!s->wide && s->D_full <= ci::TP_MAXD && T >= tp_min_t && !(pb->flags & CI_FLAG_SEQUENTIAL_SEASONAL) &&
        !(pb->flags & CI_FLAG_SEASONAL_WORKSPACE)) {
      int num_cus = 256;
      (void)hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, pb->device);
      const long long groups = ((long long)B * C + 7) / 8 * 8;
      const ci::TpLds tl = ci::make_tplds(P, s->D_full);
      if (tl.total <= 160 * 1024) {
        // the chunk grid (G virtual workgroups of TP_NWV chunks) depends on the series alone; the
        // launch size only decides how many real workgroups (Gc) share them: same bits either way
        int G = 1;
        while (G < ci::TP_MAXG && T / (2 * G * ci::TP_NWV) >= 16) G *= 2;
        int Gc = 1;
        if (!(pb->flags & CI_FLAG_NO_CLUSTER))
          while (Gc < G && groups * (2 * Gc) <= num_cus) Gc *= 2;
        s->tp = true;
        s->cluster = Gc;
        s->Lc = G;
        s->lds_bytes = tl.total;
        s->tp_ws_bytes = ci::make_tplayout(T, P, K, s->D_full, pb->has_slope, G).total;
        const int nq = ci::tp_nr(s->D_full) / 4;       // register rows of 4 nq columns
        typedef void* (*TpFn)(void);
        static const TpFn tp_fns[9] = {nullptr, nullptr, ci_gibbs_seasonal_tp_fn_nq2, ci_gibbs_seasonal_tp_fn_nq3,
                                       ci_gibbs_seasonal_tp_fn_nq4, ci_gibbs_seasonal_tp_fn_nq5,
                                       ci_gibbs_seasonal_tp_fn_nq6, ci_gibbs_seasonal_tp_fn_nq7,
                                       ci_gibbs_seasonal_tp_fn_nq8};
        s->fn = (KernelFn)tp_fns[nq]();
      }
    }
    if (s->lds_bytes > 160 * 1024) {
      return fail("seasonal model needs %zu bytes of LDS per chain (max 163840): fewer covariates "
                  "or a narrower seasonal state", s->lds_bytes);
    }
    if (!s->wide && !s->tp) s->fn = (KernelFn)ci_gibbs_seasonal_fn((s->seasonal_gws ? 1 : 0) | (bigp ? 2 : 0));
    char nm[96];
    if (s->tp) snprintf(nm, sizeof(nm), "ci::gibbs_seasonal_tp_kernel<%d> %d chunks x%d", ci::tp_nr(s->D_full) / 4,
                        s->Lc * ci::TP_NWV, s->cluster);
    else if (s->wide) snprintf(nm, sizeof(nm), bigp ? "ci::gibbs_wide_kernel<%d,%d,bigp>" : "ci::gibbs_wide_kernel<%d,%d>", D, pb->num_seasons[0]);
    else snprintf(nm, sizeof(nm), "ci::gibbs_seasonal_kernel<%s,%s>", s->seasonal_gws ? "true" : "false",
                  bigp ? "true" : "false");
    s->kernel_name = nm;
  }
  HIP_TRY(hipFuncSetAttribute((const void*)s->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)s->lds_bytes));
  if (s->fn_prof) {
    // the instrumented variant is always the four-wave kernel: its own LDS layout
    const size_t lds_prof = K == 0 ? ci::make_layout(P, D, ci::NT * s->L, s->x_in_lds).total : s->lds_bytes;
    HIP_TRY(hipFuncSetAttribute((const void*)s->fn_prof, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_prof));
  }
  HIP_TRY(pool_stream_get(&s->stream));
  HIP_TRY(pool_event_get(&s->ev0));
  HIP_TRY(pool_event_get(&s->ev1));

  const size_t BT = (size_t)B * T, BCS = (size_t)B * C * S;
  HIP_TRY(s->y.alloc(BT));
  HIP_TRY(s->mask.alloc(BT));
  HIP_TRY(s->Xt.alloc((size_t)B * P * T));
  HIP_TRY(s->xtx.alloc((size_t)B * P * P));
  HIP_TRY(s->omega.alloc((size_t)B * P * P));
  HIP_TRY(s->wps.alloc(B));
  HIP_TRY(s->sp.alloc(B));
  HIP_TRY(s->o_obs.alloc(BCS));
  HIP_TRY(s->o_lscale.alloc(BCS));
  HIP_TRY(s->o_sscale.alloc(BCS));
  HIP_TRY(s->o_w.alloc(BCS * P));
  HIP_TRY(s->o_level.alloc(BCS * T));
  HIP_TRY(s->o_slope.alloc(pb->has_slope ? BCS * T : 0));
  HIP_TRY(s->o_pm.alloc((size_t)B * C * T));
  HIP_TRY(s->o_traj.alloc(BCS * T));
  if (K > 0 || bigp) {
    HIP_TRY(s->season_change.alloc((size_t)K * T));
    HIP_TRY(s->ssp.alloc(B));
    HIP_TRY(s->p1_chol.alloc((size_t)B * s->dred * s->dred));
    s->inert_block = long_trend;
    if (!long_trend) {
      HIP_TRY(s->o_drift.alloc(BCS * K));
      HIP_TRY(s->o_seasonal.alloc(BCS * T * K));
    }
    if (s->wide) {
      HIP_TRY(s->ws.alloc((size_t)B * C * ci::wide_workspace_floats(s->dred, s->Lc)));
      // clusters: 8, 4 or 2 CUs per chain while every workgroup of the launch is resident at once
      // (the handshakes spin); needs whole 16-byte chunks of 4 steps and a regression block
      int num_cus = 256;
      (void)hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, pb->device);
      const long long groups = ((long long)B * C + 7) / 8 * 8;
      s->cluster = 1;
      if (!(pb->flags & CI_FLAG_NO_CLUSTER) && P > 0 && (T & 3) == 0) {
        if (groups * 16 <= num_cus) s->cluster = 16;
        else if (groups * 8 <= num_cus) s->cluster = 8;
        else if (groups * 4 <= num_cus) s->cluster = 4;
        else if (groups * 2 <= num_cus) s->cluster = 2;
      }
      if ((s->cluster == 16 || s->cluster == 8) && getenv("CI_WIDE_DK_GLOBAL") == nullptr) {
        // the draw's workers other than main are helpers whose LDS holds no regression matrices
        const size_t need = ci::make_wlayout(P, s->dred).big0 + ci::wide_dk_lds_bytes(s->Lc);
        if (need <= 160 * 1024 - 256) {
          s->dk_lds = 1;
          if (need > s->lds_bytes) {
            s->lds_bytes = need;
            HIP_TRY(hipFuncSetAttribute((const void*)s->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)s->lds_bytes));
          }
        }
      }
      const size_t nseg = ((size_t)(T >> 2) + ci::NT - 1) / ci::NT;
      const size_t RS = (size_t)(P > 16 ? P : 16) + 4;
      HIP_TRY(s->csync.alloc((size_t)B * C * ci::CL_INTS));
      HIP_TRY(s->cpart.alloc((size_t)B * C * (nseg > 0 ? nseg : 1) * ci::NW * RS));
      HIP_TRY(s->cw.alloc((size_t)B * C * ci::wide_cw_floats(P)));
      HIP_TRY(s->cv.alloc((size_t)B * C * ci::wide_cv_doubles(P)));
    }
    else if (s->tp) {
      HIP_TRY(s->ws.alloc((size_t)B * C * (s->tp_ws_bytes / sizeof(float))));
      HIP_TRY(s->csync.alloc((size_t)B * C * ci::TPC_INTS));
    }
    else if (s->seasonal_ws_bytes > 0) HIP_TRY(s->ws.alloc((size_t)B * C * (s->seasonal_ws_bytes / sizeof(float))));
    if (K > 0) HIP_TRY(hipMemcpy(s->season_change.p, season_change, (size_t)K * T, hipMemcpyHostToDevice));
    std::vector<ci::DevSeasonalParams> ssh(B);
    std::vector<float> ch((size_t)B * s->dred * s->dred, 0.f);
    for (int b = 0; b < B; ++b) {
      const ci_series_params& q = params[b];
      ssh[b].drift_conc = q.drift_conc; ssh[b].drift_scale = q.drift_scale; ssh[b].drift_ub = q.drift_ub;
      ssh[b].init_seasonal_scale = long_trend ? 0.0 : q.init_seasonal_scale;
      for (int k = 0; k < CI_MAX_BLOCKS; ++k) ssh[b].drift_scale0[k] = q.drift_scale0[k];
      if (long_trend) {      // the inert block's drift scale is drawn but never used
        ssh[b].drift_conc = 1.0; ssh[b].drift_scale = 1.0; ssh[b].drift_ub = 1.0;
        for (int k = 0; k < CI_MAX_BLOCKS; ++k) ssh[b].drift_scale0[k] = 0.0;
      }
      const std::vector<float> cf = prior_chol_reduced(pb, q, s->dred, long_trend);
      std::copy(cf.begin(), cf.end(), ch.begin() + (size_t)b * s->dred * s->dred);
    }
    HIP_TRY(hipMemcpy(s->ssp.p, ssh.data(), B * sizeof(ci::DevSeasonalParams), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->p1_chol.p, ch.data(), ch.size() * sizeof(float), hipMemcpyHostToDevice));
  }

  // host-side staging: zero masked outcomes, transpose X to feature-major, count observations
  std::vector<float> yh(BT);
  std::vector<ci::DevSeriesParams> sph(B);
  for (int b = 0; b < B; ++b) {
    double nobs = 0;
    for (int t = 0; t < T; ++t) {
      const size_t i = (size_t)b * T + t;
      const bool m = mask[i] != 0;
      yh[i] = m ? 0.f : y[i];
      if (!m) {
        nobs += 1;
        if (!std::isfinite(y[i])) return fail("y[%d,%d] is not finite but unmasked", b, t);
      }
    }
    sph[b] = dev_series_params(params[b], nobs);
  }
  HIP_TRY(hipMemcpy(s->y.p, yh.data(), BT * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->mask.p, mask, BT, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->sp.p, sph.data(), B * sizeof(ci::DevSeriesParams), hipMemcpyHostToDevice));
  {
    std::vector<double> wps(B);
    for (int b = 0; b < B; ++b) {
      if (!(params[b].weights_prior_scale > 0.0) || !std::isfinite(params[b].weights_prior_scale))
        return fail("params[%d].weights_prior_scale must be positive and finite (1 = the reference's prior)", b);
      wps[b] = params[b].weights_prior_scale;
    }
    HIP_TRY(hipMemcpy(s->wps.p, wps.data(), B * sizeof(double), hipMemcpyHostToDevice));
  }
  if (P > 0) {
    std::vector<float> xt((size_t)B * P * T);
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < T; ++t)
        for (int j = 0; j < P; ++j)
          xt[((size_t)b * P + j) * T + t] = X[((size_t)b * T + t) * P + j];
    HIP_TRY(hipMemcpy(s->Xt.p, xt.data(), xt.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  guard.s = nullptr;
  *out = s;
  return 0;
}

static int session_launch(ci_session* s) {
  const ci_problem& pb = s->pb;
  HIP_TRY(hipSetDevice(pb.device));
  ci::KArgs a;
  a.T = pb.T; a.P = pb.P; a.W = pb.num_warmup; a.S = pb.num_results; a.C = pb.num_chains;
  a.B = pb.num_series; a.chain_offset = pb.chain_offset;
  a.series_stream_base = (pb.flags & CI_FLAG_SHARED_SERIES_STREAMS) ? -1 : pb.series_offset;
  a.seed0 = pb.seed[0]; a.seed1 = pb.seed[1];
  a.x_in_lds = s->x_in_lds;
  a.y = s->y.p; a.mask = s->mask.p; a.Xt = s->Xt.p; a.xtx = s->xtx.p; a.omega = s->omega.p;
  a.sp = s->sp.p;
  a.out_obs = s->o_obs.p; a.out_level_scale = s->o_lscale.p; a.out_slope_scale = s->o_sscale.p;
  a.out_weights = s->o_w.p; a.out_level = s->o_level.p; a.out_slope = s->o_slope.p;
  a.out_pred_mean = s->o_pm.p; a.out_traj = s->o_traj.p;
  a.prof = nullptr;
  a.progress = s->progress_every > 0 ? s->progress : nullptr;
  a.progress_every = s->progress_every > 0 ? s->progress_every : 1;
  a.dbg = s->sched_word;
  if (s->profile) {
    if (!s->prof.p) HIP_TRY(s->prof.alloc(32));
    HIP_TRY(hipMemsetAsync(s->prof.p, 0, 32 * sizeof(long long), s->stream));
    a.prof = s->prof.p;
  }
  if (pb.P > 0) {
    hipLaunchKernelGGL(ci::setup_regression_kernel<float>, dim3(pb.num_series * pb.P * pb.P), dim3(64), 0,
                       s->stream, pb.T, pb.P, s->Xt.p, s->mask.p, s->wps.p, s->xtx.p, s->omega.p);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
  const ci_problem& kpb = s->kpb;
  if (kpb.num_blocks > 0 || kpb.P > ci::MAXP) {
    ci::SArgs sa;
    memset(&sa, 0, sizeof(sa));       // (lat_theta = NULL: the Gibbs sampler, not the latents-only mode)
    sa.k = a;
    sa.K = kpb.num_blocks; sa.has_slope = kpb.has_slope; sa.dred = s->dred;
    for (int k = 0; k < ci::SMAXK; ++k) sa.nseas[k] = k < kpb.num_blocks ? kpb.num_seasons[k] : 0;
    sa.season_change = s->season_change.p; sa.ssp = s->ssp.p; sa.p1_chol = s->p1_chol.p;
    sa.out_drift = s->o_drift.p; sa.out_seasonal = s->o_seasonal.p;
    sa.ws = s->ws.p; sa.Lc = s->Lc;
    sa.ws_stride = s->tp ? s->tp_ws_bytes : (s->wide ? 0 : s->seasonal_ws_bytes);
    sa.cluster = (s->wide || s->tp) ? s->cluster : 1;
    sa.cluster_drop = (pb.flags & CI_FLAG_TEST_DROP_HELPER) ? sa.cluster - 1 : 0;
    sa.dk_lds = s->wide ? s->dk_lds : 0;
    sa.csync = s->csync.p; sa.cpart = s->cpart.p; sa.cw = s->cw.p; sa.cv = s->cv.p;
    int grid = pb.num_series * pb.num_chains;
    if ((s->wide || s->tp) && s->cluster > 1) {
      HIP_TRY(hipMemsetAsync(s->csync.p, 0, s->csync.n * sizeof(int), s->stream));
      grid = (grid + 7) / 8 * 8 * s->cluster;       // (chain, role) <- workgroup id: see ci_wide.h
    }
    hipLaunchKernelGGL((void (*)(ci::SArgs))s->fn, dim3(grid), dim3(s->tp ? ci::TP_NT : (s->wide ? ci::NT : 64)),
                       s->lds_bytes, s->stream, sa);
  } else {
    if (s->profile && s->eight_waves && s->fn_prof8) {
      hipLaunchKernelGGL(s->fn_prof8, dim3(pb.num_series * pb.num_chains), dim3(ci::NT8), s->lds_bytes,
                         s->stream, a);
    } else if (s->profile && s->fn_prof) {
      // the instrumented variant is the four-wave kernel (its own LDS layout)
      const size_t lds4 = ci::make_layout(pb.P, pb.has_slope ? 2 : 1, ci::NT * s->L, s->x_in_lds).total;
      hipLaunchKernelGGL(s->fn_prof, dim3(pb.num_series * pb.num_chains), dim3(ci::NT), lds4,
                         s->stream, a);
    } else {
      hipLaunchKernelGGL(s->fn, dim3(pb.num_series * pb.num_chains),
                         dim3(s->eight_waves ? ci::NT8 : ci::NT), s->lds_bytes, s->stream, a);
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  return 0;
}

int ci_session_run(ci_session* s, float* kernel_ms) {
  if (!s) return fail("session is NULL");
  s->progress_every = 0;
  if (session_launch(s)) return 1;
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s->ev0, s->ev1));
  s->ran = true;
  return 0;
}

// The [B*C, S, T] arrays of a session, paired with the caller's buffers.
struct BigPair { float* dst; const float* src; size_t row; };

int ci_session_run_streamed(ci_session* s, ci_outputs* o, int32_t chunk_draws, float* kernel_ms) {
  if (!s || !o) return fail("NULL argument");
  if (chunk_draws < 1) return fail("chunk_draws must be >= 1, got %d", chunk_draws);
  const ci_problem& pb = s->pb;
  HIP_TRY(hipSetDevice(pb.device));
  const int S = pb.num_results, T = pb.T, K = pb.num_blocks;
  const size_t BC = (size_t)pb.num_series * pb.num_chains;
  if (!s->copy_stream) HIP_TRY(pool_stream_get(&s->copy_stream));
  // the register-resident kernel publishes its progress; the seasonal kernels do not (their
  // results are copied in the same chunks once the kernel has finished)
  const bool live = s->kpb.num_blocks == 0 && s->kpb.P <= ci::MAXP;
  if (live) {
    if (!s->progress)
      HIP_TRY(hipHostMalloc((void**)&s->progress, BC * sizeof(unsigned int), hipHostMallocCoherent));
    for (size_t i = 0; i < BC; ++i) s->progress[i] = 0u;
    s->progress_every = chunk_draws;
  } else {
    s->progress_every = 0;
  }
  if (session_launch(s)) { s->progress_every = 0; return 1; }
  s->progress_every = 0;
  std::vector<BigPair> big;
  if (o->level) big.push_back({o->level, s->o_level.p, (size_t)T});
  if (o->slope && pb.has_slope) big.push_back({o->slope, s->o_slope.p, (size_t)T});
  if (o->posterior_trajectories) big.push_back({o->posterior_trajectories, s->o_traj.p, (size_t)T});
  if (o->seasonal_levels && s->o_seasonal.n) big.push_back({o->seasonal_levels, s->o_seasonal.p, (size_t)T * K});
  int next = 0;
  while (next < S) {
    const int target = std::min(S, next + chunk_draws);
    if (live) {
      // wait until every chain has published `target` complete draws
      for (;;) {
        unsigned int lo = 0xFFFFFFFFu;
        volatile unsigned int* pr = s->progress;
        for (size_t i = 0; i < BC; ++i) lo = std::min(lo, (unsigned int)pr[i]);
        if (lo >= (unsigned int)target) break;
        const hipError_t q = hipStreamQuery(s->stream);
        if (q == hipSuccess) break;                 // kernel finished: everything is in HBM
        if (q != hipErrorNotReady) return fail("Gibbs kernel failed: %s", hipGetErrorString(q));
        __builtin_ia32_pause();
      }
    } else if (next == 0) {
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    for (const BigPair& b : big) {
      const size_t pitch = (size_t)S * b.row * sizeof(float);
      HIP_TRY(hipMemcpy2DAsync(b.dst + (size_t)next * b.row, pitch, b.src + (size_t)next * b.row, pitch,
                               (size_t)(target - next) * b.row * sizeof(float), BC,
                               hipMemcpyDeviceToHost, s->copy_stream));
    }
    next = target;
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s->ev0, s->ev1));
  s->ran = true;
  auto small = [&](float* dst, const DevBuf<float>& src) -> hipError_t {
    if (!dst || src.n == 0) return hipSuccess;
    return hipMemcpyAsync(dst, src.p, src.n * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream);
  };
  HIP_TRY(small(o->observation_noise_scale, s->o_obs));
  HIP_TRY(small(o->level_scale, s->o_lscale));
  HIP_TRY(small(o->slope_scale, s->o_sscale));
  HIP_TRY(small(o->weights, s->o_w));
  HIP_TRY(small(o->posterior_means, s->o_pm));
  HIP_TRY(small(o->seasonal_drift_scales, s->o_drift));
  HIP_TRY(hipStreamSynchronize(s->copy_stream));
  if (o->slope && !pb.has_slope) memset(o->slope, 0, s->o_level.n * sizeof(float));
  return 0;
}

// Order statistics of the rows of two [rows, N] matrices in one launch (M1 may be NULL: one
// matrix); out[(row / T) * R + r][row % T].  Rows of up to 16384 values (every fit_causalimpact
// shape: N = chains x draws) are selected from registers; longer rows by the L2 digit sweeps of
// summ_select_kernel.
#ifndef CI_SEL_NT
#define CI_SEL_NT 256
#endif
static hipError_t launch_select(hipStream_t stream, int N, int T, int rows, int R, const int* d_ranks,
                                const double* M0, const double* M1, double* out0, double* out1) {
  const int grid = M1 ? 2 * rows : rows;
  if (N <= 8192) {
    hipLaunchKernelGGL((ci::summ_select_reg_kernel<CI_SEL_NT, 8192 / CI_SEL_NT>), dim3(grid),
                       dim3(CI_SEL_NT), 0, stream, N, T, R, rows, d_ranks, M0, M1, out0, out1);
  } else if (N <= 16384) {
    hipLaunchKernelGGL((ci::summ_select_reg_kernel<512, 32>), dim3(grid), dim3(512), 0, stream,
                       N, T, R, rows, d_ranks, M0, M1, out0, out1);
  } else {
    hipLaunchKernelGGL(ci::summ_select_kernel, dim3(rows), dim3(256), 0, stream, N, T, R, d_ranks,
                       M0, out0);
    if (M1)
      hipLaunchKernelGGL(ci::summ_select_kernel, dim3(rows), dim3(256), 0, stream, N, T, R, d_ranks,
                         M1, out1);
  }
  return hipGetLastError();
}

int ci_session_summarize(ci_session* s, const double* scale, const double* shift,
                         const double* observed, const uint8_t* flags, int32_t num_ranks,
                         const int32_t* ranks, double* value_order, double* cum_order,
                         double* per_draw, double* per_draw_order) {
  if (!s || !scale || !shift || !observed || !flags || !ranks) return fail("NULL argument");
  if (!s->ran) return fail("ci_session_summarize needs a finished ci_session_run");
  const ci_problem& pb = s->pb;
  if (num_ranks < 1 || num_ranks > ci::SUMM_MAX_RANKS)
    return fail("num_ranks must be in [1, %d], got %d", ci::SUMM_MAX_RANKS, num_ranks);
  const int T = pb.T, N = pb.num_chains * pb.num_results, B = pb.num_series;
  for (int r = 0; r < num_ranks; ++r)
    if (ranks[r] < 0 || ranks[r] >= N) return fail("rank %d out of range [0, %d)", ranks[r], N);
  HIP_TRY(hipSetDevice(pb.device));
  const size_t BTN = (size_t)B * T * N;
  if (!s->s_value.p) {
    HIP_TRY(s->s_value.alloc(BTN));
    HIP_TRY(s->s_cum.alloc(BTN));
    HIP_TRY(s->s_obs.alloc((size_t)B * T + 2 * B));
    HIP_TRY(s->s_flags.alloc((size_t)B * T));
    HIP_TRY(s->s_ranks.alloc(ci::SUMM_MAX_RANKS));
    HIP_TRY(s->s_order.alloc((size_t)2 * B * ci::SUMM_MAX_RANKS * T));
    HIP_TRY(s->s_draw.alloc((size_t)B * 2 * N + (size_t)B * 2 * ci::SUMM_MAX_RANKS));
  }
  double* d_scale = s->s_obs.p + (size_t)B * T;
  double* d_shift = d_scale + B;
  HIP_TRY(hipMemcpyAsync(s->s_obs.p, observed, (size_t)B * T * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(d_scale, scale, B * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(d_shift, shift, B * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->s_flags.p, flags, (size_t)B * T, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->s_ranks.p, ranks, num_ranks * sizeof(int), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(ci::summ_transpose_kernel<float>, dim3((T + 63) / 64, (N + 63) / 64, B), dim3(64, 4), 0,
                     s->stream, N, T, s->o_traj.p, d_scale, d_shift, s->s_value.p);
  hipLaunchKernelGGL(ci::summ_cumsum_kernel, dim3((N + 63) / 64, B), dim3(64), 0, s->stream, N, T,
                     s->s_value.p, s->s_obs.p, s->s_flags.p, s->s_cum.p, s->s_draw.p);
  double* ord_value = s->s_order.p;
  double* ord_cum = s->s_order.p + (size_t)B * ci::SUMM_MAX_RANKS * T;
  HIP_TRY(launch_select(s->stream, N, T, B * T, num_ranks, s->s_ranks.p, s->s_value.p, s->s_cum.p,
                        ord_value, ord_cum));
  double* ord_draw = s->s_draw.p + (size_t)B * 2 * N;
  if (per_draw_order)
    HIP_TRY(launch_select(s->stream, N, 1, 2 * B, num_ranks, s->s_ranks.p, s->s_draw.p, nullptr,
                          ord_draw, nullptr));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t ord_bytes = (size_t)B * num_ranks * T * sizeof(double);
  if (value_order) HIP_TRY(hipMemcpy(value_order, ord_value, ord_bytes, hipMemcpyDeviceToHost));
  if (cum_order) HIP_TRY(hipMemcpy(cum_order, ord_cum, ord_bytes, hipMemcpyDeviceToHost));
  if (per_draw)
    HIP_TRY(hipMemcpy(per_draw, s->s_draw.p, (size_t)B * 2 * N * sizeof(double), hipMemcpyDeviceToHost));
  if (per_draw_order)
    HIP_TRY(hipMemcpy(per_draw_order, ord_draw, (size_t)B * 2 * num_ranks * sizeof(double),
                      hipMemcpyDeviceToHost));
  return 0;
}

extern "C++" {
template <class TIn>
static int summarize_draws_impl(int32_t device, int32_t num_draws, int32_t T, const TIn* trajectories,
                                double scale, double shift, const double* observed, const uint8_t* flags,
                                int32_t num_ranks, const int32_t* ranks, double* value_order,
                                double* cum_order, double* per_draw, double* per_draw_order) {
  if (!trajectories || !observed || !flags || !ranks) return fail("NULL argument");
  if (num_draws < 1 || T < 1) return fail("need num_draws >= 1 and T >= 1");
  if (num_ranks < 1 || num_ranks > ci::SUMM_MAX_RANKS)
    return fail("num_ranks must be in [1, %d], got %d", ci::SUMM_MAX_RANKS, num_ranks);
  const int N = num_draws;
  for (int r = 0; r < num_ranks; ++r)
    if (ranks[r] < 0 || ranks[r] >= N) return fail("rank %d out of range [0, %d)", ranks[r], N);
  HIP_TRY(hipSetDevice(device));
  const size_t TN = (size_t)T * N;
  DevBuf<TIn> d_traj;
  DevBuf<double> d_value, d_cum, d_obs, d_order, d_draw;
  DevBuf<uint8_t> d_flags;
  DevBuf<int> d_ranks;
  auto cleanup = [&]() {
    d_traj.release(); d_value.release(); d_cum.release(); d_obs.release(); d_order.release();
    d_draw.release(); d_flags.release(); d_ranks.release();
  };
#define CI_TRY_CLEAN(expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      cleanup();                                                                             \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                        \
  } while (0)
  CI_TRY_CLEAN(d_traj.alloc(TN));
  CI_TRY_CLEAN(d_value.alloc(TN));
  CI_TRY_CLEAN(d_cum.alloc(TN));
  CI_TRY_CLEAN(d_obs.alloc((size_t)T + 2));
  CI_TRY_CLEAN(d_flags.alloc(T));
  CI_TRY_CLEAN(d_ranks.alloc(ci::SUMM_MAX_RANKS));
  CI_TRY_CLEAN(d_order.alloc((size_t)2 * ci::SUMM_MAX_RANKS * T));
  CI_TRY_CLEAN(d_draw.alloc((size_t)2 * N + 2 * ci::SUMM_MAX_RANKS));
  const double ss[2] = {scale, shift};
  CI_TRY_CLEAN(hipMemcpy(d_traj.p, trajectories, TN * sizeof(TIn), hipMemcpyHostToDevice));
  CI_TRY_CLEAN(hipMemcpy(d_obs.p, observed, T * sizeof(double), hipMemcpyHostToDevice));
  CI_TRY_CLEAN(hipMemcpy(d_obs.p + T, ss, 2 * sizeof(double), hipMemcpyHostToDevice));
  CI_TRY_CLEAN(hipMemcpy(d_flags.p, flags, T, hipMemcpyHostToDevice));
  CI_TRY_CLEAN(hipMemcpy(d_ranks.p, ranks, num_ranks * sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ci::summ_transpose_kernel<TIn>, dim3((T + 63) / 64, (N + 63) / 64, 1), dim3(64, 4), 0,
                     0, N, T, d_traj.p, d_obs.p + T, d_obs.p + T + 1, d_value.p);
  hipLaunchKernelGGL(ci::summ_cumsum_kernel, dim3((N + 63) / 64, 1), dim3(64), 0, 0, N, T,
                     d_value.p, d_obs.p, d_flags.p, d_cum.p, d_draw.p);
  double* ord_value = d_order.p;
  double* ord_cum = d_order.p + (size_t)ci::SUMM_MAX_RANKS * T;
  CI_TRY_CLEAN(launch_select(0, N, T, T, num_ranks, d_ranks.p, d_value.p, d_cum.p, ord_value,
                             ord_cum));
  double* ord_draw = d_draw.p + (size_t)2 * N;
  if (per_draw_order)
    CI_TRY_CLEAN(launch_select(0, N, 1, 2, num_ranks, d_ranks.p, d_draw.p, nullptr, ord_draw,
                               nullptr));
  CI_TRY_CLEAN(hipGetLastError());
  CI_TRY_CLEAN(hipDeviceSynchronize());
  const size_t ord_bytes = (size_t)num_ranks * T * sizeof(double);
  if (value_order) CI_TRY_CLEAN(hipMemcpy(value_order, ord_value, ord_bytes, hipMemcpyDeviceToHost));
  if (cum_order) CI_TRY_CLEAN(hipMemcpy(cum_order, ord_cum, ord_bytes, hipMemcpyDeviceToHost));
  if (per_draw)
    CI_TRY_CLEAN(hipMemcpy(per_draw, d_draw.p, (size_t)2 * N * sizeof(double), hipMemcpyDeviceToHost));
  if (per_draw_order)
    CI_TRY_CLEAN(hipMemcpy(per_draw_order, ord_draw, (size_t)2 * num_ranks * sizeof(double),
                           hipMemcpyDeviceToHost));
#undef CI_TRY_CLEAN
  cleanup();
  return 0;
}
}  // extern "C++"

int ci_summarize_draws(int32_t device, int32_t num_draws, int32_t T, const float* trajectories,
                       double scale, double shift, const double* observed, const uint8_t* flags,
                       int32_t num_ranks, const int32_t* ranks, double* value_order,
                       double* cum_order, double* per_draw, double* per_draw_order) {
  return summarize_draws_impl<float>(device, num_draws, T, trajectories, scale, shift, observed, flags,
                                     num_ranks, ranks, value_order, cum_order, per_draw, per_draw_order);
}

int ci_summarize_draws_f64(int32_t device, int32_t num_draws, int32_t T, const double* trajectories,
                           double scale, double shift, const double* observed, const uint8_t* flags,
                           int32_t num_ranks, const int32_t* ranks, double* value_order,
                           double* cum_order, double* per_draw, double* per_draw_order) {
  return summarize_draws_impl<double>(device, num_draws, T, trajectories, scale, shift, observed, flags,
                                      num_ranks, ranks, value_order, cum_order, per_draw, per_draw_order);
}

int ci_session_fetch(ci_session* s, ci_outputs* o) {
  if (!s || !o) return fail("NULL argument");
  HIP_TRY(hipSetDevice(s->pb.device));
  auto get = [&](float* dst, const DevBuf<float>& src) -> hipError_t {
    if (!dst || src.n == 0) return hipSuccess;
    return hipMemcpy(dst, src.p, src.n * sizeof(float), hipMemcpyDeviceToHost);
  };
  HIP_TRY(get(o->observation_noise_scale, s->o_obs));
  HIP_TRY(get(o->level_scale, s->o_lscale));
  HIP_TRY(get(o->slope_scale, s->o_sscale));
  HIP_TRY(get(o->weights, s->o_w));
  HIP_TRY(get(o->level, s->o_level));
  HIP_TRY(get(o->posterior_means, s->o_pm));
  HIP_TRY(get(o->posterior_trajectories, s->o_traj));
  HIP_TRY(get(o->seasonal_drift_scales, s->o_drift));
  HIP_TRY(get(o->seasonal_levels, s->o_seasonal));
  if (o->slope) {
    if (s->pb.has_slope) HIP_TRY(get(o->slope, s->o_slope));
    else memset(o->slope, 0, s->o_level.n * sizeof(float));
  }
  return 0;
}

int ci_session_algorithmic_bytes(const ci_session* s, double* bytes) {
  if (!s || !bytes) return fail("NULL argument");
  const ci_problem& pb = s->pb;
  const double T = pb.T, P = pb.P, S = pb.num_results;
  const double chains = (double)pb.num_series * pb.num_chains;
  // per retained draw 4 T (d_out + 1) + 4 (P + 2 + slope + K), d_out = level
  // + slope + the seasonal latents the fit materialises (one per block: ci_outputs.seasonal_levels);
  // inputs once per chain: 4 T (P + 1) + T.
  const double K = pb.num_blocks;
  const double d_out = 1.0 + (pb.has_slope ? 1.0 : 0.0) + K;
  const double per_draw = 4.0 * T * (d_out + 1.0) + 4.0 * (P + 2.0 + (pb.has_slope ? 1.0 : 0.0) + K);
  const double per_chain = 4.0 * T * (P + 1.0) + T;
  *bytes = chains * (S * per_draw + per_chain);
  return 0;
}

static int copy_name(const std::string& name, char* buf, int32_t buflen) {
  if (!buf || buflen < 1) return fail("NULL / empty name buffer");
  snprintf(buf, (size_t)buflen, "%s", name.c_str());
  return 0;
}

int ci_session_kernel_name(const ci_session* s, char* buf, int32_t buflen) {
  if (!s) return fail("session is NULL");
  return copy_name(s->kernel_name, buf, buflen);
}

int ci_ll_session_kernel_name(const ci_ll_session* s, char* buf, int32_t buflen);

int ci_session_profile(ci_session* s, int enable, int64_t* cycles16) {
  if (!s) return fail("session is NULL");
  s->profile = enable != 0;
  if (cycles16) {
    if (!s->prof.p) { memset(cycles16, 0, 32 * sizeof(int64_t)); return 0; }
    HIP_TRY(hipMemcpy(cycles16, s->prof.p, 32 * sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  return 0;
}

int ci_session_destroy(ci_session* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->pb.device);
  s->y.release(); s->Xt.release(); s->o_obs.release(); s->o_lscale.release(); s->o_sscale.release();
  s->o_w.release(); s->o_level.release(); s->o_slope.release(); s->o_pm.release();
  s->o_traj.release(); s->mask.release(); s->xtx.release(); s->omega.release(); s->wps.release(); s->sp.release(); s->prof.release();
  s->season_change.release(); s->ssp.release(); s->p1_chol.release(); s->o_drift.release();
  s->o_seasonal.release(); s->ws.release(); s->csync.release(); s->cpart.release(); s->cw.release(); s->cv.release();
  s->s_value.release(); s->s_cum.release(); s->s_obs.release(); s->s_flags.release();
  s->s_ranks.release(); s->s_order.release(); s->s_draw.release();
  pool_event_put(s->ev0, s->pb.device);
  pool_event_put(s->ev1, s->pb.device);
  pool_stream_put(s->stream, s->pb.device);
  pool_stream_put(s->copy_stream, s->pb.device);
  if (s->progress) (void)hipHostFree(s->progress);
  delete s;
  return 0;
}

int ci_fit_gibbs(const ci_problem* pb, const float* y, const uint8_t* mask, const float* X,
                 const uint8_t* season_change, const ci_series_params* params, ci_outputs* outputs) {
  if (!outputs) return fail("outputs is NULL");
  ci_session* s = nullptr;
  if (ci_session_create(pb, y, mask, X, season_change, params, &s)) return 1;
  int rc = ci_session_run(s, nullptr);
  if (!rc) rc = ci_session_fetch(s, outputs);
  ci_session_destroy(s);
  return rc;
}

// ---- the float64 fit (ci_gibbs64.h): standalone, every buffer float64 -------------------------
int ci_fit_gibbs_f64(const ci_problem* pb, const double* y, const uint8_t* mask, const double* X,
                     const uint8_t* season_change, const ci_series_params* params,
                     ci_outputs_f64* o) {
  if (validate(pb)) return 1;
  if (!y || !mask || !params || !o) return fail("NULL argument");
  if (pb->num_blocks > 0 && !season_change) return fail("season_change is NULL but num_blocks > 0");
  if (pb->P > 0 && !X) return fail("X is NULL but P=%d", pb->P);
  const int T = pb->T, P = pb->P, B = pb->num_series, C = pb->num_chains, S = pb->num_results;
  const int K = pb->num_blocks, has_slope = pb->has_slope ? 1 : 0;
  int dfull = has_slope ? 2 : 1, dred = dfull;
  for (int k = 0; k < K; ++k) { dfull += pb->num_seasons[k]; dred += pb->num_seasons[k] - 1; }
  if (dfull > 64) return fail("seasonal state too wide for one wavefront: %d > 64", dfull);
  HIP_TRY(hipSetDevice(pb->device));
  // LDS first: arrays over time AND the regression block (P <= 32) when both fit, then the arrays
  // over time alone; else the HBM workspace for the arrays (regression still in LDS if small)
  int gws = 0, reg_lds = P <= 32 ? 1 : 0;
  if (ci::make_layout64(T, P, K, dfull, dred, has_slope, 0, reg_lds).total > 150 * 1024) {
    reg_lds = 0;
    if (ci::make_layout64(T, P, K, dfull, dred, has_slope, 0, 0).total > 150 * 1024) {
      gws = 1;
      reg_lds = P <= 32 ? 1 : 0;
    }
  }
  const ci::Layout64 lay = ci::make_layout64(T, P, K, dfull, dred, has_slope, gws, reg_lds);
  if (lay.total > 160 * 1024) return fail("float64 fit needs %zu bytes of LDS (max 163840)", lay.total);
  const size_t ws_stride = ci::gibbs64_ws_bytes(T, P, K, dfull, dred, has_slope, gws, reg_lds);
  const size_t BT = (size_t)B * T, BCS = (size_t)B * C * S;
  DevBuf<double> d_y, d_xt, d_xtx, d_om, d_wps, d_chol, o_obs, o_ls, o_ss, o_dr, o_w, o_lev, o_slp, o_sea,
      o_pm, o_tr;
  DevBuf<uint8_t> d_mask, d_sc, d_ws;
  DevBuf<ci::DevSeriesParams> d_sp;
  DevBuf<ci::DevSeasonalParams> d_ssp;
  auto cleanup = [&]() {
    d_y.release(); d_xt.release(); d_xtx.release(); d_om.release(); d_wps.release(); d_chol.release();
    o_obs.release(); o_ls.release(); o_ss.release(); o_dr.release(); o_w.release(); o_lev.release();
    o_slp.release(); o_sea.release(); o_pm.release(); o_tr.release(); d_mask.release(); d_sc.release();
    d_ws.release(); d_sp.release(); d_ssp.release();
  };
#define CI_TRY64(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      cleanup();                                                                             \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                        \
  } while (0)
  CI_TRY64(d_y.alloc(BT)); CI_TRY64(d_mask.alloc(BT)); CI_TRY64(d_xt.alloc((size_t)B * P * T));
  CI_TRY64(d_xtx.alloc((size_t)B * P * P)); CI_TRY64(d_om.alloc((size_t)B * P * P));
  CI_TRY64(d_wps.alloc(B)); CI_TRY64(d_sp.alloc(B)); CI_TRY64(d_ssp.alloc(B));
  CI_TRY64(d_chol.alloc((size_t)B * dred * dred)); CI_TRY64(d_sc.alloc((size_t)K * T));
  CI_TRY64(d_ws.alloc((size_t)B * C * ws_stride));
  CI_TRY64(o_obs.alloc(BCS)); CI_TRY64(o_ls.alloc(BCS)); CI_TRY64(o_ss.alloc(BCS));
  CI_TRY64(o_dr.alloc(BCS * K)); CI_TRY64(o_w.alloc(BCS * P)); CI_TRY64(o_lev.alloc(BCS * T));
  CI_TRY64(o_slp.alloc(has_slope ? BCS * T : 0)); CI_TRY64(o_sea.alloc(BCS * T * K));
  CI_TRY64(o_pm.alloc((size_t)B * C * T)); CI_TRY64(o_tr.alloc(BCS * T));
  {
    std::vector<double> yh(BT), wps(B), ch((size_t)B * dred * dred);
    std::vector<ci::DevSeriesParams> sph(B);
    std::vector<ci::DevSeasonalParams> ssh(B);
    for (int b = 0; b < B; ++b) {
      double nobs = 0;
      for (int t = 0; t < T; ++t) {
        const size_t i = (size_t)b * T + t;
        const bool m = mask[i] != 0;
        yh[i] = m ? 0.0 : y[i];
        if (!m) {
          nobs += 1;
          if (!std::isfinite(y[i])) { cleanup(); return fail("y[%d,%d] is not finite but unmasked", b, t); }
        }
      }
      const ci_series_params& q = params[b];
      if (!(q.weights_prior_scale > 0.0) || !std::isfinite(q.weights_prior_scale)) {
        cleanup();
        return fail("params[%d].weights_prior_scale must be positive and finite", b);
      }
      wps[b] = q.weights_prior_scale;
      sph[b] = dev_series_params(q, nobs);
      ssh[b].drift_conc = q.drift_conc; ssh[b].drift_scale = q.drift_scale; ssh[b].drift_ub = q.drift_ub;
      ssh[b].init_seasonal_scale = q.init_seasonal_scale;
      for (int k = 0; k < CI_MAX_BLOCKS; ++k) ssh[b].drift_scale0[k] = q.drift_scale0[k];
      const std::vector<double> cf = prior_chol_reduced_d(pb, q, dred, false);
      std::copy(cf.begin(), cf.end(), ch.begin() + (size_t)b * dred * dred);
    }
    CI_TRY64(hipMemcpy(d_y.p, yh.data(), BT * sizeof(double), hipMemcpyHostToDevice));
    CI_TRY64(hipMemcpy(d_mask.p, mask, BT, hipMemcpyHostToDevice));
    CI_TRY64(hipMemcpy(d_wps.p, wps.data(), B * sizeof(double), hipMemcpyHostToDevice));
    CI_TRY64(hipMemcpy(d_sp.p, sph.data(), B * sizeof(ci::DevSeriesParams), hipMemcpyHostToDevice));
    CI_TRY64(hipMemcpy(d_ssp.p, ssh.data(), B * sizeof(ci::DevSeasonalParams), hipMemcpyHostToDevice));
    CI_TRY64(hipMemcpy(d_chol.p, ch.data(), ch.size() * sizeof(double), hipMemcpyHostToDevice));
    if (K > 0) CI_TRY64(hipMemcpy(d_sc.p, season_change, (size_t)K * T, hipMemcpyHostToDevice));
    if (P > 0) {
      std::vector<double> xt((size_t)B * P * T);
      for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
          for (int j = 0; j < P; ++j)
            xt[((size_t)b * P + j) * T + t] = X[((size_t)b * T + t) * P + j];
      CI_TRY64(hipMemcpy(d_xt.p, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(ci::setup_regression_kernel<double>, dim3(B * P * P), dim3(64), 0, 0, T, P,
                         d_xt.p, d_mask.p, d_wps.p, d_xtx.p, d_om.p);
      CI_TRY64(hipGetLastError());
    }
  }
  ci::G64Args a;
  memset(&a, 0, sizeof(a));
  a.k.T = T; a.k.P = P; a.k.W = pb->num_warmup; a.k.S = S; a.k.C = C; a.k.B = B;
  a.k.chain_offset = pb->chain_offset;
  a.k.series_stream_base = (pb->flags & CI_FLAG_SHARED_SERIES_STREAMS) ? -1 : pb->series_offset;
  a.k.seed0 = pb->seed[0]; a.k.seed1 = pb->seed[1];
  a.k.y = d_y.p; a.k.mask = d_mask.p; a.k.Xt = d_xt.p; a.k.xtx = d_xtx.p; a.k.omega = d_om.p;
  a.k.sp = d_sp.p;
  a.k.out_obs = o_obs.p; a.k.out_level_scale = o_ls.p; a.k.out_slope_scale = o_ss.p;
  a.k.out_weights = o_w.p; a.k.out_level = o_lev.p; a.k.out_slope = o_slp.p;
  a.k.out_pred_mean = o_pm.p; a.k.out_traj = o_tr.p; a.k.prof = nullptr;
  a.K = K; a.has_slope = has_slope; a.dred = dred;
  for (int k = 0; k < ci::SMAXK; ++k) a.nseas[k] = k < K ? pb->num_seasons[k] : 0;
  a.season_change = d_sc.p; a.ssp = d_ssp.p; a.p1_chol = d_chol.p;
  a.out_drift = o_dr.p; a.out_seasonal = o_sea.p;
  a.ws = d_ws.p; a.ws_stride = ws_stride; a.lat_theta = nullptr; a.lat_S = 1; a.reg_lds = reg_lds;
  // CI_F64_PROF=1 (diagnostic): per-phase shader-clock totals of chain 0's thread 0 on stderr
  DevBuf<long long> d_prof;
  const bool want_prof = getenv("CI_F64_PROF") != nullptr;
  if (want_prof) {
    CI_TRY64(d_prof.alloc(32));
    CI_TRY64(hipMemset(d_prof.p, 0, 32 * sizeof(long long)));
    a.k.prof = d_prof.p;
  }
  {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CI_TRY64(hipEventCreate(&e0));
    CI_TRY64(hipEventCreate(&e1));
    (void)hipEventRecord(e0, 0);
    ci_launch_gibbs64(&a, B * C, lay.total, gws, 0);
    (void)hipEventRecord(e1, 0);
    const hipError_t le = hipGetLastError();
    const hipError_t se = hipDeviceSynchronize();
    g_f64_kernel_ms = 0.f;
    if (le == hipSuccess && se == hipSuccess) (void)hipEventElapsedTime(&g_f64_kernel_ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    CI_TRY64(le);
    CI_TRY64(se);
  }
  if (want_prof) {
    long long h[32];
    CI_TRY64(hipMemcpy(h, d_prof.p, sizeof(h), hipMemcpyDeviceToHost));
    const double it = (double)(pb->num_warmup + S);
    fprintf(stderr, "ci_fit_gibbs_f64 phases (cycles per iteration):");
    for (int i = 0; i < 32; ++i) if (h[i]) fprintf(stderr, " [%d] %.0f", i, (double)h[i] / it);
    fprintf(stderr, "\n");
    d_prof.release();
  }
  auto get = [&](double* dst, const DevBuf<double>& src) -> hipError_t {
    if (!dst || src.n == 0) return hipSuccess;
    return hipMemcpy(dst, src.p, src.n * sizeof(double), hipMemcpyDeviceToHost);
  };
  CI_TRY64(get(o->observation_noise_scale, o_obs)); CI_TRY64(get(o->level_scale, o_ls));
  CI_TRY64(get(o->slope_scale, o_ss)); CI_TRY64(get(o->seasonal_drift_scales, o_dr));
  CI_TRY64(get(o->weights, o_w)); CI_TRY64(get(o->level, o_lev));
  CI_TRY64(get(o->seasonal_levels, o_sea)); CI_TRY64(get(o->posterior_means, o_pm));
  CI_TRY64(get(o->posterior_trajectories, o_tr));
  if (o->slope) {
    if (has_slope) CI_TRY64(get(o->slope, o_slp));
    else memset(o->slope, 0, BCS * T * sizeof(double));
  }
#undef CI_TRY64
  cleanup();
  return 0;
}

int ci_fit_gibbs_f64_kernel_ms(float* kernel_ms) {
  if (!kernel_ms) return fail("NULL argument");
  *kernel_ms = g_f64_kernel_ms;
  return 0;
}

int ci_test_rng(int device, const uint32_t seed[2], uint32_t chain, uint32_t iter, uint32_t site,
                uint32_t sub, int32_t n, float* uniforms, float* normals, double alpha,
                double* gamma_draw) {
  if (n < 1 || n > 256) return fail("n must be in [1, 256]");
  HIP_TRY(hipSetDevice(device));
  DevBuf<float> du, dn;
  DevBuf<double> dg;
  BufGuard<DevBuf<float>, DevBuf<float>, DevBuf<double>> guard(&du, &dn, &dg);
  HIP_TRY(du.alloc(n));
  HIP_TRY(dn.alloc(2 * (size_t)n));
  HIP_TRY(dg.alloc(1));
  hipLaunchKernelGGL(ci::test_rng_kernel, dim3(1), dim3(256), 0, 0, seed[0], seed[1], chain, iter,
                     site, sub, n, du.p, dn.p, alpha, dg.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(uniforms, du.p, n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(normals, dn.p, 2 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(gamma_draw, dg.p, sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// draws per workgroup in the HMC fit's latent pass (their predictor sums stay in registers)
constexpr int HMC_LATENT_GROUP = 8;

namespace {
void launch_wide_score(int D, int ns, const ci::WideScoreArgs* a, hipStream_t st) {
#define CI_WS_CASE(NS) if (ns == NS) { if (D == 2) ci_launch_wide_score_tr2_ns##NS(a, st); else ci_launch_wide_score_tr1_ns##NS(a, st); return; }
  CI_WS_CASE(2) CI_WS_CASE(3) CI_WS_CASE(4) CI_WS_CASE(5) CI_WS_CASE(6) CI_WS_CASE(7)
#undef CI_WS_CASE
}
void launch_hmc_wide(int D, int ns, const ci::HmcWideArgs* a, hipStream_t st) {
#define CI_WH_CASE(NS) if (ns == NS) { if (D == 2) ci_launch_hmc_wide_tr2_ns##NS(a, st); else ci_launch_hmc_wide_tr1_ns##NS(a, st); return; }
  CI_WH_CASE(2) CI_WH_CASE(3) CI_WH_CASE(4) CI_WH_CASE(5) CI_WH_CASE(6) CI_WH_CASE(7)
#undef CI_WH_CASE
}
}  // namespace

struct ci_ll_session {
  int T = 0, P = 0, D = 1, L = 1, device = 0, max_evals = 0;
  float a1 = 0, p10 = 0, p11 = 0, p1e = 0;
  // seasonal blocks and / or T > 4096: the sequential one-wavefront route (ci_score_seq.h)
  bool seq = false;
  int K = 0, D_full = 1, nseas[CI_MAX_BLOCKS] = {0};
  DevBuf<uint8_t> season_change;
  DevBuf<float> seq_ws;
  size_t seq_ws_evals = 0;          // evaluations seq_ws has room for
  bool wide = false;                // ... on the time-parallel scans (ci_wide_score.h): d <= 8
  int wide_ns = 2, Lc = 0;
  int dred = 1;
  ci_problem spb;                   // the problem (geometry) for the latent pass
  DevBuf<ci::DevSeriesParams> d_sp;
  DevBuf<ci::DevSeasonalParams> d_ssp;
  DevBuf<float> p1_chol, lat_ws, h_seasonal, h_drift, h_loc;
  DevBuf<float> y, xt, level, slope, loc, traj;
  DevBuf<uint8_t> mask;
  DevBuf<double> theta, ll, grad;
  size_t draw_cap = 0;
  // on-device HMC (ci_hmc.h): the fit stays resident until ci_ll_session_hmc_fetch
  DevBuf<double> omega, h_draws, h_acc, h_eps, h_init;
  DevBuf<float> h_level, h_slope, h_part, h_traj, h_pm, h_obs, h_lscale, h_sscale, h_w;
  int h_C = 0, h_S = 0;
  bool h_ran = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  ci_series_params prm;
};

int ci_ll_session_destroy(ci_ll_session* s);
struct LlSessionGuard {
  ci_ll_session* s;
  ~LlSessionGuard() { if (s) ci_ll_session_destroy(s); }
};

int ci_ll_session_create2(const ci_problem* pb, const ci_series_params* params, const float* y,
                          const uint8_t* mask, const float* X, const uint8_t* season_change,
                          int32_t max_evals, ci_ll_session** out);

int ci_ll_session_create(const ci_problem* pb, const ci_series_params* params, const float* y,
                         const uint8_t* mask, const float* X, int32_t max_evals,
                         ci_ll_session** out) {
  if (pb && pb->num_blocks != 0)
    return fail("ci_ll_session_create: seasonal blocks need ci_ll_session_create2 (season_change)");
  return ci_ll_session_create2(pb, params, y, mask, X, nullptr, max_evals, out);
}

int ci_ll_session_create2(const ci_problem* pb, const ci_series_params* params, const float* y,
                          const uint8_t* mask, const float* X, const uint8_t* season_change,
                          int32_t max_evals, ci_ll_session** out) {
  if (validate(pb)) return 1;
  if (pb->num_blocks > 0 && !season_change) return fail("season_change is NULL but num_blocks > 0");
  if (pb->P > ci::HMC_MAXP) return fail("log-likelihood path: P must be <= %d, got %d", ci::HMC_MAXP, pb->P);
  if (!params || !y || !mask || !out || max_evals < 1) return fail("bad argument");
  if (!(params->weights_prior_scale > 0.0) || !std::isfinite(params->weights_prior_scale))
    return fail("params->weights_prior_scale must be positive and finite (1 = the reference's prior)");
  if (pb->P > 0 && !X) return fail("X is NULL but P=%d", pb->P);
  const bool seq = pb->num_blocks > 0 || steps_per_thread(pb->T) == 0;
  if (seq && pb->P > ci::MAXP)
    return fail("log-likelihood path, seasonal blocks or T > 4096: P must be <= %d, got %d", ci::MAXP, pb->P);
  int dfull = pb->has_slope ? 2 : 1;
  for (int k = 0; k < pb->num_blocks; ++k) dfull += pb->num_seasons[k];
  // trend + one block of 2-7 seasons (or a long trend-only series: an inert 2-season block) run on
  // the time-parallel scans of ci_wide_score.h; everything else sequentially (ci_score_seq.h)
  const bool wide_ll = seq && !(pb->flags & CI_FLAG_SEQUENTIAL_SEASONAL) &&
                       wide_steps_per_thread(pb->T) <= ci::WIDE_MAX_LC &&
                       (pb->num_blocks == 0 ||
                        (pb->num_blocks == 1 && pb->num_seasons[0] >= 2 && pb->num_seasons[0] <= 7));
  if (seq && !wide_ll && dfull > 64) return fail("seasonal state too wide for one wavefront: %d > 64", dfull);
  HIP_TRY(hipSetDevice(pb->device));
  ci_ll_session* s = new ci_ll_session();
  LlSessionGuard guard{s};
  s->T = pb->T; s->P = pb->P; s->D = pb->has_slope ? 2 : 1; s->L = seq ? 0 : steps_per_thread(pb->T);
  s->device = pb->device; s->max_evals = max_evals;
  s->seq = seq; s->K = pb->num_blocks; s->D_full = dfull;
  for (int k = 0; k < pb->num_blocks; ++k) s->nseas[k] = pb->num_seasons[k];
  s->p1e = (float)(params->init_seasonal_scale * params->init_seasonal_scale);
  HIP_TRY(pool_stream_get(&s->stream));
  HIP_TRY(pool_event_get(&s->ev0));
  HIP_TRY(pool_event_get(&s->ev1));
  HIP_TRY(pool_event_get(&s->ev2));
  s->a1 = (float)params->init_level_loc;
  s->p10 = (float)(params->init_level_scale * params->init_level_scale);
  s->p11 = (float)(params->init_slope_scale * params->init_slope_scale);
  const int T = s->T, P = s->P;
  HIP_TRY(s->y.alloc(T));
  HIP_TRY(s->mask.alloc(T));
  HIP_TRY(s->xt.alloc((size_t)P * T));
  const int K = pb->num_blocks;
  HIP_TRY(s->theta.alloc((size_t)max_evals * (3 + K + P)));
  HIP_TRY(s->ll.alloc(max_evals));
  HIP_TRY(s->grad.alloc((size_t)max_evals * (3 + K + P)));
  s->wide = wide_ll;
  s->wide_ns = pb->num_blocks == 1 ? pb->num_seasons[0] : 2;
  s->Lc = wide_ll ? wide_steps_per_thread(T) : 0;
  if (seq) {
    const size_t per_eval = wide_ll ? ci::wide_score_ws_floats(s->D + s->wide_ns - 1, s->Lc)
                                    : ci::seq_score_ws_floats(T, dfull);
    HIP_TRY(s->seq_ws.alloc((size_t)max_evals * per_eval));
    s->seq_ws_evals = (size_t)max_evals;
    s->spb = *pb;
    s->dred = dfull - K;
    double nobs = 0;
    for (int t = 0; t < T; ++t) nobs += mask[t] ? 0 : 1;
    const ci::DevSeriesParams dsp = dev_series_params(*params, nobs);
    ci::DevSeasonalParams dss;
    dss.drift_conc = params->drift_conc; dss.drift_scale = params->drift_scale;
    dss.drift_ub = params->drift_ub; dss.init_seasonal_scale = params->init_seasonal_scale;
    for (int k = 0; k < CI_MAX_BLOCKS; ++k) dss.drift_scale0[k] = params->drift_scale0[k];
    const std::vector<float> cf = prior_chol_reduced(pb, *params, s->dred, false);
    HIP_TRY(s->d_sp.alloc(1));
    HIP_TRY(s->d_ssp.alloc(1));
    HIP_TRY(s->p1_chol.alloc(cf.size()));
    HIP_TRY(hipMemcpy(s->d_sp.p, &dsp, sizeof(dsp), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_ssp.p, &dss, sizeof(dss), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->p1_chol.p, cf.data(), cf.size() * sizeof(float), hipMemcpyHostToDevice));
    if (K > 0) {
      HIP_TRY(s->season_change.alloc((size_t)K * T));
      HIP_TRY(hipMemcpy(s->season_change.p, season_change, (size_t)K * T, hipMemcpyHostToDevice));
    }
  }
  std::vector<float> yh(T);
  for (int t = 0; t < T; ++t) {
    yh[t] = mask[t] ? 0.f : y[t];
    if (!mask[t] && !std::isfinite(y[t])) return fail("y[%d] is not finite but unmasked", t);
  }
  HIP_TRY(hipMemcpy(s->y.p, yh.data(), T * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->mask.p, mask, T, hipMemcpyHostToDevice));
  s->prm = *params;
  if (P > 0) {
    std::vector<float> xt((size_t)P * T);
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < P; ++j) xt[(size_t)j * T + t] = X[(size_t)t * P + j];
    HIP_TRY(hipMemcpy(s->xt.p, xt.data(), xt.size() * sizeof(float), hipMemcpyHostToDevice));
    // Gaussian slab of the weights prior: Omega = 0.01 (X'X/2 + diag(X'X)/2) / T, all rows
    // (causalimpact_lib.py:451-453)
    std::vector<double> om((size_t)P * P, 0.0);
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j)
          om[(size_t)i * P + j] += (double)X[(size_t)t * P + i] * (double)X[(size_t)t * P + j];
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j)
        om[(size_t)i * P + j] = 0.01 * (i == j ? om[(size_t)i * P + j] : 0.5 * om[(size_t)i * P + j]) / T *
                                params->weights_prior_scale;
    HIP_TRY(s->omega.alloc((size_t)P * P));
    HIP_TRY(hipMemcpy(s->omega.p, om.data(), om.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  guard.s = nullptr;
  *out = s;
  return 0;
}

// The HMC fit of a session on the sequential route (seasonal blocks and / or T > 4096): the chain
// (hmc_seq_kernel, one workgroup per chain), then ONE launch of the sequential Gibbs kernel in its
// latents-only mode: a workgroup (one wavefront) per retained draw.
static int hmc_run_sequential(ci_ll_session* s, const ci_hmc_options* o, const double* init_theta,
                              float* kernel_ms) {
  const int P = s->P, C = o->num_chains, S = o->num_results, T = s->T, K = s->K;
  const size_t N = (size_t)C * S;
  const int has_slope = s->D == 2 ? 1 : 0;
  const int nsc = 2 + has_slope + K;
  const int dim = (o->prior == CI_HMC_PRIOR_HORSESHOE ? 3 * P + 2 : P) + nsc;
  if ((size_t)C > s->seq_ws_evals) {
    s->seq_ws.release();
    const size_t per_eval = s->wide ? ci::wide_score_ws_floats(s->D + s->wide_ns - 1, s->Lc)
                                    : ci::seq_score_ws_floats(T, s->D_full);
    HIP_TRY(s->seq_ws.alloc((size_t)C * per_eval));
    s->seq_ws_evals = (size_t)C;
  }
  if (init_theta) {
    if (s->h_init.n != (size_t)C * dim) { s->h_init.release(); HIP_TRY(s->h_init.alloc((size_t)C * dim)); }
    HIP_TRY(hipMemcpyAsync(s->h_init.p, init_theta, (size_t)C * dim * sizeof(double),
                           hipMemcpyHostToDevice, s->stream));
  }
  ci::HmcSeqArgs a;
  a.q.T = T; a.q.P = P; a.q.K = K; a.q.has_slope = has_slope; a.q.E = C;
  for (int k = 0; k < ci::SMAXK; ++k) a.q.nseas[k] = k < K ? s->nseas[k] : 0;
  a.q.y = s->y.p; a.q.mask = s->mask.p; a.q.Xt = s->xt.p; a.q.season_change = s->season_change.p;
  a.q.theta = nullptr; a.q.a1 = s->a1; a.q.p10 = s->p10; a.q.p11 = s->p11; a.q.p1e = s->p1e;
  a.q.out_ll = nullptr; a.q.out_grad = nullptr; a.q.ws = s->seq_ws.p;
  a.C = C; a.W = o->num_warmup; a.S = S; a.n_leap = o->num_leapfrog; a.chain_offset = o->chain_offset;
  a.prior_mode = o->prior; a.seed0 = o->seed[0]; a.seed1 = o->seed[1];
  a.omega = s->omega.p;
  const ci_series_params& q = s->prm;
  {
    int n = 0;
    a.ig_a[n] = q.obs_conc; a.ig_b[n] = q.obs_scale; a.init_log[n++] = std::log(q.obs_scale0);
    a.ig_a[n] = q.level_conc; a.ig_b[n] = q.level_scale; a.init_log[n++] = std::log(std::max(q.level_scale0, 1e-4));
    if (has_slope) {
      a.ig_a[n] = q.slope_conc; a.ig_b[n] = q.slope_scale; a.init_log[n++] = std::log(std::max(q.slope_scale0, 1e-4));
    }
    for (int k = 0; k < K; ++k) {
      a.ig_a[n] = q.drift_conc; a.ig_b[n] = q.drift_scale;
      a.init_log[n++] = std::log(std::max(q.drift_scale0[k], 1e-4));
    }
  }
  a.hs_scale0 = o->horseshoe_scale; a.target_accept = o->target_accept; a.eps0 = o->initial_step_size;
  a.init = init_theta ? s->h_init.p : nullptr;
  a.draws = s->h_draws.p; a.accept_rate = s->h_acc.p; a.step_size = s->h_eps.p;
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
  if (s->wide) {
    ci::HmcWideArgs wa;
    wa.h = a; wa.Lc = s->Lc; wa.ws = s->seq_ws.p;
    launch_hmc_wide(s->D, s->wide_ns, &wa, s->stream);
  } else {
    ci_launch_hmc_seq(&a, s->D_full, s->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  // ---- latent path + predictive trajectory of every retained draw
  const ci::SLayout in_lds = ci::make_slayout(T, P, K, s->D_full, s->dred, has_slope, 0);
  const bool gws = in_lds.total > 150 * 1024;
  const ci::SLayout lay = ci::make_slayout(T, P, K, s->D_full, s->dred, has_slope, gws ? 1 : 0);
  if (lay.total > 160 * 1024) return fail("latent pass needs %zu bytes of LDS (max 163840)", lay.total);
  const size_t ws_stride = (lay.t_total + 255) & ~(size_t)255;
  if (gws && s->lat_ws.n < N * (ws_stride / sizeof(float))) {
    s->lat_ws.release();
    HIP_TRY(s->lat_ws.alloc(N * (ws_stride / sizeof(float))));
  }
  ci::SArgs sa;
  memset(&sa, 0, sizeof(sa));
  ci::KArgs& k = sa.k;
  k.T = T; k.P = P; k.W = 0; k.S = 1; k.C = (int)N; k.B = 1; k.chain_offset = o->chain_offset;
  k.series_stream_base = -1; k.seed0 = o->seed[0]; k.seed1 = o->seed[1]; k.x_in_lds = 0;
  k.y = s->y.p; k.mask = s->mask.p; k.Xt = s->xt.p; k.xtx = nullptr; k.omega = nullptr; k.sp = s->d_sp.p;
  k.out_obs = s->h_obs.p; k.out_level_scale = s->h_lscale.p; k.out_slope_scale = s->h_sscale.p;
  k.out_weights = s->h_w.p; k.out_level = s->h_level.p; k.out_slope = s->h_slope.p;
  k.out_pred_mean = s->h_loc.p; k.out_traj = s->h_traj.p; k.prof = nullptr; k.progress = nullptr;
  k.progress_every = 1;
  sa.K = K; sa.has_slope = has_slope; sa.dred = s->dred;
  for (int kk = 0; kk < ci::SMAXK; ++kk) sa.nseas[kk] = kk < K ? s->nseas[kk] : 0;
  sa.season_change = s->season_change.p; sa.ssp = s->d_ssp.p; sa.p1_chol = s->p1_chol.p;
  sa.out_drift = s->h_drift.p; sa.out_seasonal = s->h_seasonal.p;
  sa.ws = s->lat_ws.p; sa.Lc = 0; sa.cluster = 1; sa.ws_stride = gws ? ws_stride : 0;
  sa.lat_theta = s->h_draws.p; sa.lat_S = S;
  void* fn = ci_gibbs_seasonal_fn(gws ? 1 : 0);
  HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
  hipLaunchKernelGGL((void (*)(ci::SArgs))fn, dim3((unsigned)N), dim3(64), lay.total, s->stream, sa);
  HIP_TRY(hipGetLastError());
  // per-chain mean of the noise-free predictor over the S draws (hmc_mean_kernel: groups of 1)
  hipLaunchKernelGGL(ci::hmc_mean_kernel, dim3((T + 255) / 256, C), dim3(256), 0, s->stream, C, S, S, T,
                     s->h_loc.p, s->h_pm.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev2, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) {
    HIP_TRY(hipEventElapsedTime(&kernel_ms[0], s->ev0, s->ev1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[1], s->ev1, s->ev2));
  }
  s->h_ran = true;
  return 0;
}

int ci_ll_session_hmc_run(ci_ll_session* s, const ci_hmc_options* o, const double* init_theta,
                          float* kernel_ms) {
  if (!s || !o) return fail("NULL argument");
  if (o->num_chains < 1 || o->num_results < 1 || o->num_warmup < 0 || o->num_leapfrog < 1)
    return fail("need num_chains >= 1, num_results >= 1, num_warmup >= 0, num_leapfrog >= 1");
  if (!(o->target_accept > 0.0 && o->target_accept < 1.0) || !(o->initial_step_size > 0.0))
    return fail("need 0 < target_accept < 1 and initial_step_size > 0");
  if (o->prior != CI_HMC_PRIOR_SLAB && o->prior != CI_HMC_PRIOR_HORSESHOE)
    return fail("prior must be CI_HMC_PRIOR_SLAB or CI_HMC_PRIOR_HORSESHOE, got %d", o->prior);
  if (o->prior == CI_HMC_PRIOR_HORSESHOE && !(o->horseshoe_scale > 0.0))
    return fail("horseshoe prior needs horseshoe_scale > 0");
  HIP_TRY(hipSetDevice(s->device));
  const int P = s->P, C = o->num_chains, S = o->num_results, T = s->T;
  const size_t N = (size_t)C * S;
  if (s->h_C != C || s->h_S != S) {
    s->h_draws.release(); s->h_acc.release(); s->h_eps.release();
    s->h_level.release(); s->h_slope.release(); s->h_part.release(); s->h_traj.release();
    s->h_pm.release(); s->h_obs.release(); s->h_lscale.release(); s->h_sscale.release();
    s->h_w.release();
    s->h_C = 0; s->h_S = 0;          // an allocation failing below must not leave a stale shape
    HIP_TRY(s->h_draws.alloc(N * (3 + s->K + P)));
    if (s->seq) {
      s->h_seasonal.release(); s->h_drift.release(); s->h_loc.release();
      HIP_TRY(s->h_seasonal.alloc(N * T * s->K));
      HIP_TRY(s->h_drift.alloc(N * s->K));
      HIP_TRY(s->h_loc.alloc(N * T));
    }
    HIP_TRY(s->h_acc.alloc(C));
    HIP_TRY(s->h_eps.alloc(C));
    HIP_TRY(s->h_level.alloc(N * T));
    HIP_TRY(s->h_slope.alloc(s->D == 2 ? N * T : 0));
    HIP_TRY(s->h_part.alloc((size_t)C * ((S + HMC_LATENT_GROUP - 1) / HMC_LATENT_GROUP) * T));
    HIP_TRY(s->h_traj.alloc(N * T));
    HIP_TRY(s->h_pm.alloc((size_t)C * T));
    HIP_TRY(s->h_obs.alloc(N));
    HIP_TRY(s->h_lscale.alloc(N));
    HIP_TRY(s->h_sscale.alloc(N));
    HIP_TRY(s->h_w.alloc(N * P));
    s->h_C = C; s->h_S = S;
  }
  s->h_ran = false;
  if (s->seq) return hmc_run_sequential(s, o, init_theta, kernel_ms);
  const int dim = ci::hmc_dim(P, s->D, o->prior);
  if (init_theta) {
    if (s->h_init.n != (size_t)C * dim) { s->h_init.release(); HIP_TRY(s->h_init.alloc((size_t)C * dim)); }
    HIP_TRY(hipMemcpyAsync(s->h_init.p, init_theta, (size_t)C * dim * sizeof(double),
                           hipMemcpyHostToDevice, s->stream));
  }
  ci::HmcArgs a;
  a.init = init_theta ? s->h_init.p : nullptr;
  {
    // tests only: the five-barrier driver of rounds 2-4, to compare bits with the fused one
    const char* e_ = getenv("CI_HMC_LEGACY_DRIVER");
    a.legacy_driver = (e_ && e_[0] == '1' && e_[1] == 0) ? 1 : 0;
  }
  // tools/exp_hmc_phases.py: phase cycles of chain 0 (s_memtime on its thread 0), printed to stderr
  DevBuf<long long> hprof;
  BufGuard<DevBuf<long long>> hprof_guard(&hprof);
  a.prof = nullptr;
  if (getenv("CI_HMC_PROF") != nullptr) {
    HIP_TRY(hprof.alloc(32));
    HIP_TRY(hipMemsetAsync(hprof.p, 0, 32 * sizeof(long long), s->stream));
    a.prof = hprof.p;
  }
  a.T = T; a.P = P; a.C = C; a.W = o->num_warmup; a.S = S; a.n_leap = o->num_leapfrog;
  a.chain_offset = o->chain_offset; a.seed0 = o->seed[0]; a.seed1 = o->seed[1];
  a.prior_mode = o->prior; a.hs_scale0 = o->horseshoe_scale;
  a.y = s->y.p; a.mask = s->mask.p; a.Xt = s->xt.p; a.omega = s->omega.p;
  const ci_series_params& q = s->prm;
  a.ig_a[0] = q.obs_conc; a.ig_b[0] = q.obs_scale;
  a.ig_a[1] = q.level_conc; a.ig_b[1] = q.level_scale;
  a.ig_a[2] = q.slope_conc; a.ig_b[2] = q.slope_scale;
  a.init_log[0] = std::log(q.obs_scale0);
  a.init_log[1] = std::log(std::max(q.level_scale0, 1e-4));
  a.init_log[2] = std::log(std::max(q.slope_scale0, 1e-4));
  a.a1 = s->a1; a.p10 = s->p10; a.p11 = s->p11;
  a.target_accept = o->target_accept; a.eps0 = o->initial_step_size;
  a.draws = s->h_draws.p; a.accept_rate = s->h_acc.p; a.step_size = s->h_eps.p;
  const int D = s->D, L = s->L;
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
#define CI_HMC_CASE(DD, LL) if (D == DD && L == LL) ci_launch_hmc_d##DD##_l##LL(&a, s->stream);
  CI_HMC_CASE(1, 1) CI_HMC_CASE(1, 2) CI_HMC_CASE(1, 4) CI_HMC_CASE(1, 8) CI_HMC_CASE(1, 16)
  CI_HMC_CASE(2, 1) CI_HMC_CASE(2, 2) CI_HMC_CASE(2, 4) CI_HMC_CASE(2, 8) CI_HMC_CASE(2, 16)
#undef CI_HMC_CASE
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  // latent path + posterior-predictive trajectory of every retained draw (one workgroup per
  // draw: C*S workgroups fill the chip), the per-chain predictor means and the float32 container
#define CI_LAT_CASE(DD, LL)                                                                       \
  if (D == DD && L == LL)                                                                         \
    ci_launch_latents_d##DD##_l##LL(T, P, (int)N, s->y.p, s->mask.p, s->xt.p, s->h_draws.p, s->a1, \
                                    s->p10, s->p11, o->seed[0], o->seed[1],                       \
                                    (uint32_t)o->chain_offset, 0u, S, HMC_LATENT_GROUP,           \
                                    s->h_level.p, s->h_slope.p, nullptr, s->h_traj.p,             \
                                    s->h_part.p, s->stream);
  CI_LAT_CASE(1, 1) CI_LAT_CASE(1, 2) CI_LAT_CASE(1, 4) CI_LAT_CASE(1, 8) CI_LAT_CASE(1, 16)
  CI_LAT_CASE(2, 1) CI_LAT_CASE(2, 2) CI_LAT_CASE(2, 4) CI_LAT_CASE(2, 8) CI_LAT_CASE(2, 16)
#undef CI_LAT_CASE
  hipLaunchKernelGGL(ci::hmc_mean_kernel, dim3((T + 255) / 256, C), dim3(256), 0, s->stream, C,
                     (S + HMC_LATENT_GROUP - 1) / HMC_LATENT_GROUP, S, T, s->h_part.p, s->h_pm.p);
  hipLaunchKernelGGL(ci::hmc_unpack_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s->stream,
                     (int)N, P, s->h_draws.p, s->h_obs.p, s->h_lscale.p, s->h_sscale.p, s->h_w.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev2, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) {
    HIP_TRY(hipEventElapsedTime(&kernel_ms[0], s->ev0, s->ev1));
    HIP_TRY(hipEventElapsedTime(&kernel_ms[1], s->ev1, s->ev2));
  }
  if (a.prof) {
    long long h[32];
    HIP_TRY(hipMemcpy(h, hprof.p, sizeof(h), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "ci hmc prof:");
    for (int i = 0; i < 32; ++i) std::fprintf(stderr, " %lld", h[i]);
    std::fprintf(stderr, "\n");
  }
  s->h_ran = true;
  return 0;
}

int ci_ll_session_hmc_fetch(ci_ll_session* s, double* draws, double* accept_rate, double* step_size,
                            ci_outputs* o) {
  if (!s) return fail("session is NULL");
  if (!s->h_ran) return fail("ci_ll_session_hmc_fetch needs a finished ci_ll_session_hmc_run");
  HIP_TRY(hipSetDevice(s->device));
  const int C = s->h_C;
  if (draws) HIP_TRY(hipMemcpy(draws, s->h_draws.p, s->h_draws.n * sizeof(double), hipMemcpyDeviceToHost));
  if (accept_rate) HIP_TRY(hipMemcpy(accept_rate, s->h_acc.p, C * sizeof(double), hipMemcpyDeviceToHost));
  if (step_size) HIP_TRY(hipMemcpy(step_size, s->h_eps.p, C * sizeof(double), hipMemcpyDeviceToHost));
  if (o) {
    auto get = [&](float* dst, const DevBuf<float>& src) -> hipError_t {
      if (!dst || src.n == 0) return hipSuccess;
      return hipMemcpy(dst, src.p, src.n * sizeof(float), hipMemcpyDeviceToHost);
    };
    HIP_TRY(get(o->observation_noise_scale, s->h_obs));
    HIP_TRY(get(o->level_scale, s->h_lscale));
    HIP_TRY(get(o->slope_scale, s->h_sscale));
    HIP_TRY(get(o->weights, s->h_w));
    HIP_TRY(get(o->level, s->h_level));
    HIP_TRY(get(o->posterior_means, s->h_pm));
    HIP_TRY(get(o->posterior_trajectories, s->h_traj));
    HIP_TRY(get(o->seasonal_drift_scales, s->h_drift));
    HIP_TRY(get(o->seasonal_levels, s->h_seasonal));
    if (o->slope) {
      if (s->D == 2) HIP_TRY(get(o->slope, s->h_slope));
      else memset(o->slope, 0, s->h_level.n * sizeof(float));
    }
  }
  return 0;
}

int ci_ll_session_kernel_name(const ci_ll_session* s, char* buf, int32_t buflen) {
  if (!s) return fail("session is NULL");
  char nm[64];
  if (s->wide) snprintf(nm, sizeof(nm), "ci::hmc_wide_kernel<%d,%d>", s->D, s->wide_ns);
  else if (s->seq) snprintf(nm, sizeof(nm), "ci::hmc_seq_kernel");
  else if (s->P > ci::MAXP) snprintf(nm, sizeof(nm), "ci::hmc_kernel<%d,%d,wide>", s->D, s->L);
  else snprintf(nm, sizeof(nm), "ci::hmc_kernel<%d,%d>", s->D, s->L);
  return copy_name(nm, buf, buflen);
}

int ci_ll_session_algorithmic_bytes(const ci_ll_session* s, double* bytes) {
  if (!s || !bytes) return fail("NULL argument");
  if (s->h_C < 1) return fail("no HMC fit has been configured on this session");
  // latent / trajectory draws are produced for every HMC draw, so
  // the per-draw figure is the Gibbs one: 4 T (d_out + 1) + 4 (P + 2 + slope); inputs once per chain.
  const double T = s->T, P = s->P, slope = s->D == 2 ? 1.0 : 0.0, K = s->K;
  const double per_draw = 4.0 * T * (1.0 + slope + K + 1.0) + 4.0 * (P + 2.0 + slope + K);
  const double per_chain = 4.0 * T * (P + 1.0) + T;
  *bytes = (double)s->h_C * ((double)s->h_S * per_draw + per_chain);
  return 0;
}

int ci_ll_session_eval(ci_ll_session* s, int32_t num_evals, const double* theta, double* loglik,
                       double* grad) {
  if (!s || !theta || !loglik) return fail("NULL argument");
  if (num_evals < 1 || num_evals > s->max_evals) return fail("num_evals out of range");
  HIP_TRY(hipSetDevice(s->device));
  const int T = s->T, P = s->P, D = s->D, L = s->L, E = num_evals;
  const int dimt = 3 + s->K + P;
  HIP_TRY(hipMemcpy(s->theta.p, theta, (size_t)E * dimt * sizeof(double), hipMemcpyHostToDevice));
  if (s->seq) {
    ci::SeqScoreArgs qa;
    qa.T = T; qa.P = P; qa.K = s->K; qa.has_slope = D == 2 ? 1 : 0; qa.E = E;
    for (int k = 0; k < ci::SMAXK; ++k) qa.nseas[k] = k < s->K ? s->nseas[k] : 0;
    qa.y = s->y.p; qa.mask = s->mask.p; qa.Xt = s->xt.p; qa.season_change = s->season_change.p;
    qa.theta = s->theta.p; qa.a1 = s->a1; qa.p10 = s->p10; qa.p11 = s->p11; qa.p1e = s->p1e;
    qa.out_ll = s->ll.p; qa.out_grad = grad ? s->grad.p : nullptr; qa.ws = s->seq_ws.p;
    if (s->wide) {
      ci::WideScoreArgs wa;
      wa.q = qa; wa.Lc = s->Lc; wa.ws = s->seq_ws.p;
      launch_wide_score(D, s->wide_ns, &wa, 0);
    } else {
      ci_launch_seq_score(&qa, s->D_full, 0);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(loglik, s->ll.p, E * sizeof(double), hipMemcpyDeviceToHost));
    if (grad) HIP_TRY(hipMemcpy(grad, s->grad.p, (size_t)E * dimt * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
  }
#define CI_LL_CASE(DD, LL)                                                                        \
  if (D == DD && L == LL) {                                                                       \
    if (grad)                                                                                     \
      ci_launch_llgrad_d##DD##_l##LL(T, P, E, s->y.p, s->mask.p, s->xt.p, s->theta.p, s->a1,      \
                                     s->p10, s->p11, s->ll.p, s->grad.p, 0);                      \
    else                                                                                          \
      ci_launch_loglik_d##DD##_l##LL(T, P, E, s->y.p, s->mask.p, s->xt.p, s->theta.p, s->a1,      \
                                     s->p10, s->p11, s->ll.p, 0);                                 \
  }
  CI_LL_CASE(1, 1) CI_LL_CASE(1, 2) CI_LL_CASE(1, 4) CI_LL_CASE(1, 8) CI_LL_CASE(1, 16)
  CI_LL_CASE(2, 1) CI_LL_CASE(2, 2) CI_LL_CASE(2, 4) CI_LL_CASE(2, 8) CI_LL_CASE(2, 16)
#undef CI_LL_CASE
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(loglik, s->ll.p, E * sizeof(double), hipMemcpyDeviceToHost));
  if (grad)
    HIP_TRY(hipMemcpy(grad, s->grad.p, (size_t)E * (3 + P) * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int ci_ll_session_draw_latents(ci_ll_session* s, int32_t num_draws, const double* theta,
                               const uint32_t seed[2], uint32_t rng_chain, uint32_t iter0,
                               float* level, float* slope, float* loc, float* traj) {
  if (!s || !theta || !seed || !level || !loc || !traj) return fail("NULL argument");
  if (s->seq) return fail("ci_ll_session_draw_latents: trend models with T <= 4096 only");
  if (num_draws < 1 || num_draws > s->max_evals) return fail("num_draws out of range");
  HIP_TRY(hipSetDevice(s->device));
  const int T = s->T, P = s->P, D = s->D, L = s->L, E = num_draws;
  const size_t need = (size_t)E * T;
  if (need > s->draw_cap) {
    s->level.release(); s->slope.release(); s->loc.release(); s->traj.release();
    HIP_TRY(s->level.alloc(need)); HIP_TRY(s->slope.alloc(need));
    HIP_TRY(s->loc.alloc(need)); HIP_TRY(s->traj.alloc(need));
    s->draw_cap = need;
  }
  HIP_TRY(hipMemcpy(s->theta.p, theta, (size_t)E * (3 + P) * sizeof(double), hipMemcpyHostToDevice));
#define CI_LAT_CASE(DD, LL)                                                                       \
  if (D == DD && L == LL)                                                                         \
    ci_launch_latents_d##DD##_l##LL(T, P, E, s->y.p, s->mask.p, s->xt.p, s->theta.p, s->a1,       \
                                    s->p10, s->p11, seed[0], seed[1], rng_chain, iter0, 0, 1,     \
                                    s->level.p, s->slope.p, s->loc.p, s->traj.p, nullptr, 0);
  CI_LAT_CASE(1, 1) CI_LAT_CASE(1, 2) CI_LAT_CASE(1, 4) CI_LAT_CASE(1, 8) CI_LAT_CASE(1, 16)
  CI_LAT_CASE(2, 1) CI_LAT_CASE(2, 2) CI_LAT_CASE(2, 4) CI_LAT_CASE(2, 8) CI_LAT_CASE(2, 16)
#undef CI_LAT_CASE
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(level, s->level.p, need * sizeof(float), hipMemcpyDeviceToHost));
  if (slope) {
    if (D == 2) HIP_TRY(hipMemcpy(slope, s->slope.p, need * sizeof(float), hipMemcpyDeviceToHost));
    else memset(slope, 0, need * sizeof(float));
  }
  HIP_TRY(hipMemcpy(loc, s->loc.p, need * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(traj, s->traj.p, need * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ci_ll_session_destroy(ci_ll_session* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->device);
  s->y.release(); s->xt.release(); s->mask.release(); s->theta.release(); s->ll.release();
  s->grad.release(); s->season_change.release(); s->seq_ws.release(); s->d_sp.release(); s->d_ssp.release(); s->p1_chol.release(); s->lat_ws.release(); s->h_seasonal.release(); s->h_drift.release(); s->h_loc.release(); s->level.release(); s->slope.release(); s->loc.release(); s->traj.release();
  s->omega.release(); s->h_draws.release(); s->h_acc.release(); s->h_eps.release(); s->h_init.release();
  s->h_level.release(); s->h_slope.release(); s->h_part.release(); s->h_traj.release();
  s->h_pm.release(); s->h_obs.release(); s->h_lscale.release(); s->h_sscale.release();
  s->h_w.release();
  pool_event_put(s->ev0, s->device);
  pool_event_put(s->ev1, s->device);
  pool_event_put(s->ev2, s->device);
  pool_stream_put(s->stream, s->device);
  delete s;
  return 0;
}

int ci_kalman_loglik(const ci_problem* pb, const ci_series_params* params, const float* y,
                     const uint8_t* mask, const float* X, int32_t num_evals, const double* theta,
                     double* loglik) {
  ci_ll_session* s = nullptr;
  if (ci_ll_session_create(pb, params, y, mask, X, num_evals, &s)) return 1;
  const int rc = ci_ll_session_eval(s, num_evals, theta, loglik, nullptr);
  ci_ll_session_destroy(s);
  return rc;
}

int ci_test_dk_draw(const ci_problem* pb, const ci_series_params* params, const float* resid,
                    const uint8_t* mask, const uint8_t* season_change, double obs_scale,
                    double level_scale, double slope_scale, const double* drift_scales,
                    uint32_t iter, float* out_latents) {
  (void)season_change; (void)drift_scales;
  if (validate(pb)) return 1;
  HIP_TRY(hipSetDevice(pb->device));
  const int T = pb->T, D = pb->has_slope ? 2 : 1, L = steps_per_thread(T);
  DevBuf<float> dres, dout;
  DevBuf<uint8_t> dmask;
  BufGuard<DevBuf<float>, DevBuf<float>, DevBuf<uint8_t>> guard(&dres, &dout, &dmask);
  HIP_TRY(dres.alloc(T));
  HIP_TRY(dmask.alloc(T));
  HIP_TRY(dout.alloc((size_t)T * D));
  HIP_TRY(hipMemcpy(dres.p, resid, T * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dmask.p, mask, T, hipMemcpyHostToDevice));
  const uint32_t chain = (uint32_t)pb->chain_offset;
const ci_series_params* q = params;
#define CI_DK_CASE(DD, LL)                                                                     \
  if (D == DD && L == LL)                                                                      \
    ci_launch_dk_d##DD##_l##LL(T, dres.p, dmask.p, (float)(obs_scale * obs_scale),             \
                               (float)level_scale, (float)slope_scale,                        \
                               (float)q->init_level_loc,                                       \
                               (float)(q->init_level_scale * q->init_level_scale),             \
                               (float)(q->init_slope_scale * q->init_slope_scale), pb->seed[0], \
                               pb->seed[1], chain, iter, dout.p);
  CI_DK_CASE(1, 1) CI_DK_CASE(1, 2) CI_DK_CASE(1, 4) CI_DK_CASE(1, 8) CI_DK_CASE(1, 16)
  CI_DK_CASE(2, 1) CI_DK_CASE(2, 2) CI_DK_CASE(2, 4) CI_DK_CASE(2, 8) CI_DK_CASE(2, 16)
#undef CI_DK_CASE
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_latents, dout.p, (size_t)T * D * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

#include "ci_comm.h"
