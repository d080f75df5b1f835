// ci_inst.hip -- one (D, L) instantiation of the Gibbs kernel per object file so the
// instantiations build in parallel (make -j).  Compile with -DCI_D=<1|2> -DCI_L=<1|2|4|8|16>;
// the three regression modes (ci_kernels.h "PM") are instantiated together.
#include <hip/hip_runtime.h>

#include "ci_kernels.h"
#include "ci_kernels8.h"
#include "ci_hmc.h"

#ifndef CI_D
#error "define CI_D"
#endif
#ifndef CI_L
#error "define CI_L"
#endif

#define CI_CAT_(a, b, c, d) a##b##c##d
#define CI_CAT(a, b, c, d) CI_CAT_(a, b, c, d)

extern "C" {

// Returns the device-function handle of gibbs_kernel<CI_D, CI_L, pm[, profiled]>: pm + 8 selects the
// variant instrumented with per-phase cycle counters (ci_session_profile).
void* CI_CAT(ci_gibbs_fn_d, CI_D, _l, CI_L)(int pm) {
  if (pm == 0) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 0, false>);
  if (pm == 1) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 1, false>);
  if (pm == 2) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 2, false>);
  if (pm == 8) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 0, true>);
  if (pm == 9) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 1, true>);
  if (pm == 10) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 2, true>);
#if CI_L >= 8
  // pm = 3: register-resident regression block, design streamed from L2 -- only series long enough
  // that <= 16 columns of T floats overflow LDS reach it (L >= 8 steps per thread)
  if (pm == 3) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 3, false>);
  if (pm == 11) return (void*)(&ci::gibbs_kernel<CI_D, CI_L, 3, true>);
#endif
  return nullptr;
}

// The eight-wavefront latency build of the register-resident kernels (ci_kernels8.h).  xg = 0: the
// design copied to LDS (*lds_base = its LDS bytes without the design, + P * 256 * L * 4 for it);
// xg = 1: the design read from L2 (*lds_base = everything).
void* CI_CAT(ci_gibbs8_fn_d, CI_D, _l, CI_L)(int profiled, int xg, size_t* lds_base) {
  if (lds_base) *lds_base = ci::Lay8<CI_D, CI_L>::off_x;
  if (xg) {
#if CI_L >= 8
    return profiled ? nullptr : (void*)(&ci::gibbs_kernel8<CI_D, CI_L, false, true>);
#else
    return nullptr;
#endif
  }
#if CI_L <= 8
#if CI_D == 2 && CI_L == 4
  // the instrumented variant exists for the bench shape only (ci_session_profile)
  if (profiled) return (void*)(&ci::gibbs_kernel8<CI_D, CI_L, true, false>);
#endif
  return profiled ? nullptr : (void*)(&ci::gibbs_kernel8<CI_D, CI_L, false, false>);
#else
  return nullptr;      // T > 2048: no room for the design in LDS
#endif
}

// Launches the one-draw Durbin-Koopman test kernel on the default stream.
void CI_CAT(ci_launch_dk_d, CI_D, _l, CI_L)(int T, const float* resid, const uint8_t* mask, float H,
                                           float sig0, float sig1, float a1, float p10, float p11,
                                           uint32_t k0, uint32_t k1, uint32_t chain, uint32_t iter,
                                           float* out) {
  ci::DkModel<CI_D> md;
  md.H = H;
  md.sig.v[0] = sig0;
  md.a1 = ci::Vec<CI_D>{};
  md.a1.v[0] = a1;
  md.p1.v[0] = p10;
#if CI_D == 2
  md.sig.v[1] = sig1;
  md.p1.v[1] = p11;
#else
  (void)sig1; (void)p11;
#endif
  hipLaunchKernelGGL((ci::test_dk_kernel<CI_D, CI_L>), dim3(1), dim3(ci::NT), 0, 0, T, resid, mask,
                     md, k0, k1, chain, iter, out);
}

// Launches the log-likelihood kernel for E parameter sets on `stream`.
void CI_CAT(ci_launch_loglik_d, CI_D, _l, CI_L)(int T, int P, int E, const float* y,
                                               const uint8_t* mask, const float* Xt,
                                               const double* theta, float a1, float p10,
                                               float p11, double* out, hipStream_t stream) {
  hipLaunchKernelGGL((ci::loglik_kernel<CI_D, CI_L>), dim3(E), dim3(ci::NT), 0, stream, T, P, y,
                     mask, Xt, theta, a1, p10, p11, out);
}

void CI_CAT(ci_launch_llgrad_d, CI_D, _l, CI_L)(int T, int P, int E, const float* y,
                                               const uint8_t* mask, const float* Xt,
                                               const double* theta, float a1, float p10,
                                               float p11, double* out_ll, double* out_grad,
                                               hipStream_t stream) {
  const size_t lds = sizeof(float) * (3 * ci::NW * 16 + ci::NW * (P + 4));
  hipLaunchKernelGGL((ci::loglik_grad_kernel<CI_D, CI_L>), dim3(E), dim3(ci::NT), lds, stream, T, P,
                     y, mask, Xt, theta, a1, p10, p11, out_ll, out_grad);
}

void CI_CAT(ci_launch_latents_d, CI_D, _l, CI_L)(int T, int P, int E, const float* y,
                                                const uint8_t* mask, const float* Xt,
                                                const double* theta, float a1, float p10,
                                                float p11, uint32_t k0, uint32_t k1,
                                                uint32_t rng_chain, uint32_t iter0, int per_chain,
                                                int group, float* level, float* slope, float* loc,
                                                float* traj, float* loc_sum, hipStream_t stream) {
  // E rows; per_chain > 0: chains x ceil(per_chain / group) workgroups (see latents_kernel)
  const int grid = per_chain > 0 ? (E / per_chain) * ((per_chain + group - 1) / group) : E;
  hipLaunchKernelGGL((ci::latents_kernel<CI_D, CI_L>), dim3(grid), dim3(ci::NT), 0, stream, T, P, y,
                     mask, Xt, theta, a1, p10, p11, k0, k1, rng_chain, iter0, per_chain, group, E,
                     level, slope, loc, traj, loc_sum);
}

// Runs the on-device HMC fit: one workgroup per chain.
void CI_CAT(ci_launch_hmc_d, CI_D, _l, CI_L)(const ci::HmcArgs* args, hipStream_t stream) {
  ci::HmcArgs a = *args;
  const size_t with_x = ci::hmc_lds_bytes(a.P, ci::NT * CI_L);
  a.x_in_lds = (a.P > 0 && with_x <= 150 * 1024) ? 1 : 0;
  const size_t lds = a.x_in_lds ? with_x : ci::hmc_lds_bytes(a.P, 0);
  if (a.P > ci::MAXP) {
    (void)hipFuncSetAttribute((const void*)(&ci::hmc_kernel<CI_D, CI_L, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ci::hmc_kernel<CI_D, CI_L, true>), dim3(a.C), dim3(ci::NT), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)(&ci::hmc_kernel<CI_D, CI_L, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ci::hmc_kernel<CI_D, CI_L, false>), dim3(a.C), dim3(ci::NT), lds, stream, a);
  }
}

}  // extern "C"
