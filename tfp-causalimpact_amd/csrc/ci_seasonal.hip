// ci_seasonal.hip -- object file holding the seasonal-model Gibbs kernels: arrays over time in LDS
// (short series, fastest) or in the per-chain HBM workspace (any length); P up to 52 with the
// regression block in LDS, beyond that (bit 1 of the selector) with it in the workspace too.
#include <hip/hip_runtime.h>

#include "ci_seasonal.h"
#include "ci_score_seq.h"
#include "ci_gibbs64.h"

extern "C" void* ci_gibbs_seasonal_fn(int which) {
  switch (which) {
    case 0: return (void*)(&ci::gibbs_seasonal_kernel<false, false>);
    case 1: return (void*)(&ci::gibbs_seasonal_kernel<true, false>);
    case 2: return (void*)(&ci::gibbs_seasonal_kernel<false, true>);
    default: return (void*)(&ci::gibbs_seasonal_kernel<true, true>);
  }
}

// E sequential log-likelihood / score evaluations (ci_score_seq.h), one wavefront each.
extern "C" void ci_launch_seq_score(const ci::SeqScoreArgs* args, int D, hipStream_t stream) {
  const size_t lds = ci::seq_score_lds_bytes(D, args->K);
  (void)hipFuncSetAttribute((const void*)(&ci::seq_score_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ci::seq_score_kernel, dim3(args->E), dim3(64), lds, stream, *args);
}

// The HMC fit over the sequential score (ci_score_seq.h): one workgroup per chain.
extern "C" void ci_launch_hmc_seq(const ci::HmcSeqArgs* args, int D, hipStream_t stream) {
  const size_t lds = ci::hmc_seq_lds_bytes(D, args->q.K);
  (void)hipFuncSetAttribute((const void*)(&ci::hmc_seq_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ci::hmc_seq_kernel, dim3(args->C), dim3(ci::NT), lds, stream, *args);
}

// The float64 Gibbs sampler (ci_gibbs64.h): one wavefront per chain; arrays over time in the HBM
// workspace (global_ws) or in LDS.
extern "C" void ci_launch_gibbs64(const ci::G64Args* args, int grid, size_t lds, int global_ws,
                                  hipStream_t stream) {
  if (args->K == 0 && args->lat_theta == nullptr) {
    // trend-only models: eight wavefronts per chain (gibbs64_trend_kernel)
    const void* fn = global_ws ? (const void*)(&ci::gibbs64_trend_kernel<true>)
                               : (const void*)(&ci::gibbs64_trend_kernel<false>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (global_ws) hipLaunchKernelGGL(ci::gibbs64_trend_kernel<true>, dim3(grid), dim3(ci::NT64), lds, stream, *args);
    else hipLaunchKernelGGL(ci::gibbs64_trend_kernel<false>, dim3(grid), dim3(ci::NT64), lds, stream, *args);
    return;
  }
  const void* fn = global_ws ? (const void*)(&ci::gibbs64_kernel<true>) : (const void*)(&ci::gibbs64_kernel<false>);
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (global_ws) hipLaunchKernelGGL(ci::gibbs64_kernel<true>, dim3(grid), dim3(64), lds, stream, *args);
  else hipLaunchKernelGGL(ci::gibbs64_kernel<false>, dim3(grid), dim3(64), lds, stream, *args);
}
