// ci_wide.hip -- one (TR, NS) instantiation of the time-parallel trend + seasonal Gibbs kernel
// per object file.  Compile with -DCI_TR=<1|2> -DCI_NS=<seasons>.
#include <hip/hip_runtime.h>

#define CI_SEASONAL_DECL_ONLY
#include "ci_wide.h"

#define CI_CAT_(a, b, c, d) a##b##c##d
#define CI_CAT(a, b, c, d) CI_CAT_(a, b, c, d)

extern "C" void* CI_CAT(ci_gibbs_wide_fn_tr, CI_TR, _ns, CI_NS)(void) {
  return (void*)(&ci::gibbs_wide_kernel<CI_TR, CI_NS>);
}

// Row H on the same scans (ci_wide_score.h): log-likelihood / score of E parameter sets, and the
// HMC fit over it.
#include "ci_wide_score.h"

extern "C" void CI_CAT(ci_launch_wide_score_tr, CI_TR, _ns, CI_NS)(const ci::WideScoreArgs* a,
                                                                   hipStream_t stream) {
  constexpr int D = CI_TR + CI_NS - 1;
  const size_t lds = sizeof(float) * ci::wide_score_lds_floats<D>(a->q.P);
  (void)hipFuncSetAttribute((const void*)(&ci::wide_score_kernel<CI_TR, CI_NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ci::wide_score_kernel<CI_TR, CI_NS>), dim3(a->q.E), dim3(ci::NT), lds, stream, *a);
}

extern "C" void CI_CAT(ci_launch_hmc_wide_tr, CI_TR, _ns, CI_NS)(const ci::HmcWideArgs* a,
                                                                 hipStream_t stream) {
  constexpr int D = CI_TR + CI_NS - 1;
  const size_t lds = ci::hmc_wide_lds_bytes<D>(a->h.q.P);
  (void)hipFuncSetAttribute((const void*)(&ci::hmc_wide_kernel<CI_TR, CI_NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ci::hmc_wide_kernel<CI_TR, CI_NS>), dim3(a->h.C), dim3(ci::NT), lds, stream, *a);
}
