// ci_seasonal.hip -- object file holding the seasonal-model Gibbs kernels: arrays over time in LDS
// (short series, fastest) or in the per-chain HBM workspace (any length, P up to 52).
#include <hip/hip_runtime.h>

#include "ci_seasonal.h"

extern "C" void* ci_gibbs_seasonal_fn(int global_ws) {
  return global_ws ? (void*)(&ci::gibbs_seasonal_kernel<true>) : (void*)(&ci::gibbs_seasonal_kernel<false>);
}
