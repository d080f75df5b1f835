// ci_seasonal.hip -- object file holding the seasonal-model Gibbs kernels: arrays over time in LDS
// (short series, fastest) or in the per-chain HBM workspace (any length); P up to 52 with the
// regression block in LDS, beyond that (bit 1 of the selector) with it in the workspace too.
#include <hip/hip_runtime.h>

#include "ci_seasonal.h"

extern "C" void* ci_gibbs_seasonal_fn(int which) {
  switch (which) {
    case 0: return (void*)(&ci::gibbs_seasonal_kernel<false, false>);
    case 1: return (void*)(&ci::gibbs_seasonal_kernel<true, false>);
    case 2: return (void*)(&ci::gibbs_seasonal_kernel<false, true>);
    default: return (void*)(&ci::gibbs_seasonal_kernel<true, true>);
  }
}
