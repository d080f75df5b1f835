// ci_seasonal.hip -- object file holding the seasonal-model Gibbs kernel.
#include <hip/hip_runtime.h>

#include "ci_seasonal.h"

extern "C" void* ci_gibbs_seasonal_fn(void) { return (void*)(&ci::gibbs_seasonal_kernel); }
