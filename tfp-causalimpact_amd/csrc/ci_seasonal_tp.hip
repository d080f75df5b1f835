// Instantiations of the time-parallel general seasonal kernel (ci_seasonal_tp.h): one per width of
// the register rows (4 NQ = 8, 12, ..., 32 columns).  CI_TP_NQ selects ONE of them so that the seven
// builds compile in parallel (csrc/Makefile).
#include <hip/hip_runtime.h>

#include "ci_seasonal_tp.h"

#ifndef CI_TP_NQ
#error "compile with -DCI_TP_NQ=2..8"
#endif

#define CI_TP_CAT2(a, b) a##b
#define CI_TP_CAT(a, b) CI_TP_CAT2(a, b)

extern "C" void* CI_TP_CAT(ci_gibbs_seasonal_tp_fn_nq, CI_TP_NQ)(void) {
  return (void*)&ci::gibbs_seasonal_tp_kernel<CI_TP_NQ>;
}
