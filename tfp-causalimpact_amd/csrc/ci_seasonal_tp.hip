// Instantiations of the time-parallel general seasonal kernel (ci_seasonal_tp.h): one per width of
// the register rows (8, 16, 24, 32 columns).  CI_TP_NCH selects ONE of them so that the four
// builds compile in parallel (csrc/Makefile).
#include <hip/hip_runtime.h>

#include "ci_seasonal_tp.h"

#ifndef CI_TP_NCH
#error "compile with -DCI_TP_NCH=1..4"
#endif

#define CI_TP_CAT2(a, b) a##b
#define CI_TP_CAT(a, b) CI_TP_CAT2(a, b)

extern "C" void* CI_TP_CAT(ci_gibbs_seasonal_tp_fn_nch, CI_TP_NCH)(void) {
  return (void*)&ci::gibbs_seasonal_tp_kernel<CI_TP_NCH>;
}
