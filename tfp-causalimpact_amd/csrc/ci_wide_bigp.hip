// ci_wide_bigp.hip -- the BIGP build (53+ design columns) of the time-parallel trend + one-block Gibbs
// kernel, one (TR, NS) instantiation per object file.  Compile with -DCI_TR=<1|2> -DCI_NS=<2..7>
// (NS = 2 also carries trend-only models, through the inert block).
#include <hip/hip_runtime.h>

#define CI_SEASONAL_DECL_ONLY
#include "ci_wide.h"

#define CI_CAT_(a, b, c, d) a##b##c##d
#define CI_CAT(a, b, c, d) CI_CAT_(a, b, c, d)

extern "C" void* CI_CAT(ci_gibbs_wide_bigp_fn_tr, CI_TR, _ns, CI_NS)(void) {
  return (void*)(&ci::gibbs_wide_kernel<CI_TR, CI_NS, true>);
}
