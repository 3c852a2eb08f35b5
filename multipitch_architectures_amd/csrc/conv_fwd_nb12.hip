// cout tiles of 16 and 32 (NB = 1, 2) of conv_fwd_kernel.h (see there)
#include "conv_fwd_kernel.h"

int mpa_conv_fwd_launch_nb12(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s) {
  const FwdPlan pl = fwd_plan_of(L);
  switch (pl.NB) {
    case 1: return launch_fwd_nb<1>(pl, p, s);
    case 2: return launch_fwd_nb<2>(pl, p, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}
