// cout tiles of 64 and 80 (NB = 4, 5) of conv_fwd_kernel.h (see there)
#include "conv_fwd_kernel.h"

int mpa_conv_fwd_launch_nb45(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s) {
  const FwdPlan pl = fwd_plan_of(L);
  switch (pl.NB) {
    case 4: return launch_fwd_nb<4>(pl, p, s);
    case 5: return launch_fwd_nb<5>(pl, p, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}
