// the 48- / 96-cout phase-store builds (NB = 3, 6) of conv_fwd_kernel.h (see there)
#include "conv_fwd_kernel.h"

int mpa_conv_fwd_launch_nb36(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s) {
  const FwdPlan pl = fwd_plan_of(L);
  switch (pl.NB) {
    case 3: return launch_fwd_nb3(pl, p, s);
    case 6: return launch_fwd_nb6(pl, p, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}
