// What crosses the translation units of the forward / backward-data convolution (conv_fwd.hip <-> conv_fwd_nb*.hip).
#pragma once
#include "mpa_common.h"

struct ConvFwdParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int B, Cin, H, W, Cout, OH, OW, kh, kw, sh, sw, ph, pw;
  int TH, TW, tilesY, tilesX, CK, nChunks, IH, IW, LW, CHP, COT, COTP;
  int IN64, SL64;      // LDS words of the input tile / one filter slab, rounded up to multiples of 64
  int quad;            // 16-byte LDS-DMA staging of the input tile (window origin rounded down to a multiple of 4)
  int dbg;             // diagnostics (env MPA_DEBUG_FWD): 1 = stage only once, 2 = skip the MFMA loops
  int act;
  float slope;
  long outBS, outCS;   // output batch / channel strides (floats)
  int outRS, outXmul, outCdiv;
  int outYmul, outH;   // phase stores: cout' = cin*(outXmul*outYmul) + v*outXmul + q -> row oy*outYmul+v (< outH), column ox*outXmul+q
  int chunksPer;       // input-channel chunks per blockIdx.z slice (== nChunks when the channels are not split)
  int coTiles, nTilesAll;   // cout tiles; pixel tiles over the whole batch
  float* stats;             // BatchNorm fusion: per-(pixel tile, cout) partial sums of y and y^2 -> [nTilesAll][Cout][2]
};

struct MpaFwdLaunch {              // the FwdPlan fields the launch ladder reads (conv_plan.h's types have internal linkage)
  int NB, PB, KWS, coTiles, nChunks;
  size_t lds_bytes;
};

int mpa_conv_fwd_launch_nb12(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s);
int mpa_conv_fwd_launch_nb45(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s);
int mpa_conv_fwd_launch_nb36(MpaFwdLaunch L, const ConvFwdParams& p, hipStream_t s);
