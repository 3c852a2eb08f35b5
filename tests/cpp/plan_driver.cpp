// host-only driver of the convolution planners (multipitch_architectures_amd/csrc/conv_plan.h): built with g++ alone by
// tests/test_cpu_plan.py -- no hipcc, no GPU.  Prints one line per (layer, pass) with the fields the test checks.
#include "../../multipitch_architectures_amd/csrc/conv_plan.h"

static void one(const char* name, int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw) {
  mpa_conv_desc d{B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw};
  const FwdPlan f = plan_fwd(B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw);
  printf("%s fwd ok=%d NB=%d PB=%d TH=%d TW=%d tilesY=%d tilesX=%d OH=%d OW=%d CK=%d coTiles=%d COT=%d lds=%zu KWS=%d KS=%d quad=%d\n", name,
         (int)f.ok, f.NB, f.PB, f.TH, f.TW, f.tilesY, f.tilesX, f.OH, f.OW, f.CK, f.coTiles, f.COT, f.lds_bytes, f.KWS, f.KS, f.quad);
  const BwdDataGeom g = bwd_data_geom(&d);
  if (g.ok) {
    const FwdPlan b = plan_bwd_data(&d, g);
    printf("%s dgrad ok=%d NB=%d PB=%d TH=%d TW=%d tilesY=%d tilesX=%d OH=%d OW=%d CK=%d coTiles=%d COT=%d lds=%zu KWS=%d KS=%d quad=%d\n", name,
           (int)b.ok, b.NB, b.PB, b.TH, b.TW, b.tilesY, b.tilesX, b.OH, b.OW, b.CK, b.coTiles, b.COT, b.lds_bytes, b.KWS, b.KS, b.quad);
  } else {
    printf("%s dgrad ok=0\n", name);
  }
  for (int mode = 0; mode < 2; ++mode) {          // head conv2 kernels (conv_head.hip)
    const HeadPlan h = plan_head(&d, mode);
    if (h.ok) printf("%s head%d ok=1 MT=%d WM=%d WN=%d CK=%d chunks=%d NT=%d rows=%d PXT=%d tilesP=%d XS=%d P=%d lds=%zu\n", name, mode,
                     h.MT, h.WM, h.WN, h.CK, h.nChunks, h.NT, h.Mrows, h.PXT, h.tilesP, h.XS, h.P, h.lds_bytes);
  }
  for (int mode = 2; mode < 4; ++mode) {          // tall (kh,1) filters on the same kernel
    int NG = 0;
    const HeadPlan h = plan_tall(&d, mode, &NG);
    if (h.ok) printf("%s tall%d ok=1 MT=%d WM=%d WN=%d CK=%d chunks=%d NT=%d NG=%d rows=%d PXT=%d tilesP=%d XS=%d P=%d lds=%zu\n", name,
                     mode - 2, h.MT, h.WM, h.WN, h.CK, h.nChunks, h.NT, NG, h.Mrows, h.PXT, h.tilesP, h.XS, h.P, h.lds_bytes);
  }
  {
    const HeadWgPlan h = plan_head_wgrad(&d);
    if (h.ok) printf("%s headwg ok=1 MT=%d coGroups=%d chGroups=%d S=%d NCS=%d SEG=%d NRB=%d RB=%d itemsPer=%ld XUs=%d DUs=%d lds=%zu\n",
                     name, h.MT, h.coGroups, h.chGroups, h.S, h.NCS, h.SEG, h.NRB, h.RB, h.itemsPer, h.XUs, h.DUs, h.lds_bytes);
    const FoldPlan f = plan_fold(Cout, kh, kw, sh, sw, H);
    if (f.ok) printf("%s foldf ok=1 C0=%d R=%d V=%d lds=0\n", name, f.C0, f.R, f.V);
    const FoldPlan fb = plan_fold(Cin, kh, kw, sh, sw, H);
    if (fb.ok) printf("%s foldb ok=1 C0=%d R=%d V=%d lds=0\n", name, fb.C0, fb.R, fb.V);
  }
  const Wg15Plan q = plan_wgrad15(&d);
  if (q.ok) printf("%s wgrad15 ok=1 ga=%d n32=%d has16=%d fold=%d S=%d TH=%d tilesY=%d lds=%zu\n", name, q.ga, q.n32, q.has16, q.fold_R,
                   q.S, q.TH, q.tilesY, q.lds_bytes);
  else {
    const WgPlan w = plan_wgrad(&d);
    printf("%s wgrad ok=%d NBC=%d NTW=%d TH=%d TW=%d tilesY=%d tilesX=%d S=%d ga=%d lds=%zu\n", name, (int)w.ok, w.NBC, w.NTW, w.TH, w.TW,
           w.tilesY, w.tilesX, w.S, w.ga, w.lds_bytes);
  }
}

int main() {
  one("upconv4b", 256, 16, 75, 216, 128, 15, 15, 1, 1, 7, 7);
  one("upconv4b_b32", 32, 16, 75, 216, 128, 15, 15, 1, 1, 7, 7);
  one("inc_a", 256, 6, 75, 216, 16, 15, 15, 1, 1, 7, 7);
  one("prefilt", 64, 70, 75, 216, 70, 15, 15, 1, 1, 7, 7);
  one("down2a", 256, 32, 18, 54, 64, 9, 9, 1, 1, 4, 4);
  one("down3a", 256, 64, 9, 27, 128, 5, 5, 1, 1, 2, 2);
  one("down4a", 256, 128, 4, 13, 128, 3, 3, 1, 1, 1, 1);
  one("conv2_80", 256, 128, 75, 216, 80, 3, 3, 1, 3, 1, 0);
  one("conv2_200", 256, 128, 75, 216, 200, 3, 3, 1, 3, 1, 0);
  one("conv2_80_b32", 32, 128, 75, 216, 80, 3, 3, 1, 3, 1, 0);
  one("conv2_80_b16", 16, 128, 75, 216, 80, 3, 3, 1, 3, 1, 0);
  one("conv3_T174", 64, 80, 174, 72, 50, 75, 1, 1, 1, 0, 0);
  one("conv3_T75", 64, 80, 75, 72, 50, 75, 1, 1, 1, 0, 0);
  one("strided_unsupported", 4, 8, 20, 20, 8, 3, 3, 2, 2, 1, 1);
  return 0;
}
