// Diagnostic switches of the planners and kernels, read from the environment ONCE per process (first use) into one
// struct -- host-only C++, no HIP types (conv_plan.h is also compiled by g++ alone in tests/test_cpu_plan.py).
// None of them is needed in production: with no MPA_* variable set every field has its neutral value and the planners
// take their own decisions.  Tests and scratch scripts that change a variable inside a running process call
// mpa_diag_reload() (C ABI) afterwards.
//
// The kernel-side debug switches (dbg_*: "stage once", "skip the MFMA loop" ... -- timing experiments that produce wrong
// results) only exist in -DMPA_DIAG builds (`python -m multipitch_architectures_amd.build --diag` -> libmpa_hip_diag.so,
// loaded when MPA_DIAG_LIB=1): in the release library MPA_DBG(p) is the constant 0 and the branches are compiled out.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef MPA_DIAG
#define MPA_DBG(p) ((p).dbg)
#else
#define MPA_DBG(p) 0
#endif

struct MpaDiag {
  // forward / backward-data planner (plan_fwd)
  int fwd_nb = 0, fwd_pb = 0;        // MPA_FWD_FORCE="NB,PB": restrict the tile search
  int fwd_th = 0, fwd_tw = 0;        // MPA_FWD_TILE="TH,TW": one pixel-tile shape (0 = any)
  bool fwd_no_minw = false;          // MPA_FWD_NO_MINW: <2,12,9> always as the one-workgroup-per-CU build
  double fwd_halo_nb1 = -1.0;        // MPA_FWD_HALO_NB1: halo weight of the cost model for 16-cout tiles (-1: the planner's own)
  int fwd_ks_max = 16;               // MPA_FWD_KS_MAX
  int fwd_ks_force = 0;              // MPA_FWD_KS_FORCE
  bool fold_off = false;             // MPA_FOLD_OFF: no cout-remainder fold for the 15x15 layers
  // generic backward-weight planner (plan_wgrad)
  int wg_variant = -1;               // MPA_WG_VARIANT: one wave-tile variant 0..4
  bool wg_costnorm_old = false;      // MPA_WG_COSTNORM=0
  int wg_txn = 0;                    // MPA_WG_TXN
  bool wg_no_ga = false, wg_force_ga = false;   // MPA_WG_GA = "0" / "force": dY from global memory never / whenever feasible
  bool wg_s_old = false;             // MPA_WG_S_OLD
  long wg_s = 0;                     // MPA_WG_S: slice count
  // 15x15 backward-weight planner (plan_wgrad15)
  bool wg15_lds_dy = false;          // MPA_WG15_LDS_DY
  bool wg15_nofold = false;          // MPA_WG15_NOFOLD
  bool wg15_s_old = false;           // MPA_WG15_S_OLD
  long wg15_s = 0;                   // MPA_WG15_S
  // head conv2 / tall conv3 planners
  bool head_off = false;             // MPA_HEAD_OFF
  long head_fwd_min_wgs = -1;        // MPA_HEAD_FWD_MIN_WGS (-1: the planner's own threshold)
  int head_ck = 0;                   // MPA_HEAD_CK
  bool tall_off = false;             // MPA_TALL_OFF
  long head_wg_s = 0;                // MPA_HEAD_WG_S
  bool head_wg_rolled = false;       // MPA_HEAD_WG_ROLLED
  // split-bf16 kernels
  int bfx_r = 0;                     // MPA_BFX_R
  int bfx_wg_s = 0;                  // MPA_BFX_WG_S
  bool attn_valu = false;            // MPA_ATTN_VALU: head dimension 16 on the VALU kernels instead of the MFMA ones
  // GEMM planner
  int gemm_variant = -1, gemm_splits = 0;   // MPA_GEMM_FORCE="variant,splits"
  int gemm_panel_wgs = 0;            // MPA_GEMM_PANEL_WGS: workgroups the panel kernel aims for
  bool gemm_no_mask_fuse = false;    // MPA_GEMM_NO_MASK_FUSE: mpa_gemm_masked as product + masking pass for every shape
  bool gemm_no_panel = false;        // MPA_GEMM_NO_PANEL: short-K products on gemm_kernel instead of gemm_panel_kernel
  // kernel debug switches (honoured by -DMPA_DIAG builds only)
  int dbg_fwd = 0, dbg_wg15 = 0, dbg_head = 0, dbg_bfx = 0;   // MPA_DEBUG_FWD, MPA_DEBUG_WG15, MPA_HEAD_DEBUG, MPA_BFX_DEBUG
};

inline MpaDiag mpa_diag_read() {
  MpaDiag g;
  auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
  auto set = [](const char* name) { return getenv(name) != nullptr; };
  if (const char* e = getenv("MPA_FWD_FORCE")) sscanf(e, "%d,%d", &g.fwd_nb, &g.fwd_pb);
  if (const char* e = getenv("MPA_FWD_TILE")) sscanf(e, "%d,%d", &g.fwd_th, &g.fwd_tw);
  if (const char* e = getenv("MPA_FWD_HALO_NB1")) g.fwd_halo_nb1 = atof(e);
  g.fwd_no_minw = set("MPA_FWD_NO_MINW");
  g.fwd_ks_max = (int)num("MPA_FWD_KS_MAX", 16);
  g.fwd_ks_force = (int)num("MPA_FWD_KS_FORCE", 0);
  g.fold_off = set("MPA_FOLD_OFF");
  g.wg_variant = (int)num("MPA_WG_VARIANT", -1);
  g.wg_costnorm_old = set("MPA_WG_COSTNORM") && num("MPA_WG_COSTNORM", 1) == 0;
  g.wg_txn = (int)num("MPA_WG_TXN", 0);
  if (const char* e = getenv("MPA_WG_GA")) { g.wg_no_ga = e[0] == '0'; g.wg_force_ga = e[0] == 'f'; }
  g.wg_s_old = set("MPA_WG_S_OLD");
  g.wg_s = num("MPA_WG_S", 0);
  g.wg15_lds_dy = set("MPA_WG15_LDS_DY");
  g.wg15_nofold = set("MPA_WG15_NOFOLD");
  g.wg15_s_old = set("MPA_WG15_S_OLD");
  g.wg15_s = num("MPA_WG15_S", 0);
  g.head_off = set("MPA_HEAD_OFF");
  g.head_fwd_min_wgs = num("MPA_HEAD_FWD_MIN_WGS", -1);
  if (set("MPA_HEAD_CK")) g.head_ck = num("MPA_HEAD_CK", 8) == 4 ? 4 : 8;
  g.tall_off = set("MPA_TALL_OFF");
  g.head_wg_s = num("MPA_HEAD_WG_S", 0);
  g.head_wg_rolled = set("MPA_HEAD_WG_ROLLED");
  g.attn_valu = set("MPA_ATTN_VALU");
  g.bfx_r = (int)num("MPA_BFX_R", 0);
  g.bfx_wg_s = (int)num("MPA_BFX_WG_S", 0);
  g.gemm_no_panel = set("MPA_GEMM_NO_PANEL");
  g.gemm_no_mask_fuse = set("MPA_GEMM_NO_MASK_FUSE");
  if (const char* e = getenv("MPA_GEMM_PANEL_WGS")) g.gemm_panel_wgs = atoi(e);
  if (const char* e = getenv("MPA_GEMM_FORCE")) {
    int v = -1, s = 0;
    if (sscanf(e, "%d,%d", &v, &s) == 2) { g.gemm_variant = v; g.gemm_splits = s; }
  }
  g.dbg_fwd = (int)num("MPA_DEBUG_FWD", 0);
  g.dbg_wg15 = (int)num("MPA_DEBUG_WG15", 0);
  g.dbg_head = (int)num("MPA_HEAD_DEBUG", 0);
  g.dbg_bfx = (int)num("MPA_BFX_DEBUG", 0);
  return g;
}

inline MpaDiag& mpa_diag_mutable() {
  static MpaDiag g = mpa_diag_read();
  return g;
}
inline const MpaDiag& mpa_diag() { return mpa_diag_mutable(); }
