// Experiment knobs (include/tzrec_hip.h: tzr_tune).  Defaults are chosen inside each launcher.
#include <string.h>

#include "tzr_common.h"

extern int g_tzr_fwd_tile_b;
extern int g_tzr_fwd_variant;
extern int g_tzr_bwd_force_prep;
extern int g_tzr_bwd_ch;
extern int g_tzr_bwd_one_wg_heavy;
extern int g_tzr_bwd_debug;
extern int g_tzr_bwd_apply_waves;
extern int g_tzr_bwd_apply_fast;
extern int g_tzr_bwd_no_fuse_sort;
extern int g_tzr_bwd_scan_slices;
extern int g_tzr_bwd_direct_ch;
extern int g_tzr_bwd_direct;
extern int g_tzr_fwd_plan;
extern int g_tzr_fwd_plan_order;
extern int g_tzr_bwd_direct_debug;
extern int g_tzr_bwd_direct_hot;
extern int g_tzr_ia_bwd_plain;
extern int g_tzr_ia_bwd_wgs;
extern int g_tzr_ia_gen_wgs;
extern int g_tzr_ia_fwd_wgs;
extern int g_tzr_it_wgs;
extern int g_tzr_it_stagger;
extern int g_tzr_wg_debug;
extern int g_tzr_it_fwd_stagger;
extern int g_tzr_mlp_mfma;
extern int g_tzr_linear_bwd_wg;
extern int g_tzr_gemm_rows_wg;

extern "C" int tzr_tune(const char* name, int value) {
  if (!name) return TZR_ERR_INVALID;
  if (!strcmp(name, "fwd_tile_b")) {
    g_tzr_fwd_tile_b = value;
    return TZR_OK;
  }
  if (!strcmp(name, "fwd_variant")) {
    g_tzr_fwd_variant = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_ch")) {
    g_tzr_bwd_ch = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_one_wg_heavy")) {
    g_tzr_bwd_one_wg_heavy = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_apply_waves")) {
    g_tzr_bwd_apply_waves = value;
    return TZR_OK;
  }
  if (!strcmp(name, "fwd_plan_order")) {
    g_tzr_fwd_plan_order = value;
    return TZR_OK;
  }
  if (!strcmp(name, "fwd_plan")) {
    g_tzr_fwd_plan = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_direct")) {
    g_tzr_bwd_direct = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_direct_debug")) {
    g_tzr_bwd_direct_debug = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_direct_ch")) {
    g_tzr_bwd_direct_ch = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_debug")) {
    g_tzr_bwd_debug = value;
    return TZR_OK;
  }
  if (!strcmp(name, "ia_bwd_plain")) {
    g_tzr_ia_bwd_plain = value;
    return TZR_OK;
  }
  if (!strcmp(name, "ia_fwd_wgs")) {
    g_tzr_ia_fwd_wgs = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_scan_slices")) {
    g_tzr_bwd_scan_slices = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_no_fuse_sort")) {
    g_tzr_bwd_no_fuse_sort = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_apply_fast")) {
    g_tzr_bwd_apply_fast = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_direct_hot")) {
    g_tzr_bwd_direct_hot = value;
    return TZR_OK;
  }
  if (!strcmp(name, "mlp_mfma")) {
    g_tzr_mlp_mfma = value;
    return TZR_OK;
  }
  if (!strcmp(name, "linear_bwd_wg")) {
    g_tzr_linear_bwd_wg = value;
    return TZR_OK;
  }
  if (!strcmp(name, "gemm_rows_wg")) {
    g_tzr_gemm_rows_wg = value;
    return TZR_OK;
  }
  if (!strcmp(name, "it_stagger")) {
    g_tzr_it_stagger = value;
    return TZR_OK;
  }
  if (!strcmp(name, "it_fwd_stagger")) {
    g_tzr_it_fwd_stagger = value;
    return TZR_OK;
  }
  if (!strcmp(name, "wg_debug")) {
    g_tzr_wg_debug = value;
    return TZR_OK;
  }
  if (!strcmp(name, "it_wgs")) {
    g_tzr_it_wgs = value;
    return TZR_OK;
  }
  if (!strcmp(name, "ia_gen_wgs")) {
    g_tzr_ia_gen_wgs = value;
    return TZR_OK;
  }
  if (!strcmp(name, "ia_bwd_wgs")) {
    g_tzr_ia_bwd_wgs = value;
    return TZR_OK;
  }
  if (!strcmp(name, "bwd_force_prep")) {
    g_tzr_bwd_force_prep = value;
    return TZR_OK;
  }
  return TZR_ERR_INVALID;
}
