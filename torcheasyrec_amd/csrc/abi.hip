// Library identity of libtzrec_hip.so (include/tzrec_hip.h).
#include "tzr_common.h"

extern "C" const char* tzr_backend(void) { return "hip-gfx950"; }
extern "C" int tzr_abi_version(void) { return 15; }  // 15: tzr_pooled_fwd_cells_plan (the forward and the backward's index plan as one launch); 14: tzr_relu_bwd_colsum_parts, TZR_ADAM_SRC_TENSOR with a source pointer; 13: tzr_linear_rows, tzr_linear_rows_wgrad (tall-input Linear layers on MFMA); 12: tzr_bwd_cells_geometry, tzr_pooled_bwd_cells_plan / _apply (one-launch index plan); 11: tzr_linear_bwd_relu, tzr_head_bwd_relu, tzr_skinny_linear_*, tzr_moe_mix_*, TZR_GRAD_HOT_ROWS; 10: tzr_comm_* / tzr_step_* (native step driver), tzr_zch_remap_ring; 9: tzr_din_* (jagged DIN target attention); 8: tzr_dot_interaction_top_wgrad; 7: tzr_pooled_bwd_direct; 2: zch, dense glue, positional candidates; 3: TzrSparseOptim grew (sparse Adam); 4: delta tracker; 5: tzr_mlp2_* / tzr_mlp_tail; 6: tzr_dot_interaction_top_*
