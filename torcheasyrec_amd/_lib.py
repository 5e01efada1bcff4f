"""ctypes binding of the C-ABI library ``libtzrec_hip.so`` (``include/tzrec_hip.h``).

The product path has exactly one compute backend: the hipcc-built gfx950 library that sits next to
this file.  If it is missing, ``lib()`` raises -- there is no CPU fallback.  (Tests may point the
loader at the CPU lane-emulator build of the same kernels with ``use_library``; that build reports
``tzr_backend() == "emu"`` and accepts CPU tensors only.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtzrec_hip.so")

TZR_OK = 0
TZR_MAX_DST = 8
TZR_MAX_FEAT_DST = 4
POOL_SUM, POOL_MEAN = 0, 1
DT_F32, DT_F16 = 0, 1
FWD_MIXED_DTYPE = 1
OPT_SGD, OPT_ADAGRAD, OPT_ROWWISE_ADAGRAD, OPT_ACCUMULATE, OPT_ADAM = 0, 1, 2, 3, 4
ABI_VERSION = 15  # struct layouts below match include/tzrec_hip.h of this version
WD_NONE, WD_L2, WD_DECOUPLE = 0, 1, 2
BOUNDS_FATAL, BOUNDS_WARNING, BOUNDS_IGNORE = 0, 1, 2

_ERR = {-1: "TZR_ERR_INVALID", -2: "TZR_ERR_LAUNCH", -3: "TZR_ERR_WORKSPACE", -4: "TZR_ERR_UNSUPPORTED"}

# numpy mirrors of the header structs (host-side construction, then uploaded as bytes)
TABLE_DT = np.dtype(
    [("w", "<u8"), ("m", "<u8"), ("rows", "<i8"), ("dim", "<i4"), ("w_stride", "<i4"),
     ("m_stride", "<i4"), ("first_order", "<i4"), ("n_feats", "<i4"), ("w_dtype", "<i4")]
)
FEATURE_DT = np.dtype(
    [("table", "<i4"), ("key", "<i4"), ("pooling", "<i4"), ("n_dst", "<i4"),
     ("dst", "<i4", (TZR_MAX_FEAT_DST,)), ("col", "<i4", (TZR_MAX_FEAT_DST,)), ("order", "<i4"),
     ("reserved", "<i4", (3,))]
)
SLOT_DT = np.dtype([("feature", "<i4"), ("chunk", "<i4"), ("dst", "<i4"), ("col", "<i4")])
assert TABLE_DT.itemsize == 48 and FEATURE_DT.itemsize == 64 and SLOT_DT.itemsize == 16


class TzrDst(C.Structure):
    _fields_ = [("ptr", C.c_uint64), ("stride", C.c_int64)]


GRAD_HOT_ROWS = 0x100  # TZR_GRAD_HOT_ROWS: flag in tzr_pooled_bwd_direct's grad_mode
MOE_MAX_EXPERTS, MOE_MAX_TASKS = 8, 4


class TzrMoeMix(C.Structure):
    _fields_ = [
        ("B", C.c_int64), ("H", C.c_int32), ("n_experts", C.c_int32), ("n_tasks", C.c_int32), ("reserved", C.c_int32),
        ("expert", C.c_uint64 * MOE_MAX_EXPERTS), ("expert_stride", C.c_int64 * MOE_MAX_EXPERTS),
        ("d_expert", C.c_uint64 * MOE_MAX_EXPERTS), ("d_expert_stride", C.c_int64 * MOE_MAX_EXPERTS),
        ("logits", C.c_uint64 * MOE_MAX_TASKS), ("logits_stride", C.c_int64 * MOE_MAX_TASKS),
        ("probs", C.c_uint64 * MOE_MAX_TASKS),
        ("out", C.c_uint64 * MOE_MAX_TASKS), ("out_stride", C.c_int64 * MOE_MAX_TASKS),
        ("grad_out", C.c_uint64 * MOE_MAX_TASKS), ("grad_out_stride", C.c_int64 * MOE_MAX_TASKS),
        ("d_logits", C.c_uint64 * MOE_MAX_TASKS), ("d_logits_stride", C.c_int64 * MOE_MAX_TASKS),
    ]


class TzrSparseOptim(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("weight_decay_mode", C.c_int32),
        ("d_lr", C.c_uint64),
        ("eps", C.c_float),
        ("weight_decay", C.c_float),
        ("max_gradient", C.c_float),
        ("gradient_clipping", C.c_int32),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("d_adam", C.c_uint64),
    ]


class TzrAdamTensor(C.Structure):
    _fields_ = [("param", C.c_uint64), ("grad", C.c_uint64), ("exp_avg", C.c_uint64), ("exp_avg_sq", C.c_uint64),
                ("state", C.c_uint64), ("numel", C.c_int64)]


class TzrAdamSource(C.Structure):
    _fields_ = [("kind", C.c_int32), ("G", C.c_int32), ("P", C.c_int32), ("col", C.c_int32), ("parts", C.c_uint64), ("reserved", C.c_uint64)]


class TzrWgradParts(C.Structure):
    _fields_ = [("opaque", C.c_uint64 * 24)]


ADAM_SRC_TENSOR, ADAM_SRC_ROWS, ADAM_SRC_WGRAD = 0, 1, 2


class TzrZchModule(C.Structure):
    _fields_ = [("keys", C.c_uint64), ("rows", C.c_uint64), ("counts", C.c_uint64), ("last_iter", C.c_uint64),
                ("capacity", C.c_int64), ("zch_size", C.c_int64), ("reserved", C.c_int64 * 2)]


DELTA_SEG_DT = np.dtype([("bitmap", "<u8"), ("rows", "<i8"), ("key", "<i4"), ("reserved", "<i4")])
assert DELTA_SEG_DT.itemsize == 24

ZCH_EMPTY = (1 << 63) - 1


class TzrError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None
_backend: Optional[str] = None

_vp, _i64, _i32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t

_SIGNATURES = {
    "tzr_backend": (C.c_char_p, []),
    "tzr_abi_version": (_i32, []),
    "tzr_tune": (_i32, [C.c_char_p, _i32]),
    "tzr_lengths_to_offsets_workspace": (_sz, [_i64]),
    "tzr_lengths_to_offsets": (_i32, [_vp, _i32, _i64, _vp, _vp, _sz, _vp]),
    "tzr_bounds_check": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "tzr_kjt_permute_workspace": (_sz, [_i64, _i64]),
    "tzr_kjt_permute": (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _i64, _vp, _sz, _vp]),
    "tzr_block_bucketize_workspace": (_sz, [_i64, _i64, _i32]),
    "tzr_block_bucketize": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp,
                                   _vp, _vp, _vp, _sz, _vp]),
    "tzr_exchange_bucketize_workspace": (_sz, [_i32, _i64, _i32]),
    "tzr_exchange_bucketize": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "tzr_exchange_message_stride": (_i64, [_i32, _i64]),
    "tzr_exchange_bucketize_capped": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "tzr_exchange_pad": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "tzr_exchange_owner_segments": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "tzr_pooled_fwd": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i64, C.POINTER(TzrDst),
                              _i32, _i32, _vp]),
    "tzr_pooled_fwd_ex": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i64, C.POINTER(TzrDst),
                                 _i32, _i32, _i32, _vp]),
    "tzr_pooled_bwd_workspace": (_sz, [_i64, _i64, _i32, _i32, _i64, _i32]),
    "tzr_pooled_bwd_plan_view": (_i32, [_i64, _i64, _i32, _i32, _i32, _vp]),
    "tzr_pooled_bwd_plan": (_i32, [_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i64, _i64, _i64,
                                   _i32, _vp, _sz, _vp]),
    "tzr_pooled_bwd_direct_supported": (_i32, [_i64, _i32, _i32, _i32, _i32]),
    "tzr_pooled_bwd_direct_workspace": (_sz, [_i64, _i32, _i32]),
    "tzr_pooled_bwd_direct": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                     C.POINTER(TzrDst), _i32, C.POINTER(TzrSparseOptim), _vp, _sz, _vp]),
    "tzr_sparse_adam_tick": (_i32, [_vp, C.c_float, C.c_float, _vp]),
    "tzr_dense_rows_update": (_i32, [_vp, _i32, _vp, _i64, _vp, _i32, C.POINTER(TzrSparseOptim), _vp]),
    "tzr_dense_rows_update_clear": (_i32, [_vp, _i32, _vp, _i64, _vp, _i32, C.POINTER(TzrSparseOptim), _vp]),
    "tzr_rows_gather": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _vp]),
    "tzr_lookup_grads": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, C.POINTER(TzrDst), _i32, _vp,
                                _i64, _i32, _vp]),
    "tzr_pooled_bwd_apply": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                    C.POINTER(TzrDst), _i32, C.POINTER(TzrSparseOptim), _vp, _sz,
                                    _vp]),
    "tzr_bwd_cells_geometry": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _vp, _sz, C.POINTER(C.c_int64)]),
    "tzr_pooled_bwd_cells_plan": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "tzr_pooled_fwd_cells_plan_supported": (_i32, [_i32, _i64]),
    "tzr_pooled_fwd_cells_plan": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "tzr_pooled_bwd_cells_apply": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i64, _i64, _i32, C.POINTER(TzrDst), _i32,
                                          C.POINTER(TzrSparseOptim), _vp, _vp, _vp, _sz, _vp]),
    "tzr_dense_adam_fused": (_i32, [C.POINTER(TzrAdamTensor), C.POINTER(TzrAdamSource), _i32, C.POINTER(TzrWgradParts), _vp, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "tzr_mlp2_bwd_parts": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _sz, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), _vp]),
    "tzr_dot_interaction_top_wgrad_parts": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _vp, _i64,
                                                   C.POINTER(TzrWgradParts), _vp]),
    "tzr_dot_interaction_fwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _i32,
                                       _i32, _vp]),
    "tzr_dot_interaction_bwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _i32,
                                       _i32, _vp, _i64, _vp, _i64, _vp]),
    "tzr_dot_interaction_top_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "tzr_dot_interaction_top_fwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _i64,
                                           _vp, _i64, _vp]),
    "tzr_dot_interaction_top_bwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp,
                                           _i64, _vp, _i64, _vp]),
    "tzr_dot_interaction_top_wgrad_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "tzr_dot_interaction_top_wgrad": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp,
                                             _i64, _vp]),
    "tzr_jagged_to_padded_dense": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, C.c_float, _vp, _vp]),
    "tzr_padded_dense_to_jagged": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "tzr_quantize_rows_q8f16": (_i32, [_vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp]),
    "tzr_dequantize_rows_q8f16": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp]),
    "tzr_segment_reduce_fwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "tzr_segment_reduce_bwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "tzr_jagged_segment_ids": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "tzr_din_assemble_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "tzr_din_assemble_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp]),
    "tzr_din_assemble2_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "tzr_din_assemble2_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp]),
    "tzr_din_attn_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "tzr_din_attn_bwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "tzr_comm_available": (_i32, [C.c_char_p]),
    "tzr_comm_version": (_i32, [C.c_char_p]),
    "tzr_comm_unique_id": (_i32, [C.c_char_p, _vp, _sz]),
    "tzr_comm_create": (_i32, [C.c_char_p, _vp, _sz, _i32, _i32, C.POINTER(C.c_void_p)]),
    "tzr_comm_destroy": (_i32, [_vp]),
    "tzr_comm_all_to_all": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "tzr_comm_all_reduce": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "tzr_step_create": (_i32, [C.POINTER(C.c_void_p)]),
    "tzr_step_destroy": (_i32, [_vp]),
    "tzr_step_add_graph": (_i32, [_vp, _vp]),
    "tzr_step_add_all_to_all": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32]),
    "tzr_step_add_all_reduce": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32]),
    "tzr_step_add_wait": (_i32, [_vp, _i32]),
    "tzr_step_num_ops": (_i32, [_vp]),
    "tzr_step_run": (_i32, [_vp, _vp]),
    "tzr_bce_logits_workspace": (_sz, [_i64]),
    "tzr_bce_logits": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "tzr_relu_bwd_colsum_workspace": (_sz, [_i64, _i32]),
    "tzr_relu_bwd_colsum": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_relu_bwd_colsum_parts": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _sz, _vp, _vp]),
    "tzr_head_bwd_workspace": (_sz, [_i64, _i32]),
    "tzr_head_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_head_bwd_relu_workspace": (_sz, [_i64, _i32]),
    "tzr_head_bwd_relu": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_moe_mix_fwd": (_i32, [_vp, _vp]),
    "tzr_moe_mix_bwd": (_i32, [_vp, _vp]),
    "tzr_skinny_linear_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "tzr_skinny_linear_bwd_workspace": (_sz, [_i64, _i32, _i32]),
    "tzr_skinny_linear_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_skinny_linear_bwd_parts": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _sz, _vp, _vp, _vp]),
    "tzr_linear_rows_supported": (_i32, [_i32, _i32]),
    "tzr_linear_rows": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _i64, _vp]),
    "tzr_linear_rows_wgrad_supported": (_i32, [_i32, _i32]),
    "tzr_linear_rows_wgrad_workspace": (_sz, [_i64, _i32, _i32]),
    "tzr_linear_rows_wgrad": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _sz, _vp]),
    "tzr_linear_bwd_relu_supported": (_i32, [_i32, _i32]),
    "tzr_linear_bwd_relu_workspace": (_sz, [_i64, _i32]),
    "tzr_linear_bwd_relu": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_mlp_workspace": (_sz, []),
    "tzr_mlp2_fwd": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp]),
    "tzr_mlp2_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                            _vp, _sz, _vp]),
    "tzr_mlp_tail": (_i32, [_vp, _i64, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                            _vp, _vp, _vp, _sz, _vp]),
    "tzr_dense_adam": (_i32, [_vp, _i32, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "tzr_zch_remap": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _i64, _i64, _i32, _vp, _vp, _vp]),
    "tzr_zch_remap_ring": (_i32, [_vp, _vp, _i32, _vp, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp]),
    "tzr_zch_build": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "tzr_zch_update": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "tzr_zch_select_hist": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, C.c_double, _i32, _i32, _i32, C.c_uint64, _i32,
                                   C.c_uint64, _vp, _vp]),
    "tzr_zch_select_mark": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, C.c_double, _i32, C.c_uint64, _i32, C.c_uint64,
                                   _vp, _vp, _vp]),
    "tzr_delta_mark": (_i32, [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tzr_delta_collect_workspace": (_sz, [_i64]),
    "tzr_delta_count": (_i32, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "tzr_delta_collect": (_i32, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _sz, _vp]),
    "tzr_fm_fwd": (_i32, [_vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp]),
    "tzr_fm_bwd": (_i32, [_vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp, _i64, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _bind(path: str) -> C.CDLL:
    handle = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError -> a symbol the header declares is missing
        fn.restype = res
        fn.argtypes = args
    return handle


def use_library(path: str) -> None:
    """Load an explicit library file (tests: the lane-emulator build)."""
    global _lib, _backend
    loaded = _bind(path)
    have = loaded.tzr_abi_version()
    if have != ABI_VERSION:
        raise TzrError(f"{path} speaks C-ABI version {have}, this package version {ABI_VERSION}: rebuild it "
                       "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = loaded
    _backend = _lib.tzr_backend().decode()
    apply_env_tune()


def apply_env_tune() -> None:
    """Experiment knobs from the environment: TZR_TUNE="name=value,name=value" (tzr_tune of include/tzrec_hip.h)."""
    spec = os.environ.get("TZR_TUNE", "")
    if not spec or _lib is None:
        return
    for kv in spec.split(","):
        name, _, val = kv.partition("=")
        if _lib.tzr_tune(name.strip().encode(), int(val)) != TZR_OK:
            raise TzrError(f"TZR_TUNE: unknown knob {name!r}")


def use_native() -> None:
    """Load the gfx950 library, compiling it first when the in-tree .so is missing or older than
    its sources (a fresh checkout on a GPU box)."""
    from . import _build

    use_library(_build.build())


def lib() -> C.CDLL:
    global _lib, _backend
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TzrError(
                f"{LIB_PATH} is missing: build the gfx950 kernels first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback."
            )
        use_library(LIB_PATH)
    return _lib


def backend() -> str:
    lib()
    return _backend  # type: ignore[return-value]


def check(rc: int, what: str) -> None:
    if rc != TZR_OK:
        raise TzrError(f"{what} failed: {_ERR.get(rc, rc)}")


def check_device(t: torch.Tensor) -> None:
    """The HIP library takes device pointers only; the emulator host pointers only."""
    if backend() == "emu":
        if t.device.type != "cpu":
            raise TzrError("emulator library loaded but tensor is on " + str(t.device))
    elif t.device.type != "cuda":
        raise TzrError(
            f"tensor on {t.device}: the gfx950 library needs HIP device memory (no CPU path)"
        )


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    check_device(t)
    return t.data_ptr()


def stream_ptr(device: torch.device) -> Optional[int]:
    """raw handle of torch's current stream on `device` (one C call: `torch.cuda.current_stream(device).cuda_stream` builds
    a Stream object through three layers of Python per kernel launch -- 35 us of a sharded step whose host time is its
    duration, profiles/r04r)"""
    if device.type != "cuda":
        return None
    idx = device.index
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def _probe_raw_stream():
    """`torch._C._cuda_getCurrentRawStream` is a private binding (what torch's own inductor / triton glue calls per launch).
    Used only when it exists AND agrees with the public path on this build; otherwise the public path, slower, is taken."""
    fn = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None)
    if fn is None or not torch.cuda.is_available():
        return None
    try:
        d = torch.cuda.current_device()
        return fn if int(fn(d)) == int(torch.cuda.current_stream(d).cuda_stream) else None
    except Exception:
        return None


class _LazyRawStream:
    """probed at the first call on a process that has a GPU (importing this module must not initialise HIP)"""

    def __init__(self):
        self._fn, self._probed = None, False

    def __call__(self, idx):
        if not self._probed:
            self._fn, self._probed = _probe_raw_stream(), True
        if self._fn is None:
            return torch.cuda.current_stream(idx).cuda_stream
        return self._fn(idx)


_RAW_STREAM = _LazyRawStream()


def upload_struct(arr: np.ndarray, device: torch.device) -> torch.Tensor:
    """Structured numpy array -> uint8 device tensor holding the same bytes."""
    raw = np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def zeroed_workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """256-byte aligned, zero-filled, meant to be KEPT by the caller (tzr_pooled_bwd_direct's self-resetting counters)."""
    t = torch.zeros(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:]


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """256-byte aligned scratch from the torch caching allocator."""
    t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:]
