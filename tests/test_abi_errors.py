"""Error behaviour of the C ABI: bad arguments come back as negative status
codes -- never a crash, never a launch -- and degenerate sizes (empty batch, zero ids) are TZR_OK
no-ops.  Called straight through ctypes, on the emulator build here and on the gfx950 library under
`-m gpu`."""
import ctypes as C
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd import _lib  # noqa: E402

OK, INVALID, LAUNCH, WORKSPACE, UNSUPPORTED = 0, -1, -2, -3, -4


def _buf(dev, n, dtype=torch.float32):
    return torch.zeros(max(n, 1), dtype=dtype, device=dev)


def test_index_ops_reject_bad_arguments(dev):
    L = _lib.lib()
    lens = _buf(dev, 8, torch.int32)
    off = _buf(dev, 9, torch.int64)
    ws = _lib.workspace(L.tzr_lengths_to_offsets_workspace(8), dev)
    p = _lib.ptr
    assert L.tzr_lengths_to_offsets(p(lens), 4, 8, None, p(ws), ws.numel(), None) == INVALID
    assert L.tzr_lengths_to_offsets(p(lens), 3, 8, p(off), p(ws), ws.numel(), None) == INVALID
    assert L.tzr_lengths_to_offsets(None, 4, 8, p(off), p(ws), ws.numel(), None) == INVALID
    assert L.tzr_lengths_to_offsets(p(lens), 4, 8, p(off), None, 0, None) == WORKSPACE
    assert L.tzr_lengths_to_offsets(p(lens), 4, 8, p(off), p(ws) + 8, ws.numel() - 8, None) == WORKSPACE  # misaligned
    assert L.tzr_lengths_to_offsets(p(lens), 4, 8, p(off), p(ws), ws.numel(), None) == OK
    assert L.tzr_lengths_to_offsets(None, 4, 0, p(off), p(ws), ws.numel(), None) == OK  # n = 0: offsets = [0]
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert int(off[0]) == 0


def test_pooled_forward_rejects_bad_destinations(dev):
    L = _lib.lib()
    tables = _buf(dev, 48, torch.uint8)
    feats = _buf(dev, 64, torch.uint8)
    slots = _buf(dev, 16, torch.uint8)
    vals = _buf(dev, 4, torch.int64)
    out = _buf(dev, 64)
    d = (_lib.TzrDst * 9)()
    for i in range(9):
        d[i].ptr, d[i].stride = _lib.ptr(out), 16
    p = _lib.ptr
    assert L.tzr_pooled_fwd(p(tables), p(feats), 1, p(slots), 1, p(vals), None, None, 4, d, 9, 1, None) == INVALID  # > TZR_MAX_DST
    assert L.tzr_pooled_fwd(None, p(feats), 1, p(slots), 1, p(vals), None, None, 4, d, 1, 1, None) == INVALID
    assert L.tzr_pooled_fwd(p(tables), p(feats), 1, p(slots), 1, p(vals), None, None, 4, d, 1, 0, None) == INVALID  # jagged without offsets
    d[0].stride = 18
    assert L.tzr_pooled_fwd(p(tables), p(feats), 1, p(slots), 1, p(vals), None, None, 4, d, 1, 1, None) == INVALID  # stride % 4
    d[0].stride, d[0].ptr = 16, _lib.ptr(out) + 4
    assert L.tzr_pooled_fwd(p(tables), p(feats), 1, p(slots), 1, p(vals), None, None, 4, d, 1, 1, None) == INVALID  # 16-byte alignment
    d[0].ptr = _lib.ptr(out)
    assert L.tzr_pooled_fwd(p(tables), p(feats), 1, p(slots), 1, None, None, None, 0, d, 1, 1, None) == OK  # empty batch


def test_forward_that_carries_the_plan_argument_checks(dev):
    L = _lib.lib()
    tables, feats, slots = _buf(dev, 48, torch.uint8), _buf(dev, 64, torch.uint8), _buf(dev, 16, torch.uint8)
    vals, out = _buf(dev, 4, torch.int64), _buf(dev, 64)
    d = (_lib.TzrDst * 1)()
    d[0].ptr, d[0].stride = _lib.ptr(out), 16
    p = _lib.ptr
    assert L.tzr_pooled_fwd_cells_plan_supported(104, 65536) == 1 and L.tzr_pooled_fwd_cells_plan_supported(104, 8192) == 0
    assert L.tzr_pooled_fwd_cells_plan_supported(129, 65536) == 0 and L.tzr_pooled_fwd_cells_plan_supported(0, 65536) == 0
    args = [p(tables), p(feats), 1, p(slots), 1, d, 1, p(tables), 1, p(feats), 1, 16, p(vals), 4, 4, None, None, None, 0, None]
    assert L.tzr_pooled_fwd_cells_plan(*args) == UNSUPPORTED  # a batch this small: the two calls
    bad = list(args)
    bad[0] = None
    assert L.tzr_pooled_fwd_cells_plan(*bad) == INVALID
    assert L.tzr_tune(b"fwd_plan", 0) == OK
    try:
        assert L.tzr_pooled_fwd_cells_plan_supported(104, 65536) == 0
    finally:
        L.tzr_tune(b"fwd_plan", 1)
    assert L.tzr_tune(b"fwd_plan", 2) == OK
    try:
        assert L.tzr_pooled_fwd_cells_plan(*args) in (INVALID, WORKSPACE)  # supported at any size now: no geometry, no workspace
    finally:
        L.tzr_tune(b"fwd_plan", 1)


def test_backward_plan_limits(dev):
    L = _lib.lib()
    tables, feats = _buf(dev, 48, torch.uint8), _buf(dev, 64, torch.uint8)
    vals = _buf(dev, 4, torch.int64)
    ws = _lib.workspace(L.tzr_pooled_bwd_workspace(4, 4, 1, 1, 4, 16), dev)
    p = _lib.ptr
    assert L.tzr_pooled_bwd_workspace(-1, 4, 1, 1, 4, 16) == 0
    assert L.tzr_pooled_bwd_plan(p(tables), 1, p(feats), 1, 1, 10, 16, p(vals), None, 1 << 32, 4, 4, 1, p(ws), ws.numel(), None) == UNSUPPORTED
    assert L.tzr_pooled_bwd_plan(p(tables), 1, p(feats), 1, 1, 10, 16, p(vals), None, 4, 4, 4, 0, p(ws), ws.numel(), None) == INVALID  # jagged, no offsets
    assert L.tzr_pooled_bwd_plan(p(tables), 1, p(feats), 1, 1, 10, 16, p(vals), None, 4, 4, 4, 1, p(ws), 64, None) == WORKSPACE
    assert L.tzr_pooled_bwd_plan(p(tables), 1, p(feats), 1, 1, 10, 16, p(vals), None, 4, 4, 4, 1, None, 0, None) == WORKSPACE
    assert L.tzr_pooled_bwd_plan(p(tables), 1, p(feats), 1, 1, 10, 16, None, None, 0, 0, 4, 1, p(ws), ws.numel(), None) == OK  # no ids


def test_interaction_and_jagged_shape_limits(dev):
    L = _lib.lib()
    x, out = _buf(dev, 4096), _buf(dev, 4096)
    p = _lib.ptr
    assert L.tzr_dot_interaction_fwd(None, 0, p(x), 26 * 6, 26, 6, 2, p(out), 400, 0, 0, None) == UNSUPPORTED  # D % 4
    assert L.tzr_dot_interaction_fwd(None, 0, p(x), 200 * 128, 200, 128, 2, p(out), 20000, 0, 0, None) == UNSUPPORTED  # one sample > 60 KB of LDS
    assert L.tzr_dot_interaction_fwd(None, 0, p(x), 16, 1, 16, 2, p(out), 16, 0, 0, None) == UNSUPPORTED  # a single row has no pairs
    assert L.tzr_dot_interaction_fwd(None, 0, p(x), 26 * 8, 26, 8, 2, p(out), 400, 0, 0, None) == OK  # general kernel (D != 16)
    assert L.tzr_dot_interaction_fwd(None, 0, None, 0, 26, 16, 2, p(out), 400, 0, 0, None) == INVALID
    assert L.tzr_fm_fwd(p(x), 26 * 6, 26, 6, 2, p(out), 8, None) == UNSUPPORTED  # D % 4
    off = _buf(dev, 3, torch.int64)
    assert L.tzr_jagged_to_padded_dense(p(x), 6, p(off), 2, 4, 6, 0.0, p(out), None) == UNSUPPORTED  # dim % 4
    assert L.tzr_jagged_to_padded_dense(p(x), 8, None, 2, 4, 8, 0.0, p(out), None) == INVALID


def test_zch_and_dense_glue_argument_checks(dev):
    L = _lib.lib()
    keys = _buf(dev, 24, torch.int64)
    rows = _buf(dev, 24, torch.int32)
    m = _lib.TzrZchModule()
    m.keys, m.rows, m.capacity, m.zch_size = _lib.ptr(keys), _lib.ptr(rows), 24, 8  # not a power of two
    p = _lib.ptr
    assert L.tzr_zch_build(C.byref(m), None, None, 0, None) == INVALID
    m.capacity = 16
    assert L.tzr_zch_build(C.byref(m), p(keys), p(rows), 9, None) == INVALID  # load factor > 1/2
    assert L.tzr_zch_build(C.byref(m), None, None, 0, None) == OK
    x, y = _buf(dev, 64), _buf(dev, 64)
    ws = _lib.workspace(L.tzr_relu_bwd_colsum_workspace(4, 8), dev)
    assert L.tzr_relu_bwd_colsum(p(x), 6, p(y), 6, 4, 6, p(x), 6, p(y), p(ws), ws.numel(), None) == UNSUPPORTED  # N % 4
    assert L.tzr_relu_bwd_colsum(p(x), 8, p(y), 8, 4, 8, p(x), 8, p(y), None, 0, None) == WORKSPACE
    assert L.tzr_relu_bwd_colsum(p(x), 8, p(y), 8, 0, 8, p(x), 8, p(y), p(ws), ws.numel(), None) == INVALID  # B <= 0
    wb = _lib.workspace(L.tzr_bce_logits_workspace(4), dev)
    assert L.tzr_bce_logits(p(x), p(y), 2, 1, None, 4, p(x), p(y), p(wb), wb.numel(), None) == UNSUPPORTED  # fp16 labels
    assert L.tzr_bce_logits(p(x), p(y), 4, 1, None, 0, p(x), p(y), p(wb), wb.numel(), None) == INVALID
    t = (_lib.TzrAdamTensor * 1)()
    assert L.tzr_dense_adam(t, 1, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == INVALID  # null tensor pointers
    assert L.tzr_dense_adam(None, 0, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == INVALID
    assert L.tzr_tune(b"no_such_knob", 1) == INVALID and L.tzr_tune(None, 1) == INVALID


def test_module_level_errors_are_python_exceptions(dev):
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    ebc = EmbeddingBagCollection([EmbeddingBagConfig("t", 8, 10, ["k"])], device=dev)
    kjt = KeyedJaggedTensor(["other"], torch.zeros(2, dtype=torch.int64), torch.ones(2, dtype=torch.int32)).to(dev)
    with pytest.raises(KeyError):
        ebc(kjt)
    with pytest.raises(_lib.TzrError):  # host tensor into the device library (or the reverse on the emulator)
        other = torch.device("cpu") if dev.type == "cuda" else None
        if other is None:
            raise _lib.TzrError("n/a on the emulator: it only ever sees host tensors")
        _lib.ptr(torch.zeros(1, device=other))
    # out-of-range ids never fault: they read row 0 (K4 reports them)
    bad = KeyedJaggedTensor(["k"], torch.tensor([3, 10**12, -5]), torch.ones(3, dtype=torch.int32), uniform_length=1).to(dev)
    out = ebc(bad).values()
    w = ebc.table_weights()["t"].detach()
    assert torch.equal(out[1], w[0]) and torch.equal(out[2], w[0]) and torch.equal(out[0], w[3])


def test_export_segment_reduce_and_adam_argument_checks(dev):
    L = _lib.lib()
    p = _lib.ptr
    x, out = _buf(dev, 256), _buf(dev, 256)
    q = _buf(dev, 512, torch.uint8)
    bad = _buf(dev, 3, torch.int64)
    off = _buf(dev, 5, torch.int64)
    # INT8 export
    assert L.tzr_quantize_rows_q8f16(p(x), _lib.DT_F32, 16, 4, 16, p(q), None, None) == INVALID          # no status scratch
    assert L.tzr_quantize_rows_q8f16(p(x), _lib.DT_F32, 16, 4, 6, p(q), p(bad), None) == UNSUPPORTED     # dim % 4
    assert L.tzr_quantize_rows_q8f16(p(x), _lib.DT_F32, 8, 4, 16, p(q), p(bad), None) == UNSUPPORTED     # stride < dim
    assert L.tzr_quantize_rows_q8f16(p(x), 7, 16, 4, 16, p(q), p(bad), None) == UNSUPPORTED              # unknown dtype
    assert L.tzr_quantize_rows_q8f16(None, _lib.DT_F32, 16, 4, 16, p(q), p(bad), None) == INVALID
    assert L.tzr_quantize_rows_q8f16(None, _lib.DT_F32, 16, 0, 16, None, p(bad), None) == OK              # no rows
    assert L.tzr_dequantize_rows_q8f16(p(q), 4, 16, p(out), 8, None) == UNSUPPORTED                       # out stride < dim
    assert L.tzr_dequantize_rows_q8f16(None, 4, 16, p(out), 16, None) == INVALID
    assert L.tzr_dequantize_rows_q8f16(None, 0, 16, None, 16, None) == OK
    # segment reduce
    assert L.tzr_segment_reduce_fwd(p(x), 16, None, 4, 16, 0, p(out), 16, None) == INVALID                # no offsets
    assert L.tzr_segment_reduce_fwd(p(x), 16, p(off), 4, 16, 2, p(out), 16, None) == INVALID              # mode
    assert L.tzr_segment_reduce_fwd(p(x), 16, p(off), 4, 10, 0, p(out), 16, None) == UNSUPPORTED          # dim % 4
    assert L.tzr_segment_reduce_fwd(p(x), 16, p(off), 4, 16, 1, None, 16, None) == INVALID                # no output
    assert L.tzr_segment_reduce_fwd(p(x), 16, p(off), 0, 16, 1, None, 16, None) == OK                     # no segments
    assert L.tzr_segment_reduce_bwd(p(out), 16, p(off), 4, 16, 0, p(x), 8, None) == UNSUPPORTED           # stride < dim
    # sparse Adam step counter
    st = _buf(dev, 4)
    assert L.tzr_sparse_adam_tick(None, 0.9, 0.999, None) == INVALID
    assert L.tzr_sparse_adam_tick(p(st), 1.0, 0.999, None) == INVALID                                     # beta in [0, 1)
    assert L.tzr_sparse_adam_tick(p(st), 0.9, -0.1, None) == INVALID
    assert L.tzr_sparse_adam_tick(p(st), 0.5, 0.75, None) == OK
    assert L.tzr_sparse_adam_tick(p(st), 0.5, 0.75, None) == OK
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert st.cpu().tolist()[:3] == [2.0, 0.75, 0.4375]
