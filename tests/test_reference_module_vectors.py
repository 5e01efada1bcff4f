"""Floating-point stages pinned to the reference's OWN modules.

tests/golden/reference_module_vectors.npz holds inputs, parameters, outputs and autograd gradients
produced by running /root/reference/tzrec/modules/{interaction,fm,mlp,sequence,mmoe}.py on CPU
(tests/golden/make_reference_module_vectors.py, run in the authoring container).  Two layers:

* the oracle restatements (oracle/tzrec_oracle.py) reproduce them (pins the oracle);
* the product modules -- the HIP kernels through the C ABI (emulated on CPU, native under -m gpu)
  and the host modules around them -- reproduce them directly.

fp32, 1e-5 relative (the tolerance BASELINE.json's north_star states).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402

RTOL, ATOL = 1e-5, 1e-5
_Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_module_vectors.npz"))


def T(key):
    return torch.from_numpy(_Z[key].copy())


def layers(prefix):
    out, i = [], 0
    while f"{prefix}/W{i}" in _Z:
        out.append((T(f"{prefix}/W{i}"), T(f"{prefix}/b{i}")))
        i += 1
    return out


def close(a, b, scale=1.0):
    torch.testing.assert_close(a.detach().cpu(), b, rtol=RTOL, atol=ATOL * scale)


def load_mlp(mlp, prefix, dev):
    """reference keys mlp.<i>.perceptron.0.* -> this package's mlp.<2i>.* (Linear, ReLU pairs)"""
    lins = [m for m in mlp.mlp if isinstance(m, torch.nn.Linear)]
    ws = layers(prefix)
    assert len(lins) == len(ws)
    with torch.no_grad():
        for lin, (w, b) in zip(lins, ws):
            assert lin.weight.shape == w.shape
            lin.weight.copy_(w)
            lin.bias.copy_(b)
    return mlp.to(dev)


# ---- the oracle against the reference ---------------------------------------------------------


@pytest.mark.parametrize("tag", ["fm_a", "fm_b"])
def test_oracle_fm(tag):
    x = T(f"{tag}/x").requires_grad_(True)
    y = orc.fm(x)
    close(y, T(f"{tag}/y"))
    y.backward(T(f"{tag}/gy"))
    close(x.grad, T(f"{tag}/gx"))


@pytest.mark.parametrize("tag", ["ia_27", "ia_5", "ia_17"])
def test_oracle_dot_interaction(tag):
    x = T(f"{tag}/x").requires_grad_(True)
    y = orc.dot_interaction(x)
    close(y, T(f"{tag}/y"))
    y.backward(T(f"{tag}/gy"))
    close(x.grad, T(f"{tag}/gx"))


def test_oracle_mlp_din_dlrm():
    close(orc.mlp(T("mlp/x"), layers("mlp")), T("mlp/y"))
    y = orc.din_encoder(T("din/query"), T("din/sequence"), T("din/length"), layers("din/mlp"),
                        (T("din/linW"), T("din/linb")), max_seq_length=6)
    close(y, T("din/y"))
    p = {"dim": 16, "dense_mlp": layers("dlrm/dense_mlp"), "final_mlp": layers("dlrm/final_mlp"),
         "output": (T("dlrm/outW"), T("dlrm/outb")), "arch_with_sparse": True}
    logits = orc.dlrm_forward(T("dlrm/dense"), T("dlrm/sparse"), p)
    close(logits, T("dlrm/logits"))
    close(orc.bce_with_logits(logits, T("dlrm/labels")), T("dlrm/loss"))


# ---- the product against the reference --------------------------------------------------------


@pytest.mark.parametrize("tag", ["fm_a", "fm_b"])
def test_fm_kernel(dev, tag):
    from torcheasyrec_amd.interaction import FactorizationMachine

    x = T(f"{tag}/x").to(dev).requires_grad_(True)
    y = FactorizationMachine()(x)
    close(y, T(f"{tag}/y"))
    y.backward(T(f"{tag}/gy").to(dev))
    close(x.grad, T(f"{tag}/gx"))


@pytest.mark.parametrize("tag", ["ia_27", "ia_5", "ia_17"])
def test_interaction_kernel(dev, tag):
    """ia_27 / ia_17: the MFMA kernels; ia_5 (D = 8): the general kernel"""
    from torcheasyrec_amd.interaction import InteractionArch

    x = T(f"{tag}/x").to(dev).requires_grad_(True)
    y = InteractionArch(x.shape[1])(x)
    close(y, T(f"{tag}/y"))
    y.backward(T(f"{tag}/gy").to(dev))
    close(x.grad, T(f"{tag}/gx"))


def test_mlp_module(dev):
    from torcheasyrec_amd.dlrm import MLP

    mlp = load_mlp(MLP(13, [24, 16]), "mlp", dev)
    x = T("mlp/x").to(dev).requires_grad_(True)
    y = mlp(x)
    close(y, T("mlp/y"))
    y.backward(T("mlp/gy").to(dev))
    close(x.grad, T("mlp/gx"))
    lins = [m for m in mlp.mlp if isinstance(m, torch.nn.Linear)]
    for i, lin in enumerate(lins):
        close(lin.weight.grad, T(f"mlp/gW{i}"))
        close(lin.bias.grad, T(f"mlp/gb{i}"))


def test_din_encoder_module(dev):
    from torcheasyrec_amd.sequence import DINEncoder

    din = DINEncoder(sequence_dim=16, query_dim=12, input="g", attn_mlp={"hidden_units": [20, 8]}, max_seq_length=6)
    load_mlp(din.mlp, "din/mlp", dev)
    with torch.no_grad():
        din.linear.weight.copy_(T("din/linW"))
        din.linear.bias.copy_(T("din/linb"))
    din.to(dev)
    q = T("din/query").to(dev).requires_grad_(True)
    s = T("din/sequence").to(dev).requires_grad_(True)
    y = din({"g.query": q, "g.sequence": s, "g.sequence_length": T("din/length").to(dev)})
    close(y, T("din/y"))
    y.backward(T("din/gy").to(dev))
    close(q.grad, T("din/gquery"))
    close(s.grad, T("din/gsequence"))


@pytest.mark.parametrize("tag,cfg", [("din", dict(sequence_dim=16, query_dim=12, hidden=[20, 8], max_seq_length=6)),
                                     ("din_taobao", dict(sequence_dim=48, query_dim=48, hidden=[256, 64], max_seq_length=0))])
def test_din_encoder_jagged_path(dev, tag, cfg):
    """The DEFAULT config-4 evaluation -- DINEncoder.forward_jagged: csrc/din_attention.hip + linear_bwd.hip on the rows of the
    unpooled lookup, no padded tensor -- directly against outputs of the reference's DINEncoder (tzrec/modules/sequence.py:65-128)
    run on the padded tensor: output, query gradient, the gradient of every valid sequence row, and the gradient of every
    attention-MLP layer and of the score layer, elementwise 1e-5.  The jagged rows are the padded vectors' valid positions
    (what to_padded_dense was given: tzrec/modules/embedding.py:1480).  `din` pads with random rows: a sample WITHOUT positions
    averages them in the reference (uniform softmax over masked scores) -- a property of that fixture's padding, not of the path
    (the lookup pads with zeros), so that one sample's output is compared against zero and its padding rows are left out."""
    from torcheasyrec_amd.sequence import DINEncoder

    din = DINEncoder(sequence_dim=cfg["sequence_dim"], query_dim=cfg["query_dim"], input="g", attn_mlp={"hidden_units": cfg["hidden"]},
                     max_seq_length=cfg["max_seq_length"])
    load_mlp(din.mlp, f"{tag}/mlp", dev)
    with torch.no_grad():
        din.linear.weight.copy_(T(f"{tag}/linW"))
        din.linear.bias.copy_(T(f"{tag}/linb"))
    din.to(dev)
    assert din.jagged_capable()
    din.row_bucket = 16  # (rows of the MLP input rounded up: zero rows behind the last position)
    seq, lens = T(f"{tag}/sequence"), T(f"{tag}/length")
    B, L, D = seq.shape
    valid = torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)  # [B, L]
    rows = seq[valid]                                          # sample-major, position order: the unpooled lookup's rows
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    q = T(f"{tag}/query").to(dev).requires_grad_(True)
    v = rows.to(dev).requires_grad_(True)
    y = din({"g.query": q, "g.sequence_jagged": v, "g.sequence_offsets": off.to(dev), "g.sequence_max_len": L,
             "g.sequence_length": lens.to(dev)})
    y_ref = T(f"{tag}/y").clone()
    y_ref[lens == 0] = 0.0
    close(y, y_ref)
    y.backward(T(f"{tag}/gy").to(dev))
    close(q.grad, T(f"{tag}/gquery"))
    eff = valid if not cfg["max_seq_length"] else valid & (torch.arange(L).unsqueeze(0) < cfg["max_seq_length"])
    g_ref = T(f"{tag}/gsequence")
    close(v.grad, (g_ref * eff.unsqueeze(2))[valid])  # (a row behind max_seq_length takes no part: zero gradient, as in the reference)
    lins = [m for m in din.mlp.mlp if isinstance(m, torch.nn.Linear)]
    for i, lin in enumerate(lins):
        close(lin.weight.grad, T(f"{tag}/gW{i}"))
        close(lin.bias.grad, T(f"{tag}/gb{i}"))
    close(din.linear.weight.grad, T(f"{tag}/glinW"))
    close(din.linear.bias.grad, T(f"{tag}/glinb"))


def test_oracle_din_param_gradients():
    """oracle/tzrec_oracle.din_encoder differentiated by autograd against the reference module's parameter gradients (pins the
    oracle's backward, which the config-level DIN tests compare the HIP path with)"""
    for tag, msl in (("din", 6), ("din_taobao", 0)):
        ls = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in layers(f"{tag}/mlp")]
        lw, lb = T(f"{tag}/linW").requires_grad_(True), T(f"{tag}/linb").requires_grad_(True)
        q, s = T(f"{tag}/query").requires_grad_(True), T(f"{tag}/sequence").requires_grad_(True)
        y = orc.din_encoder(q, s, T(f"{tag}/length"), ls, (lw, lb), max_seq_length=msl)
        close(y, T(f"{tag}/y"))
        y.backward(T(f"{tag}/gy"))
        close(q.grad, T(f"{tag}/gquery"))
        close(s.grad, T(f"{tag}/gsequence"))
        for i, (w, b) in enumerate(ls):
            close(w.grad, T(f"{tag}/gW{i}"))
            close(b.grad, T(f"{tag}/gb{i}"))
        close(lw.grad, T(f"{tag}/glinW"))
        close(lb.grad, T(f"{tag}/glinb"))


def test_dlrm_dense_half(dev):
    """dense MLP -> fused dot interaction + concatenation -> final MLP -> logits -> BCE, against the
    reference modules wired as tzrec/models/dlrm.py:101-135 wires them (DLRM-Criteo shapes)."""
    from torcheasyrec_amd.dlrm import MLP, OutputLinear, bce_with_logits
    from torcheasyrec_amd.interaction import dot_interaction

    dense_mlp = load_mlp(MLP(13, [64, 16]), "dlrm/dense_mlp", dev)
    final_mlp = load_mlp(MLP(783, [64, 32]), "dlrm/final_mlp", dev)
    out = OutputLinear(32, 1)
    with torch.no_grad():
        out.weight.copy_(T("dlrm/outW"))
        out.bias.copy_(T("dlrm/outb"))
    out.to(dev)
    sparse = T("dlrm/sparse").to(dev).requires_grad_(True)
    d = dense_mlp(T("dlrm/dense").to(dev))
    logits = out(final_mlp(dot_interaction(d, sparse, 16, cat_dense=True, cat_sparse=True))).squeeze(1)
    close(logits, T("dlrm/logits"))
    loss = bce_with_logits(logits, T("dlrm/labels").to(dev))
    close(loss, T("dlrm/loss"))
    loss.backward()
    close(sparse.grad, T("dlrm/gsparse"))
    close([m for m in final_mlp.mlp if isinstance(m, torch.nn.Linear)][0].weight.grad, T("dlrm/g_final_W0"))
    close(out.weight.grad, T("dlrm/g_outW"))


@pytest.mark.parametrize("tag", ["mmoe_gate", "mmoe_plain"])
def test_mmoe_module(dev, tag):
    """rank_model.MMoE (what ConfigMMoE runs) against the reference's MMoE (tzrec/modules/mmoe.py:20-76)"""
    from torcheasyrec_amd.rank_model import MMoE

    mm = MMoE(12, {"hidden_units": [16, 8]}, num_expert=3, num_task=2, gate_mlp={"hidden_units": [6]} if tag == "mmoe_gate" else None)
    for e in range(3):
        load_mlp(mm.expert_mlps[e], f"{tag}/expert{e}", dev)
    with torch.no_grad():
        for t in range(2):
            if tag == "mmoe_gate":
                load_mlp(mm.gate_mlps[t], f"{tag}/gate{t}", dev)
            mm.gate_finals[t].weight.copy_(T(f"{tag}/final{t}W"))
            mm.gate_finals[t].bias.copy_(T(f"{tag}/final{t}b"))
    mm.to(dev)
    assert mm.output_dim() == 8
    ys = mm(T(f"{tag}/x").to(dev))
    for t in range(2):
        close(ys[t], T(f"{tag}/y{t}"))
