"""Generate tests/golden/reference_module_vectors.npz by RUNNING the reference's own pure-torch modules.

Run in the authoring container only (needs /root/reference; the GPU box has none):

    python tests/golden/make_reference_module_vectors.py

The floating-point modules of the path that are written in the reference repo itself are imported
from where they lie and executed on CPU in fp32:

    tzrec/modules/interaction.py:57-91   InteractionArch
    tzrec/modules/fm.py:17-42            FactorizationMachine
    tzrec/modules/mlp.py:21-177          MLP / Perceptron (defaults: bias, ReLU, no bn / ln / dropout)
    tzrec/modules/sequence.py:65-128     DINEncoder
    tzrec/modules/mmoe.py:20-76          MMoE

`import tzrec` itself fails here (its __init__ pulls every model and with them torchrec / fbgemm_gpu /
pyfg and the protoc-generated `tzrec.protos.*_pb2`, all absent), so the package objects `tzrec` and
`tzrec.modules` are pre-seeded as bare namespaces and the absent third-party / generated modules
are satisfied by an attribute-less placeholder.  None of the code that computes the vectors touches
a placeholder: the five modules above are plain torch.  The dense half of DLRM is wired from these
modules exactly as tzrec/models/dlrm.py:101-135 wires it (that file cannot be imported: it needs
the torchrec-backed EmbeddingGroup).

Inputs, parameters, outputs and autograd gradients are stored, so the tests need neither the
reference nor a seed convention.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/tzrec"


class _Meta(type):
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Meta(k, (), {})


class _Placeholder(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Meta(k, (), {})


class _AbsentDeps(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.startswith("tzrec.protos") or name == "torchrec" or name.startswith("torchrec."):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Placeholder(spec.name)

    def exec_module(self, module):
        pass


def install_reference_imports():
    for name, path in (("tzrec", REF), ("tzrec.modules", REF + "/modules")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    sys.meta_path.insert(0, _AbsentDeps())


def _np(t):
    return t.detach().cpu().numpy().copy()


def _mlp_params(out, prefix, mlp):
    """Reference MLP -> W0, b0, W1, b1, ... (state_dict keys mlp.<i>.perceptron.0.{weight,bias})."""
    for i, layer in enumerate(mlp.mlp):
        lin = layer.perceptron[0]
        out[f"{prefix}/W{i}"] = _np(lin.weight)
        out[f"{prefix}/b{i}"] = _np(lin.bias)


def _din_param_grads(out, tag, din):
    """gradients of the attention MLP's layers and of the score layer (tzrec/modules/sequence.py:92-99)"""
    for i, layer in enumerate(din.mlp.mlp):
        lin = layer.perceptron[0]
        out[f"{tag}/gW{i}"], out[f"{tag}/gb{i}"] = _np(lin.weight.grad), _np(lin.bias.grad)
    out[f"{tag}/glinW"], out[f"{tag}/glinb"] = _np(din.linear.weight.grad), _np(din.linear.bias.grad)


def main():
    install_reference_imports()
    fm_mod = importlib.import_module("tzrec.modules.fm")
    ia_mod = importlib.import_module("tzrec.modules.interaction")
    mlp_mod = importlib.import_module("tzrec.modules.mlp")
    seq_mod = importlib.import_module("tzrec.modules.sequence")
    mmoe_mod = importlib.import_module("tzrec.modules.mmoe")
    torch.manual_seed(20260925)
    torch.set_num_threads(1)
    out = {}

    # FactorizationMachine: [B, F, D] -> [B, D]
    for tag, (B, F, D) in {"fm_a": (6, 5, 8), "fm_b": (3, 26, 16)}.items():
        x = torch.randn(B, F, D, requires_grad=True)
        y = fm_mod.FactorizationMachine()(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/gy"], out[f"{tag}/gx"] = _np(x), _np(y), _np(gy), _np(x.grad)

    # InteractionArch: [B, N, D] -> [B, N(N-1)/2]
    for tag, (B, N, D) in {"ia_27": (4, 27, 16), "ia_5": (3, 5, 8), "ia_17": (2, 17, 16)}.items():
        x = torch.randn(B, N, D, requires_grad=True)
        y = ia_mod.InteractionArch(N)(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/gy"], out[f"{tag}/gx"] = _np(x), _np(y), _np(gy), _np(x.grad)

    # MLP with the defaults the example configs use
    mlp = mlp_mod.MLP(13, [24, 16])
    x = torch.randn(7, 13, requires_grad=True)
    y = mlp(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    _mlp_params(out, "mlp", mlp)
    out["mlp/x"], out["mlp/y"], out["mlp/gy"], out["mlp/gx"] = _np(x), _np(y), _np(gy), _np(x.grad)
    for i, layer in enumerate(mlp.mlp):
        out[f"mlp/gW{i}"], out[f"mlp/gb{i}"] = _np(layer.perceptron[0].weight.grad), _np(layer.perceptron[0].bias.grad)

    # DINEncoder: query narrower than the sequence rows, clamped lengths, an empty sequence
    din = seq_mod.DINEncoder(sequence_dim=16, query_dim=12, input="g", attn_mlp={"hidden_units": [20, 8]}, max_seq_length=6)
    q = torch.randn(5, 12, requires_grad=True)
    s = torch.randn(5, 8, 16, requires_grad=True)
    L = torch.tensor([0, 3, 8, 6, 1], dtype=torch.int64)
    y = din({"g.query": q, "g.sequence": s, "g.sequence_length": L})
    gy = torch.randn_like(y)
    y.backward(gy)
    _mlp_params(out, "din/mlp", din.mlp)
    out["din/linW"], out["din/linb"] = _np(din.linear.weight), _np(din.linear.bias)
    out["din/query"], out["din/sequence"], out["din/length"] = _np(q), _np(s), _np(L)
    out["din/y"], out["din/gy"], out["din/gquery"], out["din/gsequence"] = _np(y), _np(gy), _np(q.grad), _np(s.grad)
    _din_param_grads(out, "din", din)

    # MMoE with and without gate MLPs
    for tag, gate in (("mmoe_gate", {"hidden_units": [6]}), ("mmoe_plain", None)):
        mm = mmoe_mod.MMoE(in_features=12, expert_mlp={"hidden_units": [16, 8]}, num_expert=3, num_task=2, gate_mlp=gate)
        x = torch.randn(9, 12)
        ys = mm(x)
        out[f"{tag}/x"] = _np(x)
        for e, em in enumerate(mm.expert_mlps):
            _mlp_params(out, f"{tag}/expert{e}", em)
        for t in range(2):
            if gate is not None:
                _mlp_params(out, f"{tag}/gate{t}", mm.gate_mlps[t])
            out[f"{tag}/final{t}W"], out[f"{tag}/final{t}b"] = _np(mm.gate_finals[t].weight), _np(mm.gate_finals[t].bias)
            out[f"{tag}/y{t}"] = _np(ys[t])

    # dense half of DLRM (tzrec/models/dlrm.py:101-135), DLRM-Criteo shapes: 13 dense, 26 x 16 sparse
    B, Fs, D = 4, 26, 16
    dense_mlp = mlp_mod.MLP(13, [64, D])
    ia = ia_mod.InteractionArch(Fs + 1)
    final_mlp = mlp_mod.MLP(ia.output_dim() + D + Fs * D, [64, 32])
    output = torch.nn.Linear(32, 1)
    dense = torch.randn(B, 13)
    sparse = torch.randn(B, Fs * D, requires_grad=True)
    d = dense_mlp(dense)                                            # dlrm.py:118-121
    feat = torch.cat([d.unsqueeze(1), sparse.reshape(B, Fs, D)], dim=1)   # :112-116, :122
    allf = torch.cat([ia(feat), d, sparse], dim=-1)                 # :124-129 (arch_with_sparse)
    logits = output(final_mlp(allf)).squeeze(1)                     # :130-134
    labels = torch.tensor([1.0, 0.0, 0.0, 1.0])
    loss = torch.nn.BCEWithLogitsLoss()(logits, labels)             # rank_model.py:190-191
    loss.backward()
    _mlp_params(out, "dlrm/dense_mlp", dense_mlp)
    _mlp_params(out, "dlrm/final_mlp", final_mlp)
    out["dlrm/outW"], out["dlrm/outb"] = _np(output.weight), _np(output.bias)
    out["dlrm/dense"], out["dlrm/sparse"], out["dlrm/labels"] = _np(dense), _np(sparse), _np(labels)
    out["dlrm/logits"], out["dlrm/loss"], out["dlrm/gsparse"] = _np(logits), _np(loss), _np(sparse.grad)
    out["dlrm/g_final_W0"] = _np(final_mlp.mlp[0].perceptron[0].weight.grad)
    out["dlrm/g_outW"] = _np(output.weight.grad)

    # (added in round 6, BEHIND everything above so that the earlier arrays keep their random draws)
    # DINEncoder at the shapes of examples/multi_tower_din_taobao.config: three 16-wide sequence features = 48-wide rows, query of
    # the same width, attn_mlp [256, 64], sequence_length 50 here (100 there); lengths incl. 0, 1 and the full length
    din = seq_mod.DINEncoder(sequence_dim=48, query_dim=48, input="g", attn_mlp={"hidden_units": [256, 64]})
    q = torch.randn(12, 48, requires_grad=True)
    L = torch.tensor([0, 1, 50, 50, 17, 33, 2, 49, 16, 31, 5, 40], dtype=torch.int64)
    # (zero rows behind a sample's length, as to_padded_dense leaves them -- tzrec/modules/embedding.py:1480: the sample without
    # any position then has output 0, which is also what an evaluation on the jagged rows gives)
    s = (torch.randn(12, 50, 48) * (torch.arange(50).unsqueeze(0) < L.unsqueeze(1)).unsqueeze(2)).requires_grad_(True)
    y = din({"g.query": q, "g.sequence": s, "g.sequence_length": L})
    gy = torch.randn_like(y)
    y.backward(gy)
    _mlp_params(out, "din_taobao/mlp", din.mlp)
    out["din_taobao/linW"], out["din_taobao/linb"] = _np(din.linear.weight), _np(din.linear.bias)
    out["din_taobao/query"], out["din_taobao/sequence"], out["din_taobao/length"] = _np(q), _np(s), _np(L)
    out["din_taobao/y"], out["din_taobao/gy"] = _np(y), _np(gy)
    out["din_taobao/gquery"], out["din_taobao/gsequence"] = _np(q.grad), _np(s.grad)
    _din_param_grads(out, "din_taobao", din)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_module_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
