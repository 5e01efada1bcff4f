"""`torch.ops.tzrec_hip.*` (torcheasyrec_amd/ops.py): every op is registered with a fake
implementation, passes `torch.library.opcheck`, shows up in FX / `torch.export` graphs of the modules, and computes
what the module path computes (same library underneath)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd import ops as tops  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.interaction import FactorizationMachine, InteractionArch, dot_interaction  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402


def test_every_op_is_registered_with_a_fake_impl():
    for name in tops.OPS:
        assert hasattr(torch.ops.tzrec_hip, name), name
    from torch._subclasses.fake_tensor import FakeTensorMode

    with FakeTensorMode():
        sp, de = torch.empty(5, 26 * 16), torch.empty(5, 16)
        assert torch.ops.tzrec_hip.dot_interaction_fwd(de, sp, 16, True, True).shape == (5, 783)
        gd, gs = torch.ops.tzrec_hip.dot_interaction_bwd(None, sp, torch.empty(5, 325), 16, False, False)
        assert gd.shape == (0,) and gs.shape == sp.shape
        assert torch.ops.tzrec_hip.fm_fwd(torch.empty(5, 6, 8)).shape == (5, 8)
        outs = torch.ops.tzrec_hip.pooled_fwd([torch.empty(10, 16)], torch.empty(48, dtype=torch.uint8), torch.empty(64, dtype=torch.uint8),
                                              torch.empty(64, dtype=torch.uint8), torch.empty(5, dtype=torch.int64), None, None, 5, [16, 4], False)
        assert [tuple(o.shape) for o in outs] == [(5, 16), (5, 4)]
        lens = torch.empty(3 * 5, dtype=torch.int32)
        r = torch.ops.tzrec_hip.kjt_permute(torch.empty(2, dtype=torch.int32), lens, torch.empty(16, dtype=torch.int64),
                                            torch.empty(20, dtype=torch.int64), None, 3, 5, 20)
        assert r[0].shape == (10,) and r[1].shape == (11,) and r[2].shape == (20,) and r[3].shape == (0,)
        r = torch.ops.tzrec_hip.block_bucketize(torch.empty(3, dtype=torch.int64), None, lens, torch.empty(16, dtype=torch.int64),
                                                torch.empty(20, dtype=torch.int64), None, 5, 2)
        assert r[0].shape == (30,) and r[1].shape == (31,) and r[2].shape == (20,) and r[4].shape == (20,)


def test_interaction_and_fm_ops_match_the_modules_and_pass_opcheck(dev):
    torch.manual_seed(0)
    B, F, D = 9, 26, 16
    sp = torch.randn(B, F * D).to(dev).requires_grad_(True)
    de = torch.randn(B, D).to(dev).requires_grad_(True)
    want = dot_interaction(de, sp, D)
    got = torch.ops.tzrec_hip.dot_interaction_fwd(de, sp, D, True, True)
    assert torch.equal(got, want)
    g = torch.randn_like(got)
    gw = torch.autograd.grad(want, (de, sp), g)
    gg = torch.autograd.grad(got, (de, sp), g)  # through the registered autograd formula = dot_interaction_bwd
    assert torch.equal(gw[0], gg[0]) and torch.equal(gw[1], gg[1])
    x = torch.randn(B, 6, 8).to(dev).requires_grad_(True)
    assert torch.equal(torch.ops.tzrec_hip.fm_fwd(x), FactorizationMachine()(x))
    if dev.type == "cpu":  # opcheck: schema, fake impl vs real, autograd registration, aot dispatch
        torch.library.opcheck(torch.ops.tzrec_hip.dot_interaction_fwd, (de.detach(), sp.detach(), D, True, True))
        torch.library.opcheck(torch.ops.tzrec_hip.fm_fwd, (x.detach(),))


def test_fx_and_export_graphs_show_the_ops(dev):
    """what tzrec's export path needs: tracing the modules yields tzrec_hip nodes, and the exported program runs"""
    from torch.fx.experimental.proxy_tensor import make_fx

    class Tower(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ia, self.fm = InteractionArch(5), FactorizationMachine()

        def forward(self, x):  # x [B, 5, 8]
            return torch.cat([self.ia(x), self.fm(x)], dim=1)

    m = Tower()
    x = torch.randn(4, 5, 8).to(dev)
    gm = make_fx(m, tracing_mode="fake")(x)
    targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
    assert any("tzrec_hip.dot_interaction_fwd" in t for t in targets) and any("tzrec_hip.fm_fwd" in t for t in targets), targets
    assert torch.equal(gm(x), m(x))
    ep = torch.export.export(m, (x,))
    text = str(ep.graph_module.graph)
    assert "tzrec_hip.dot_interaction_fwd" in text and "tzrec_hip.fm_fwd" in text
    assert torch.equal(ep.module()(x), m(x))


def test_pooled_ops_match_the_collection(dev):
    """pooled_fwd / pooled_bwd_adagrad driven with the descriptors of an EmbeddingBagCollection: same outputs and the
    same updated tables as the module's own forward / backward"""
    rng = np.random.default_rng(0)
    rows, keys, B, D, lr = [300, 7, 1000], ["a", "b", "c"], 33, 16, 0.1

    def make():
        g = torch.Generator().manual_seed(5)
        return EmbeddingBagCollection(
            [EmbeddingBagConfig(f"t{i}", D, r, [k], init_fn=lambda w, g=g: w.copy_((torch.rand(w.shape, generator=g) - 0.5) * 0.2))
             for i, (r, k) in enumerate(zip(rows, keys))], device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=lr), row_layout="split")

    ref, mine = make(), make()
    ids = np.concatenate([rng.integers(0, r, size=B) for r in rows]).astype(np.int64)
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(ids), torch.ones(3 * B, dtype=torch.int32), uniform_length=1).to(dev)
    out_ref = ref(kjt).values()
    g = torch.randn(B, 3 * D, generator=torch.Generator().manual_seed(1)).to(dev)
    (out_ref * g).sum().backward()
    meta = mine._meta(kjt.keys(), mine._default_layout())
    tables, states = list(mine.table_weights().values()), list(mine.table_states().values())
    (out,) = torch.ops.tzrec_hip.pooled_fwd(tables, meta.d_tables, meta.d_feats, meta.d_slots, kjt.values(), None, None, B, [3 * D], False)
    assert torch.equal(out, out_ref.detach())
    torch.ops.tzrec_hip.pooled_bwd_adagrad(tables, states, meta.d_bwd_tables, meta.d_bwd_feats, kjt.values(), None, None, [g],
                                           torch.full((1,), lr, device=dev), B, len(keys), max(rows), 1e-8)
    for n in ref.table_weights():
        assert torch.equal(mine.table_weights()[n], ref.table_weights()[n]), n
        assert torch.equal(mine.table_states()[n], ref.table_states()[n]), n
