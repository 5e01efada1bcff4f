"""MixedShardedEmbeddingBagCollection and the sharded delta tracker on ONE rank: every kernel and every
collective call of the lanes executes (the all-to-alls are self copies), so the result must equal the
unsharded collection.  The `hip` variant runs through a world-size-1 RCCL group on the GPU; the `emu`
variant is the same body over gloo + the lane emulator.

(Named test_zz_* like tests/test_zz_delta_embedding_dump.py: written without GPU minutes; sorted last.)"""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def test_mixed_lanes_on_one_rank_match_the_unsharded_collection(dev):
    from torcheasyrec_amd import delta_embedding_dump as dd
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sharding import MixedShardedEmbeddingBagCollection
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    def seeded(t):
        def f(w):
            g = torch.Generator().manual_seed(100 + t)
            w.copy_((torch.rand(w.shape, generator=g) - 0.5) * 0.2)
        return f

    spec = [("wide_a", 4, 50, ["a"]), ("wide_b", 4, 301, ["b"]), ("deep_a", 16, 50, ["a"]), ("deep_b", 16, 301, ["b"]),
            ("cw_c", 16, 120, ["c"]), ("plain_d", 16, 7, ["d"])]
    cfgs = lambda: [EmbeddingBagConfig(n, d, r, f, init_fn=seeded(t)) for t, (n, d, r, f) in enumerate(spec)]  # noqa: E731
    groups = {"wide": ["a@wide_a", "b@wide_b"], "deep": ["a@deep_a", "b@deep_b", "c", "d"]}
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.1)
    # one rank: the plan puts both column shards of cw_c on rank 0 -> two dim-8 lanes next to the dim-4 and dim-16 ones
    plan = {n: {"sharding_type": "row_wise", "block": r, "rot": 0, "ranks": [0]} for n, _, r, _ in spec}
    plan["cw_c"] = {"sharding_type": "column_wise", "ranks": [0, 0]}
    plan["deep_a"] = {"sharding_type": "table_wise", "block": 50, "rot": 0, "ranks": [0]}
    with tempfile.TemporaryDirectory() as d:
        if dev.type == "cuda":
            dist.init_process_group("nccl", init_method=f"file://{d}/init", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("gloo", init_method=f"file://{d}/init", rank=0, world_size=1)
        try:
            class Holder(torch.nn.Module):
                def __init__(self, ebc):
                    super().__init__()
                    self.ebc = ebc

            sh = MixedShardedEmbeddingBagCollection(cfgs(), device=dev, optimizer=opt, groups=groups, plan=plan)
            assert len(sh.lanes) == 4 and sh.sharding_plan()["cw_c"] == {"sharding_type": "column_wise", "ranks": [0, 0], "shard_dim": 8}
            ref = EmbeddingBagCollection(cfgs(), device=dev, optimizer=opt, groups=groups)
            keys, rows, B = ["a", "b", "c", "d"], [50, 301, 120, 7], (256 if dev.type == "cuda" else 24)
            rng = np.random.default_rng(1)
            for step, jagged in enumerate([False, True]):
                lens = rng.integers(0, 4, size=(4, B)).astype(np.int32) if jagged else np.ones((4, B), dtype=np.int32)
                vals = np.concatenate([rng.integers(0, rows[f], size=int(lens[f].sum())) for f in range(4)]).astype(np.int64)
                kjt = KeyedJaggedTensor(keys, torch.from_numpy(vals), torch.from_numpy(lens.reshape(-1)),
                                        uniform_length=None if jagged else 1).to(dev)
                out, out_ref = sh.forward_grouped(kjt), ref.forward_grouped(kjt)
                for g in groups:
                    assert out[g].shape == out_ref[g].shape == (B, 8 if g == "wide" else 64)
                    if jagged:
                        torch.testing.assert_close(out[g].detach(), out_ref[g].detach(), rtol=1e-6, atol=1e-7, msg=g)
                    else:
                        assert torch.equal(out[g].detach(), out_ref[g].detach()), g  # one id per bag: a copy on both paths
                gw = torch.randn(B, 8, generator=torch.Generator().manual_seed(9 + step)).to(dev)
                gd = torch.randn(B, 64, generator=torch.Generator().manual_seed(19 + step)).to(dev)
                ((out["wide"] * gw).sum() + (out["deep"] * gd).sum()).backward()
                ((out_ref["wide"] * gw).sum() + (out_ref["deep"] * gd).sum()).backward()
                w, w_ref = sh.table_weights(), ref.table_weights()
                for name, _, r, _ in spec:
                    if name == "cw_c":
                        got = torch.cat([w[s].detach()[:r] for s in sh.column_shards(name)], dim=1)
                    else:
                        got = w[name].detach()[:r]
                    torch.testing.assert_close(got, w_ref[name].detach(), rtol=1e-5, atol=1e-7, msg=name)
            # the delta tracker refuses the column-wise table, and follows the lanes once it is gone
            try:
                dd.ModelDeltaTracker(Holder(sh))
                raise AssertionError("column-wise table accepted by the delta tracker")
            except ValueError as e:
                assert "does not support column-wise embedding sharding" in str(e)
            plan2 = {n: p for n, p in plan.items()}
            plan2["cw_c"] = {"sharding_type": "row_wise", "block": 120, "rot": 0, "ranks": [0]}
            sh2 = MixedShardedEmbeddingBagCollection(cfgs(), device=dev, optimizer=opt, groups=groups, plan=plan2)
            tr = dd.ModelDeltaTracker(Holder(sh2))
            sh2.forward_grouped(kjt)["deep"].sum().backward()
            got = {k: v.cpu().numpy() for k, v in tr.get_unique_ids().items()}
            off = np.concatenate([[0], np.cumsum(lens.reshape(-1).astype(np.int64))])
            tab_feat = {"wide_a": 0, "deep_a": 0, "wide_b": 1, "deep_b": 1, "cw_c": 2, "plain_d": 3}
            assert set(got) == {f"ebc.embedding_bags.{n}" for n in tab_feat}
            for n, f in tab_feat.items():
                np.testing.assert_array_equal(got[f"ebc.embedding_bags.{n}"], np.unique(vals[off[f * B]:off[(f + 1) * B]]))
            # sparse Adam over lanes: every lane ticks its own step counter, a checkpoint carries all of them
            # (only lane 0's used to be saved -> wrong bias correction in lanes 1..k after a resume)
            from torcheasyrec_amd.checkpoint import restore_checkpoint, save_checkpoint

            aopt = SparseOptimizerConfig(kind="adam", lr=0.01)
            sa = MixedShardedEmbeddingBagCollection(cfgs(), device=dev, optimizer=aopt, groups=groups, plan=plan)
            n_adam = 3 if dev.type == "cuda" else 1
            for _ in range(n_adam):
                o = sa.forward_grouped(kjt)
                (o["wide"].sum() + o["deep"].sum()).backward()
            assert [float(lane.fused_optimizer.adam_state(dev)[0]) for lane in sa.lanes] == [float(n_adam)] * len(sa.lanes)
            save_checkpoint(os.path.join(d, "ck_adam"), Holder(sa))
            sb = MixedShardedEmbeddingBagCollection(cfgs(), device=dev, optimizer=aopt, groups=groups, plan=plan)
            restore_checkpoint(os.path.join(d, "ck_adam"), Holder(sb))
            assert [float(lane.fused_optimizer.adam_state(dev)[0]) for lane in sb.lanes] == [float(n_adam)] * len(sb.lanes)
            oa, ob = sa.forward_grouped(kjt), sb.forward_grouped(kjt)
            (oa["wide"].sum() + oa["deep"].sum()).backward()
            (ob["wide"].sum() + ob["deep"].sum()).backward()
            for name in sa.table_weights():
                assert torch.equal(sa.table_weights()[name], sb.table_weights()[name]), name  # the 4th step agrees bit for bit
        finally:
            dist.destroy_process_group()
