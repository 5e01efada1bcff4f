"""Sharding planner (torcheasyrec_amd/planner.py).

1. The reference's own known-answer scenarios for its DynamicProgrammingProposer
   (/root/reference/tzrec/utils/plan_util_test.py:297-447) -- same fake options / topologies, same
   expected picks -- run against this package's proposer (duck-typed the same way).
2. Property test: the vectorised reachable-cell DP equals the dense oracle on the reference's
   seeds and problem sizes (plan_util_test.py:278-295, 424-438) and on larger random ones.
3. The MI355X enumerator + partitioner + search on the DLRM-Criteo tables.
"""
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.planner_oracle import dense_dp_proposals  # noqa: E402
from torcheasyrec_amd.planner import (GB, DynamicProgrammingProposer, EmbeddingEnumerator, GreedyPartitioner,  # noqa: E402
                                      PlannerError, TableSpec, Topology,
                                      dp_proposals, plan_tables, plan_to_json)


def _opt(fqn, hbm, ddr, perf):  # plan_util_test.py:158-167: one shard holding the whole option
    st = SimpleNamespace(hbm=hbm, ddr=ddr)
    return SimpleNamespace(fqn=fqn, shards=[SimpleNamespace(storage=st)], total_storage=st, total_perf=perf)


def _topo(n, hbm, ddr, local=None):  # plan_util_test.py:170-181
    return SimpleNamespace(devices=[SimpleNamespace(storage=SimpleNamespace(hbm=hbm, ddr=ddr)) for _ in range(n)],
                           local_world_size=local or n)


def _run(space, topo):  # plan_util_test.py:300-315
    p = DynamicProgrammingProposer(hbm_bins_per_device=20, ddr_bins_per_device=20)
    p.load(space)
    p.propose()
    p.feedback(partitionable=True, storage_constraint=topo)
    out = []
    prop = p.propose()
    while prop:
        out.append(prop)
        p.feedback(partitionable=True, storage_constraint=topo)
        prop = p.propose()
    return out


def test_reference_scenario_generous_ddr_prefers_fast_option():  # :317-333
    opts = [_opt("table_a", 1000, 0, 50.0), _opt("table_a", 100, 900, 80.0), _opt("table_a", 100, 1000, 10.0)]
    props = _run(opts, _topo(2, 2000, 2000))
    assert min(props, key=lambda p: sum(o.total_perf for o in p))[0].total_perf == 10.0


def test_reference_scenario_tight_ddr_rejects_it():  # :335-347
    opts = [_opt("table_a", 100, 900, 80.0), _opt("table_a", 100, 1000, 10.0)]
    props = _run(opts, _topo(1, 2000, 950))
    assert props and all(p[0].total_perf == 80.0 for p in props)


def test_reference_scenario_tied_options():  # :349-366
    opts = [_opt("table_a", 1000, 0, 50.0), _opt("table_a", 1000, 1000, 50.0)]
    props = _run(opts, _topo(1, 1100, 2000))
    assert props and all(p[0].total_perf == 50.0 for p in props)


def test_reference_scenario_joint_budget_forces_mixed_plan():  # :368-392
    opts = [_opt("table_a", 1500, 0, 50.0), _opt("table_a", 100, 1500, 40.0),
            _opt("table_b", 1500, 0, 50.0), _opt("table_b", 100, 1500, 40.0)]
    props = _run(opts, _topo(1, 2000, 2000))
    best = min(props, key=lambda p: sum(o.total_perf for o in p))
    assert sorted("hbm" if o.shards[0].storage.ddr == 0 else "ddr" for o in best) == ["ddr", "hbm"]


def test_reference_scenario_per_machine_ddr_prune():  # :394-415
    topo = _topo(4, 2000, 500, local=2)
    assert _run([_opt("t", 100, 1500, 10.0)], topo) == []
    assert len(_run([_opt("t", 100, 900, 10.0)], topo)) > 0


def test_reference_scenario_empty_search_space():  # :440-455
    p = DynamicProgrammingProposer()
    p.load([])
    assert p.propose() == []
    p.feedback(partitionable=True, storage_constraint=_topo(1, 1000, 1000))
    assert p.propose() is None


def _random_opts(seed, tables=5, options=4, hbm_bins=8, ddr_bins=8):  # plan_util_test.py:278-295
    rng = random.Random(seed)
    out = []
    for _ in range(tables):
        hbm = np.asarray([rng.uniform(0, hbm_bins / 2) for _ in range(options)], dtype=np.float32)
        ddr = np.asarray([rng.uniform(0, ddr_bins / 2) for _ in range(options)], dtype=np.float32)
        perf = np.asarray([rng.uniform(1, 100) for _ in range(options)], dtype=np.float32)
        out.append((hbm, ddr, perf, np.arange(options, dtype=np.int32)))
    return out


@pytest.mark.parametrize("seed", [0, 1, 7, 42, 1337])
def test_dp_equals_dense_oracle_reference_seeds(seed):  # :424-438
    opts = _random_opts(seed)
    assert {tuple(p) for p in dp_proposals(opts, 8, 8)} == {tuple(p) for p in dense_dp_proposals(opts, 8, 8)}


@pytest.mark.parametrize("seed", range(20))
def test_dp_equals_dense_oracle_wider(seed):
    rng = random.Random(1000 + seed)
    tables, options = rng.randint(1, 9), rng.randint(1, 6)
    hb, db = rng.choice([1, 5, 16, 40]), rng.choice([1, 3, 12])
    opts = _random_opts(seed, tables, options, hb * 1.3, db * 1.3)
    got, want = dp_proposals(opts, hb, db), dense_dp_proposals(opts, hb, db)
    assert got == want  # same proposals in the same (decreasing HBM) order
    for picks in got:  # every proposal respects the budgets
        assert sum(float(opts[t][0][j]) for t, j in enumerate(picks)) < hb


def _criteo():
    from torcheasyrec_amd.criteo import CRITEO_ROWS

    return [TableSpec(f"cat_{i}_emb", r, 16, [f"cat_{i}"]) for i, r in enumerate(CRITEO_ROWS)]


def test_criteo_plan_on_one_mi355x_node():
    plan = plan_tables(_criteo(), Topology(8), batch_size=8192)
    kinds = {n: p["sharding_type"] for n, p in plan.items()}
    big = [t.name for t in _criteo() if t.num_embeddings == 40_000_000]
    tiny = [t.name for t in _criteo() if t.num_embeddings <= 1_000]
    assert all(kinds[n] == "row_wise" for n in big)  # replication would all-reduce 2.5 GB per table per step
    assert all(kinds[n] == "data_parallel" for n in tiny)  # no exchange for the tables that hold most lookups
    assert all(p["block"] == 5_000_000 and p["ranks"] == list(range(8)) for n, p in plan.items() if n in big)
    js = plan_to_json(plan)
    assert '"compute_kernel": "fused"' in js and '"sharding_type": "row_wise"' in js


def test_plan_respects_hbm_and_constraints():
    tabs = [TableSpec("huge", 1_500_000_000, 16, ["a"]), TableSpec("small", 1000, 16, ["b"])]
    # 1.5 G rows x (64 B weights + 64 B Adagrad) = 192 GB: fits one 288 GB device only before the
    # reserve, so table_wise is out once the exchange buffers are counted; row_wise fits
    plan = plan_tables(tabs, Topology(8), batch_size=8192)
    assert plan["huge"]["sharding_type"] == "row_wise"
    plan = plan_tables(tabs, Topology(8), batch_size=8192, constraints={"small": ["table_wise"]})
    assert plan["small"]["sharding_type"] == "table_wise" and len(plan["small"]["ranks"]) == 1
    with pytest.raises(PlannerError):
        plan_tables(tabs, Topology(2, hbm_cap=40 * GB), batch_size=8192)
    with pytest.raises(PlannerError):  # torchrec types this runtime cannot execute (feature.proto:8)
        plan_tables(tabs, Topology(8), batch_size=8192, constraints={"small": ["key_value_wise"]})
    assert plan_tables(tabs, Topology(8), batch_size=8192, constraints={"small": ["column_wise"]})["small"]["sharding_type"] == "column_wise"


def _drive(proposer, space, topo):
    """the reference's test loop (plan_util_test.py:60-77): every proposal through the partitioner"""
    import copy

    proposer.load(space)
    best, best_plan, n = float("inf"), None, 0
    p = proposer.propose()
    while p:
        n += 1
        try:
            GreedyPartitioner().partition(copy.deepcopy(p), topo)
            perf = sum(o.total_perf for o in p)
            if perf < best:
                best, best_plan = perf, {o.fqn: o for o in p}
        except PlannerError:
            pass
        proposer.feedback(partitionable=True, storage_constraint=topo)
        p = proposer.propose()
    return best, best_plan, n


def _ref_tables():
    # plan_util_test.py:44-52: rows 1000**i, dim 10*i (dims rounded up to the kernels' multiple of 4)
    return [TableSpec(f"table_{i}", 1000 ** i, 4 * ((10 * i + 3) // 4), [f"feature_{i}"]) for i in range(1, 4)]


def test_dp_best_equals_grid_search_best():
    """plan_util_test.py:39-100 with this package's enumerator / partitioner: the best partitionable
    DP proposal has the perf of the best combination an exhaustive grid search finds."""
    import copy
    import itertools

    topo = Topology(2)
    space = EmbeddingEnumerator(topo, batch_size=8196).enumerate(_ref_tables())
    best_dp, plan_dp, n = _drive(DynamicProgrammingProposer(), space, topo)
    assert n >= 2 and plan_dp is not None
    by_table = {}
    for o in space:
        by_table.setdefault(o.fqn, []).append(o)
    best_grid, plan_grid = float("inf"), None
    for combo in itertools.product(*by_table.values()):
        try:
            GreedyPartitioner().partition(copy.deepcopy(list(combo)), topo)
        except PlannerError:
            continue
        perf = sum(o.total_perf for o in combo)
        if perf < best_grid:
            best_grid, plan_grid = perf, {o.fqn: o for o in combo}
    assert best_dp == pytest.approx(best_grid)
    assert {k: v.sharding_type for k, v in plan_dp.items()} == {k: v.sharding_type for k, v in plan_grid.items()}


def test_dp_with_prune_shards_the_table_no_single_device_holds():
    """plan_util_test.py:102-146: a device cap that the biggest table exceeds on its own.  (Sizes adapted:
    this planner also counts the optimizer state, so the table is 5e8 x 32 floats = 64 GB weights + 64 GB
    Adagrad state against 100 GB devices.)  table_wise is pruned before the knapsack; the best plan shards
    the table row_wise."""
    tabs = _ref_tables()[:2] + [TableSpec("table_3", 500_000_000, 32, ["feature_3"])]
    topo = Topology(2, hbm_cap=100 * GB)
    space = EmbeddingEnumerator(topo, batch_size=8196).enumerate(tabs)
    assert any(o.fqn == "table_3" and o.sharding_type == "table_wise" for o in space)  # enumerated, then pruned
    _, plan, n = _drive(DynamicProgrammingProposer(), space, topo)
    assert plan is not None and n >= 2
    assert plan["table_3"].sharding_type == "row_wise"


def test_column_wise_is_enumerated_only_on_request():
    """column_wise (feature.proto:8) is executable (sharding.MixedShardedEmbeddingBagCollection) but costs
    one exchange lane per column shard, so the enumerator offers it only when a constraint names it."""
    from torcheasyrec_amd.planner import EmbeddingEnumerator, TableSpec, Topology, plan_tables

    top = Topology(8)
    t = TableSpec("wide", 1_000_000, 64, ["w"])
    kinds = {o.sharding_type for o in EmbeddingEnumerator(top, 1024).enumerate([t])}
    assert kinds == {"data_parallel", "table_wise", "row_wise"}
    (o,) = EmbeddingEnumerator(top, 1024, {"wide": ["column_wise"]}).enumerate([t])
    assert o.sharding_type == "column_wise" and len(o.shards) == 8  # 64 / 8 = 8 columns per shard (a multiple of 4)
    assert [s.size for s in o.shards] == [(1_000_000, 8)] * 8 and [s.offset for s in o.shards] == [(0, 8 * j) for j in range(8)]
    # weights + elementwise Adagrad state of one column shard, plus its exchange buffers
    assert all(s.storage.hbm >= 1_000_000 * 8 * 4 * 2 for s in o.shards)
    assert all(s.perf >= 3 * top.collective_latency for s in o.shards)  # every shard pays its own three collectives
    (o12,) = EmbeddingEnumerator(Topology(8), 1024, {"t": ["column_wise"]}).enumerate([TableSpec("t", 100, 12, ["f"])])
    assert [s.size[1] for s in o12.shards] == [4, 4, 4]  # 12 columns: three shards of width 4
    plan = plan_tables([t, TableSpec("small", 50, 16, ["s"])], top, 1024, constraints={"wide": ["column_wise"]})
    assert plan["wide"]["sharding_type"] == "column_wise" and plan["wide"]["shard_dim"] == 8 and len(plan["wide"]["ranks"]) == 8
    assert sorted(plan["wide"]["ranks"]) == list(range(8))  # equal shards spread over the least-loaded ranks


def test_hierarchical_types_have_their_single_node_meaning():
    """table_row_wise / table_column_wise / grid_shard (feature.proto:8) on one node = row_wise / column_wise /
    column shards that are row-wise; across hosts they are refused."""
    from torcheasyrec_amd.planner import PlannerError, TableSpec, Topology, plan_tables

    tabs = [TableSpec("a", 1000, 16, ["a"]), TableSpec("g", 5000, 32, ["g"]), TableSpec("c", 400, 16, ["c"])]
    p = plan_tables(tabs, Topology(4), 256, constraints={"g": ["grid_shard"], "a": ["table_row_wise"], "c": ["table_column_wise"]})
    assert (p["a"]["sharding_type"], p["a"]["block"], p["a"]["ranks"]) == ("table_row_wise", 250, [0, 1, 2, 3])
    assert (p["g"]["sharding_type"], p["g"]["shard_dim"], p["g"]["ranks"]) == ("grid_shard", 8, [0, 1, 2, 3])
    assert (p["c"]["sharding_type"], p["c"]["shard_dim"], len(p["c"]["ranks"])) == ("table_column_wise", 4, 4)
    with pytest.raises(PlannerError, match="across hosts"):
        plan_tables(tabs, Topology(8, local_world_size=4), 256, constraints={"g": ["grid_shard"]})


def test_rowwise_adagrad_state_is_priced_as_allocated():
    """under the default interleaved row layout a row-wise Adagrad table holds a second D-wide half per row
    (its scalar state next to the weights): the storage estimate must count it, or plans that pass the HBM check OOM.
    split layout and FP16 tables keep the [rows] state array."""
    from torcheasyrec_amd.planner import EmbeddingEnumerator, TableSpec, Topology

    en = EmbeddingEnumerator(Topology(2), 1024)
    rows, D = 1_000_000, 16
    inter = TableSpec("t", rows, D, ["f"], optimizer="rowwise_adagrad")
    split = TableSpec("t", rows, D, ["f"], optimizer="rowwise_adagrad", row_layout="split")
    half = TableSpec("t", rows, D, ["f"], optimizer="rowwise_adagrad", bytes_per_element=2)
    assert en._state_bytes(inter, rows) == rows * D * 4
    assert en._state_bytes(split, rows) == rows * 4 and en._state_bytes(half, rows) == rows * 4
    tw = {o.sharding_type: o for o in en.enumerate([inter])}["table_wise"]
    assert tw.shards[0].storage.hbm >= 2 * rows * D * 4  # weights + the padded state half
