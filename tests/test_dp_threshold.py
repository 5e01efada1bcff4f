"""planner.pick_dp_max_rows: the row count up to which a table is replicated, from the wire + kernel arithmetic (host logic)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd.criteo import CRITEO_ROWS  # noqa: E402
from torcheasyrec_amd.planner import dp_threshold_costs, pick_dp_max_rows  # noqa: E402


def test_dlrm_criteo_at_8_ranks_replicates_the_small_tables_only():
    best, costs = pick_dp_max_rows(CRITEO_ROWS, 16, 8, 8192)
    by = {c["dp_max_rows"]: c for c in costs}
    # the round-4 constant (65 536 -> every table up to 39 060 rows replicated) pays 89 us of all-reduce per step for 7.75 MB
    assert abs(by[39060]["replica_all_reduce_us"] - 88.7) < 0.5 and by[39060]["replicated_tables"] == 18
    assert best < 39060 and by[best]["total_us"] == min(c["total_us"] for c in costs)
    assert by[best]["wire_us"] <= 30.0  # (projection.wire_total_us <= 30)
    assert by[best]["total_us"] < by[39060]["total_us"] - 20.0
    # monotone pieces: more replicas = more all-reduce, less all-to-all and fewer exchange-path kernels
    for a, b in zip(costs, costs[1:]):
        assert b["replica_all_reduce_us"] >= a["replica_all_reduce_us"] and b["all_to_all_us"] <= a["all_to_all_us"] + 1e-9
        assert b["kernels_us"] <= a["kernels_us"] + 1e-9


def test_one_rank_and_batch_dependence():
    assert pick_dp_max_rows(CRITEO_ROWS, 16, 1, 8192)[0] == 65536 and all(c["wire_us"] == 0.0 for c in dp_threshold_costs(CRITEO_ROWS, 16, 1, 8192))
    # a larger per-rank batch makes the exchange dearer (per lookup) and the all-reduce no dearer (per row): replicate more
    assert pick_dp_max_rows(CRITEO_ROWS, 16, 8, 65536)[0] >= pick_dp_max_rows(CRITEO_ROWS, 16, 8, 8192)[0]
    assert pick_dp_max_rows(CRITEO_ROWS, 16, 8, 512)[0] <= pick_dp_max_rows(CRITEO_ROWS, 16, 8, 8192)[0]
