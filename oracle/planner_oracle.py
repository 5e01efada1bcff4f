"""CPU oracle of the sharding-plan proposer (TEST INFRASTRUCTURE ONLY -- nothing under
torcheasyrec_amd/ may import this file).

Restates, as a dense table walk in plain Python, the multi-choice knapsack the reference's
`DynamicProgrammingProposer` solves (/root/reference/tzrec/utils/plan_util.py:257-356 is the
production form; /root/reference/tzrec/utils/plan_util_test.py:184-276 holds the dense form its own
property test trusts).  Pinned by the reference's known-answer proposer scenarios transcribed in
tests/test_planner.py.

State after table t: for every (floor(hbm), floor(ddr)) cell, the lowest summed perf of any choice of
one option per table 0..t whose un-floored sums land in that cell, with those sums.  A candidate
replaces a cell's holder only if its perf is strictly lower (first come wins ties; candidates are
visited in ascending (source cell, option) order).  Result: for every HBM bin of the last layer, the
lowest-perf cell over its DDR bins, reported largest HBM bin first, each as [option id per table].
"""
from typing import List, Sequence, Tuple

import numpy as np


def dense_dp_proposals(table_opts: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]], hbm_bins: int,
                       ddr_bins: int) -> List[List[int]]:
    T = len(table_opts)
    if T == 0:
        return []
    INF = float("inf")
    # layer[h][d] = (perf, hbm_sum, ddr_sum); trail[t][h][d] = (option id, source h, source d)
    layer = [[None] * ddr_bins for _ in range(hbm_bins)]
    trail = []
    sources = [(0, 0, 0.0, 0.0, 0.0)]  # (h, d, perf, hbm_sum, ddr_sum): the empty plan
    for t in range(T):
        oh, od, op, oid = table_opts[t]
        nxt = [[None] * ddr_bins for _ in range(hbm_bins)]
        tr = [[None] * ddr_bins for _ in range(hbm_bins)]
        for (sh, sd, sp, shs, sds) in sources:
            for j in range(len(op)):
                h = shs + float(oh[j])
                d = sds + float(od[j])
                if h >= hbm_bins or d >= ddr_bins:
                    continue
                p = sp + float(op[j])
                hi, di = int(h), int(d)
                cur = nxt[hi][di]
                if cur is None or p < cur[0]:
                    nxt[hi][di] = (p, h, d)
                    tr[hi][di] = (int(oid[j]), sh, sd)
        trail.append(tr)
        layer = nxt
        sources = [(h, d) + layer[h][d] for h in range(hbm_bins) for d in range(ddr_bins) if layer[h][d] is not None]
        if not sources:
            return []
    out = []
    for h in range(hbm_bins - 1, -1, -1):
        best, best_d = INF, -1
        for d in range(ddr_bins):
            if layer[h][d] is not None and layer[h][d][0] < best:
                best, best_d = layer[h][d][0], d
        if best_d < 0:
            continue
        picks = [0] * T
        ch, cd = h, best_d
        for t in range(T - 1, -1, -1):
            picks[t], ch, cd = trail[t][ch][cd]
        out.append(picks)
    return out
