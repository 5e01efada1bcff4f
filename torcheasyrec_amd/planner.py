"""Sharding planner: which sharding type, and which rank(s), each embedding table gets.

SURVEY.md row a10.  The reference builds a torchrec `EmbeddingShardingPlanner` out of its own
enumerator / storage estimator / `DynamicProgrammingProposer`
(/root/reference/tzrec/utils/plan_util.py:93-206, 359-556, 917-1225) and persists the chosen
`{sharding_type, compute_kernel, ranks}` per table next to the checkpoint
(/root/reference/tzrec/utils/checkpoint_util.py:1152-1167).  torchrec is not part of this stack, so
the planner here is self-contained and speaks MI355X:

  Topology                8 x 288 GB HBM3E, xGMI full mesh (7 links x ~153 GB/s per GPU), host DDR
  EmbeddingEnumerator     per table, one ShardingOption for every sharding type the runtime can
                          execute (data_parallel | table_wise | row_wise; column_wise when a
                          constraint names it), with storage (weights +
                          fused-optimizer state + exchange buffers) and a perf estimate (seconds per
                          step from the measured gather / read-modify-write ceilings and link rate)
  DynamicProgrammingProposer
                          the reference's proposer semantics: multi-choice knapsack over discretised
                          (HBM, DDR) totals; keeps every reachable (hbm_bin, ddr_bin) cell with its
                          best perf; emits one proposal per HBM bin (best over DDR), largest HBM first
  GreedyPartitioner       places shards on ranks under the per-device HBM cap, least-loaded first
  plan_tables             search loop -> the plan dict `sharding.ShardedEmbeddingBagCollection` executes

The proposer is duck-typed like the reference's (options expose `.fqn`, `.shards[i].storage.{hbm,
ddr}`, `.total_storage`, `.total_perf`; topologies expose `.devices[i].storage`, optional
`.local_world_size`), so the reference's own proposer tests translate one-to-one
(tests/test_planner.py).
"""
from __future__ import annotations

import json
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

GB = 1 << 30


class PlannerError(RuntimeError):
    pass


@dataclass
class Storage:
    hbm: int = 0
    ddr: int = 0

    def __add__(self, o: "Storage") -> "Storage":
        return Storage(self.hbm + o.hbm, self.ddr + o.ddr)

    def fits_in(self, o: "Storage") -> bool:
        return self.hbm <= o.hbm and self.ddr <= o.ddr


@dataclass
class Shard:
    size: Tuple[int, int]  # rows, cols
    offset: Tuple[int, int]
    storage: Storage = field(default_factory=Storage)
    perf: float = 0.0
    rank: Optional[int] = None


@dataclass
class ShardingOption:
    fqn: str
    sharding_type: str
    compute_kernel: str
    shards: List[Shard]

    @property
    def total_storage(self) -> Storage:
        s = Storage()
        for sh in self.shards:
            s = s + sh.storage
        return s

    @property
    def total_perf(self) -> float:
        return float(sum(sh.perf for sh in self.shards))

    def __str__(self) -> str:
        return f"{self.fqn}:{self.sharding_type}:{self.compute_kernel}:{len(self.shards)}"


@dataclass
class Device:
    rank: int
    storage: Storage
    perf: float = 0.0


class Topology:
    """One node of MI355X by default: per-device HBM cap with a reserve for activations / dense
    parameters / RCCL buffers, host DDR split evenly over the local ranks."""

    def __init__(self, world_size: int, hbm_cap: int = 288 * GB, ddr_cap: int = 0, local_world_size: Optional[int] = None,
                 hbm_reserve: float = 0.15, hbm_gather_bw: float = 3.97e12, hbm_rmw_bw: float = 4.94e12,
                 link_bw: float = 153e9, links_per_device: int = 7, collective_latency: float = 20e-6) -> None:
        self.world_size = world_size
        self.local_world_size = local_world_size or world_size
        self.devices = [Device(r, Storage(int(hbm_cap * (1 - hbm_reserve)), ddr_cap)) for r in range(world_size)]
        self.hbm_gather_bw, self.hbm_rmw_bw = hbm_gather_bw, hbm_rmw_bw
        # all-to-all on a full mesh: every peer has its own link, so a rank's injection rate is
        # min(W-1, links) links in parallel
        self.a2a_bw = link_bw * max(1, min(world_size - 1, links_per_device))
        # all-reduce: RCCL lays one ring per link; the 2(W-1)/W traffic factor is applied by the caller
        self.ring_bw = self.a2a_bw
        # launch + rendezvous cost of one collective.  The runtime batches ALL exchanged tables into
        # one all-to-all per direction and ALL replicated tables into one all-reduce, so this is a
        # per-step constant, not a per-table cost: it is reported, never added to an option.
        self.collective_latency = collective_latency


@dataclass
class TableSpec:
    name: str
    num_embeddings: int
    embedding_dim: int
    feature_names: Sequence[str] = ()
    pooling_factor: float = 1.0  # ids per bag
    optimizer: str = "adagrad"  # adagrad | rowwise_adagrad | sgd
    bytes_per_element: int = 4
    # storage layout of the collection the plan is for (EmbeddingBagCollection(row_layout=...), default "interleaved"):
    # interleaved fp32 rows are [w(D) | state] with a 2 D row stride -- ALSO for row-wise Adagrad, whose one scalar per
    # row then occupies D floats of padding (embedding.py: the state sits in the sector next to its weights)
    row_layout: str = "interleaved"


class EmbeddingEnumerator:
    """All (table, sharding type) options with storage and perf filled in."""

    SHARDING_TYPES = ("data_parallel", "table_wise", "row_wise")

    def __init__(self, topology: Topology, batch_size: int, constraints: Optional[Dict[str, Sequence[str]]] = None) -> None:
        self.topology, self.batch_size, self.constraints = topology, int(batch_size), constraints or {}

    @staticmethod
    def _state_bytes_dim(t: TableSpec, rows: int, dim: int) -> int:
        """optimizer-state bytes of `rows` rows of `dim` columns AS ALLOCATED (the planner priced row-wise
        Adagrad at 4 bytes per row while the default interleaved layout allocates a second D-wide half per row)"""
        if t.optimizer == "adagrad":
            return rows * dim * 4
        if t.optimizer == "rowwise_adagrad":
            padded = t.row_layout == "interleaved" and t.bytes_per_element == 4  # (FP16 tables are never interleaved)
            return rows * dim * 4 if padded else rows * 4
        if t.optimizer == "adam":
            return rows * dim * 8
        return 0

    def _state_bytes(self, t: TableSpec, rows: int) -> int:
        return self._state_bytes_dim(t, rows, t.embedding_dim)

    # torchrec's hierarchical types priced with their single-node meaning (sharding.MixedShardedEmbeddingBagCollection)
    _SINGLE_NODE_ALIAS = {"table_row_wise": "row_wise", "table_column_wise": "column_wise"}

    def _option(self, t: TableSpec, kind: str) -> ShardingOption:
        top, W, B = self.topology, self.topology.world_size, self.batch_size
        if kind in self._SINGLE_NODE_ALIAS or kind == "grid_shard":
            if top.local_world_size != W:
                raise PlannerError(f"{t.name}: {kind} across hosts is not executable by this runtime (one node: local_world_size == world_size)")
            if kind == "grid_shard":
                # column shards that are each row-wise over the node: row_wise traffic and balance, one exchange
                # lane (three collectives) per column shard
                base = self._option(t, "row_wise")
                q4 = t.embedding_dim // 4
                k = max(c for c in range(1, min(W, q4) + 1) if q4 % c == 0)
                for sh in base.shards:
                    sh.perf += 3 * k * top.collective_latency / W
                return ShardingOption(t.name, "grid_shard", "fused", base.shards)
            base = self._option(t, self._SINGLE_NODE_ALIAS[kind])
            return ShardingOption(t.name, kind, "fused", base.shards)
        D, eb = t.embedding_dim, t.bytes_per_element
        nfeat = max(1, len(t.feature_names))
        ids = B * nfeat * t.pooling_factor  # lookups of this table issued by ONE rank per step
        row_b = D * eb
        rmw_b = 2 * row_b + 2 * self._state_bytes(t, 1)  # read + write of weight and state per touched row

        def weights(rows):
            return rows * row_b + self._state_bytes(t, rows)

        # Shard.perf = seconds the shard adds to the step's critical path.  The W shards of a
        # row_wise / data_parallel option work concurrently, so each carries 1/W of the per-rank
        # time; the single shard of a table_wise option serialises the lookups of all W ranks on its
        # owner.  (Sum over shards = the option's critical-path time, torchrec's total_perf.)
        if kind == "data_parallel":
            # local gather; backward = exact row sums into a dense [rows, D] buffer, all-reduce of
            # that buffer, dense update of every row
            dense = t.num_embeddings * D * 4
            per_rank = (ids * row_b / top.hbm_gather_bw + ids * (8 + row_b) / top.hbm_rmw_bw
                        + (2 * (W - 1) / W * dense / top.ring_bw if W > 1 else 0.0)
                        + t.num_embeddings * rmw_b / top.hbm_rmw_bw)
            shards = [Shard((t.num_embeddings, D), (0, 0), Storage(weights(t.num_embeddings) + dense + int(ids * 8)),
                            per_rank / W) for _ in range(W)]
        elif kind == "table_wise":
            # the owner serves the lookups of all W ranks; ids in, rows out, gradient rows in
            n = ids * W
            wire = (n * (8 + 2 * row_b) * (W - 1) / W) / top.a2a_bw
            perf = n * row_b / top.hbm_gather_bw + n * rmw_b / top.hbm_rmw_bw + wire
            shards = [Shard((t.num_embeddings, D), (0, 0), Storage(weights(t.num_embeddings) + int(n * (8 + 2 * row_b))), perf)]
        elif kind == "row_wise":
            blk = -(-t.num_embeddings // W)
            wire = (ids * (8 + 2 * row_b) * (W - 1) / W) / top.a2a_bw
            per_rank = ids * row_b / top.hbm_gather_bw + ids * rmw_b / top.hbm_rmw_bw + wire
            shards = []
            for q in range(W):
                rows = max(0, min(blk, t.num_embeddings - q * blk))
                shards.append(Shard((rows, D), (q * blk, 0), Storage(weights(rows) + int(ids * (8 + 2 * row_b))), per_rank / W))
        elif kind == "column_wise":
            # k column shards of width D/k (sharding.MixedShardedEmbeddingBagCollection): every shard
            # owner serves every id of the table's features with 1/k of the row, and every shard runs its
            # OWN exchange lane -- three collectives (ids, rows, gradient rows) that no other table
            # shares, so their launch + rendezvous latency is this option's to pay.  Enumerated only when
            # a constraint names it.
            q4 = D // 4
            k = max(c for c in range(1, min(W, q4) + 1) if q4 % c == 0)
            d = D // k
            n = ids * W
            piece_b = d * eb
            piece_rmw = 2 * piece_b + 2 * (d * 4 if t.optimizer == "adagrad" else 4 if t.optimizer == "rowwise_adagrad" else 0)
            wire = (n * (8 + 2 * piece_b) * (W - 1) / W) / top.a2a_bw
            perf = n * piece_b / top.hbm_gather_bw + n * piece_rmw / top.hbm_rmw_bw + wire + 3 * top.collective_latency
            st = t.num_embeddings * piece_b + self._state_bytes_dim(t, t.num_embeddings, d)
            shards = [Shard((t.num_embeddings, d), (0, j * d), Storage(st + int(n * (8 + 2 * piece_b))), perf) for j in range(k)]
        else:
            raise PlannerError(f"{t.name}: sharding type {kind!r} is not executable by this runtime")
        return ShardingOption(t.name, kind, "fused", shards)

    def enumerate(self, tables: Sequence[TableSpec]) -> List[ShardingOption]:
        out = []
        for t in tables:
            kinds = self.constraints.get(t.name, self.SHARDING_TYPES)
            for k in kinds:
                if k == "data_parallel" and self.topology.world_size == 1 and len(kinds) > 1:
                    continue  # one rank: replication is just table_wise with an extra buffer
                out.append(self._option(t, k))
        return out


def _best_per_cell(cell: np.ndarray, perf: np.ndarray) -> np.ndarray:
    """For every distinct cell, the index of its lowest-perf candidate (first one on ties);
    returned in ascending cell order."""
    if cell.size == 0:
        return np.zeros(0, np.int64)
    uniq, inv = np.unique(cell, return_inverse=True)
    best = np.full(uniq.size, np.inf, dtype=perf.dtype)
    np.minimum.at(best, inv, perf)
    hit = np.nonzero(perf == best[inv])[0]  # ascending candidate index
    _, first = np.unique(inv[hit], return_index=True)  # first hit of every cell
    return hit[first]


def dp_proposals(table_opts: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]], hbm_bins: int,
                 ddr_bins: int) -> List[List[int]]:
    """Multi-choice knapsack over (hbm, ddr) in bin units (the reference's
    `_sparse_dp_proposor_numpy`, plan_util.py:257-356, restated over reachable cells).

    table_opts[t] = (hbm[], ddr[], perf[], option id[]).  A state is a reachable cell
    (floor(hbm_sum), floor(ddr_sum)) holding the lowest perf that reaches it together with the
    un-floored float32 sums of that winner.  Output: one proposal per reachable HBM bin of the last
    layer (best perf over its DDR bins), largest HBM bin first; a proposal is one option id per
    table."""
    T = len(table_opts)
    if T == 0:
        return []
    st_h = np.zeros(1, np.float32)
    st_d = np.zeros(1, np.float32)
    st_p = np.zeros(1, np.float32)
    choice: List[np.ndarray] = []
    parent: List[np.ndarray] = []
    for t in range(T):
        oh, od, op, oid = (np.asarray(a) for a in table_opts[t])
        if st_p.size == 0 or op.size == 0:
            return []
        h = (st_h[:, None] + oh[None, :].astype(np.float32)).ravel()
        d = (st_d[:, None] + od[None, :].astype(np.float32)).ravel()
        p = (st_p[:, None] + op[None, :].astype(np.float32)).ravel()
        ok = np.nonzero((h < hbm_bins) & (d < ddr_bins))[0]
        if ok.size == 0:
            return []
        cell = h[ok].astype(np.int32).astype(np.int64) * ddr_bins + d[ok].astype(np.int32)
        win = ok[_best_per_cell(cell, p[ok])]
        n_opt = op.size
        choice.append(np.asarray(oid)[win % n_opt].astype(np.int64))
        parent.append((win // n_opt).astype(np.int64))
        st_h, st_d, st_p = h[win], d[win], p[win]
    last = _best_per_cell(st_h.astype(np.int32).astype(np.int64), st_p)[::-1]
    out = []
    for s in last:
        picks = [0] * T
        s = int(s)
        for t in range(T - 1, -1, -1):
            picks[t] = int(choice[t][s])
            s = int(parent[t][s])
        out.append(picks)
    return out


class DynamicProgrammingProposer:
    """Same protocol as the reference's proposer (plan_util.py:359-556): `load` the search space,
    `propose` -> list of options (first: the smallest option of every table), `feedback` advances;
    the DP runs on the first feedback, which must carry the topology."""

    def __init__(self, hbm_bins_per_device: int = 100, ddr_bins_per_device: int = 25) -> None:
        self._hbm_bins = max(int(hbm_bins_per_device), 1)
        self._ddr_bins = max(int(ddr_bins_per_device), 1)
        self._by_table: "OrderedDict[str, list]" = OrderedDict()
        self._proposals: List[List[int]] = []
        self._cursor = -1
        self._ran = False

    def load(self, search_space: Sequence, enumerator=None) -> None:
        self._by_table, self._proposals, self._cursor, self._ran = OrderedDict(), [], -1, False

        def footprint(o):
            return (o.total_storage.hbm or 0) + (o.total_storage.ddr or 0)

        for o in sorted(search_space, key=footprint):
            self._by_table.setdefault(o.fqn, []).append(o)

    def propose(self) -> Optional[list]:
        if not self._ran:
            return [opts[0] for opts in self._by_table.values()]
        if self._cursor < 0:
            return None
        picks = self._proposals[self._cursor]
        return [opts[i] for opts, i in zip(self._by_table.values(), picks)]

    def feedback(self, partitionable: bool, plan=None, perf_rating: Optional[float] = None, storage_constraint=None) -> None:
        if self._ran:
            self._cursor += 1
            if self._cursor >= len(self._proposals):
                self._cursor = -1
            return
        self._ran = True
        if storage_constraint is None:
            raise PlannerError("the first feedback must carry the topology (storage_constraint)")
        if not self._by_table:
            return
        devs = storage_constraint.devices
        n = len(devs)
        hbm_total = sum(d.storage.hbm or 0 for d in devs)
        ddr_total = sum(d.storage.ddr or 0 for d in devs)
        dev_hbm_max = max(d.storage.hbm or 0 for d in devs)
        # HBM is private to a device; host DDR is shared by the ranks of one machine
        per_host = max(getattr(storage_constraint, "local_world_size", None) or n, 1)
        host_ddr_max = max(sum(devs[i].storage.ddr or 0 for i in range(s, min(s + per_host, n))) for s in range(0, n, per_host))
        hbm_bins = self._hbm_bins * n if hbm_total > 0 else 1
        ddr_bins = self._ddr_bins * n if ddr_total > 0 else 1
        hbm_unit = hbm_total / hbm_bins if hbm_total > 0 else 1.0
        ddr_unit = ddr_total / ddr_bins if ddr_total > 0 else 1.0
        table_opts = []
        for opts in self._by_table.values():
            rows = []
            for j, o in enumerate(opts):
                if hbm_total > 0 and max((s.storage.hbm or 0) for s in o.shards) > dev_hbm_max:
                    continue  # one shard alone overflows a device
                if ddr_total > 0 and max((s.storage.ddr or 0) for s in o.shards) > host_ddr_max:
                    continue
                rows.append(((o.total_storage.hbm or 0) / hbm_unit if hbm_total > 0 else 0.0,
                             (o.total_storage.ddr or 0) / ddr_unit if ddr_total > 0 else 0.0, o.total_perf, j))
            a = np.asarray(rows, dtype=np.float64).reshape(-1, 4)
            table_opts.append((a[:, 0].astype(np.float32), a[:, 1].astype(np.float32), a[:, 2].astype(np.float32),
                               a[:, 3].astype(np.int32)))
        self._proposals = dp_proposals(table_opts, hbm_bins, ddr_bins)
        self._cursor = 0 if self._proposals else -1


class GreedyPartitioner:
    """Places the shards of a proposal: multi-shard options (row_wise, data_parallel) pin shard q to
    a fixed rank; single-shard options go to the device with the lowest accumulated perf that still
    has room, biggest first.  Raises PlannerError when something does not fit."""

    def partition(self, proposal: Sequence[ShardingOption], topology: Topology) -> List[ShardingOption]:
        W = len(topology.devices)
        free = [Storage(d.storage.hbm, d.storage.ddr) for d in topology.devices]
        load = [0.0] * W

        def take(rank, shard):
            if not shard.storage.fits_in(free[rank]):
                raise PlannerError(f"shard of {shard.size} does not fit on rank {rank}")
            free[rank] = Storage(free[rank].hbm - shard.storage.hbm, free[rank].ddr - shard.storage.ddr)
            load[rank] += shard.perf
            shard.rank = rank

        for o in proposal:
            if o.sharding_type in ("row_wise", "data_parallel", "table_row_wise", "grid_shard"):
                if len(o.shards) != W:
                    raise PlannerError(f"{o.fqn}: {o.sharding_type} needs one shard per rank")
                for q, sh in enumerate(o.shards):
                    take(q, sh)
        singles = [o for o in proposal if o.sharding_type not in ("row_wise", "data_parallel", "table_row_wise", "grid_shard")]
        for o in sorted(singles, key=lambda o: -(o.shards[0].storage.hbm + o.shards[0].storage.ddr)):
            for sh in o.shards:
                fits = [r for r in range(W) if sh.storage.fits_in(free[r])]
                if not fits:
                    raise PlannerError(f"{o.fqn}: no rank has room for a {sh.storage.hbm / GB:.1f} GB shard")
                take(min(fits, key=lambda r: (load[r], r)), sh)
        return list(proposal)


def plan_tables(tables: Sequence[TableSpec], topology: Topology, batch_size: int,
                constraints: Optional[Dict[str, Sequence[str]]] = None,
                proposer: Optional[DynamicProgrammingProposer] = None) -> Dict[str, dict]:
    """Search: every proposal of the DP that can be partitioned is scored by its summed perf; the best
    one becomes the plan.  Output entries carry what `ShardedEmbeddingBagCollection` needs
    (`sharding_type`, `block`, `rot`, `ranks`; column_wise: `ranks` = owner of every column shard, `shard_dim`;
    executed by `MixedShardedEmbeddingBagCollection`) plus what the reference persists
    (`compute_kernel`) and the estimates (`perf`, `hbm`)."""
    space = EmbeddingEnumerator(topology, batch_size, constraints).enumerate(tables)
    proposer = proposer or DynamicProgrammingProposer()
    part = GreedyPartitioner()
    proposer.load(space)
    best, best_perf = None, float("inf")
    prop = proposer.propose()
    while prop:
        try:
            placed = part.partition(prop, topology)
            perf = sum(o.total_perf for o in placed)
            if perf < best_perf:
                best_perf = perf
                best = [(o, [s.rank for s in o.shards]) for o in placed]
        except PlannerError:
            pass
        proposer.feedback(partitionable=True, storage_constraint=topology)
        prop = proposer.propose()
    if best is None:
        raise PlannerError("no sharding plan fits the topology: "
                           f"{sum(o.total_storage.hbm for o in proposer.propose() or []) / GB:.0f} GB needed at the minimum")
    rows = {t.name: t.num_embeddings for t in tables}
    W = topology.world_size
    plan: Dict[str, dict] = {}
    for o, ranks in best:
        e = {"sharding_type": o.sharding_type, "compute_kernel": o.compute_kernel, "perf": o.total_perf,
             "hbm": o.total_storage.hbm}
        if o.sharding_type in ("row_wise", "table_row_wise"):
            e.update({"block": max(1, -(-rows[o.fqn] // W)), "rot": 0, "ranks": list(range(W))})
        elif o.sharding_type == "grid_shard":
            q4 = o.shards[0].size[1] // 4
            e.update({"ranks": list(range(W)), "shard_dim": o.shards[0].size[1] // max(c for c in range(1, min(W, q4) + 1) if q4 % c == 0)})
        elif o.sharding_type == "table_column_wise":
            e.update({"ranks": list(ranks), "shard_dim": o.shards[0].size[1]})
        elif o.sharding_type == "table_wise":
            e.update({"block": max(1, rows[o.fqn]), "rot": ranks[0], "ranks": [ranks[0]]})
        elif o.sharding_type == "column_wise":
            e.update({"ranks": list(ranks), "shard_dim": o.shards[0].size[1]})
        else:
            e.update({"ranks": list(range(W))})
        plan[o.fqn] = e
    return {t.name: plan[t.name] for t in tables}


# ---- replicate or exchange: the row count up to which a table is data_parallel ------------------------------------------------
# A replicated table costs wire in proportion to its ROWS (its dense row-sum all-reduce: 2 (W-1)/W x rows x D x 4 bytes over
# one xGMI link, whatever the batch), a row-wise one in proportion to its LOOKUPS (one id out, one row back, one gradient row
# out per lookup, spread over all W-1 links) plus the exchange path's kernels per id.  sharding.make_plan takes the threshold
# as `dp_max_rows`; round 4 passed the constant 65 536 (18 of DLRM-Criteo's 26 tables replicated: a 7.75 MB all-reduce, 89 of
# the step's 100 us of wire at 8 ranks).  This is the arithmetic that picks it.
#
# Constants: 153 GB/s per xGMI link (MI355X_MICROARCH.md).  Kernel time of the two paths per 1 000 lookups: their AVERAGE
# cost on the 8 192-per-rank step is 1.07 (exchange: bucketize 15 + rows gather 10 + pooled gather 6 + gradient rows 7 +
# owners' update 33 us for 65 k ids) against 0.28 us (replicas: lookup 6 + row sums 26 + dense update 9 us for 147 k
# lookups, profiles/r05q/timeline.txt) -- but these kernels are latency-bound at that size, and what the threshold moves is
# the MARGINAL cost: the step measured 0.2636 / 0.2620 / 0.2647 ms with 8 / 13 / 15 exchanged features (profiles/r05s), i.e.
# +0.02 us per 1 000 ids moved from the replicas to the exchange.  The marginal figures are what the model uses.
XGMI_LINK_BYTES_PER_S = 153e9
EXCHANGE_US_PER_1K_IDS = 0.30
REPLICA_US_PER_1K_IDS = 0.28


def dp_threshold_costs(rows: Sequence[int], dim: int, world: int, batch_per_rank: int, ids_per_sample: float = 1.0,
                       capacity_factor: float = 1.25) -> List[dict]:
    """For every candidate threshold (0 and each distinct row count): the modelled per-step microseconds of wire and of the
    two lookup paths' kernels at `world` ranks.  [{dp_max_rows, replicated_tables, replicated_rows, wire_us, kernels_us, total_us}]"""
    W = max(int(world), 1)
    link = XGMI_LINK_BYTES_PER_S
    out = []
    for cand in [0] + sorted(set(int(r) for r in rows)):
        rep = [r for r in rows if r <= cand]
        exch = [r for r in rows if r > cand]
        n_rep_ids = len(rep) * batch_per_rank * ids_per_sample
        n_ex_ids = len(exch) * batch_per_rank * ids_per_sample
        if W > 1:
            ar = 2.0 * (W - 1) / W * sum(rep) * dim * 4.0 / link * 1e6
            per_peer = capacity_factor * n_ex_ids / W
            a2a = (8.0 * per_peer + 2 * 4.0 * dim * per_peer) / link * 1e6  # ids out, rows back, gradient rows out: every link at once
        else:
            ar = a2a = 0.0
        kern = EXCHANGE_US_PER_1K_IDS * n_ex_ids / 1e3 + REPLICA_US_PER_1K_IDS * n_rep_ids / 1e3
        out.append({"dp_max_rows": cand, "replicated_tables": len(rep), "replicated_rows": int(sum(rep)), "wire_us": ar + a2a,
                    "replica_all_reduce_us": ar, "all_to_all_us": a2a, "kernels_us": kern, "total_us": ar + a2a + kern})
    return out


def pick_dp_max_rows(rows: Sequence[int], dim: int, world: int, batch_per_rank: int, ids_per_sample: float = 1.0,
                     capacity_factor: float = 1.25) -> Tuple[int, List[dict]]:
    """(threshold with the smallest modelled wire + kernel time, the table of all candidates).  One rank: everything that can
    be replicated is local anyway -- the largest candidate below 2^16 rows, as before."""
    costs = dp_threshold_costs(rows, dim, world, batch_per_rank, ids_per_sample, capacity_factor)
    if world <= 1:
        return 65536, costs
    best = min(costs, key=lambda c: (c["total_us"], c["dp_max_rows"]))
    return int(best["dp_max_rows"]), costs


def plan_to_json(plan: Dict[str, dict]) -> str:
    """The `plan` file the reference writes next to a checkpoint (checkpoint_util.py:1152-1167):
    {table: {sharding_type, compute_kernel, ranks}}."""
    return json.dumps({k: {"sharding_type": v["sharding_type"], "compute_kernel": v.get("compute_kernel", "fused"),
                           "ranks": v["ranks"]} for k, v in plan.items()}, indent=1)
