"""Row-wise sharded EmbeddingBagCollection: one process per GPU, RCCL all-to-all over xGMI.

What it replaces.  In the reference, `DistributedModelParallel(module, plan, sharders)`
(/root/reference/tzrec/utils/dist_util.py:164-195, /root/reference/tzrec/main.py:783-804) swaps the
EBC for torchrec's sharded EBC [upstream 1.7.0]: KJTAllToAll (input dist) -> per-shard TBE lookup
-> PooledEmbeddingsReduceScatter / AllToAll (output dist), with the sharding type per table chosen
by the planner (`row_wise | table_wise | ...`, /root/reference/tzrec/protos/feature.proto:6-13).

MI355X-first design.  xGMI is a point-to-point mesh, so bytes per link are what matters.  torchrec's
row-wise output dist returns a dense [B_local, F*D] partial from every rank to every rank (7/8 of it
zeros when bags hold one id).  Here the exchange is at id granularity:

  forward   requester: K2 bucketize ids by owner rank            -> all-to-all (ids)
            owner:     tzr_rows_gather, one embedding row per id -> all-to-all (rows)
            requester: K5 pooled gather over the received rows (ids = unbucketize positions),
                       straight into feature-group layout
  backward  requester: tzr_lookup_grads, one gradient row per id -> all-to-all (rows)
            owner:     K6 plan + K7 fused optimizer with per-id gradients (grad_mode 1)

Placement ("plan").  Every table is split in contiguous blocks of ceil(rows/W) rows (torchrec
row-wise geometry, /root/reference/tzrec/utils/plan_util.py:1049-1060); block q of table t lives on
rank (q + rot[t]) mod W.  rot spreads tables with fewer rows than ranks over the node; with
block = rows it degenerates to table-wise placement on rank rot[t].  Dense parameters are
data-parallel: gradients are all-reduced and averaged like DDP (dist_util.py:170); sparse
gradients are NOT divided by the world size (torchrec behaviour).

One host sync per step (the per-peer id counts that size the all-to-all), as in torchrec.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from .dlrm import MLP
from .embedding import (EmbeddingBagCollection, EmbeddingBagConfig, FusedSparseOptimizer,
                        SparseOptimizerConfig, _OPT_KIND, _WD_MODE)
from .interaction import dot_interaction
from .sparse import KeyedJaggedTensor, block_bucketize


def row_wise_plan(rows: Sequence[int], world: int) -> Tuple[List[int], List[int]]:
    """(block_size[t], rot[t]).  Tables with fewer blocks than ranks are rotated so their blocks
    land on different ranks (greedy by rows already placed)."""
    blocks = [max(1, -(-r // world)) for r in rows]
    load = [0] * world
    rot = []
    for r, b in zip(rows, blocks):
        nblk = -(-r // b)
        if nblk >= world:
            rot.append(0)
            for q in range(world):
                load[q] += min(b, max(0, r - q * b))
            continue
        best, best_cost = 0, None
        for o in range(world):
            cost = max(load[(q + o) % world] + min(b, r - q * b) for q in range(nblk))
            if best_cost is None or cost < best_cost:
                best, best_cost = o, cost
        rot.append(best)
        for q in range(nblk):
            load[(q + best) % world] += min(b, r - q * b)
    return blocks, rot


class _ShardedLookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, kjt, dst_names, hook):
        outs, state = mod._forward_impl(kjt, dst_names)
        ctx.mod, ctx.state = mod, state
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.mod._backward_impl(ctx.state, grads)
        return None, None, None, None


class ShardedEmbeddingBagCollection(nn.Module):
    """Row-wise sharded pooled lookup with the optimizer fused in backward.

    Same constructor surface as `EmbeddingBagCollection` (tables are the GLOBAL configs); all tables
    must share one embedding_dim (DLRM / the fm+deep groups of DeepFM)."""

    def __init__(
        self,
        tables: Sequence[EmbeddingBagConfig],
        device: torch.device,
        optimizer: Optional[SparseOptimizerConfig] = None,
        groups: Optional[Dict[str, List[str]]] = None,
        row_layout: str = "interleaved",
        process_group: Optional[dist.ProcessGroup] = None,
    ) -> None:
        super().__init__()
        self.pg = process_group
        self.W = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        self._device = torch.device(device)
        self._global = list(tables)
        dims = {t.embedding_dim for t in tables}
        if len(dims) != 1:
            raise ValueError("sharded lookup needs one embedding_dim for all tables")
        self.dim = dims.pop()
        self._opt_cfg = optimizer
        self.block, self.rot = row_wise_plan([t.num_embeddings for t in tables], self.W)
        # local shard of every table: block q = (rank - rot) mod W
        local_cfgs = []
        for t, cfg in enumerate(self._global):
            q = (self.rank - self.rot[t]) % self.W
            lo = q * self.block[t]
            n = max(0, min(self.block[t], cfg.num_embeddings - lo))
            init = None
            if cfg.init_fn is not None:  # init the global table deterministically, keep my block
                def init(w, cfg=cfg, lo=lo, n=n):  # noqa: E306
                    full = torch.empty(cfg.num_embeddings, cfg.embedding_dim)
                    cfg.init_fn(full)
                    if n > 0:
                        w[:n].copy_(full[lo:lo + n])
            else:
                def init(w, rows=cfg.num_embeddings):  # noqa: E306
                    a = (1.0 / max(rows, 1)) ** 0.5  # same distribution as the unsharded table
                    w.uniform_(-a, a)
            local_cfgs.append(EmbeddingBagConfig(cfg.name, cfg.embedding_dim, max(n, 1),
                                                 list(cfg.feature_names), cfg.pooling, init))
        self.shard_rows = [c.num_embeddings for c in local_cfgs]
        # owner-side storage + optimizer state reuse the single-GPU module (storage only)
        self.local = EmbeddingBagCollection(local_cfgs, device=self._device, optimizer=optimizer,
                                            row_layout=row_layout)
        self.fused_optimizer = self.local.fused_optimizer
        # requester-side pooling runs K5 over the received rows: one pseudo table read by every key
        self._lookups = self.local._lookups  # (key, table, out_key) in table-then-feature order
        self._groups = groups
        self._hook = torch.zeros(0, requires_grad=True, device=self._device)
        self._req_meta: Dict[Tuple, dict] = {}
        self._own_meta: Optional[dict] = None
        self._rows_buf: Dict[int, tuple] = {}
        self._timers = None

    # -- descriptors ---------------------------------------------------------------------------
    def _layout_for(self, dst_names):
        if dst_names == ("__all__",):
            return [("__all__", [lk.out_key for lk in self._lookups])]
        return [(g, self._groups[g]) for g in dst_names]

    def _requester_meta(self, kjt_keys, layout) -> dict:
        ck = (tuple(kjt_keys), tuple((n, tuple(ks)) for n, ks in layout))
        m = self._req_meta.get(ck)
        if m is not None:
            return m
        key_index = {k: i for i, k in enumerate(kjt_keys)}
        Fn = len(self._lookups)
        feats = np.zeros(Fn, dtype=_lib.FEATURE_DT)
        feats["dst"] = -1
        by_out = {lk.out_key: i for i, lk in enumerate(self._lookups)}
        pool = {c.name: c.pooling for c in self._global}
        for i, lk in enumerate(self._lookups):
            feats[i]["table"] = 0
            feats[i]["key"] = key_index[lk.key]
            feats[i]["pooling"] = _lib.POOL_MEAN if pool[self._global[lk.table].name].lower() == "mean" else _lib.POOL_SUM
            feats[i]["order"] = i
        slots = []
        for d, (_, out_keys) in enumerate(layout):
            col = 0
            for ok in out_keys:
                i = by_out[ok]
                n = int(feats[i]["n_dst"])
                feats[i]["dst"][n] = d
                feats[i]["col"][n] = col
                feats[i]["n_dst"] = n + 1
                for c in range(self.dim // 4):
                    slots.append((i, c, d, col + 4 * c))
                col += self.dim
        # per-key bucketize geometry (keys not served by this module get block = 2^62: rank 0, unused)
        blk = np.full(len(kjt_keys), 1 << 62, dtype=np.int64)
        rot = np.zeros(len(kjt_keys), dtype=np.int32)
        served = np.zeros(len(kjt_keys), dtype=bool)
        key_table = np.zeros(len(kjt_keys), dtype=np.int32)
        for lk in self._lookups:
            k = key_index[lk.key]
            if served[k]:
                raise ValueError(f"key {lk.key} feeds more than one sharded table")
            served[k] = True
            blk[k], rot[k], key_table[k] = self.block[lk.table], self.rot[lk.table], lk.table
        if not served.all():
            raise ValueError("sharded lookup expects a KJT holding exactly the keys it serves")
        m = {
            "feats_np": feats, "slots_np": np.array(slots, dtype=_lib.SLOT_DT),
            "d_feats": _lib.upload_struct(feats, self._device),
            "d_slots": _lib.upload_struct(np.array(slots, dtype=_lib.SLOT_DT), self._device),
            "blk": torch.from_numpy(blk).to(self._device), "rot": torch.from_numpy(rot).to(self._device),
            "key_table": key_table, "widths": [len(ks) * self.dim for _, ks in layout],
        }
        self._req_meta[ck] = m
        return m

    def _owner_meta(self, key_table: np.ndarray) -> dict:
        """Descriptors of the owner side: W*F received keys (source-major), key (s, f) -> table."""
        if self._own_meta is not None:
            return self._own_meta
        F, W, T = len(key_table), self.W, len(self._global)
        K = W * F
        base = self.local._meta([lk.key for lk in self._lookups], self.local._default_layout())
        tables = base.tables_np.copy()
        feats = np.zeros(K, dtype=_lib.FEATURE_DT)
        feats["dst"] = -1
        kt = np.tile(key_table, W).astype(np.int32)
        # table-major order of the received keys: (table, source, key)
        order = np.lexsort((np.arange(K), kt))
        rank_of = np.empty(K, dtype=np.int32)
        rank_of[order] = np.arange(K, dtype=np.int32)
        feats["table"], feats["key"], feats["order"] = kt, np.arange(K, dtype=np.int32), rank_of
        for t in range(T):
            mine = np.nonzero(kt == t)[0]
            tables[t]["first_order"] = int(rank_of[mine].min()) if len(mine) else 0
            tables[t]["n_feats"] = len(mine)
        self._own_meta = {
            "d_tables": _lib.upload_struct(tables, self._device),
            "d_feats": _lib.upload_struct(feats, self._device),
            "d_key_table": torch.from_numpy(kt).to(self._device), "K": K,
            "max_rows": int(max(self.shard_rows)),
        }
        return self._own_meta

    def _recv_rows_buffer(self, n: int):
        """Persistent [n, D] buffer for the rows coming back from their owners plus the one-table
        descriptor that lets K5 pool over it (cached: no per-step upload)."""
        hit = self._rows_buf.get(n)
        if hit is None:
            D = self.dim
            rows_in = torch.empty(max(n, 1), D, dtype=torch.float32, device=self._device)
            pt = np.zeros(1, dtype=_lib.TABLE_DT)
            pt[0]["w"], pt[0]["rows"], pt[0]["dim"], pt[0]["w_stride"] = rows_in.data_ptr(), max(n, 1), D, D
            pt[0]["n_feats"] = len(self._lookups)
            hit = (rows_in, _lib.upload_struct(pt, self._device))
            if len(self._rows_buf) > 8:
                self._rows_buf.clear()
            self._rows_buf[n] = hit
        return hit

    # -- exchange ------------------------------------------------------------------------------
    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits) -> None:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.pg)

    def _forward_impl(self, kjt: KeyedJaggedTensor, dst_names):
        L = _lib.lib()
        dev, W, D = self._device, self.W, self.dim
        layout = self._layout_for(dst_names)
        rm = self._requester_meta(kjt.keys(), layout)
        om = self._owner_meta(rm["key_table"])
        F, B = len(kjt.keys()), kjt.stride()
        N = kjt.values().numel()
        stream = _lib.stream_ptr(dev)
        # 1. requester: bucketize by owner rank
        bkt, unb = block_bucketize(kjt, rm["blk"], W, return_permute=True, rank_offsets=rm["rot"])
        # per (dest rank, key) id counts -> owners (they become the owners' key segments)
        send_cnt = (bkt.offsets()[B::B] - bkt.offsets()[:-1:B]).contiguous()  # [W*F]
        recv_cnt = torch.empty_like(send_cnt)
        self._a2a(recv_cnt, send_cnt, None, None)
        both = torch.stack([send_cnt.view(W, F).sum(1), recv_cnt.view(W, F).sum(1)]).cpu()  # host sync
        send_splits, recv_splits = both[0].tolist(), both[1].tolist()
        n_recv = int(sum(recv_splits))
        # 2. ids to their owners
        recv_ids = torch.empty(n_recv, dtype=torch.int64, device=dev)
        self._a2a(recv_ids, bkt.values(), recv_splits, send_splits)
        key_start = torch.zeros(W * F + 1, dtype=torch.int64, device=dev)
        torch.cumsum(recv_cnt, 0, out=key_start[1:])
        # 3. owner: one row per received id
        rows_out = torch.empty(max(n_recv, 1), D, dtype=torch.float32, device=dev)
        _lib.check(L.tzr_rows_gather(_lib.ptr(om["d_tables"]), _lib.ptr(om["d_key_table"]),
                                     _lib.ptr(key_start), om["K"], _lib.ptr(recv_ids), n_recv,
                                     _lib.ptr(rows_out), D, D, stream), "tzr_rows_gather")
        # 4. rows back to the requesters (bucketized order)
        rows_in, d_pt = self._recv_rows_buffer(N)
        self._a2a(rows_in[:N], rows_out[:n_recv], send_splits, recv_splits)
        # 5. requester: pooled gather over the received rows, ids = position in bucketized order
        uniform = kjt.uniform_length() == 1
        offsets = None if uniform else kjt.offsets()
        outs = [torch.empty(B, w, dtype=torch.float32, device=dev) for w in rm["widths"]]
        dsts = (_lib.TzrDst * len(outs))()
        for i, o in enumerate(outs):
            dsts[i].ptr, dsts[i].stride = _lib.ptr(o), o.stride(0)
        _lib.check(L.tzr_pooled_fwd(_lib.ptr(d_pt), _lib.ptr(rm["d_feats"]), len(self._lookups),
                                    _lib.ptr(rm["d_slots"]), len(rm["slots_np"]), _lib.ptr(unb),
                                    _lib.ptr(offsets), _lib.ptr(kjt.weights_or_none()), B, dsts,
                                    len(outs), 1 if uniform else 0, stream), "tzr_pooled_fwd")
        state = {"kjt": kjt, "rm": rm, "om": om, "unb": unb, "recv_ids": recv_ids, "key_start": key_start,
                 "send_splits": send_splits, "recv_splits": recv_splits, "n_recv": n_recv,
                 "keep": (rows_in, d_pt)}
        return outs, state

    def _backward_impl(self, st, grads) -> None:
        if self.fused_optimizer is None:
            return
        L = _lib.lib()
        dev, W, D = self._device, self.W, self.dim
        kjt, rm, om = st["kjt"], st["rm"], st["om"]
        B, N, n_recv = kjt.stride(), kjt.values().numel(), st["n_recv"]
        stream = _lib.stream_ptr(dev)
        uniform = kjt.uniform_length() == 1
        offsets = None if uniform else kjt.offsets()
        gl = []
        for g, w in zip(grads, rm["widths"]):
            if g is None:
                g = torch.zeros(B, w, dtype=torch.float32, device=dev)
            gl.append(g.contiguous().float())
        gd = (_lib.TzrDst * len(gl))()
        for i, g in enumerate(gl):
            gd[i].ptr, gd[i].stride = _lib.ptr(g), g.stride(0)
        # 1. requester: one gradient row per id, in bucketized order
        grow = torch.empty(max(N, 1), D, dtype=torch.float32, device=dev)
        _lib.check(L.tzr_lookup_grads(_lib.ptr(rm["d_feats"]), len(self._lookups), _lib.ptr(offsets),
                                      _lib.ptr(kjt.weights_or_none()), B, 1 if uniform else 0,
                                      _lib.ptr(st["unb"]), gd, len(gl), _lib.ptr(grow), D, D, stream),
                   "tzr_lookup_grads")
        # 2. to the owners
        grecv = torch.empty(max(n_recv, 1), D, dtype=torch.float32, device=dev)
        self._a2a(grecv[:n_recv], grow[:N], st["recv_splits"], st["send_splits"])
        if n_recv == 0:
            return
        # 3. owner: sort by (table,row) + fused optimizer, gradients addressed per id
        K, T = om["K"], len(self._global)
        nbytes = L.tzr_pooled_bwd_workspace(n_recv, n_recv, K, T, 1, D)
        ws = _lib.workspace(nbytes, dev)
        _lib.check(L.tzr_pooled_bwd_plan(_lib.ptr(om["d_tables"]), T, _lib.ptr(om["d_feats"]), K, K,
                                         om["max_rows"], D, _lib.ptr(st["recv_ids"]),
                                         _lib.ptr(st["key_start"]), n_recv, n_recv, 1, 0,
                                         _lib.ptr(ws), ws.numel(), stream), "tzr_pooled_bwd_plan")
        cfg = self._opt_cfg
        opt = _lib.TzrSparseOptim()
        opt.kind = _OPT_KIND[cfg.kind]
        opt.weight_decay_mode = _WD_MODE[cfg.weight_decay_mode.lower()]
        opt.d_lr = _lib.ptr(self.fused_optimizer.lr_device(dev))
        opt.eps, opt.weight_decay, opt.max_gradient = cfg.eps, cfg.weight_decay, cfg.max_gradient
        opt.gradient_clipping = 1 if cfg.gradient_clipping else 0
        g1 = (_lib.TzrDst * 1)()
        g1[0].ptr, g1[0].stride = _lib.ptr(grecv), grecv.stride(0)
        _lib.check(L.tzr_pooled_bwd_apply(_lib.ptr(om["d_tables"]), _lib.ptr(om["d_feats"]), K, T, D,
                                          _lib.ptr(st["key_start"]), None, n_recv, n_recv, 1, 0, 1, g1, 1,
                                          opt, _lib.ptr(ws), ws.numel(), stream), "tzr_pooled_bwd_apply")

    # -- public API ------------------------------------------------------------------------------
    def forward_grouped(self, features: KeyedJaggedTensor, group_names=None) -> Dict[str, torch.Tensor]:
        names = tuple(group_names) if group_names is not None else tuple(self._groups)
        if torch.is_grad_enabled() and self.fused_optimizer is not None and self.training:
            outs = list(_ShardedLookupFn.apply(self, features, names, self._hook))
        else:
            outs, _ = self._forward_impl(features, names)
        return dict(zip(names, outs))

    def shard_of(self, name: str) -> Tuple[int, int]:
        """(first global row, rows) of table `name` held by this rank."""
        t = [c.name for c in self._global].index(name)
        q = (self.rank - self.rot[t]) % self.W
        lo = q * self.block[t]
        return lo, max(0, min(self.block[t], self._global[t].num_embeddings - lo))

    def table_weights(self) -> Dict[str, torch.Tensor]:
        return self.local.table_weights()

    def table_states(self) -> Dict[str, torch.Tensor]:
        return self.local.table_states()


class ShardedDLRM(nn.Module):
    """DLRM with row-wise sharded tables and data-parallel MLPs."""

    def __init__(self, tables, sparse_features, dense_dim, dense_mlp=(64, 16), final_mlp=(64, 32),
                 arch_with_sparse=True, device=None, sparse_optimizer=None, row_layout="interleaved",
                 process_group=None) -> None:
        super().__init__()
        self.pg = process_group
        self.dim = tables[0].embedding_dim
        self.num_sparse = len(sparse_features)
        self.arch_with_sparse = arch_with_sparse
        self.ebc = ShardedEmbeddingBagCollection(
            tables, device=device, optimizer=sparse_optimizer, groups={"sparse": list(sparse_features)},
            row_layout=row_layout, process_group=process_group)
        self.dense_mlp = MLP(dense_dim, dense_mlp).to(device)
        n = self.num_sparse + 1
        feat = n * (n - 1) // 2 + self.dim + (self.num_sparse * self.dim if arch_with_sparse else 0)
        self.final_mlp = MLP(feat, final_mlp).to(device)
        self.output_mlp = nn.Linear(final_mlp[-1], 1).to(device)
        # same dense parameters on every rank (DDP broadcasts rank 0's at construction)
        for p in self.dense_parameters():
            dist.broadcast(p.data, src=0, group=self.pg)
        self._flat = None

    def describe(self) -> str:
        w = self.ebc.W
        return (f"{w} ranks: all tables row-wise (block=ceil(rows/{w}), small tables rotated), id-granularity "
                f"all-to-all (ids, rows, grads) over RCCL; MLPs data-parallel with all-reduce")

    def dense_parameters(self):
        for m in (self.dense_mlp, self.final_mlp, self.output_mlp):
            yield from m.parameters()

    def forward(self, dense: torch.Tensor, sparse_features: KeyedJaggedTensor) -> torch.Tensor:
        sparse = self.ebc.forward_grouped(sparse_features)["sparse"]
        d = self.dense_mlp(dense)
        allf = dot_interaction(d, sparse, self.dim, cat_dense=True, cat_sparse=self.arch_with_sparse)
        return self.output_mlp(self.final_mlp(allf)).squeeze(1)

    def allreduce_dense_grads(self) -> None:
        """DDP semantics: average dense gradients over ranks, one flat all-reduce (217 KB)."""
        ps = [p for p in self.dense_parameters() if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, group=self.pg)
        flat.div_(dist.get_world_size(self.pg))
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
