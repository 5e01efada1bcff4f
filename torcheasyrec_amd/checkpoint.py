"""Checkpoint I/O for models whose tables live in this package's storage.

What the reference writes (/root/reference/tzrec/utils/checkpoint_util.py:1109-1167): a directory
with `model/` and `optimizer/` (torch.distributed.checkpoint of the sharded state dicts, so a run
can resume under a different plan) and a `plan` JSON `{module_path: {param: {sharding_type,
compute_kernel, ranks}}}`.  The same directory contract here, over plain files:

  <dir>/model/rank<r>.pt       {"dense": {param: tensor},
                                "tables": {"<module path>/<table>": {"lo", "n", "weight"[n, D]}},   rows [lo, lo+n)
                                           (module path: "ebc", "embedding_group.ebc", "embedding_group.ecs.<dim>")
                                "zch": {"iter", "sharded", "world_size", "tables": {table: {row_ids, counts, last_iter}}} | None}
                                        (unsharded maps: rank 0's file; sharded: every rank's file holds its own share)
  <dir>/optimizer/rank<r>.pt   {"tables": {table: {"lo", "n", "momentum1"}}, "sparse_lr", "adam_steps": {module path: step},
                                "dense": optimizer.state_dict()}        (sparse Adam: momentum1 = [exp_avg | exp_avg_sq])
  <dir>/plan                   the reference's plan JSON (rank 0)
  <dir>/meta.json              world size, {table: [rows, dim]}, format version (rank 0)

`save_checkpoint(..., tables_format="dcp")` puts the bulk -- table rows, their optimizer state and the dense
parameters -- through torch.distributed.checkpoint itself, as the reference does (`save(model.state_dict(),
checkpoint_id=<dir>/model)`, :1127-1133): `<dir>/model/dcp/` and `<dir>/optimizer/dcp/` hold `.metadata` + `__<rank>_0.distcp`
with one entry per table named like torchrec's parameters, `<module path>.embedding_bags.<table>.weight` (/ `.momentum1`):
a ShardedTensor whose shards are the row ranges the ranks own (replicated tables and dense parameters: plain tensors,
written once), so DCP's own planner re-shards on load.  The small remainder (ZCH maps, step counters, dense optimizer
state, plan, meta) stays in the files above.

Names in the DCP containers follow the REFERENCE's module paths (what `model.state_dict()` of tzrec's TrainWrapper ->
torchrec modules yields, read off /root/reference/tzrec/models/model.py:244-256 and tzrec/modules/embedding.py:194-195,
855-864,1193-1207; key-for-key equality with a torchrec run cannot be checked here: no torchrec wheel):
  pooled tables     model.embedding_group.emb_impls.__BASE__.ebc.embedding_bags.<table>.weight
                    (under a managed-collision wrapper: ...emb_impls.__BASE__.mc_ebc._embedding_module.embedding_bags.<table>.weight;
                    a model that holds its collection directly, like this package's DLRM: model.ebc.embedding_bags.<table>.weight)
  sequence tables   model.embedding_group.seq_emb_impls.__BASE__.ec_dict.<dim>.embeddings.<table>.weight
  dense parameters  model.<parameter path>
  optimizer state   state.<weight key>.<table>.momentum1                     (torchrec's fused-optimizer state naming)
  ZCH maps          ...mc_ebc._managed_collision_collection._managed_collision_modules.<table>._tzr_raw_ids / _tzr_counts /
                    _tzr_last_access_iter / _tzr_rows / _tzr_rows_state: the occupied entries of ALL ranks, keyed by RAW ID
                    (not by row): a restore at another world size routes every id to its new owner (splitmix64(id) mod W, as
                    the exchange does), gives it a row there and moves its embedding row and optimizer state along
                    (torchrec's `_mch_sorted_raw_ids` / `_mch_remapped_ids_mapping` are this list sorted by id; the
                    reference re-distributes them on a world-size change the same way, checkpoint_util.py:731-903).
The "files" format keeps the maps in the rank files, tied to the world size they were saved at.

Row shards are saved by their owner; replicated (data_parallel) tables and dense parameters by rank
0 only.  `restore_checkpoint` reads whichever row ranges the CURRENT placement needs from whichever
files hold them, so world size and sharding types may change between save and restore (the
re-sharding torch DCP gives the reference).  Tensors are loaded with mmap, a shard is never
materialised twice on the host.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

FORMAT_VERSION = 2  # 2: DCP entries under the reference's module paths (DCP_NAMES); "files" checkpoints of format 1 still load
DCP_NAMES = "reference-2"  # naming scheme of the DCP entries; a checkpoint written under another one is refused by name


def _zch_empty() -> int:
    from . import _lib

    return _lib.ZCH_EMPTY


def _rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _collections(model: nn.Module):
    """[(module path, collection)]: every table-holding collection of the model -- the pooled one
    (`ebc`, plain or sharded) and, for config-built models with SEQUENCE groups, the unpooled ones."""
    out = []
    for path, holder in (("", model), ("embedding_group.", getattr(model, "embedding_group", None))):
        if holder is None:
            continue
        ebc = getattr(holder, "ebc", None)
        if ebc is not None and not any(c is ebc for _, c in out):
            out.append((path + "ebc", ebc))
        for d, ec in getattr(holder, "ecs", {}).items():
            out.append((f"{path}ecs.{d}", getattr(ec, "sharded", None) or ec._store))
    if not out:
        raise ValueError("model has no `.ebc` / `.embedding_group.ebc` (EmbeddingBagCollection or ShardedEmbeddingBagCollection)")
    return out


def _mc_of(model: nn.Module):
    """(managed-collision (ZCH) wrapper or None, sharded?).  Unsharded: `model.mc` /
    `model.embedding_group.mc`.  Sharded: the group keeps the wrapper of the rank's OWN share of the
    maps under `_sharded_zch.mc` (embedding_group.py) -- every rank owns a different raw id -> row map."""
    for holder in (model, getattr(model, "embedding_group", None)):
        if holder is None:
            continue
        mc = getattr(holder, "mc", None)
        if mc is not None and hasattr(mc, "modules_by_table"):
            return mc, False
        sz = getattr(holder, "_sharded_zch", None)
        if sz is not None and getattr(sz, "mc", None) is not None:
            return sz.mc, True
    return None, False


def _fused_optimizers(path: str, col):
    """[(key, FusedSparseOptimizer)] of a collection.  A MixedShardedEmbeddingBagCollection runs one
    optimizer -- and one sparse-Adam step counter -- per exchange lane (sharding.py), each ticking in
    its own backward: all of them are saved, or lanes 1..k resume with a zero count under warm moments."""
    lanes = getattr(col, "lanes", None)
    if lanes is not None:
        return [(f"{path}#lane{i}", lane.fused_optimizer) for i, lane in enumerate(lanes) if lane.fused_optimizer is not None]
    f = getattr(col, "fused_optimizer", None)
    return [(path, f)] if f is not None else []


def _placement(ebc) -> Dict[str, Tuple[int, int, int, str]]:
    """{table: (first row held here, rows held here, total rows, sharding type)}"""
    out = {}
    if hasattr(ebc, "shard_of"):  # sharded module
        rows = {c.name: c.num_embeddings for c in ebc._global}
        for name, p in ebc.plan().items():
            lo, n = ebc.shard_of(name)
            out[name] = (lo, n, rows[name], p["sharding_type"])
    else:
        for c in ebc.embedding_bag_configs():
            out[c.name] = (0, c.num_embeddings, c.num_embeddings, "table_wise")
    return out


def _dense_state(model: nn.Module) -> Dict[str, torch.Tensor]:
    tables = {id(w) for _, c in _collections(model) for w in c.table_weights().values()}
    return {n: p.detach().cpu() for n, p in model.named_parameters() if id(p) not in tables and ".embedding_bags." not in n}


def _dcp_entry(local: Optional[torch.Tensor], lo: int, n: int, full_shape, sharded: bool):
    """A table (or its state) as torch.distributed.checkpoint wants it: a ShardedTensor whose one local shard is rows
    [lo, lo + n) of `full_shape` when the table is spread over ranks (collective: every rank calls, owners of no rows
    with no shard), the plain tensor otherwise."""
    if not sharded:
        return local
    from torch.distributed._shard.sharded_tensor import Shard, ShardedTensor, ShardMetadata

    rank = dist.get_rank()
    shards = []
    if n > 0:
        shards.append(Shard(local, ShardMetadata(shard_offsets=[lo] + [0] * (len(full_shape) - 1),
                                                 shard_sizes=[n] + list(full_shape[1:]), placement=f"rank:{rank}/cpu")))
    return ShardedTensor._init_from_local_shards(shards, list(full_shape))


BASE_GROUP = "__BASE__"  # tzrec/datasets/utils.py:28


def _ref_module_path(path: str, has_mc: bool) -> Tuple[str, str]:
    """(reference module path of the collection `path`, name of its table container)"""
    if path == "ebc":
        return "model.ebc", "embedding_bags"
    if path == "embedding_group.ebc":
        if has_mc:
            return f"model.embedding_group.emb_impls.{BASE_GROUP}.mc_ebc._embedding_module", "embedding_bags"
        return f"model.embedding_group.emb_impls.{BASE_GROUP}.ebc", "embedding_bags"
    if path.startswith("embedding_group.ecs."):
        return f"model.embedding_group.seq_emb_impls.{BASE_GROUP}.ec_dict.{path.rsplit('.', 1)[1]}", "embeddings"
    if path.startswith("ecs."):
        return f"model.ec_dict.{path.rsplit('.', 1)[1]}", "embeddings"
    return "model." + path, "embedding_bags"


def _dcp_key(path: str, name: str, field: str, has_mc: bool = False) -> str:
    mod, box = _ref_module_path(path, has_mc)
    w = f"{mod}.{box}.{name}.weight"
    return w if field == "weight" else f"state.{w}.{name}.{field}"


def _zch_prefix(path: str, name: str) -> str:
    mod, _ = _ref_module_path(path, True)
    root = mod[:-len("._embedding_module")] if mod.endswith("._embedding_module") else mod
    return f"{root}._managed_collision_collection._managed_collision_modules.{name}"


def route_rank(raw_ids: torch.Tensor, world: int) -> torch.Tensor:
    """owner rank of raw ids under hash routing: splitmix64(id) mod W (csrc/index_ops.hip, K2 hash mode)"""
    import numpy as np

    x = raw_ids.cpu().numpy().astype(np.int64)
    with np.errstate(over="ignore"):
        z = x.view(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return torch.from_numpy((z % np.uint64(world)).astype(np.int64))


def _concat_entry(local: torch.Tensor, world: int):
    """rows contributed by every rank, concatenated in rank order, as DCP wants them: a ShardedTensor over the global axis
    (collective: all_gather of the counts); world 1: the tensor itself.  Returns (entry, total rows)."""
    n = int(local.shape[0])
    if world == 1:
        return local, n
    counts = [None] * world
    dist.all_gather_object(counts, n)
    lo = sum(counts[:dist.get_rank()])
    total = sum(counts)
    return _dcp_entry(local, lo, n, [total] + list(local.shape[1:]), True), total


def save_checkpoint(checkpoint_dir: str, model: nn.Module, dense_optimizer: Optional[torch.optim.Optimizer] = None,
                    tables_format: str = "files") -> None:
    if tables_format not in ("files", "dcp"):
        raise ValueError("tables_format: 'files' or 'dcp'")
    use_dcp = tables_format == "dcp"
    dcp_model, dcp_optim = {}, {}
    rank, world = _rank_world()
    cols = _collections(model)
    for sub in ("model", "optimizer"):
        os.makedirs(os.path.join(checkpoint_dir, sub), exist_ok=True)
    m_tables, o_tables, plan_js, dims = {}, {}, {}, {}
    mc, mc_sharded = _mc_of(model)
    zch_names = set(mc.modules_by_table) if mc is not None else set()
    zch_totals = {}
    for path, col in cols:
        weights, states = col.table_weights(), col.table_states()
        for name, (lo, n, total, kind) in _placement(col).items():
            # the reference keeps a table under `mc_ebc._embedding_module` only when IT is managed-collision; the other tables
            # of the same data group stay under `ebc` (tzrec/modules/embedding.py:855-864).  Decided per table.
            has_mc = name in zch_names
            if use_dcp and name in zch_names:
                # a zero-collision-hash table: its rows mean something only through the raw id -> row map, so they travel
                # BY RAW ID (occupied rows only, all ranks concatenated): world-size independent
                mod = mc.modules_by_table[name]
                contributes = mc_sharded or rank == 0
                occ = torch.nonzero(mod.row_ids != _zch_empty()).squeeze(1) if contributes else torch.zeros(0, dtype=torch.int64)
                ids = mod.row_ids[occ].cpu()
                order = torch.argsort(ids)
                occ, ids = occ[order.to(occ.device)], ids[order]
                pre = _zch_prefix(path, name)
                parts = {"_tzr_raw_ids": ids, "_tzr_counts": mod.counts[occ].cpu(), "_tzr_last_access_iter": mod.last_iter[occ].cpu(),
                         "_tzr_rows": weights[name].detach()[occ].cpu().contiguous()}
                for k_, t_ in parts.items():
                    dcp_model[f"{pre}.{k_}"], zch_totals[name] = _concat_entry(t_.contiguous(), world)
                # the shared row (ids without a row are served -- and trained -- there): one per contributing rank
                Zl = mod.cfg.zch_size
                shared = weights[name].detach()[Zl - 1:Zl].cpu().contiguous() if contributes else weights[name].detach()[:0].cpu()
                dcp_model[f"{pre}._tzr_shared_rows"], _ = _concat_entry(shared, world)
                if name in states:
                    dcp_optim[f"state.{pre}._tzr_rows_state"], _ = _concat_entry(states[name].detach()[occ].cpu().contiguous(), world)
                    sh_st = states[name].detach()[Zl - 1:Zl].cpu().contiguous() if contributes else states[name].detach()[:0].cpu()
                    dcp_optim[f"state.{pre}._tzr_shared_rows_state"], _ = _concat_entry(sh_st, world)
                continue
            if use_dcp:  # every rank takes part for every table (ShardedTensor construction is a collective)
                spread = world > 1 and kind != "data_parallel" and hasattr(col, "shard_of")
                w = weights[name].detach()[:n].cpu().contiguous()
                dcp_model[_dcp_key(path, name, "weight", has_mc)] = _dcp_entry(w, lo, n, [total] + list(w.shape[1:]), spread)
                if name in states:
                    m = states[name].detach()[:n].cpu().contiguous()
                    dcp_optim[_dcp_key(path, name, "momentum1", has_mc)] = _dcp_entry(m, lo, n, [total] + list(m.shape[1:]), spread)
                continue
            if n == 0 or (kind == "data_parallel" and rank != 0):
                continue
            key = f"{path}/{name}"
            m_tables[key] = {"lo": lo, "n": n, "weight": weights[name].detach()[:n].cpu().contiguous()}
            if name in states:
                o_tables[key] = {"lo": lo, "n": n, "momentum1": states[name].detach()[:n].cpu().contiguous()}
        plan = col.plan() if hasattr(col, "plan") else {n: {"sharding_type": "table_wise", "ranks": [0]} for n in _placement(col)}
        plan_js[path] = {n: {"sharding_type": p["sharding_type"], "compute_kernel": p.get("compute_kernel", "fused"),
                             "ranks": list(p["ranks"])} for n, p in plan.items()}
        for c in (col._global if hasattr(col, "_global") else col.embedding_bag_configs()):
            dims[f"{path}/{c.name}"] = [c.num_embeddings, c.embedding_dim]
    zch = None
    if mc is not None and use_dcp:
        zch = {"iter": mc._iter, "by_raw_id": True, "world_size": world, "tables": {}}  # the maps themselves: DCP, by raw id
    elif mc is not None and (mc_sharded or rank == 0):
        # raw id / access count / last access of every row + the step counter.  Sharded: every rank saves
        # the map of its own share (ids are routed by hash mod world size, so the maps only fit this world size)
        zch = {"iter": mc._iter, "sharded": mc_sharded, "world_size": world,
               "tables": {n: {"row_ids": m.row_ids.cpu(), "counts": m.counts.cpu(), "last_iter": m.last_iter.cpu()}
                          for n, m in mc.modules_by_table.items()}}
    if use_dcp:
        import torch.distributed.checkpoint as dcp

        for n_, t in _dense_state(model).items():
            dcp_model[f"model.{n_}"] = t
        dcp.save(dcp_model, checkpoint_id=os.path.join(checkpoint_dir, "model", "dcp"), no_dist=world == 1)
        if dcp_optim:
            dcp.save(dcp_optim, checkpoint_id=os.path.join(checkpoint_dir, "optimizer", "dcp"), no_dist=world == 1)
    torch.save({"dense": _dense_state(model) if (rank == 0 and not use_dcp) else {}, "tables": m_tables, "zch": zch},
               os.path.join(checkpoint_dir, "model", f"rank{rank}.pt"))
    fo = getattr(cols[0][1], "fused_optimizer", None)
    adam_steps = {}
    for path, col in cols:  # sparse Adam: the step count of every fused optimizer that ticks one
        for key, f in _fused_optimizers(path, col):
            if getattr(f, "cfg", None) is not None and f.cfg.kind == "adam" and f._adam is not None:
                adam_steps[key] = float(f._adam[0])
    torch.save({"tables": o_tables, "sparse_lr": None if fo is None else fo.param_groups[0]["lr"], "adam_steps": adam_steps,
                "dense": dense_optimizer.state_dict() if (dense_optimizer is not None and rank == 0) else None},
               os.path.join(checkpoint_dir, "optimizer", f"rank{rank}.pt"))
    if rank == 0:
        with open(os.path.join(checkpoint_dir, "plan"), "w") as f:
            json.dump(plan_js, f)
        with open(os.path.join(checkpoint_dir, "meta.json"), "w") as f:
            json.dump({"format": FORMAT_VERSION, "world_size": world, "tables": dims, "tables_format": tables_format,
                       "dcp_optimizer_state": bool(dcp_optim), "dcp_names": DCP_NAMES, "zch_entries": zch_totals,
                       "zch_shared_rows": (world if mc_sharded else 1) if zch_totals else 0}, f)
    if world > 1:
        dist.barrier()


def _fill(dst: torch.Tensor, lo: int, n: int, pieces, field: str, name: str) -> int:
    """Copy the parts of [lo, lo+n) that `pieces` (list of saved shards) hold into dst[:n]."""
    done = 0
    for p in pieces:
        s, e = max(lo, p["lo"]), min(lo + n, p["lo"] + p["n"])
        if s < e:
            dst[s - lo:e - lo].copy_(p[field][s - p["lo"]:e - p["lo"]])
            done += e - s
    return done


def restore_checkpoint(checkpoint_dir: str, model: nn.Module, dense_optimizer: Optional[torch.optim.Optimizer] = None,
                       strict: bool = True) -> None:
    rank, world = _rank_world()
    meta = json.load(open(os.path.join(checkpoint_dir, "meta.json")))
    fmt = meta.get("format")
    is_dcp = meta.get("tables_format", "files") == "dcp"
    if fmt not in (1, FORMAT_VERSION):
        raise ValueError(f"checkpoint format {fmt}: this version reads formats 1 (tables_format 'files') and {FORMAT_VERSION}")
    if is_dcp and (fmt != FORMAT_VERSION or meta.get("dcp_names") != DCP_NAMES):
        # earlier revisions wrote the DCP entries under other names (`dense.*`, `<path>.embedding_bags.*`, later every table
        # of a collection with a ZCH table under `mc_ebc`): the template below would not find them
        raise ValueError(f"DCP checkpoint written with entry names {meta.get('dcp_names', 'legacy')!r} (format {fmt}); this version "
                         f"reads {DCP_NAMES!r} (format {FORMAT_VERSION}).  Restore it with the revision that wrote it and save again, "
                         "or save with tables_format='files', which is unchanged.")
    saved_world = int(meta["world_size"])
    cols = _collections(model)
    for path, col in cols:
        for name, (_, _, total, _) in _placement(col).items():
            key = f"{path}/{name}"
            if key not in meta["tables"]:
                if strict:
                    raise KeyError(f"checkpoint has no table {key}")
                continue
            if meta["tables"][key][0] != total:
                raise ValueError(f"{key}: checkpoint has {meta['tables'][key][0]} rows, model {total}")
    m_files = [torch.load(os.path.join(checkpoint_dir, "model", f"rank{r}.pt"), mmap=True, weights_only=True)
               for r in range(saved_world)]
    o_files = [torch.load(os.path.join(checkpoint_dir, "optimizer", f"rank{r}.pt"), mmap=True, weights_only=True)
               for r in range(saved_world)]
    use_dcp = meta.get("tables_format", "files") == "dcp"
    if use_dcp:
        _restore_dcp(checkpoint_dir, model, cols, meta, strict, world)
    with torch.no_grad():
        for path, col in cols:
            if use_dcp:
                break
            weights, states = col.table_weights(), col.table_states()
            for name, (lo, n, _, _) in _placement(col).items():
                key = f"{path}/{name}"
                if n == 0 or key not in meta["tables"]:
                    continue
                got = _fill(weights[name].detach(), lo, n, [f["tables"][key] for f in m_files if key in f["tables"]], "weight", key)
                if got != n:
                    raise ValueError(f"{key}: rows [{lo}, {lo + n}) only partly present in the checkpoint ({got} of {n})")
                if name in states:
                    pcs = [f["tables"][key] for f in o_files if key in f["tables"]]
                    if pcs:
                        _fill(states[name].detach(), lo, n, pcs, "momentum1", key)
                    elif strict:
                        raise KeyError(f"checkpoint has no optimizer state for {key}")
        dense = {} if use_dcp else m_files[0]["dense"]
        mine = dict(model.named_parameters())
        for n_, t in dense.items():
            if n_ in mine:
                mine[n_].data.copy_(t)
            elif strict:
                raise KeyError(f"checkpoint parameter {n_} not in the model")
    mc, mc_sharded = _mc_of(model)
    zch = None
    if mc is not None and use_dcp and (m_files[0].get("zch") or {}).get("by_raw_id"):
        mc.load_iter(int(m_files[0]["zch"]["iter"]))
        mc = None  # the maps came in through _restore_dcp, by raw id
    if mc is not None:
        if mc_sharded:
            if saved_world != world:
                raise ValueError(f"sharded ZCH maps were saved at world size {saved_world}: raw ids are routed by hash mod "
                                 f"world size, they cannot be restored at world size {world}")
            zch = m_files[rank].get("zch")
            if zch is not None and not zch.get("sharded", False):
                raise ValueError("the checkpoint holds an unsharded ZCH map; this model shards it")
        else:
            zch = m_files[0].get("zch")
            if zch is not None and zch.get("sharded", False):
                raise ValueError("the checkpoint holds per-rank ZCH maps; this model keeps one unsharded map")
        if zch is None and strict:
            raise KeyError("checkpoint has no zch state for this rank")
    if mc is not None and zch is not None:
        mc.load_iter(int(zch["iter"]))
        for n, m in mc.modules_by_table.items():
            if n in zch["tables"]:
                z = zch["tables"][n]
                m.row_ids.copy_(z["row_ids"])
                m.counts.copy_(z["counts"])
                m.last_iter.copy_(z["last_iter"])
                m.rebuild()
            elif strict:
                raise KeyError(f"checkpoint has no zch state for {n}")
    fo = getattr(cols[0][1], "fused_optimizer", None)
    if fo is not None and o_files[0].get("sparse_lr") is not None:
        fo.param_groups[0]["lr"] = o_files[0]["sparse_lr"]
    saved_steps = o_files[0].get("adam_steps") or {}
    for path, col in cols:
        for key, f in _fused_optimizers(path, col):
            # (older files hold one count per collection: every lane of it gets that one)
            t = saved_steps.get(key, saved_steps.get(path))
            if t is not None and f.cfg.kind == "adam":
                f.set_adam_step(float(t))
    if dense_optimizer is not None and o_files[0].get("dense") is not None:
        dense_optimizer.load_state_dict(o_files[0]["dense"])
    if world > 1:
        dist.barrier()


def _restore_dcp(checkpoint_dir: str, model: nn.Module, cols, meta: dict, strict: bool, world: int) -> None:
    """Tables, their optimizer state and the dense parameters from `<dir>/{model,optimizer}/dcp`: the template names the
    rows THIS placement holds, torch.distributed.checkpoint reads them from whichever saved shards overlap.  Zero-
    collision-hash tables come back by raw id (`_restore_zch_by_raw_id`)."""
    import torch.distributed.checkpoint as dcp

    m_dir, o_dir = os.path.join(checkpoint_dir, "model", "dcp"), os.path.join(checkpoint_dir, "optimizer", "dcp")
    saved = set(dcp.FileSystemReader(m_dir).read_metadata().state_dict_metadata)
    mc, mc_sharded = _mc_of(model)
    zch_names = set(mc.modules_by_table) if mc is not None else set()
    tm, to, back = {}, {}, []
    for path, col in cols:
        weights, states = col.table_weights(), col.table_states()
        for name, (lo, n, total, kind) in _placement(col).items():
            has_mc = name in zch_names  # per table, as in save_checkpoint
            if f"{path}/{name}" not in meta["tables"]:
                continue
            if name in zch_names and name in meta.get("zch_entries", {}):
                _restore_zch_by_raw_id(m_dir, o_dir, path, name, mc, mc_sharded, weights[name], states.get(name),
                                       int(meta["zch_entries"][name]), bool(meta.get("dcp_optimizer_state")), world,
                                       int(meta.get("zch_shared_rows", 0)))
                continue
            spread = world > 1 and kind != "data_parallel" and hasattr(col, "shard_of")
            w = torch.empty((n,) + tuple(weights[name].shape[1:]), dtype=weights[name].dtype)
            tm[_dcp_key(path, name, "weight", has_mc)] = _dcp_entry(w, lo, n, [total] + list(w.shape[1:]), spread)
            back.append((weights[name], w, n))
            if name in states and meta.get("dcp_optimizer_state"):
                m = torch.empty((n,) + tuple(states[name].shape[1:]), dtype=states[name].dtype)
                to[_dcp_key(path, name, "momentum1", has_mc)] = _dcp_entry(m, lo, n, [total] + list(m.shape[1:]), spread)
                back.append((states[name], m, n))
    dense = {f"model.{n_}": torch.empty_like(t) for n_, t in _dense_state(model).items()}
    table_like = (".embedding_bags.", ".embeddings.", "._managed_collision_modules.")
    missing = [k for k in dense if k not in saved]
    if missing and strict:
        raise KeyError(f"checkpoint has no dense parameter(s) {missing}")
    extra = [k for k in saved if k.startswith("model.") and k not in dense and not any(t in k for t in table_like)]
    if extra and strict:
        raise KeyError(f"checkpoint parameter(s) {extra} not in the model")
    dense = {k: t for k, t in dense.items() if k in saved}
    tm.update(dense)
    dcp.load(tm, checkpoint_id=m_dir, no_dist=world == 1)
    if to:
        dcp.load(to, checkpoint_id=o_dir, no_dist=world == 1)
    with torch.no_grad():
        for dst, src, n in back:
            if n > 0:
                dst.detach()[:n].copy_(src)
        mine = dict(model.named_parameters())
        for k, t in dense.items():
            mine[k[len("model."):]].data.copy_(t)


def _restore_zch_by_raw_id(m_dir: str, o_dir: str, path: str, name: str, mc, mc_sharded: bool, weight: torch.Tensor,
                           state: Optional[torch.Tensor], total: int, has_state: bool, world: int, n_shared: int = 0) -> None:
    """One zero-collision-hash table from its by-raw-id form: every rank reads the id list, keeps the ids the hash routes to
    it at THIS world size (all of them for an unsharded map), gives them rows 0 .. k-1 in id order, and takes their
    access statistics, embedding rows and optimizer state along.  (The whole list is read by every rank: a re-shard is
    not a hot path; rows of a 200 M-row table are 25 GB through the page cache.)"""
    import torch.distributed.checkpoint as dcp

    mod = mc.modules_by_table[name]
    pre = _zch_prefix(path, name)
    D = weight.shape[1:]
    tm = {f"{pre}._tzr_raw_ids": torch.empty(total, dtype=torch.int64), f"{pre}._tzr_counts": torch.empty(total, dtype=torch.int64),
          f"{pre}._tzr_last_access_iter": torch.empty(total, dtype=torch.int64),
          f"{pre}._tzr_rows": torch.empty((total,) + tuple(D), dtype=weight.dtype)}
    if total:
        dcp.load(tm, checkpoint_id=m_dir, no_dist=world == 1)
    ids = tm[f"{pre}._tzr_raw_ids"]
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    mine = (route_rank(ids, world) == rank) if (mc_sharded and world > 1) else torch.ones(total, dtype=torch.bool)
    sel = torch.nonzero(mine).squeeze(1)
    sel = sel[torch.argsort(ids[sel])]
    k = int(sel.numel())
    Z = mod.cfg.zch_size
    if k > Z - 1:
        raise ValueError(f"{name}: {k} ids hash to rank {rank} but its share of the table holds {Z - 1} rows (+ the shared row): "
                         "the checkpoint's occupancy does not fit this world size")
    with torch.no_grad():
        mod.row_ids.fill_(_zch_empty())
        mod.counts.zero_()
        mod.last_iter.zero_()
        if k:
            mod.row_ids[:k] = ids[sel].to(mod.row_ids.device)
            mod.counts[:k] = tm[f"{pre}._tzr_counts"][sel].to(mod.counts.device)
            mod.last_iter[:k] = tm[f"{pre}._tzr_last_access_iter"][sel].to(mod.last_iter.device)
            weight.detach()[:k].copy_(tm[f"{pre}._tzr_rows"][sel])
        mod.rebuild()
        if n_shared:  # the shared row of rank r at the saved world size goes to rank r mod that size
            sh = {f"{pre}._tzr_shared_rows": torch.empty((n_shared,) + tuple(D), dtype=weight.dtype)}
            dcp.load(sh, checkpoint_id=m_dir, no_dist=world == 1)
            weight.detach()[Z - 1].copy_(sh[f"{pre}._tzr_shared_rows"][rank % n_shared])
            if state is not None and has_state:
                ss = {f"state.{pre}._tzr_shared_rows_state": torch.empty((n_shared,) + tuple(state.shape[1:]), dtype=state.dtype)}
                dcp.load(ss, checkpoint_id=o_dir, no_dist=world == 1)
                state.detach()[Z - 1].copy_(ss[f"state.{pre}._tzr_shared_rows_state"][rank % n_shared])
        if state is not None and has_state and total:
            st = {f"state.{pre}._tzr_rows_state": torch.empty((total,) + tuple(state.shape[1:]), dtype=state.dtype)}
            dcp.load(st, checkpoint_id=o_dir, no_dist=world == 1)
            if k:
                state.detach()[:k].copy_(st[f"state.{pre}._tzr_rows_state"][sel])


def read_plan(checkpoint_dir: str) -> Dict[str, dict]:
    return json.load(open(os.path.join(checkpoint_dir, "plan")))
