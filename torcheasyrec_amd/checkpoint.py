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

FORMAT_VERSION = 1


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


def _dcp_key(path: str, name: str, field: str) -> str:
    return f"{path}.embedding_bags.{name}.{field}"


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
    for path, col in cols:
        weights, states = col.table_weights(), col.table_states()
        for name, (lo, n, total, kind) in _placement(col).items():
            if use_dcp:  # every rank takes part for every table (ShardedTensor construction is a collective)
                spread = world > 1 and kind != "data_parallel" and hasattr(col, "shard_of")
                w = weights[name].detach()[:n].cpu().contiguous()
                dcp_model[_dcp_key(path, name, "weight")] = _dcp_entry(w, lo, n, [total] + list(w.shape[1:]), spread)
                if name in states:
                    m = states[name].detach()[:n].cpu().contiguous()
                    dcp_optim[_dcp_key(path, name, "momentum1")] = _dcp_entry(m, lo, n, [total] + list(m.shape[1:]), spread)
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
    mc, mc_sharded = _mc_of(model)
    zch = None
    if mc is not None and (mc_sharded or rank == 0):
        # raw id / access count / last access of every row + the step counter.  Sharded: every rank saves
        # the map of its own share (ids are routed by hash mod world size, so the maps only fit this world size)
        zch = {"iter": mc._iter, "sharded": mc_sharded, "world_size": world,
               "tables": {n: {"row_ids": m.row_ids.cpu(), "counts": m.counts.cpu(), "last_iter": m.last_iter.cpu()}
                          for n, m in mc.modules_by_table.items()}}
    if use_dcp:
        import torch.distributed.checkpoint as dcp

        for n_, t in _dense_state(model).items():
            dcp_model[f"dense.{n_}"] = t
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
                       "dcp_optimizer_state": bool(dcp_optim)}, f)
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
    if meta.get("format") != FORMAT_VERSION:
        raise ValueError(f"checkpoint format {meta.get('format')} != {FORMAT_VERSION}")
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
        mc._iter = int(zch["iter"])
        mc._cand = []
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
    rows THIS placement holds, torch.distributed.checkpoint reads them from whichever saved shards overlap."""
    import torch.distributed.checkpoint as dcp

    m_dir, o_dir = os.path.join(checkpoint_dir, "model", "dcp"), os.path.join(checkpoint_dir, "optimizer", "dcp")
    saved = set(dcp.FileSystemReader(m_dir).read_metadata().state_dict_metadata)
    tm, to, back = {}, {}, []
    for path, col in cols:
        weights, states = col.table_weights(), col.table_states()
        for name, (lo, n, total, kind) in _placement(col).items():
            if f"{path}/{name}" not in meta["tables"]:
                continue
            spread = world > 1 and kind != "data_parallel" and hasattr(col, "shard_of")
            w = torch.empty((n,) + tuple(weights[name].shape[1:]), dtype=weights[name].dtype)
            tm[_dcp_key(path, name, "weight")] = _dcp_entry(w, lo, n, [total] + list(w.shape[1:]), spread)
            back.append((weights[name], w, n))
            if name in states and meta.get("dcp_optimizer_state"):
                m = torch.empty((n,) + tuple(states[name].shape[1:]), dtype=states[name].dtype)
                to[_dcp_key(path, name, "momentum1")] = _dcp_entry(m, lo, n, [total] + list(m.shape[1:]), spread)
                back.append((states[name], m, n))
    dense = {f"dense.{n_}": torch.empty_like(t) for n_, t in _dense_state(model).items()}
    missing = [k for k in dense if k not in saved]
    if missing and strict:
        raise KeyError(f"checkpoint has no dense parameter(s) {missing}")
    extra = [k for k in saved if k.startswith("dense.") and k not in dense]
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
            mine[k[len("dense."):]].data.copy_(t)


def read_plan(checkpoint_dir: str) -> Dict[str, dict]:
    return json.load(open(os.path.join(checkpoint_dir, "plan")))
