"""Zero-collision hash (managed collision) in front of the embedding lookup.

Reference wiring: `BaseFeature.mc_module` builds one torchrec `MCHManagedCollisionModule` per
`zch {...}` feature config (/root/reference/tzrec/features/feature.py:693-736; proto
/root/reference/tzrec/protos/feature.proto:31-47; policies and admission filters documented in
/root/reference/docs/source/feature/zch.md) and `ManagedCollisionEmbeddingBagCollection(ebc, mcc)`
remaps every KJT before the lookup (/root/reference/tzrec/modules/embedding.py:856-864).  torchrec is
not part of this stack; the semantics below restate its published behaviour and are fixed by
`oracle/zch_oracle.py` (parity with torchrec itself is UNPINNED: no golden vectors exist in the
reference for it):

  * a table of `zch_size` rows; row `zch_size-1` is the shared row of ids that have no row (yet);
  * every training step looks ids up (K13 `tzr_zch_remap`: open-addressing map in HBM), bumps
    `count` / `last_iter` of the rows hit and records absent ids as admission candidates;
  * every `eviction_interval` steps: candidates are coalesced (unique + counts), filtered by the
    admission function, and compete with the resident ids by score
        lfu           count
        lru           1 / max(iter - last_iter, 1) ** decay_exponent
        distance_lfu  count / max(iter - last_iter, 1) ** decay_exponent
    for the `zch_size-1` real rows.  Order: score descending, residents before candidates, raw id
    ascending.  Admitted candidates take the free / evicted rows in ascending row order; the rows
    that changed owner are reported so the caller can re-initialise them.

The periodic admission / eviction is a handful of torch sorts on the device (plumbing, off the hot
path); the per-step remap is the HIP kernel.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from .embedding import EmbeddingBagCollection
from .sparse import KeyedJaggedTensor, KeyedTensor

EMPTY = _lib.ZCH_EMPTY


@torch.no_grad()
def dynamic_threshold_filter(id_counts: torch.Tensor, threshold_skew_multiplier: float = 10.0):
    """Admit ids seen more than `multiplier` times the mean count (docs/source/feature/zch.md)."""
    threshold = id_counts.sum() * (threshold_skew_multiplier / max(id_counts.numel(), 1))
    return id_counts > threshold, threshold


@torch.no_grad()
def average_threshold_filter(id_counts: torch.Tensor):
    threshold = id_counts.float().mean()
    return id_counts.float() > threshold, threshold


@torch.no_grad()
def probabilistic_threshold_filter(id_counts: torch.Tensor, per_id_probability: float = 0.01):
    score = 1 - torch.pow(torch.full_like(id_counts, 1 - per_id_probability, dtype=torch.float), id_counts)
    threshold = torch.rand(id_counts.size(), device=id_counts.device)
    return score > threshold, threshold


def _first_zero_rows(flags: torch.Tensor, n: int, min_window: int = 1 << 20) -> torch.Tensor:
    """`torch.nonzero(flags == 0).squeeze(1)[:n]` without materialising every zero position first: a 200 M-row table that is
    mostly EMPTY has ~200 M rows without a keeper, and listing them all (1.6 GB of int64 + the allocation) was 90 of the
    97-128 ms of an admission round at BASELINE config 5's scale (profiles/r05u), to keep the first few hundred thousand.
    Windows of growing size until n rows are found: one window on a sparse table, log2 windows = one pass on a full one."""
    Z = flags.numel()
    if n <= 0 or Z == 0:
        return torch.zeros(0, dtype=torch.int64, device=flags.device)
    out, got, a, step = [], 0, 0, max(4 * n, min_window)
    while a < Z and got < n:
        b = min(Z, a + step)
        f = torch.nonzero(flags[a:b] == 0).squeeze(1)
        if f.numel():
            f = f[:n - got]
            out.append(f + a if a else f)
            got += f.numel()
        a, step = b, step * 2
    if not out:
        return torch.zeros(0, dtype=torch.int64, device=flags.device)
    return torch.cat(out) if len(out) > 1 else out[0]


def zch_config_from_msg(z) -> "ZchConfig":
    """`zch {...}` block of a feature config (protos/feature.proto:31-47) -> ZchConfig.  The
    admission filter is a lambda STRING evaluated with the documented helpers in scope, exactly as
    the reference does (tzrec/features/feature.py:700-714)."""
    policy, decay = "lfu", 1.0
    for k in ("lfu", "lru", "distance_lfu"):
        if z.has(k):
            policy = k
            body = z.one(k)
            decay = float(body.one("decay_exponent", 1.0)) if hasattr(body, "one") else 1.0
    func = None
    if z.has("threshold_filtering_func"):
        from functools import partial

        func = eval(str(z.one("threshold_filtering_func")), {  # noqa: S307 - config-provided, as in the reference
            "partial": partial, "nn": nn, "torch": torch, "average_threshold_filter": average_threshold_filter,
            "dynamic_threshold_filter": dynamic_threshold_filter,
            "probabilistic_threshold_filter": probabilistic_threshold_filter})
    return ZchConfig(int(z.one("zch_size")), int(z.one("eviction_interval", 5)), policy, decay, func)


@dataclass
class ZchConfig:
    zch_size: int
    eviction_interval: int = 5
    policy: str = "lfu"  # lfu | lru | distance_lfu
    decay_exponent: float = 1.0
    threshold_filtering_func: Optional[Callable] = None


class ManagedCollisionModule:
    """State of one ZCH table, all in HBM: the id->row map (cells), and per row its raw id, access
    count and last access iteration."""

    def __init__(self, cfg: ZchConfig, device: torch.device) -> None:
        if cfg.zch_size < 2:
            raise ValueError("zch_size must be >= 2 (the last row is the shared row of unseen ids)")
        if cfg.policy not in ("lfu", "lru", "distance_lfu"):
            raise ValueError(f"unknown eviction policy {cfg.policy!r}")
        self.cfg, self.device = cfg, torch.device(device)
        Z = cfg.zch_size
        cap = 1
        while cap < 2 * Z:
            cap *= 2
        self.capacity = cap
        self.keys = torch.full((cap,), EMPTY, dtype=torch.int64, device=device)
        self.rows = torch.zeros(cap, dtype=torch.int32, device=device)
        self.counts = torch.zeros(Z, dtype=torch.int64, device=device)
        self.last_iter = torch.zeros(Z, dtype=torch.int64, device=device)
        self.row_ids = torch.full((Z,), EMPTY, dtype=torch.int64, device=device)  # raw id held by every row
        self._event_trackers: list = []  # register_post_zch_event_tracker_fn
        self._tombstones = 0  # upper bound of the evicted cells still in the map (tzr_zch_update)

    def lookup_rows(self, raw_ids: torch.Tensor) -> torch.Tensor:
        """Row every raw id is served from right now (`zch_size - 1`, the shared row, for ids without a
        row): K13 without profiling -- what the delta-embedding dump publishes
        (/root/reference/tzrec/utils/delta_embedding_dump.py:1043-1094)."""
        raw_ids = raw_ids.contiguous()
        n = raw_ids.numel()
        out = torch.empty_like(raw_ids)
        if n == 0:
            return out
        mods = torch.frombuffer(bytearray(bytes(self.struct())), dtype=torch.uint8).to(self.device)
        km = torch.zeros(1, dtype=torch.int32, device=self.device)
        off = torch.tensor([0, n], dtype=torch.int64, device=self.device)
        _lib.check(_lib.lib().tzr_zch_remap(_lib.ptr(mods), _lib.ptr(km), 1, _lib.ptr(raw_ids), _lib.ptr(off), 1, 0, n, 0, 0,
                                            _lib.ptr(out), None, _lib.stream_ptr(self.device)), "tzr_zch_remap")
        return out

    def struct(self) -> "_lib.TzrZchModule":
        m = _lib.TzrZchModule()
        m.keys, m.rows, m.counts, m.last_iter = (_lib.ptr(self.keys), _lib.ptr(self.rows), _lib.ptr(self.counts),
                                                 _lib.ptr(self.last_iter))
        m.capacity, m.zch_size = self.capacity, self.cfg.zch_size
        return m

    def sorted_raw_ids(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sorted raw ids padded with EMPTY, their rows): the view torchrec keeps as
        `_mch_sorted_raw_ids` / `_mch_remapped_ids_mapping` (zch_util.py:29)."""
        ids, order = torch.sort(self.row_ids, stable=True)
        return ids, order

    def rebuild(self) -> None:
        occ = torch.nonzero(self.row_ids != EMPTY).squeeze(1)
        ids = self.row_ids[occ].contiguous()
        rows = occ.to(torch.int32).contiguous()
        s = self.struct()
        _lib.check(_lib.lib().tzr_zch_build(_lib.C.byref(s), _lib.ptr(ids), _lib.ptr(rows), ids.numel(),
                                            _lib.stream_ptr(self.device)), "tzr_zch_build")
        self._tombstones = 0

    def _notify(self, evicted: torch.Tensor, admitted: torch.Tensor, absent: torch.Tensor) -> None:
        for fn in self._event_trackers:
            fn(self, evicted, admitted, absent)

    def _select_kept(self, new_ids: torch.Tensor, new_cnt: torch.Tensor, cur_iter: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(row_kept uint8[Z - 1], new_kept uint8[n]): 1 = the resident keeps its row / the candidate gets one.
        MSB-first radix selection of the D-th smallest "drop key" (score image, kind, id image): every pass is one
        streaming read of row_ids / counts / last_iter into 2 048 bins that the host narrows (<= 13 passes)."""
        L, dev, cfg = _lib.lib(), self.device, self.cfg
        Zp, n = cfg.zch_size - 1, new_ids.numel()
        new_ids, new_cnt = new_ids.contiguous(), new_cnt.contiguous()
        s = self.struct()
        pol = {"lfu": 0, "lru": 1, "distance_lfu": 2}[cfg.policy]
        bins = torch.empty(2048, dtype=torch.int64, device=dev)
        row_kept = torch.empty(Zp, dtype=torch.uint8, device=dev)
        new_kept = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        FULL = (1 << 64) - 1

        def hist(field, shift, bits, t1, t2, t3):
            _lib.check(L.tzr_zch_select_hist(_lib.C.byref(s), _lib.ptr(self.row_ids), _lib.ptr(new_ids), _lib.ptr(new_cnt), n, cur_iter,
                                             pol, float(cfg.decay_exponent), field, shift, bits, t1, t2, t3, _lib.ptr(bins),
                                             _lib.stream_ptr(dev)), "tzr_zch_select_hist")
            return bins[:1 << bits].cpu().numpy()  # the one host wait of a pass

        def mark(drop_none, t1=0, t2=0, t3=0):
            _lib.check(L.tzr_zch_select_mark(_lib.C.byref(s), _lib.ptr(self.row_ids), _lib.ptr(new_ids), _lib.ptr(new_cnt), n, cur_iter,
                                             pol, float(cfg.decay_exponent), drop_none, t1, t2, t3, _lib.ptr(row_kept),
                                             _lib.ptr(new_kept), _lib.stream_ptr(dev)), "tzr_zch_select_mark")
            return row_kept, new_kept[:n]

        def narrow(h, need):
            c = np.cumsum(h)
            d = int(np.searchsorted(c, need, side="left"))  # first bin whose running count reaches `need`
            return d, need - (int(c[d - 1]) if d else 0), int(h[d])

        digits = ((53, 11), (42, 11), (31, 11), (20, 11), (9, 11), (0, 9))
        need, t1 = None, 0
        for shift, bits in digits:  # the score image
            h = hist(0, shift, bits, t1, 0, 0)
            if need is None:
                need = int(h.sum()) - Zp  # residents + candidates - rows = how many lose
                if need <= 0:
                    return mark(1)
            d, need, in_bin = narrow(h, need)
            t1 = (t1 << bits) | d
            if in_bin == need:  # the whole bin loses: the threshold is its largest possible key
                return mark(0, ((t1 + 1) << shift) - 1, 1, FULL)
        h = hist(1, 0, 1, t1, 0, 0)  # equal scores: candidates lose before residents ...
        t2 = 0
        if need > int(h[0]):
            need, t2 = need - int(h[0]), 1
        elif need == int(h[0]):
            return mark(0, t1, 0, FULL)
        t3 = 0
        for shift, bits in digits:  # ... and inside a kind the larger raw id first
            d, need, in_bin = narrow(hist(2, shift, bits, t1, t2, t3), need)
            t3 = (t3 << bits) | d
            if in_bin == need:
                return mark(0, t1, t2, ((t3 + 1) << shift) - 1)
        raise AssertionError("drop keys are unique: the last digit always resolves")

    @torch.no_grad()
    def update_and_evict(self, cand_ids: torch.Tensor, cur_iter: int) -> torch.Tensor:
        """Admit candidates / evict residents.  Returns the rows whose owner changed."""
        cfg, Z = self.cfg, self.cfg.zch_size
        dev = self.device
        new_ids, new_cnt = torch.unique(cand_ids, return_counts=True)
        absent = new_ids  # every id looked up without a row since the last round, before the admission filter
        if cfg.threshold_filtering_func is not None and new_ids.numel():
            keep, _ = cfg.threshold_filtering_func(new_cnt)
            new_ids, new_cnt = new_ids[keep], new_cnt[keep]
        if new_ids.numel() == 0:
            self._notify(absent[:0], absent[:0], absent)
            return torch.zeros(0, dtype=torch.int64, device=dev)
        # Who stays: residents and candidates ranked together (score desc, residents first, raw id asc), the first
        # Z - 1 keep / get a row.  At most len(new_ids) entries can lose, and which ones is a rank query -- radix
        # selection on the reversed order over the per-row arrays in place (csrc/zch_evict.hip) instead of sorting
        # residents + candidates three times.
        row_kept, new_kept = self._select_kept(new_ids, new_cnt, cur_iter)
        kept_new = torch.nonzero(new_kept).squeeze(1)  # ascending raw id (torch.unique's order) ...
        # ... into admission order: score descending (a candidate's age is 1: its score is its count, or 1 for lru)
        if cfg.policy != "lru" and kept_new.numel() > 1:
            kept_new = kept_new[torch.sort(new_cnt[kept_new], descending=True, stable=True).indices]
        free = _first_zero_rows(row_kept, kept_new.numel())  # ascending rows
        old = self.row_ids[free]
        admitted = new_ids[kept_new].contiguous()
        if self._event_trackers:
            self._notify(old[old != EMPTY], admitted, absent)
        self.row_ids[free] = admitted
        self.counts[free] = new_cnt[kept_new]
        self.last_iter[free] = cur_iter
        # the id -> row map: only the rows that changed hands (evicted ids leave a tombstone that insertions reuse);
        # a full rebuild once tombstones could fill an eighth of the cells
        self._tombstones += free.numel()
        if self._tombstones > self.capacity // 8:
            self.rebuild()
        else:
            s, old, rows32 = self.struct(), old.contiguous(), free.to(torch.int32).contiguous()
            _lib.check(_lib.lib().tzr_zch_update(_lib.C.byref(s), _lib.ptr(old), _lib.ptr(admitted), _lib.ptr(rows32), free.numel(),
                                                 _lib.stream_ptr(self.device)), "tzr_zch_update")
        return free


def register_post_zch_event_tracker_fn(mc_module: ManagedCollisionModule, fn) -> None:
    """The reference's hook of the same name (/root/reference/tzrec/utils/zch_util.py:105-134): after every
    admission / eviction round `fn(mc_module, evicted_raw_ids, admitted_raw_ids, absent_raw_ids)` -- the ids
    that lost their row, gained one, and (third argument, this library's addition: here lookups are recorded
    as rows, so ids WITHOUT a row have to come from the ZCH module) the ids looked up without a row."""
    mc_module._event_trackers.append(fn)


class ManagedCollisionEmbeddingBagCollection(nn.Module):
    """`forward(kjt) -> (KeyedTensor, remapped kjt)` like torchrec's module of the same name: the keys
    of tables listed in `zch` are remapped, then the wrapped collection does the lookup (its tables
    must have `num_embeddings == zch_size`)."""

    def __init__(self, ebc: EmbeddingBagCollection, zch: Dict[str, ZchConfig], reset_evicted_rows: bool = False) -> None:
        super().__init__()
        self.ebc = ebc
        ebc.expect_hot_rows = True  # every id without a slot reads the table's LAST row: thousands of lookups of one row per step
        self._device = ebc.device
        cfgs = {c.name: c for c in ebc.embedding_bag_configs()}
        self.modules_by_table: Dict[str, ManagedCollisionModule] = {}
        for name, z in zch.items():
            if name not in cfgs:
                raise KeyError(f"zch config for unknown table {name}")
            if cfgs[name].num_embeddings != z.zch_size:
                raise ValueError(f"{name}: num_embeddings {cfgs[name].num_embeddings} != zch_size {z.zch_size}")
            self.modules_by_table[name] = ManagedCollisionModule(z, self._device)
        self._order = list(self.modules_by_table)
        self._key_module: Dict[str, int] = {}
        for name in self._order:
            for f in cfgs[name].feature_names:
                self._key_module[f] = self._order.index(name)
        self._d_mods: Optional[torch.Tensor] = None
        self._km_cache: Dict[Tuple[str, ...], torch.Tensor] = {}
        self._iter = 0
        self._cand: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []  # (candidates, module per key, ids per key)
        self._reset = reset_evicted_rows
        self.last_evicted: Dict[str, torch.Tensor] = {}
        # Ring mode (`device_profile = True`, uniform bags): the step can be replayed from a hipGraph.  The iteration
        # number the remap stamps rows with is a device counter bumped inside the step; a step's candidates go to slot
        # `iter % slots` of a device ring (slots = the largest eviction interval), so nothing on the host names a step.
        # The host's `_iter` follows: `remap_step` counts eager steps, `replayed()` counts graph replays (and runs the
        # admission / eviction round when one is due -- eagerly, between replays).
        self.device_profile = False
        self._d_iter: Optional[torch.Tensor] = None
        self._ring: Optional[torch.Tensor] = None
        self._ring_meta: Optional[dict] = None

    @property
    def fused_optimizer(self):
        return self.ebc.fused_optimizer

    def _modules_device(self) -> torch.Tensor:
        if self._d_mods is None:
            raw = b"".join(bytes(self.modules_by_table[n].struct()) for n in self._order)
            self._d_mods = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self._device)
        return self._d_mods

    def remap(self, kjt: KeyedJaggedTensor, profile: bool) -> KeyedJaggedTensor:
        keys = tuple(kjt.keys())
        km = self._km_cache.get(keys)
        if km is None:
            km = torch.tensor([self._key_module.get(k, -1) for k in keys], dtype=torch.int32, device=self._device)
            self._km_cache[keys] = km
        values = kjt.values()
        n = values.numel()
        uniform = kjt.uniform_length() or 0
        out = torch.empty_like(values)
        if self.device_profile and profile:
            if uniform:
                return self._remap_ring(kjt, keys, km, out)
            if self._device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("not capturable: a training step with a zero-collision hash replays from a hipGraph in ring mode, "
                                   "which needs uniform bags (one id list shape per step); this batch has jagged bags")
            # (jagged bags outside a capture: the ordinary record below; `_evict` reads both)
        cand = torch.empty_like(values) if profile else None
        _lib.check(_lib.lib().tzr_zch_remap(
            _lib.ptr(self._modules_device()), _lib.ptr(km), len(keys), _lib.ptr(values),
            _lib.ptr(None if uniform else kjt.offsets()), kjt.stride(), uniform, n, self._iter, 1 if profile else 0,
            _lib.ptr(out), _lib.ptr(cand), _lib.stream_ptr(self._device)), "tzr_zch_remap")
        if profile and self._d_iter is not None:
            self._d_iter.fill_(self._iter)  # (this step was counted by the host: ring mode's device counter follows)
        if profile:  # positional candidates of this step + where each key's segment ends
            B = kjt.stride()
            seg = (torch.full((len(keys),), B * uniform, dtype=torch.int64, device=self._device) if uniform
                   else kjt.offsets()[::B].diff())
            self._cand.append((cand, km, seg))
        return KeyedJaggedTensor(kjt.keys(), out, kjt.lengths(), kjt.weights_or_none(), kjt._offsets, kjt.stride(),
                                 uniform_length=kjt.uniform_length())

    # -- ring mode -----------------------------------------------------------------------------------------------------
    def _remap_ring(self, kjt: KeyedJaggedTensor, keys, km: torch.Tensor, out: torch.Tensor) -> KeyedJaggedTensor:
        uniform = kjt.uniform_length() or 0
        if not uniform:
            raise ValueError("ManagedCollisionEmbeddingBagCollection.device_profile needs uniform bags (one shape per step)")
        B = kjt.stride()
        zk = [i for i, k in enumerate(keys) if self._key_module.get(k, -1) >= 0]
        meta = self._ring_meta
        capturing = self._device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing and (self._d_iter is None or meta is None or meta["keys"] != keys or meta["per_key"] != B * uniform):
            # (the counter and the ring are made OUTSIDE any graph: a fill captured with the step would reset them at every replay)
            raise RuntimeError("device_profile: run one eager training step of this shape before capturing it")
        if meta is None or meta["keys"] != keys or meta["per_key"] != B * uniform:
            if meta is not None and self._ring is not None and bool((self._ring != EMPTY).any()):
                raise ValueError("device_profile: the shape of the step changed while candidates are pending")
            slots = max(self.modules_by_table[n].cfg.eviction_interval for n in self._order)
            key_cand = torch.full((len(keys),), -1, dtype=torch.int32)
            for j, i in enumerate(zk):
                key_cand[i] = j
            self._ring = torch.full((slots, max(len(zk), 1), B * uniform), EMPTY, dtype=torch.int64, device=self._device)
            self._ring_meta = meta = {"keys": keys, "per_key": B * uniform, "slots": slots, "zk": zk,
                                      "key_cand": key_cand.to(self._device),
                                      "zk_module": [self._key_module[keys[i]] for i in zk]}
        if self._d_iter is None:  # (an eager step: `remap_step` has already counted it)
            self._d_iter = torch.full((1,), self._iter - 1, dtype=torch.int64, device=self._device)
        elif not capturing:  # (eager steps of the other record -- jagged bags -- may have been counted by the host alone)
            self._d_iter.fill_(self._iter - 1)
        self._d_iter.add_(1)  # (inside the step: captured with it)
        _lib.check(_lib.lib().tzr_zch_remap_ring(
            _lib.ptr(self._modules_device()), _lib.ptr(km), len(keys), _lib.ptr(kjt.values()), B, uniform, kjt.values().numel(),
            _lib.ptr(self._d_iter), 1, _lib.ptr(out), _lib.ptr(meta["key_cand"]), len(zk), _lib.ptr(self._ring), meta["slots"],
            _lib.stream_ptr(self._device)), "tzr_zch_remap_ring")
        return KeyedJaggedTensor(kjt.keys(), out, kjt.lengths(), kjt.weights_or_none(), kjt._offsets, kjt.stride(),
                                 uniform_length=kjt.uniform_length())

    def load_iter(self, n: int) -> None:
        """The iteration count of restored maps (checkpoint.restore_checkpoint): the host's count, the device counter of ring mode
        (the captured step bumps and stamps `last_iter` with it: it has to restart where the maps were saved, not where this
        process happened to be) -- and no candidate of the steps before the restore is left pending in either form."""
        self._iter = int(n)
        self._cand = []
        self._pending_evict = False
        if self._d_iter is not None:
            self._d_iter.fill_(int(n))  # (== the host's count between steps: the step itself adds the one)
        if self._ring is not None:
            self._ring.fill_(EMPTY)

    def replayed(self) -> None:
        """One replay of a captured training step has been queued: the host's iteration count follows the device's, and the
        admission / eviction round runs when it is due (eagerly, on the current stream, behind the replay)."""
        self._iter += 1
        if any(self._iter % self.modules_by_table[n].cfg.eviction_interval == 0 for n in self._order):
            self._evict()

    def _ring_candidates(self, j: int) -> torch.Tensor:
        """raw ids recorded for module j in the ring (live cells only)"""
        meta = self._ring_meta
        cols = [c for c, m in enumerate(meta["zk_module"]) if m == j] if meta else []
        if not cols:
            return torch.zeros(0, dtype=torch.int64, device=self._device)
        c = self._ring[:, cols, :].reshape(-1)
        return c[c != EMPTY]

    def _reset_rows(self, name: str, mod: ManagedCollisionModule, changed: torch.Tensor) -> None:
        if self._reset and changed.numel():
            w = self.ebc.table_weights()[name]
            a = (1.0 / mod.cfg.zch_size) ** 0.5
            w.data[changed] = torch.empty(changed.numel(), w.shape[1], device=w.device).uniform_(-a, a)
            st = self.ebc.table_states().get(name)
            if st is not None:
                st[changed] = 0

    @torch.no_grad()
    def _evict(self) -> None:
        """The admission / eviction round of every module that is due.  Candidates come from both records: the device ring
        (ring-mode steps) and the positional lists of ordinary eager steps -- a model can see both (warm-up steps with jagged
        bags before a capturable shape), and neither may be dropped."""
        ids = mods = None
        if self._cand:
            ids = torch.cat([c for c, _, _ in self._cand])
            mods = torch.cat([torch.repeat_interleave(km, seg) for _, km, seg in self._cand])
            live = ids != EMPTY
            ids, mods = ids[live], mods[live]
        meta = self._ring_meta if self._ring is not None else None
        if ids is None and meta is None:
            return
        due = [self._iter % self.modules_by_table[n].cfg.eviction_interval == 0 for n in self._order]
        for j, name in enumerate(self._order):
            if not due[j]:
                continue
            mod = self.modules_by_table[name]
            parts = []
            if meta is not None:
                parts.append(self._ring_candidates(j))
            if ids is not None:
                parts.append(ids[mods == j])
            changed = mod.update_and_evict(parts[0] if len(parts) == 1 else torch.cat(parts), self._iter)
            self.last_evicted[name] = changed
            self._reset_rows(name, mod, changed)
            if meta is not None:
                cols = [c for c, m in enumerate(meta["zk_module"]) if m == j]
                if cols:  # consumed: a module with a shorter interval than the ring must not meet them again
                    self._ring[:, cols, :] = EMPTY
        if ids is None:
            return
        if all(due):
            self._cand = []
        else:  # keep only the candidates of the modules that did not run
            keep = torch.ones_like(mods, dtype=torch.bool)
            for j, d in enumerate(due):
                if d:
                    keep &= mods != j
            ids, mods = ids[keep], mods[keep]
            uniq_mods = torch.arange(len(self._order), dtype=torch.int32, device=self._device)
            self._cand = [(ids, uniq_mods[mods.long()], torch.ones_like(ids))] if ids.numel() else []

    def pending_candidates(self, table: str) -> torch.Tensor:
        """Raw ids looked up without a row since `table`'s last admission round (not consumed)."""
        j = self._order.index(table)
        parts = [self._ring_candidates(j)] if self._ring is not None else []
        for c, km, seg in self._cand:
            mods = torch.repeat_interleave(km, seg)
            parts.append(c[(mods == j) & (c != EMPTY)])
        return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=self._device)

    def remap_step(self, kjt: KeyedJaggedTensor) -> KeyedJaggedTensor:
        """First half of a step for callers that run the lookup themselves (EmbeddingGroup): remap, and
        in training count the iteration and profile."""
        training = self.training and torch.is_grad_enabled()
        # (a step that is being CAPTURED is not a step yet: its replays are counted by `replayed()`, which also runs the
        # rounds that fall due -- never inside the graph)
        capturing = self._device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing and training and not self.device_profile:
            raise RuntimeError("a training step with a zero-collision hash is capturable in ring mode only: set "
                               "ManagedCollisionEmbeddingBagCollection.device_profile = True before the first step")
        if training and not capturing:
            self._iter += 1
        self._pending_evict = training and not capturing and any(
            self._iter % self.modules_by_table[n].cfg.eviction_interval == 0 for n in self._order)
        return self.remap(kjt, profile=training)

    def finish_step(self) -> None:
        """Second half: admission / eviction when due (after the lookup read the old mapping)."""
        if getattr(self, "_pending_evict", False):
            self._evict()
            self._pending_evict = False

    def forward(self, kjt: KeyedJaggedTensor) -> Tuple[KeyedTensor, KeyedJaggedTensor]:
        remapped = self.remap_step(kjt)
        out = self.ebc(remapped)
        self.finish_step()
        return out, remapped


class ShardedManagedCollisionEmbeddingBagCollection(nn.Module):
    """ZCH tables across ranks (BASELINE config 5: `zch {...}` features on 8 GPUs).

    torchrec shards an MCH module by raw-id value range and the embedding table row-wise next to it.
    Same ownership here, spelled for the id-granularity exchange of `sharding.py`: a raw id belongs
    to rank `splitmix64(id) mod W` (hash routing, `tzr_block_bucketize` with block size 0 -- raw ids
    are arbitrary 64-bit values, a value range would pile them on one rank); that rank holds
    `ceil(zch_size / W)` rows of the table AND the map raw id -> row for them.  So a lookup travels as a
    raw id, the owner remaps it (K13) right before its row gather, and admission / eviction is a
    purely local affair of every rank (no collective), run after the step's sparse update.

    `forward_grouped(kjt)` like the wrapped module; tables named in `zch` must have
    `num_embeddings == zch_size`; every rank's share needs >= 2 rows (one is the shared row of ids
    without a row)."""

    def __init__(self, tables, zch: Dict[str, ZchConfig], device, optimizer=None, groups=None, process_group=None,
                 dp_max_rows: int = 65536, reset_evicted_rows: bool = False) -> None:
        super().__init__()
        from .embedding import EmbeddingBagConfig  # noqa: F401
        from .sharding import ShardedEmbeddingBagCollection

        cfgs = {c.name: c for c in tables}
        for name, z in zch.items():
            if name not in cfgs:
                raise KeyError(f"zch config for unknown table {name}")
            if cfgs[name].num_embeddings != z.zch_size:
                raise ValueError(f"{name}: num_embeddings {cfgs[name].num_embeddings} != zch_size {z.zch_size}")
        self.sharded = ShardedEmbeddingBagCollection(tables, device=device, optimizer=optimizer, groups=groups,
                                                     process_group=process_group, dp_max_rows=dp_max_rows,
                                                     constraints={n: "row_wise" for n in zch})
        sh = self.sharded
        sh._hash_routed = set(zch)
        sh._owner_remap = self._owner_remap
        sh._after_backward = self._after_backward
        local = {}
        for name, z in zch.items():
            n = sh.shard_of(name)[1]
            if n < 2:
                raise ValueError(f"{name}: rank {sh.rank} would hold {n} rows; zch_size must be >= 2 * world size")
            local[name] = ZchConfig(n, z.eviction_interval, z.policy, z.decay_exponent, z.threshold_filtering_func)
        # the owner-side map + bookkeeping of this rank's share; its `ebc` is the local shard collection
        self.mc = ManagedCollisionEmbeddingBagCollection(sh.local, local, reset_evicted_rows=reset_evicted_rows)
        self._table_of_key = {f: c.name for c in tables for f in c.feature_names}
        self._pseudo_keys: Optional[List[str]] = None
        self._train_step = False

    @property
    def fused_optimizer(self):
        return self.sharded.fused_optimizer

    def plan(self):
        return self.sharded.plan()

    def _owner_remap(self, st: dict) -> torch.Tensor:
        """Raw ids received from the requesters -> rows of my shard (keys of plain tables pass through)."""
        sh = self.sharded
        if self._pseudo_keys is None:
            F = st["rm"]["rw_n"]
            rw_keys = [k for k, t, _ in sh._lookups if sh._global[t].name in sh.block]
            assert len(rw_keys) == F
            self._pseudo_keys = [f"{k}@from{src}" for src in range(sh.W) for k in rw_keys]
            for src in range(sh.W):
                for k in rw_keys:
                    t = self._table_of_key[k]
                    if t in self.mc.modules_by_table:
                        self.mc._key_module[f"{k}@from{src}"] = self.mc._order.index(t)
        training = self._train_step  # decided outside: autograd.Function.forward runs with grad mode off
        if training:
            self.mc._iter += 1
        # the received lookups as a one-sample KJT: key (source, feature) = segment of key_start
        pseudo = KeyedJaggedTensor(self._pseudo_keys, st["recv_ids"], st["recv_cnt"], None, st["key_start"], 1)
        st["zch_training"] = training
        return self.mc.remap(pseudo, profile=training).values()

    def _after_backward(self, st: dict) -> None:
        if st.get("zch_training") and any(self.mc._iter % m.cfg.eviction_interval == 0 for m in self.mc.modules_by_table.values()):
            self.mc._evict()

    def forward_grouped(self, features: KeyedJaggedTensor, group_names=None):
        self.sharded.train(self.training)
        self._train_step = self.training and torch.is_grad_enabled() and self.sharded.fused_optimizer is not None
        return self.sharded.forward_grouped(features, group_names)
