"""CPU oracle of the zero-collision-hash remap (TEST INFRASTRUCTURE ONLY -- nothing under
torcheasyrec_amd/ may import this file).

PARITY UNPINNED against torchrec's MCHManagedCollisionModule (un-vendored dependency
torchrec==1.7.0, absent here; /root/reference holds no golden vectors for it): this file DEFINES the
semantics the HIP path is held to, from what the reference documents
(/root/reference/docs/source/feature/zch.md: score formulas and admission filters;
/root/reference/tzrec/protos/feature.proto:31-47: fields; /root/reference/tzrec/utils/zch_util.py:29:
empty-slot sentinel) plus the assumptions listed in torcheasyrec_amd/zch.py (shared last row,
distance clamped to >= 1, total order of the competition, free rows handed out ascending).

Plain dictionaries and Python sorts; small cases only.
"""
from typing import Callable, Dict, List, Optional, Sequence

EMPTY = (1 << 63) - 1


class ZchTable:
    def __init__(self, zch_size: int, eviction_interval: int = 5, policy: str = "lfu", decay_exponent: float = 1.0,
                 threshold_filtering_func: Optional[Callable] = None) -> None:
        self.Z, self.interval, self.policy, self.decay = zch_size, eviction_interval, policy, decay_exponent
        self.filter = threshold_filtering_func
        self.row_of: Dict[int, int] = {}
        self.row_ids = [EMPTY] * zch_size
        self.counts = [0] * zch_size
        self.last_iter = [0] * zch_size
        self.candidates: List[int] = []

    def remap(self, ids: Sequence[int], cur_iter: int, profile: bool) -> List[int]:
        out = []
        for x in ids:
            x = int(x)
            r = self.row_of.get(x)
            if r is None:
                out.append(self.Z - 1)
                if profile and x != EMPTY:
                    self.candidates.append(x)
            else:
                out.append(r)
                if profile:
                    self.counts[r] += 1
                    self.last_iter[r] = cur_iter
        return out

    def _score(self, cnt: int, last: int, cur_iter: int) -> float:
        dist = float(max(cur_iter - last, 1))
        if self.policy == "lfu":
            return float(cnt)
        age = dist if self.decay == 1.0 else dist ** self.decay
        return 1.0 / age if self.policy == "lru" else float(cnt) / age

    def update_and_evict(self, cur_iter: int) -> List[int]:
        uniq = sorted(set(self.candidates))
        cnt = {x: 0 for x in uniq}
        for x in self.candidates:
            cnt[x] += 1
        self.candidates = []
        if self.filter is not None and uniq:
            import torch

            keep, _ = self.filter(torch.tensor([cnt[x] for x in uniq]))
            uniq = [x for x, k in zip(uniq, keep.tolist()) if k]
        if not uniq:
            return []
        entries = []  # (-score, is_new, id, row or None, count)
        for r in range(self.Z - 1):
            if self.row_ids[r] != EMPTY:
                entries.append((-self._score(self.counts[r], self.last_iter[r], cur_iter), 0, self.row_ids[r], r, 0))
        for x in uniq:
            entries.append((-self._score(cnt[x], cur_iter, cur_iter), 1, x, None, cnt[x]))
        entries.sort(key=lambda e: (e[0], e[1], e[2]))
        kept = entries[: self.Z - 1]
        held = {e[3] for e in kept if e[1] == 0}
        free = [r for r in range(self.Z - 1) if r not in held]
        changed = []
        for e, r in zip([e for e in kept if e[1] == 1], free):
            old = self.row_ids[r]
            if old != EMPTY:
                del self.row_of[old]
            self.row_ids[r], self.counts[r], self.last_iter[r] = e[2], e[4], cur_iter
            changed.append(r)
        # residents that lost keep their row only if no candidate took it; a resident outside `kept` whose
        # row stayed free is still resident (nobody needed the row)
        self.row_of = {x: r for r, x in enumerate(self.row_ids) if x != EMPTY}
        return changed
