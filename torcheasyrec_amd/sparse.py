"""KeyedJaggedTensor / KeyedTensor containers with torchrec's field semantics.

tzrec hands the embedding stack a ``torchrec.sparse.jagged_tensor.KeyedJaggedTensor`` built at
/root/reference/tzrec/datasets/data_parser.py:576-585 (keys, values int64 key-major, lengths per
(key, sample), optional weights, stride = B) and reads ``.keys() / .values() / .lengths() /
.offsets() / .weights_or_none() / .stride() / .length_per_key()`` from it; pooled output comes back
as a ``KeyedTensor`` (``.keys() / .length_per_key() / .values()``,
/root/reference/tzrec/modules/embedding.py:943-947).  These classes keep those names and meanings;
the offsets scan (K3) and the key permute (K1) run on the gfx950 kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _lib


class KeyedTensor:
    """Dense [B, sum(length_per_key)] tensor with named column blocks."""

    def __init__(self, keys: Sequence[str], length_per_key: Sequence[int], values: torch.Tensor):
        self._keys = list(keys)
        self._length_per_key = [int(x) for x in length_per_key]
        self._values = values

    def keys(self) -> List[str]:
        return self._keys

    def length_per_key(self) -> List[int]:
        return self._length_per_key

    def values(self) -> torch.Tensor:
        return self._values

    def offset_per_key(self) -> List[int]:
        out = [0]
        for n in self._length_per_key:
            out.append(out[-1] + n)
        return out

    def to_dict(self) -> Dict[str, torch.Tensor]:
        off = self.offset_per_key()
        return {k: self._values[:, off[i] : off[i + 1]] for i, k in enumerate(self._keys)}

    def to(self, device, non_blocking: bool = False) -> "KeyedTensor":
        return KeyedTensor(self._keys, self._length_per_key, self._values.to(device, non_blocking=non_blocking))

    def record_stream(self, stream) -> None:
        if self._values.is_cuda:
            self._values.record_stream(stream)

    @staticmethod
    def regroup_as_dict(
        keyed_tensors: Sequence["KeyedTensor"], groups: Sequence[Sequence[str]], keys: Sequence[str]
    ) -> Dict[str, torch.Tensor]:
        """Column-concat of each group's blocks (torchrec KeyedTensor.regroup_as_dict as used at
        /root/reference/tzrec/modules/embedding.py:972-976).  The pooled-embedding blocks never take
        this route in this package (the forward kernel writes group layout directly); it serves
        the dense KeyedTensor and tests."""
        blocks: Dict[str, torch.Tensor] = {}
        for kt in keyed_tensors:
            blocks.update(kt.to_dict())
        return {name: torch.cat([blocks[k] for k in group], dim=1) for name, group in zip(keys, groups)}


class JaggedTensor:
    """values ([N, D] embedding rows, or [N] ids) + lengths [B] + offsets [B+1] of one key (torchrec
    JaggedTensor fields)."""

    def __init__(self, values: torch.Tensor, lengths: torch.Tensor, offsets: torch.Tensor,
                 weights: Optional[torch.Tensor] = None) -> None:
        self._values, self._lengths, self._offsets, self._weights = values, lengths, offsets, weights

    def values(self) -> torch.Tensor:
        return self._values

    def lengths(self) -> torch.Tensor:
        return self._lengths

    def offsets(self) -> torch.Tensor:
        return self._offsets

    def to_padded_dense(self, desired_length: int, padding_value: float = 0.0) -> torch.Tensor:
        from .sequence import jagged_to_padded_dense  # K12; float rows only

        return jagged_to_padded_dense(self._values, self._offsets, desired_length, padding_value)

    def weights_or_none(self) -> Optional[torch.Tensor]:
        return self._weights

    # Pipelineable pieces a Batch needs for `sequence_dense_features`
    def to(self, device, non_blocking: bool = False) -> "JaggedTensor":
        return JaggedTensor(self._values.to(device, non_blocking=non_blocking), self._lengths.to(device, non_blocking=non_blocking),
                            self._offsets.to(device, non_blocking=non_blocking))

    def record_stream(self, stream) -> None:
        for t in (self._values, self._lengths, self._offsets):
            if t.is_cuda:
                t.record_stream(stream)

    def pin_memory(self) -> "JaggedTensor":
        return JaggedTensor(self._values.pin_memory(), self._lengths.pin_memory(), self._offsets.pin_memory())

    @staticmethod
    def from_lengths(values: torch.Tensor, lengths: torch.Tensor) -> "JaggedTensor":
        off = torch.zeros(lengths.numel() + 1, dtype=torch.int64, device=lengths.device)
        torch.cumsum(lengths.to(torch.int64), 0, out=off[1:])
        return JaggedTensor(values, lengths, off)


# the key permutations of a model are a handful of fixed tuples: their device copies are made once (a host list -> device copy
# is a synchronising transfer, and not allowed while a hipGraph is being captured)
_PERM_CACHE: Dict[tuple, torch.Tensor] = {}


def _perm_tensor(indices: tuple, dev: torch.device) -> torch.Tensor:
    key = (indices, str(dev))
    t = _PERM_CACHE.get(key)
    if t is None:
        t = torch.tensor(list(indices), dtype=torch.int32, device=dev)
        _PERM_CACHE[key] = t
    return t


class KeyedJaggedTensor:
    """Jagged ids for F keys x B samples, key-major (torchrec KJT field semantics)."""

    def __init__(
        self,
        keys: Sequence[str],
        values: torch.Tensor,
        lengths: Optional[torch.Tensor] = None,
        weights: Optional[torch.Tensor] = None,
        offsets: Optional[torch.Tensor] = None,
        stride: Optional[int] = None,
        length_per_key: Optional[Sequence[int]] = None,
        uniform_length: Optional[int] = None,
    ) -> None:
        self._keys = list(keys)
        # The kernels read `values` as int64, `lengths` as int32 / int64, `offsets` as int64 and
        # `weights` as float32, all dense: torchrec accepts int32 ids and strided views, which would be
        # read here as garbage / past the end of the buffer -- normalise once, at construction.
        if values.dtype != torch.int64:
            if values.is_floating_point() or values.dtype == torch.bool:
                raise TypeError(f"KeyedJaggedTensor values must be integer ids, got {values.dtype}")
            values = values.to(torch.int64)
        if lengths is not None and lengths.dtype not in (torch.int32, torch.int64):
            if lengths.is_floating_point():
                raise TypeError(f"KeyedJaggedTensor lengths must be integers, got {lengths.dtype}")
            lengths = lengths.to(torch.int32)
        if offsets is not None and offsets.dtype != torch.int64:
            offsets = offsets.to(torch.int64)
        if weights is not None and weights.dtype != torch.float32:
            weights = weights.to(torch.float32)
        self._values = values.contiguous()
        self._weights = weights.contiguous() if weights is not None else None
        self._lengths = lengths.contiguous() if lengths is not None else None
        self._offsets = offsets.contiguous() if offsets is not None else None
        values, lengths, offsets, weights = self._values, self._lengths, self._offsets, self._weights
        if stride is None:
            if lengths is not None:
                stride = lengths.numel() // max(len(self._keys), 1)
            elif offsets is not None:
                stride = (offsets.numel() - 1) // max(len(self._keys), 1)
            else:
                raise ValueError("KeyedJaggedTensor needs lengths, offsets or stride")
        self._stride = int(stride)
        self._length_per_key = list(length_per_key) if length_per_key is not None else None
        # Host-side hint: every bag has exactly this many ids (1 for Criteo).  Known for free when
        # the batch is assembled on CPU (dataloader workers); lets the kernels skip the offsets.
        # ... and PER KEY (a batch whose one KeyedJaggedTensor holds sequence keys next to one-id-per-sample keys, as tzrec builds
        # it: data_parser.py:576-585): the keys known to hold exactly one id per bag.  `permute` to a subset of them gives a
        # uniform KeyedJaggedTensor again -- the pooled collection of such a model then takes the kernels' one-id-per-bag forms.
        self._uniform_keys: frozenset = frozenset()
        if uniform_length is None and lengths is not None and lengths.device.type == "cpu":
            n = lengths.numel()
            if n > 0 and values.numel() == n and bool((lengths == 1).all()):
                uniform_length = 1
            elif n > 0 and self._stride > 0 and n == len(self._keys) * self._stride:
                ones = (lengths.view(len(self._keys), self._stride) == 1).all(dim=1).tolist()
                self._uniform_keys = frozenset(k for k, u in zip(self._keys, ones) if u)
        self._uniform_length = uniform_length

    def _carry(self, new: "KeyedJaggedTensor", *others: "KeyedJaggedTensor") -> "KeyedJaggedTensor":
        """the per-key hint of `self` (and `others`) on a KeyedJaggedTensor derived from them (restricted to its keys)"""
        hint = set(self._uniform_keys)
        for o in others:
            hint |= o._uniform_keys
        if hint and new._uniform_length is None:
            new._uniform_keys = frozenset(k for k in new._keys if k in hint)
        return new

    # -- torchrec accessors ----------------------------------------------------------------
    def keys(self) -> List[str]:
        return self._keys

    def values(self) -> torch.Tensor:
        return self._values

    def weights_or_none(self) -> Optional[torch.Tensor]:
        return self._weights

    def weights(self) -> torch.Tensor:
        if self._weights is None:
            raise ValueError("KeyedJaggedTensor has no weights")
        return self._weights

    def stride(self) -> int:
        return self._stride

    def uniform_length(self) -> Optional[int]:
        return self._uniform_length

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:
            assert self._offsets is not None
            self._lengths = (self._offsets[1:] - self._offsets[:-1]).to(torch.int32)
        return self._lengths

    def offsets(self) -> torch.Tensor:
        """[0] + cumsum(lengths), int64[F*B+1]: K3 on the device (fbgemm
        asynchronous_complete_cumsum in the reference)."""
        if self._offsets is None:
            self._offsets = lengths_to_offsets(self.lengths())
        return self._offsets

    def offsets_or_none(self) -> Optional[torch.Tensor]:
        return self._offsets

    def length_per_key(self) -> List[int]:
        if self._length_per_key is None:
            if self._lengths is not None and self._lengths.device.type == "cpu" and self._stride > 0:
                # host bookkeeping on a batch still in host memory (dataloader side)
                self._length_per_key = [int(v) for v in self._lengths.view(len(self._keys), self._stride).sum(dim=1).tolist()]
                return self._length_per_key
            off = self.offsets()[:: self._stride].cpu().tolist() if self._stride > 0 else [0] * (len(self._keys) + 1)
            self._length_per_key = [int(off[i + 1] - off[i]) for i in range(len(self._keys))]
        return self._length_per_key

    @property
    def device(self) -> torch.device:
        return self._values.device

    @staticmethod
    def from_lengths_sync(keys, values, lengths, weights=None) -> "KeyedJaggedTensor":
        return KeyedJaggedTensor(keys=keys, values=values, lengths=lengths, weights=weights)

    @staticmethod
    def from_offsets_sync(keys, values, offsets, weights=None) -> "KeyedJaggedTensor":
        return KeyedJaggedTensor(keys=keys, values=values, offsets=offsets.to(torch.int64), weights=weights)

    @staticmethod
    def empty(device=None, values_dtype=torch.int64, lengths_dtype=torch.int32) -> "KeyedJaggedTensor":
        return KeyedJaggedTensor([], torch.empty(0, dtype=values_dtype, device=device),
                                 torch.empty(0, dtype=lengths_dtype, device=device), stride=0)

    def lengths_or_none(self) -> Optional[torch.Tensor]:
        return self._lengths

    def offset_per_key(self) -> List[int]:
        out = [0]
        for n in self.length_per_key():
            out.append(out[-1] + n)
        return out

    # -- host-side views (torchrec KJT.to_dict / __getitem__ / split / concat): slices, no kernels --
    def _key_slice(self, lo_key: int, hi_key: int) -> "KeyedJaggedTensor":
        B, opk = self._stride, self.offset_per_key()
        lo, hi = opk[lo_key], opk[hi_key]
        off = None if self._offsets is None else self._offsets[lo_key * B:hi_key * B + 1] - lo
        return self._carry(KeyedJaggedTensor(
            self._keys[lo_key:hi_key], self._values[lo:hi], self.lengths()[lo_key * B:hi_key * B],
            None if self._weights is None else self._weights[lo:hi], off, B, self.length_per_key()[lo_key:hi_key],
            self._uniform_length))

    def __getitem__(self, key: str) -> JaggedTensor:
        i = self._keys.index(key)
        sub = self._key_slice(i, i + 1)
        return JaggedTensor(sub._values, sub.lengths(), sub.offsets(), sub._weights)

    def to_dict(self) -> Dict[str, JaggedTensor]:
        return {k: self[k] for k in self._keys}

    def split(self, segments: Sequence[int]) -> List["KeyedJaggedTensor"]:
        """Consecutive groups of `segments[i]` keys each."""
        if sum(segments) != len(self._keys):
            raise ValueError(f"split segments {list(segments)} do not cover {len(self._keys)} keys")
        out, k = [], 0
        for n in segments:
            out.append(self._key_slice(k, k + n))
            k += n
        return out

    @staticmethod
    def concat(kjt_list: Sequence["KeyedJaggedTensor"]) -> "KeyedJaggedTensor":
        """Keys of all inputs, in order; all inputs share the stride (torchrec KJT.concat)."""
        if not kjt_list:
            raise ValueError("concat of no KeyedJaggedTensors")
        strides = {k.stride() for k in kjt_list if len(k.keys())}
        if len(strides) > 1:
            raise ValueError(f"concat needs one stride, got {sorted(strides)}")
        any_w = any(k.weights_or_none() is not None for k in kjt_list)
        if any_w and not all(k.weights_or_none() is not None for k in kjt_list):
            raise ValueError("concat: either every KeyedJaggedTensor carries weights or none does")
        keys = [x for k in kjt_list for x in k.keys()]
        uni = {k.uniform_length() for k in kjt_list}
        out = KeyedJaggedTensor(
            keys, torch.cat([k.values() for k in kjt_list]), torch.cat([k.lengths() for k in kjt_list]),
            torch.cat([k.weights() for k in kjt_list]) if any_w else None, None,
            strides.pop() if strides else 0, None, uni.pop() if len(uni) == 1 else None)
        if out._uniform_length is None:  # (inputs that are uniform as a whole count key by key)
            hint = set()
            for k in kjt_list:
                hint |= set(k._keys) if k._uniform_length == 1 else set(k._uniform_keys)
            out._uniform_keys = frozenset(hint)
        return out

    # -- Pipelineable contract (tzrec Batch.to / record_stream, datasets/utils.py:344-408) ----
    def to(self, device, non_blocking: bool = False) -> "KeyedJaggedTensor":
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        # ids per key: free while the batch is still in host memory, a device read-back (and the end of any hipGraph capture)
        # once it is not -- torchrec's KJT carries the same cache across `.to()`
        if self._length_per_key is None and self._values.device.type == "cpu" and torch.device(device).type != "cpu" \
                and self._lengths is not None:
            self.length_per_key()
        return self._carry(KeyedJaggedTensor(
            self._keys, mv(self._values), mv(self._lengths), mv(self._weights), mv(self._offsets),
            self._stride, self._length_per_key, self._uniform_length,
        ))

    def pin_memory(self) -> "KeyedJaggedTensor":
        pin = lambda t: None if t is None else t.pin_memory()  # noqa: E731
        return self._carry(KeyedJaggedTensor(
            self._keys, pin(self._values), pin(self._lengths), pin(self._weights), pin(self._offsets),
            self._stride, self._length_per_key, self._uniform_length,
        ))

    def record_stream(self, stream) -> None:
        for t in (self._values, self._lengths, self._weights, self._offsets):
            if t is not None and t.is_cuda:
                t.record_stream(stream)

    def uniform_prefix(self, keys: Sequence[str]) -> Optional["KeyedJaggedTensor"]:
        """A uniform (one id per bag) VIEW that holds `keys`, or None: when `keys` and every key in front of the last of them are
        known to hold one id per bag (the per-key hint), the first `last + 1` keys ARE a uniform KeyedJaggedTensor -- key k's ids
        are values[k B : (k + 1) B] -- sharing this one's storage: no kernel, no copy, no read-back."""
        if self._uniform_length is not None or not self._uniform_keys or not keys:
            return None
        idx = {k: i for i, k in enumerate(self._keys)}
        if any(k not in idx for k in keys):
            return None
        last = max(idx[k] for k in keys)
        if any(self._keys[i] not in self._uniform_keys for i in range(last + 1)):
            return None
        B, n = self._stride, last + 1
        if self._values.numel() < n * B:
            return None
        lengths = None if self._lengths is None else self._lengths[:n * B]
        offsets = None if self._offsets is None else self._offsets[:n * B + 1]
        weights = None if self._weights is None else self._weights[:n * B]
        return KeyedJaggedTensor(self._keys[:n], self._values[:n * B], lengths, weights, offsets, B, [B] * n, 1)

    def narrow_ids(self) -> "WireKeyedJaggedTensor":
        """The same ids as int32 for the trip across PCIe (`Batch.narrow_ids`): 4 instead of 8 bytes per id."""
        return WireKeyedJaggedTensor(self)

    # -- K1 --------------------------------------------------------------------------------
    def permute(self, indices: Sequence[int]) -> "KeyedJaggedTensor":
        """Output key t = input key indices[t] (torchrec KJT.permute -> fbgemm
        permute_2D_sparse_data)."""
        B, F, T = self._stride, len(self._keys), len(indices)
        dev = self.device
        lengths = self.lengths()
        in_off = self.offsets()
        lpk = self._length_per_key
        uni_out = self._uniform_length
        if uni_out is None and T > 0 and self._uniform_keys and all(self._keys[i] in self._uniform_keys for i in indices):
            uni_out = 1  # every selected key holds one id per bag (the per-key hint)
        if uni_out is not None and self._uniform_length is None:
            n_out = T * B * uni_out
        elif lpk is not None:
            n_out = int(sum(lpk[i] for i in indices))
        elif self._uniform_length is not None:
            n_out = T * B * self._uniform_length
        elif sorted(indices) == list(range(F)):
            n_out = self._values.numel()
        else:
            n_out = int(sum(self.length_per_key()[i] for i in indices))  # host sync, like torchrec
        perm = _perm_tensor(tuple(int(i) for i in indices), dev)
        out_lengths = torch.empty(T * B, dtype=lengths.dtype, device=dev)
        out_offsets = torch.empty(T * B + 1, dtype=torch.int64, device=dev)
        out_values = torch.empty(n_out, dtype=torch.int64, device=dev)
        out_weights = None if self._weights is None else torch.empty(n_out, dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws_bytes = L.tzr_kjt_permute_workspace(T, B)
        ws = _lib.workspace(ws_bytes, dev)
        rc = L.tzr_kjt_permute(
            _lib.ptr(perm), T, F, B, _lib.ptr(lengths), lengths.element_size(), _lib.ptr(in_off),
            _lib.ptr(self._values), _lib.ptr(self._weights), _lib.ptr(out_lengths),
            _lib.ptr(out_offsets), _lib.ptr(out_values), _lib.ptr(out_weights), n_out,
            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev),
        )
        _lib.check(rc, "tzr_kjt_permute")
        return self._carry(KeyedJaggedTensor(
            [self._keys[i] for i in indices], out_values, out_lengths, out_weights, out_offsets, B,
            None if lpk is None else [lpk[i] for i in indices], uni_out,
        ))


def lengths_to_offsets(lengths: torch.Tensor) -> torch.Tensor:
    """K3: exclusive scan of int32/int64 lengths into int64 offsets[n+1]."""
    assert lengths.dtype in (torch.int32, torch.int64) and lengths.is_contiguous()
    n = lengths.numel()
    dev = lengths.device
    out = torch.empty(n + 1, dtype=torch.int64, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_lengths_to_offsets_workspace(n), dev)
    rc = L.tzr_lengths_to_offsets(
        _lib.ptr(lengths), lengths.element_size(), n, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
        _lib.stream_ptr(dev),
    )
    _lib.check(rc, "tzr_lengths_to_offsets")
    return out


def block_bucketize(
    kjt: KeyedJaggedTensor, block_sizes: torch.Tensor, world_size: int, return_permute: bool = False,
    rank_offsets: Optional[torch.Tensor] = None,
):
    """K2: split every bag by owning rank (row-wise sharding).  Returns a KJT with W*F keys
    (rank-major) and, optionally, unbucketize_permute (fbgemm block_bucketize_sparse_features)."""
    B, F, W = kjt.stride(), len(kjt.keys()), int(world_size)
    dev = kjt.device
    values, weights = kjt.values(), kjt.weights_or_none()
    offsets = kjt.offsets()
    lengths = kjt.lengths()
    n = values.numel()
    new_lengths = torch.empty(W * F * B, dtype=lengths.dtype, device=dev)
    new_offsets = torch.empty(W * F * B + 1, dtype=torch.int64, device=dev)
    new_values = torch.empty(n, dtype=torch.int64, device=dev)
    new_weights = None if weights is None else torch.empty(n, dtype=torch.float32, device=dev)
    unbucketize = torch.empty(n, dtype=torch.int64, device=dev) if return_permute else None
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_block_bucketize_workspace(F, B, W), dev)
    rc = L.tzr_block_bucketize(
        _lib.ptr(block_sizes), _lib.ptr(rank_offsets), F, B, W, _lib.ptr(offsets), _lib.ptr(values), _lib.ptr(weights), n,
        _lib.ptr(new_lengths), lengths.element_size(), _lib.ptr(new_offsets), _lib.ptr(new_values),
        _lib.ptr(new_weights), _lib.ptr(unbucketize), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev),
    )
    _lib.check(rc, "tzr_block_bucketize")
    keys = [f"{k}@{r}" for r in range(W) for k in kjt.keys()]
    out = KeyedJaggedTensor(keys, new_values, new_lengths, new_weights, new_offsets, B)
    return (out, unbucketize) if return_permute else out


class WireKeyedJaggedTensor:
    """A KeyedJaggedTensor on its way from the dataloader to the device with its ids narrowed to int32.

    The kernels read ids as int64 (the KJT contract, /root/reference/tzrec/datasets/data_parser.py:576-585), but 13.6 of
    the 17.5 MB a DLRM-Criteo batch of 65 536 samples moves host -> device are ids of tables that all hold fewer than 2^31
    rows: on the slower hosts of the pool (25 GB/s) that copy, not the 0.58 ms step, bounded the end-to-end rate
    (NOTES.md).  This is the WIRE form only -- host side of `Batch.to / pin_memory`, never handed to a lookup: `.to(device)`
    moves the int32 ids and widens them on the device (one elementwise kernel behind the copy, on the copy's stream) into
    a regular KeyedJaggedTensor.  Raw 64-bit ids (zero-collision-hash features) do not fit: `narrow_ids` refuses them."""

    def __init__(self, kjt: "KeyedJaggedTensor", _values32: Optional[torch.Tensor] = None) -> None:
        v = kjt.values()
        if _values32 is None:
            if v.numel() and (int(v.max()) >= (1 << 31) or int(v.min()) < 0):
                raise ValueError("ids outside [0, 2^31) cannot travel as int32 (raw ids of a zero-collision-hash feature?)")
            _values32 = v.to(torch.int32)
        self._kjt, self._values32 = kjt, _values32

    def keys(self):
        return self._kjt.keys()

    def stride(self) -> int:
        return self._kjt.stride()

    def uniform_length(self):
        return self._kjt.uniform_length()

    def wire_values(self) -> torch.Tensor:
        return self._values32

    def lengths_or_none(self):
        return self._kjt.lengths_or_none()

    def offsets_or_none(self):
        return self._kjt.offsets_or_none()

    def weights_or_none(self):
        return self._kjt.weights_or_none()

    def pin_memory(self) -> "WireKeyedJaggedTensor":
        k = self._kjt
        pin = lambda t: None if t is None else t.pin_memory()  # noqa: E731
        meta = k._carry(KeyedJaggedTensor(k._keys, torch.zeros(0, dtype=torch.int64), pin(k._lengths), pin(k._weights), pin(k._offsets), k._stride,
                                          k._length_per_key, k._uniform_length))
        meta._values = k._values  # (kept for `widen()` on the host; never copied)
        return WireKeyedJaggedTensor(meta, self._values32.pin_memory())

    def to(self, device, non_blocking: bool = False) -> "KeyedJaggedTensor":
        k = self._kjt
        if torch.device(device).type == "cpu":
            return self.widen()
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        vals = self._values32.to(device, non_blocking=non_blocking).to(torch.int64)  # widened on the device, behind the copy
        return k._carry(KeyedJaggedTensor(k._keys, vals, mv(k._lengths), mv(k._weights), mv(k._offsets), k._stride, k._length_per_key,
                                          k._uniform_length))

    def widen(self) -> "KeyedJaggedTensor":
        k = self._kjt
        return k._carry(KeyedJaggedTensor(k._keys, self._values32.to(torch.int64), k._lengths, k._weights, k._offsets, k._stride,
                                          k._length_per_key, k._uniform_length))

    def record_stream(self, stream) -> None:
        pass  # host tensors only
