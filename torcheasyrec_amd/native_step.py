"""Python face of the native step driver (csrc/step_driver.hip): a communicator of the library's own (RCCL reached directly)
and step programs -- a sharded train step's captured hipGraphs and the collectives between them, queued by ONE C call.

Replaces the host side of a steady-state step of tzrec's train pipeline (/root/reference/tzrec/utils/dist_util.py:221-303),
which issues every collective through torch.distributed from Python: at 8 192 samples per rank that host work (0.35 ms)
is longer than the step's kernels (0.24 ms).  torch.distributed is used ONCE here: to carry the communicator's unique id
from rank 0 to the others."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from . import _lib


def librccl_path() -> bytes:
    """the RCCL library this process already holds (torch's own copy): the driver looks its symbols up there.
    `TZR_RCCL_PATH` names another one (a site's own RCCL build; the CPU suite's shared-memory stand-in, tests/emu/rccl_stub.cpp)."""
    env = os.environ.get("TZR_RCCL_PATH", "")
    if env:
        return env.encode()
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else b""


def available() -> bool:
    """can this process make a communicator of the library's own?  On the GPU: whenever RCCL is reachable.  Under the lane
    emulator (tests): only when a library was named explicitly."""
    if _lib.backend() != "hip-gfx950" and not os.environ.get("TZR_RCCL_PATH"):
        return False
    return bool(_lib.lib().tzr_comm_available(librccl_path()))


class NativeComm:
    """One RCCL communicator over the ranks of `process_group` (default: the world), created from a unique id that rank 0
    makes and `torch.distributed.broadcast_object_list` carries -- the only torch.distributed call it ever makes.  The
    calling thread's current device is the communicator's device."""

    def __init__(self, process_group=None, device: Optional[torch.device] = None) -> None:
        import torch.distributed as dist

        L = _lib.lib()
        path = librccl_path()
        if not L.tzr_comm_available(path):
            raise _lib.TzrError("RCCL is not reachable from libtzrec_hip.so (tzr_comm_available)")
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        else:
            self.world, self.rank = 1, 0
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.set_device(device)
        box: List[Optional[bytes]] = [None]
        if self.rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(L.tzr_comm_unique_id(path, C.cast(buf, C.c_void_p), 128), "tzr_comm_unique_id")
            box[0] = bytes(buf)
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
        uid = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        _lib.check(L.tzr_comm_create(path, C.cast(uid, C.c_void_p), 128, self.world, self.rank, C.byref(h)), "tzr_comm_create")
        self._h = h
        self.version = int(L.tzr_comm_version(path))
        # RCCL sets its peer connections up lazily, on the first call of a kind -- a host-side handshake between the ranks.  One
        # eager call of each collective this class offers, here, so that none of that happens while a stream is being captured
        # (the step's first native call IS a capture: sharded_step._capture_native).
        dev = torch.device(device) if device is not None else None
        if dev is not None and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            a = torch.zeros(self.world * 64, dtype=torch.int64, device=dev)
            b = torch.empty_like(a)
            r = torch.zeros(1024, dtype=torch.float32, device=dev)
            self.all_to_all(a, b)
            self.all_reduce(r, average=False)
            self.all_reduce(r, average=True)
            torch.cuda.current_stream(dev).synchronize()

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def all_to_all(self, send: torch.Tensor, recv: torch.Tensor, stream: Optional[int] = None) -> None:
        """equal splits along dim 0, in stream order on `stream` (default: the current stream)"""
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel() and send.numel() % self.world == 0
        _lib.check(_lib.lib().tzr_comm_all_to_all(self._h, _lib.ptr(send), _lib.ptr(recv), send.numel() * send.element_size() // self.world,
                                                  _lib.stream_ptr(send.device) if stream is None else stream), "tzr_comm_all_to_all")

    def all_reduce(self, buf: torch.Tensor, average: bool = False, stream: Optional[int] = None) -> None:
        assert buf.is_contiguous() and buf.dtype == torch.float32
        _lib.check(_lib.lib().tzr_comm_all_reduce(self._h, _lib.ptr(buf), buf.numel(), 1 if average else 0,
                                                  _lib.stream_ptr(buf.device) if stream is None else stream), "tzr_comm_all_reduce")

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().tzr_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # (best effort: a communicator left to the process exit is reclaimed by the driver)
        try:
            self.close()
        except Exception:
            pass


class StepProgram:
    """ops recorded once, replayed by `run`: graphs on the caller's stream, collectives on the program's communication stream
    behind everything queued so far; `sync=False` collectives are waited for by a later `add_wait`.  Keeps every tensor and
    graph it was given alive."""

    def __init__(self) -> None:
        h = C.c_void_p()
        _lib.check(_lib.lib().tzr_step_create(C.byref(h)), "tzr_step_create")
        self._h = h
        self._keep: list = []
        self._graphs: list = []

    def _op(self, rc: int, what: str) -> int:
        if rc < 0:
            _lib.check(rc, what)
        return rc

    def add_graph(self, graph: "torch.cuda.CUDAGraph") -> int:
        self._keep.append(graph)
        if hasattr(graph, "check"):
            self._graphs.append(graph)
        return self._op(_lib.lib().tzr_step_add_graph(self._h, C.c_void_p(graph.raw_cuda_graph_exec())), "tzr_step_add_graph")

    def add_all_to_all(self, comm: NativeComm, send: torch.Tensor, recv: torch.Tensor, sync: bool = False) -> int:
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel() and send.numel() % comm.world == 0
        self._keep += [comm, send, recv]
        return self._op(_lib.lib().tzr_step_add_all_to_all(self._h, comm.handle, _lib.ptr(send), _lib.ptr(recv),
                                                           send.numel() * send.element_size() // comm.world, 1 if sync else 0),
                        "tzr_step_add_all_to_all")

    def add_all_reduce(self, comm: NativeComm, buf: torch.Tensor, average: bool = False, sync: bool = False) -> int:
        assert buf.is_contiguous() and buf.dtype == torch.float32
        self._keep += [comm, buf]
        return self._op(_lib.lib().tzr_step_add_all_reduce(self._h, comm.handle, _lib.ptr(buf), buf.numel(), 1 if average else 0,
                                                           1 if sync else 0), "tzr_step_add_all_reduce")

    def add_wait(self, op: int) -> int:
        return self._op(_lib.lib().tzr_step_add_wait(self._h, op), "tzr_step_add_wait")

    def __len__(self) -> int:
        return int(_lib.lib().tzr_step_num_ops(self._h))

    def run(self, stream: Optional[int]) -> None:
        _lib.check(_lib.lib().tzr_step_run(self._h, stream), "tzr_step_run")
        for g in self._graphs:  # (stand-ins for graphs that run host code -- the CPU suite's -- hand their exceptions over here)
            g.check()

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().tzr_step_destroy(self._h)
            self._h = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
