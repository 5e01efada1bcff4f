"""Training step of the sharded DLRM, organised the way MI355X wants it launched.

A sharded step cannot be ONE hipGraph: the all-to-all split sizes depend on the ids and RCCL wants
them on the host.  Issued op by op it is ~150 launches and host-bound (profiles/r01d).  So the step
is cut along the data-dependent boundary:

  input dist   (ids only)   bucketize -> counts all-to-all -> host sync -> ids all-to-all, then the
                            backward index plans (K6) of the owners and of the replicas: they need
                            ids only, and next to the GEMM-heavy dense segment of the previous batch
                            they cost nothing (-105 us per step on the 1-rank proxy).
                            Runs one batch AHEAD on a side HIP stream, so the host sync waits for
                            work that was queued behind nothing, while the main stream is busy with
                            the previous batch (reference: TrainPipelineSparseDist,
                            /root/reference/tzrec/utils/dist_util.py:221-303).
  lookup       (C-ABI)      owner row gather -> rows all-to-all -> pooled gather, straight into
                            the static input buffer of ...
  dense segment (hipGraph)  bottom MLP, dot interaction, top MLP, BCE loss and the whole autograd
                            backward: static shapes, captured once per batch size, ONE launch.  Its
                            outputs are d(loss)/d(pooled embeddings) and the dense gradients.
  sparse bwd   (C-ABI)      per-id gradient rows -> all-to-all -> sort + fused optimizer on the
                            owners; replicated tables: accumulate -> all-reduce -> dense update.
  dense sync                one flat all-reduce (AVG) of the MLP gradients, fused Adam.

Weights are only read by `lookup`, which runs after the previous step's sparse update on the same
stream, so prefetching the input dist changes no value (tests/test_sharded_gloo.py runs both ways).

`step_graph=True` (with `ShardedEmbeddingBagCollection(exchange="capacity")`): the capacity-bounded exchange
has fixed split sizes, so everything after the input dist -- lookup with its rows all-to-all, dense segment,
gradient all-to-all, sort + fused optimizer, dense all-reduce, Adam -- is captured as ONE hipGraph per
pipeline slot (two slots: batch i+1 is laid out by its input dist while the graph of batch i runs) -- in
practice three graphs with the RCCL calls issued eagerly between them, five when the collectives are issued async
so that they fly under the neighbouring graph (`_step_whole`, the default).  The
overflow word of a batch is read on the host BEFORE its graph is launched (the input dist ran a batch
earlier), so a batch that does not fit simply takes the eager exact path; nothing is ever undone.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Optional

import torch

from . import _lib
from .dlrm import bce_with_logits
from .sharding import ShardedDLRM
from .sparse import KeyedJaggedTensor


# Captures here run in a process that owns a process group: its watchdog thread polls the completion events of the
# eager collectives (hipEventQuery) whenever it wakes up, also while this thread is capturing.  In the default "global"
# capture mode such a call from ANOTHER thread is an error that takes the process down (one unexplained crash of
# tests/test_sharded_gpu.py in round 2 fits: rare, timing dependent); "thread_local" only polices the capturing thread.
_CAPTURE_MODE = "thread_local"


_FR_STATE = {"on": None, "stuck": set()}  # flight recorder recording? (None: not probed); entries given up on
QUIESCE_FALLBACK_S = 0.35   # without the recorder: a few periods of the watchdog loop (kWatchdogThreadSleepMillis = 100)
QUIESCE_TIMEOUT_S = 3.0


def _unretired_collectives() -> Optional[set]:
    """Ids of the collectives the process group's WATCHDOG still lists, read from the flight recorder it feeds: an entry's
    `retired` flag is set by the watchdog loop (`FlightRecorder::retire_id`) at the moment it finds the work finished,
    right before it drops the work from its list -- not by anything this thread does (`onlyActive` would not do: the dump
    itself queries the events and marks finished entries completed).  None when the recorder is unavailable or off
    (TORCH_FR_BUFFER_SIZE = 0)."""
    import pickle

    try:
        from torch._C._distributed_c10d import _dump_nccl_trace
    except ImportError:
        return None
    try:
        ents = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries", ())
    except Exception:
        return None
    if _FR_STATE["on"] is None:  # collectives were issued before any capture (parameter broadcasts): the dump has them
        _FR_STATE["on"] = bool(ents) and all("retired" in e and "record_id" in e for e in ents[:1])
    if not _FR_STATE["on"]:
        return None
    return {e["record_id"] for e in ents if not e["retired"]} - _FR_STATE["stuck"]


def _quiesce_process_group(device) -> float:
    """Call right before opening a hipGraph capture next to an RCCL process group.

    The failure this guards against (round 3: 1 run in 4 - 6 of the step-graph tests; round 4: reproduced at will by
    scripts/capture_stress.py): torch runs a collective issued with async_op=False ON the current stream and records its
    completion event there; the group's watchdog thread polls the events of every collective it still lists (every
    ~100 ms, hipEventQuery); on this ROCm stack a query of an event whose stream is CAPTURING fails with
    hipErrorCapturedEvent -- the watchdog throws, the process aborts (and the capture is invalidated on the way).

    Two mechanisms, neither of them a timer:
      1. every collective of this package is issued async (`sharding.stream_collective`): it runs on RCCL's own stream,
         which nothing ever captures, so the watchdog may poll its event at any time;
      2. for collectives issued by anybody else on the stream about to capture: device synchronize -- they HAVE
         completed -- then a handshake with the watchdog itself: wait until the flight recorder shows every entry
         retired, i.e. the watchdog has seen each of them complete and dropped it (`_unretired_collectives`).  Nothing is
         issued between here and the capture, so its list stays empty.
    Only when the recorder is off does (2) fall back to waiting a few watchdog periods.  Returns the seconds waited."""
    import time

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() == "gloo":
        return 0.0
    torch.cuda.synchronize(device)
    t0 = time.monotonic()
    left = _unretired_collectives()
    if left is None:
        time.sleep(QUIESCE_FALLBACK_S)
        return time.monotonic() - t0
    while left:
        if time.monotonic() - t0 > QUIESCE_TIMEOUT_S:  # entries that never retire (works nobody enqueued): do not hang, and
            _FR_STATE["stuck"] |= left                # do not wait for them again
            break
        time.sleep(0.005)
        left = _unretired_collectives()
    return time.monotonic() - t0


def _consumes_rng(model) -> bool:
    """does a training step of `model` draw random numbers?  (the stochastic layers this package builds: nn.Dropout of dlrm.MLP)"""
    mods = model.modules() if hasattr(model, "modules") else []
    return any(isinstance(m, (torch.nn.Dropout, torch.nn.Dropout1d, torch.nn.Dropout2d, torch.nn.AlphaDropout)) and m.p > 0 for m in mods)


class _Segment:
    """Static buffers + captured graph of the dense segment for one batch size."""

    def __init__(self) -> None:
        self.graph = None
        self.dense = self.label = self.sparse = self.loss = self.logits = None
        self.grads = None


class ShardedTrainStep:
    def __init__(self, model: ShardedDLRM, dense_optimizer: torch.optim.Optimizer,
                 loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = bce_with_logits,
                 use_graph: Optional[bool] = None, prefetch: bool = True, warmup_iters: int = 2,
                 plan_ahead: bool = True, step_graph: bool = False, graph_input_dist: bool = False,
                 overlap_collectives: Optional[bool] = None, native_driver: Optional[bool] = None,
                 input_dist_stream: Optional[str] = None, graph_factory: Optional[Callable] = None) -> None:
        """`input_dist_stream`: "side" -- the next batch's input dist replays on a second stream next to the current step -- or
        "main" -- on the step's own stream, in front of the step.  None: "main" when the native driver runs on more than one
        rank (see `_input_dist_on_main`), else "side".
        `graph_factory(body) -> graph` (an object with `replay()` and `raw_cuda_graph_exec()`): what stands in for a hipGraph
        capture of `body`'s launches on a device without graphs -- the CPU suite hands in tests/emu's recorded host functions
        so that the native driver's program order runs at world size 2 / 4 on gloo; never set on a GPU."""
        self.model, self.opt, self.loss_fn = model, dense_optimizer, loss_fn
        self.device = model.ebc._device
        self.cuda = self.device.type == "cuda"
        self._graph_factory = graph_factory
        can_graph = self.cuda or graph_factory is not None
        self.use_graph = can_graph if use_graph is None else (use_graph and can_graph)
        if input_dist_stream not in (None, "side", "main"):
            raise ValueError("input_dist_stream: 'side', 'main' or None")
        self.input_dist_stream = input_dist_stream
        self.prefetch = prefetch
        self.plan_ahead = plan_ahead
        self.warmup_iters = warmup_iters
        self.params = list(model.dense_parameters())
        # the root gradient of the backward pass exists (and holds its 1.0) before any capture: created inside one it would
        # be uninitialised until that graph's first replay
        from .dense import unit_gradient
        unit_gradient(torch.zeros((), dtype=torch.float32, device=self.device))
        self._seg: Dict[int, _Segment] = {}
        self._seen: Dict[int, int] = {}
        self._side = torch.cuda.Stream(self.device) if self.cuda else None
        # (round 2 set tzr_tune("bwd_one_wg_heavy") here for its side-stream plans; round 3 could not reproduce that
        # failure -- round 2's own binary passes 200 / 200 on the round-3 boxes, the current one 3 000 iterations in
        # every stream arrangement, NOTES.md "Side-stream plan" -- and the knob is gone from the default path)
        self._lr_targets = None  # objects whose learning rate is mirrored to the device before a replay (collected once)
        self._ahead: Optional[tuple] = None  # (kjt, state) of the batch whose input dist already ran
        # whole-step graphs: static inputs + one captured graph per pipeline slot
        self.step_graph = bool(step_graph)
        # also replay the input dist's two kernel runs from hipGraphs (5 host calls instead of 17): bit-identical, but
        # not faster on the 1-rank proxy (profiles/r02r: 0.58 vs ~0.55 ms at 8192 on the same slow box) -> opt-in
        self.graph_input_dist = bool(graph_input_dist)
        if self.step_graph and model.ebc.exchange != "capacity":
            raise ValueError("step_graph needs the capacity-bounded exchange (fixed all-to-all split sizes)")
        self._slots: Dict[tuple, dict] = {}
        self._next_slot = 0
        self.graph_steps = self.eager_steps = 0
        # whole-step path: five graphs instead of three, so that the gradient all-to-all flies under the replicated
        # tables' row sums and their all-reduce (+ the dense gradients') under the owners' sort + fused optimizer --
        # the order of issue the eager exact path has always had (`_backward_impl`).  On one rank there is nothing to
        # hide (RCCL's self copies take 12 us), and the two extra graph launches cost nothing measurable either
        # (0.408 vs 0.412 ms at 8 192, profiles/r03bx) -- so it is the default everywhere and the one-rank GPU tests
        # run the order the ranks of a real job run.
        if overlap_collectives is None:
            overlap_collectives = True
        self.overlap_collectives = bool(overlap_collectives)
        # Whole-step path, steady state: the slot's six graphs and the collectives between them queued by ONE native call
        # (native_step.StepProgram over csrc/step_driver.hip) on a communicator of the library's own -- no torch ProcessGroup
        # call, no Python between the graphs.  None = whenever the library can reach RCCL (a GPU build next to torch's
        # librccl); the warm-up and capture steps run the Python sequence, which is also what the gloo tests run.
        self.native_driver = native_driver
        self.native_error: Optional[str] = None  # why the native driver was given up at run time (`_native_failed`)
        self._comm = None
        self.native_steps = 0
        # (the ids all-to-all of batch i+1 is issued from the side stream on the collection's own process group, like the
        # exact exchange's: every RCCL call of the step is eager, torch orders them on the group's stream in issue order --
        # `ebc.input_dist_group` can name another communicator for it)

    # -- graphs and streams ----------------------------------------------------------------------
    def _can_graph(self) -> bool:
        return self.cuda or self._graph_factory is not None

    def _capture(self, body: Callable[[], None], stream=None, pool=None):
        """`body`'s launches as one hipGraph (not run).  A capture that fails leaves no open capture behind (torch ends it)."""
        if self._graph_factory is not None:
            return self._graph_factory(body)
        g = torch.cuda.CUDAGraph()
        _quiesce_process_group(self.device)
        kw = {} if pool is None else {"pool": pool}
        with torch.cuda.graph(g, stream=stream if stream is not None else torch.cuda.current_stream(self.device),
                              capture_error_mode=_CAPTURE_MODE, **kw):
            body()
        return g

    def _raw_stream(self, stream=None):
        if not self.cuda:
            return None
        return (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream

    def _input_dist_on_main(self) -> bool:
        """The native driver's two communicators -- the step's, and the input dist's, which works a batch ahead -- must not have
        collectives in flight at the same time in an order that differs between ranks (the documented hazard of concurrent
        NCCL communicators: each rank's kernels of one communicator wait for their peers while the other communicator's
        cannot start).  On two streams nothing orders them.  The ordering edge: with more than one rank the input dist of batch
        i + 1 is queued on the STEP's stream, in front of step i -- every rank then runs `in(i+1), step(i), in(i+2), ...` in one
        stream order, and the two communicators are never concurrent.  What it gives up is the overlap of the ~30 us input dist
        with the step; what it gains besides safety: the batch's overflow word reaches the host a whole step earlier."""
        if self.input_dist_stream is not None:
            return self.input_dist_stream == "main"
        env = os.environ.get("TZR_INPUT_DIST_STREAM", "")
        if env in ("main", "side"):
            return env == "main"
        import torch.distributed as dist

        world = dist.get_world_size(self.model.pg) if dist.is_available() and dist.is_initialized() else 1
        return world > 1 and self._use_native_driver()

    def _in_stream(self):
        """the stream the input dist is queued on (None off the GPU)"""
        if not self.cuda:
            return None
        return torch.cuda.current_stream(self.device) if self._input_dist_on_main() else self._side

    # -- dense segment ---------------------------------------------------------------------------
    def _split_bottom(self) -> bool:
        """can the bottom MLP run ahead of the rest of the dense segment (under the rows all-to-all)?"""
        return self.loss_fn is bce_with_logits and hasattr(self.model, "dense_loss") and hasattr(self.model, "dense_bottom")

    def _dense_fwd_bwd(self, dense, sparse, label, d=None):
        if self.loss_fn is bce_with_logits and hasattr(self.model, "dense_loss"):
            # the top MLP's tail + loss + their backward: one launch (`d`: the bottom MLP's output, already computed)
            loss, logits = self.model.dense_loss(dense, sparse, label) if d is None else self.model.dense_loss(dense, sparse, label, d=d)
        else:
            logits = self.model.dense_forward(dense, sparse)
            loss = self.loss_fn(logits, label)
        from .dense import root_loss, unit_gradient

        with root_loss():  # the loss itself is differentiated: its incoming gradient is 1.0, nothing to scale
            grads = torch.autograd.grad(loss, [sparse] + self.params, grad_outputs=unit_gradient(loss))
        return loss.detach(), logits.detach(), grads

    def _drop_grads(self) -> None:
        """With FusedDenseAdam(fuse_finish=True) the parameters do not keep their `.grad` (views of the all-reduced buffer) past the
        optimizer's step: a backward leaves its gradients as partial sums only for parameters that hold none (dense._defer_finish --
        autograd's accumulation would read the unwritten tensor), and this step takes its gradients by `autograd.grad` anyway."""
        from . import dense

        if dense.FUSE_FINISH:
            for p in self.params:
                p.grad = None

    def _segment(self, dense, label, width) -> _Segment:
        B = dense.shape[0]
        seg = self._seg.get(B)
        if seg is None:
            seg = _Segment()
            seg.sparse = torch.zeros(B, width, dtype=torch.float32, device=self.device, requires_grad=True)
            if self.use_graph:
                seg.dense, seg.label = torch.empty_like(dense), torch.empty_like(label)
            self._seg[B] = seg
        return seg

    def _run_dense(self, seg: _Segment, dense, label) -> None:
        B = dense.shape[0]
        if not self.use_graph:
            seg.loss, seg.logits, seg.grads = self._dense_fwd_bwd(dense, seg.sparse, label)
            return
        seg.dense.copy_(dense)
        seg.label.copy_(label)
        if seg.graph is None:
            n = self._seen.get(B, 0)
            self._seen[B] = n + 1
            if n < self.warmup_iters:  # eager: TunableOp / lazy inits must not happen under capture
                seg.loss, seg.logits, seg.grads = self._dense_fwd_bwd(seg.dense, seg.sparse, seg.label)
                return
            if self.cuda and torch.cuda.current_stream(self.device) == torch.cuda.default_stream(self.device):
                raise RuntimeError("ShardedTrainStep captures on the current stream: run the training loop under a "
                                   "non-default stream (torch.cuda.set_stream)")
            seg.loss = seg.logits = seg.grads = None

            def body():
                seg.loss, seg.logits, seg.grads = self._dense_fwd_bwd(seg.dense, seg.sparse, seg.label)

            seg.graph = self._capture(body)
        seg.graph.replay()

    # -- input dist, possibly one batch ahead ------------------------------------------------------
    def _static_kjt(self, kjt: KeyedJaggedTensor) -> tuple:
        """(slot id, KJT over the slot's static id buffer) for a uniform one-id-per-bag batch, else (None, kjt)"""
        if not self.step_graph or kjt.uniform_length() != 1 or kjt.weights_or_none() is not None:
            return None, kjt
        k = self._next_slot
        self._next_slot ^= 1
        key = (k, tuple(kjt.keys()), kjt.stride())
        sl = self._slots.get(key)
        if sl is None:
            vals = torch.empty_like(kjt.values(), device=self.device)
            sl = {"kjt": KeyedJaggedTensor(list(kjt.keys()), vals, kjt.lengths().to(self.device).clone(), uniform_length=1),
                  "graph": None, "seen": 0, "dense": None, "label": None}
            self._slots[key] = sl
        sl["kjt"].values().copy_(kjt.values(), non_blocking=True)
        return key, sl["kjt"]

    def _begin(self, kjt: KeyedJaggedTensor, after: Optional["torch.cuda.Event"] = None) -> dict:
        ebc = self.model.ebc
        if not self.cuda:
            key, skjt = self._static_kjt(kjt)
            if key is not None and ebc.cap_eligible(skjt, ("sparse",)):
                return self._begin_graphed(self._slots[key], key, skjt)
            st = ebc.input_dist_begin(skjt, ("sparse",), slot=None if key is None else key[0])
            st["slot_key"] = key
            return st
        # the ids must exist before the side stream reads them: either everything queued on the main
        # stream so far, or (prefetch) just the point where this step started -- NOT the step's own
        # work, or the prefetch would queue behind the dense segment it is meant to overlap
        ins = self._in_stream()
        if ins is self._side:
            if after is not None:
                self._side.wait_event(after)
            else:
                self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(ins):
            key, skjt = self._static_kjt(kjt)
            if key is not None and ebc.cap_eligible(skjt, ("sparse",)):
                return self._begin_graphed(self._slots[key], key, skjt)
            st = ebc.input_dist_begin(skjt, ("sparse",), slot=None if key is None else key[0])
            st["slot_key"] = key
            return st

    def _begin_graphed(self, sl: dict, key: tuple, skjt: KeyedJaggedTensor) -> dict:
        """Capacity-bounded input dist of a slot with its two kernel runs replayed from hipGraphs (bucketize | ids
        all-to-all | owner segments, overflow word to the host, both backward plans): 5 host calls instead of 17.
        Runs on the side stream (the caller's stream context)."""
        ebc = self.model.ebc
        st = {"kjt": skjt, "rm": ebc._requester_meta(skjt.keys(), ebc._layout_for(("sparse",))), "uniform": True,
              "dst_names": ("sparse",), "slot": key[0], "slot_key": key}
        ebc.cap_state(st, key[0])
        ig = sl.get("in_graphs")
        if ig is not None and len(ig) == 1:  # native driver: ONE graph, the ids all-to-all inside (`_capture_native`)
            ebc.cap_flag_arm(st)
            ig[0].replay()
            for k in ("ws_dp", "ws_rw"):
                if k in sl["in_st"]:
                    st[k] = sl["in_st"][k]
            st["planned"] = True
        elif ig is not None:
            ig[0].replay()
            ebc.cap_exchange(st)
            ebc.cap_flag_arm(st)  # (the replayed D2H copy of the overflow word lands on this sentinel)
            ig[1].replay()
            for k in ("ws_dp", "ws_rw"):  # the plans live in the slot's workspaces
                if k in sl["in_st"]:
                    st[k] = sl["in_st"][k]
            st["planned"] = True
        else:
            sl["seen_in"] = sl.get("seen_in", 0) + 1
            ins = self._in_stream()
            g01 = None
            if sl["seen_in"] <= self.warmup_iters + 1 or not (self.graph_input_dist and self.use_graph and self._can_graph()):
                ebc.cap_bucketize(st)
                ebc.cap_exchange(st)
                ebc.cap_segments(st)
            elif self._use_native_driver():
                # bucketize | ids all-to-all (the library's own communicator, on the capturing stream) | owner segments,
                # overflow word to the host, both backward plans: one graph, one launch per batch
                def body01():
                    ebc.cap_bucketize(st)
                    comm.all_to_all(st["msg"][0], st["msg"][1], stream=self._raw_stream(ins))
                    ebc.cap_segments(st)
                    if self.plan_ahead:
                        ebc.plan_ahead(st)

                try:
                    comm = self._native_comm_in()
                    ebc.cap_flag_arm(st)
                    g01 = self._capture(body01, stream=ins)
                except Exception as e:  # RCCL not capturable on this stack: the two-graph form with torch's all-to-all between
                    self._native_failed("input dist capture", e)
                    g01 = None
            if g01 is not None:
                ebc.cap_flag_arm(st)
                g01.replay()
                sl["in_graphs"], sl["in_st"] = (g01,), st
                st["planned"] = True
            elif not (sl["seen_in"] <= self.warmup_iters + 1 or not (self.graph_input_dist and self.use_graph and self._can_graph())):
                g0 = self._capture(lambda: ebc.cap_bucketize(st), stream=ins)
                g0.replay()
                ebc.cap_exchange(st)

                def body1():
                    ebc.cap_segments(st)
                    if self.plan_ahead:
                        ebc.plan_ahead(st)

                g1 = self._capture(body1, stream=ins)
                g1.replay()
                sl["in_graphs"], sl["in_st"] = (g0, g1), st
                st["planned"] = True
        ebc.cap_flag_event(st)
        return st

    def _end(self, st: dict) -> dict:
        ebc = self.model.ebc
        if not self.cuda:
            st.pop("_deferred", None)
            return ebc.input_dist_end(st)
        ins = self._in_stream()
        with torch.cuda.stream(ins):
            spec = st.pop("_deferred", False)
            st2 = ebc.input_dist_end(st)
            if spec and st2 is st:
                return st  # the batch fitted: its plans and its `ready` event were queued a step ago (`_begin_ahead`)
            st = st2
            if self.plan_ahead and not st.get("planned"):
                st = ebc.plan_ahead(st)  # K6 of both backward halves needs ids only
            ev = torch.cuda.Event()
            ev.record(ins)
        st["ready"] = ev
        return st

    def _drain_discarded_ahead(self) -> None:
        """A prefetched input dist that is NOT the batch now stepped (the caller changed its mind about the next batch): its
        overflow word's D2H copy may still be in flight towards the pinned word of a slot the new batch is about to re-arm --
        landing after the sentinel it would be read as the new batch's word.  Wait for it before the slot is reused."""
        if self._ahead is None:
            return
        st = self._ahead[1]
        self._ahead = None
        ev = st.get("flag_event") if isinstance(st, dict) else None
        if ev is not None:
            ev.synchronize()
        elif self.cuda and isinstance(st, dict) and "flag_host" in st:
            torch.cuda.synchronize(self.device)

    def _begin_ahead(self, kjt: KeyedJaggedTensor, after) -> dict:
        """Input dist of the NEXT batch, queued behind the start of the current step on the side stream.  Capacity-bounded
        exchange: nothing of it needs the host -- the backward plans are queued right behind it (they depend on ids only)
        and the overflow word is looked at when the batch's own step begins (`step` -> `_end`), a whole step after its
        D2H copy was queued: no host wait inside a step (round 3 blocked here until the side stream had run the bucketize +
        ids all-to-all + flag copy).  A batch that did overflow is redone through the exact exchange then, plans included.
        Exact exchange: its all-to-all split sizes live on the host, the wait stays where it is."""
        st = self._begin(kjt, after)
        if "cap" not in st or "flag_host" not in st:
            return self._end(st)
        if self.cuda:
            ins = self._in_stream()
            with torch.cuda.stream(ins):
                if self.plan_ahead and not st.get("planned"):
                    st = self.model.ebc.plan_ahead(st)
                    st["planned"] = True
                ev = torch.cuda.Event()
                ev.record(ins)
            st["ready"] = ev
        st["_deferred"] = True
        return st

    def _consume(self, st: dict) -> None:
        """Main stream takes over the tensors the side stream produced."""
        if not self.cuda:
            return
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(st["ready"])
        if st.get("slot_key") is not None and "cap" in st:
            return  # every buffer of a pipeline slot is persistent: nothing for the allocator to learn (15 record_stream calls
            #         per step were ~40 us of a step whose HOST time is its duration: profiles/r04o)
        for v in st.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
        sub = st.get("sub")
        if sub is not None:
            for t in (sub.values(), sub.lengths(), sub.weights_or_none(), sub._offsets):
                if t is not None and t.is_cuda:
                    t.record_stream(cur)

    # -- one step --------------------------------------------------------------------------------
    def step(self, dense: torch.Tensor, kjt: KeyedJaggedTensor, label: torch.Tensor,
             next_kjt: Optional[KeyedJaggedTensor] = None) -> torch.Tensor:
        """Forward, backward, sparse + dense optimizer for one batch; returns the (detached) loss.
        `next_kjt` lets the input dist of the following batch overlap this one."""
        model, ebc = self.model, self.model.ebc
        if self.cuda:  # the captured update kernels read learning rates from device scalars: refresh them outside capture
            if self._lr_targets is None:
                from .dense import lr_sync_targets

                self._lr_targets = lr_sync_targets(model, self.opt)
            for o in self._lr_targets:
                o.sync_lr()
        t0 = None
        if self.cuda:
            t0 = torch.cuda.Event()
            t0.record(torch.cuda.current_stream(self.device))
        if self._ahead is not None and self._ahead[0] is kjt:
            st = self._ahead[1]
            if st.get("_deferred"):
                st = self._end(st)  # the overflow word of a batch whose input dist was queued a step ago
        else:
            self._drain_discarded_ahead()
            st = self._end(self._begin(kjt))
        self._ahead = None
        self._consume(st)
        if self.step_graph and "cap" in st and st.get("slot_key") is not None:
            return self._step_whole(st, dense, label, next_kjt, t0)
        if self.step_graph:
            self.eager_steps += 1
        seg = self._segment(dense, label, st["rm"]["widths"][0])
        ebc.lookup(st, [seg.sparse.detach()])
        self._run_dense(seg, dense, label)
        pending = self._begin_ahead(next_kjt, t0) if (next_kjt is not None and self.prefetch) else None
        ebc.backward(st, [seg.grads[0]])
        from .sharding import allreduce_flat_average, dense_grad_views, pack_dense_grads

        pg = list(seg.grads[1:])
        flat = pack_dense_grads(pg)  # DDP semantics: one flat all-reduce (AVG); the optimizer reads views of it
        allreduce_flat_average(flat, model.pg)
        for p, g in zip(self.params, dense_grad_views(flat, pg)):
            p.grad = g
        self.opt.step()
        self._drop_grads()
        if pending is not None:
            self._ahead = (next_kjt, pending)
        return seg.loss

    # -- whole-step graphs ------------------------------------------------------------------------------
    # The step after the input dist = three runs of kernels on static buffers (one hipGraph each) with the RCCL
    # calls between them issued eagerly:
    #   G0  owner row gather, replicas' pooled lookup
    #       rows all-to-all
    #   G1  pooled gather, dense forward + backward, per-id gradient rows, replicas' row sums, dense gradients packed
    #       gradient all-to-all, all-reduce of the replicas' row sums, all-reduce of the dense gradients
    #   G2  owners' sort + fused optimizer, replicas' dense row update, dense gradients unpacked, Adam
    # `overlap_collectives` (the default) cuts G1 behind the per-id gradient rows and G2 behind the owners'
    # update, five graphs: G1a | gradient all-to-all issued | G1b (replicas' row sums, pack) | both all-reduces issued,
    # all-to-all waited for | G2a (owners' sort + optimizer) | all-reduces waited for | G2b.
    def _seg0(self, st: dict, sl: dict) -> None:
        self.model.ebc.seg_owner_rows(st, [sl["sparse"].detach()])

    # the overlapped order cuts G0 behind the owners' row gather: the rows all-to-all is issued there, and the work that does
    # not depend on it -- the replicated tables' pooled lookup and the bottom MLP -- runs while it is in flight (G0b)
    def _seg0_rw(self, st: dict, sl: dict) -> None:
        self.model.ebc.seg_owner_rows(st, [sl["sparse"].detach()], dp=False)

    def _seg0b(self, st: dict, sl: dict) -> None:
        self.model.ebc.seg_dp_pool(st, [sl["sparse"].detach()])
        sl["d"] = self.model.dense_bottom(sl["dense"]) if self._split_bottom() else None

    def _seg1a(self, st: dict, sl: dict) -> None:
        ebc = self.model.ebc
        ebc.seg_pool(st, [sl["sparse"].detach()])
        sl["loss"], sl["logits"], grads = self._dense_fwd_bwd(sl["dense"], sl["sparse"], sl["label"], d=sl.pop("d", None))
        ebc.seg_grads_rw(st, [grads[0]])
        sl["grads"] = list(grads[1:])

    def _seg1b(self, st: dict, sl: dict) -> None:
        from .sharding import pack_dense_grads

        self.model.ebc.seg_grads_dp(st)
        sl["flat"] = pack_dense_grads(sl["grads"])

    def _seg1(self, st: dict, sl: dict) -> None:
        self._seg1a(st, sl)
        self._seg1b(st, sl)

    def _seg2a(self, st: dict, sl: dict) -> None:
        self.model.ebc.seg_apply_rw(st)

    def _seg2b(self, st: dict, sl: dict) -> None:
        from .sharding import dense_grad_views

        self.model.ebc.seg_apply_dp(st)
        for p, g in zip(self.params, dense_grad_views(sl["flat"], sl["grads"])):  # no unpack launch: views of the averaged buffer
            p.grad = g
        self.opt.step()
        self._drop_grads()

    def _seg2(self, st: dict, sl: dict) -> None:
        self._seg2a(st, sl)
        self._seg2b(st, sl)

    def _coll0(self, st: dict, sl: dict) -> None:
        self.model.ebc.coll_rows(st)

    def _coll0_issue(self, st: dict, sl: dict) -> None:
        sl["works"] = [self.model.ebc.coll_rows(st, async_op=self.cuda)]

    def _coll0_wait(self, st: dict, sl: dict) -> None:
        self._wait(sl.pop("works"))

    def _coll1(self, st: dict, sl: dict) -> None:
        from .sharding import allreduce_flat_average

        self.model.ebc.coll_grads(st)
        allreduce_flat_average(sl["flat"], self.model.pg)

    # the overlapped order: a collective is ISSUED (async: RCCL's stream takes it from here) right behind the graph
    # that produced its input, and WAITED FOR (the current stream, not the host) right in front of the graph that
    # reads its output
    def _coll1a(self, st: dict, sl: dict) -> None:
        sl["works"] = [self.model.ebc.coll_grads_rw(st, async_op=self.cuda)]

    def _coll1b(self, st: dict, sl: dict) -> None:
        from .sharding import allreduce_flat_average

        rows = sl["works"]
        sl["works"] = [self.model.ebc.coll_grads_dp(st, async_op=self.cuda), allreduce_flat_average(sl["flat"], self.model.pg, async_op=self.cuda)]
        self._wait(rows)

    def _coll2a(self, st: dict, sl: dict) -> None:
        self._wait(sl.pop("works"))

    @staticmethod
    def _wait(works) -> None:
        for w in works:
            if w is not None:
                w.wait()

    def _step_whole(self, st: dict, dense, label, next_kjt, t0) -> torch.Tensor:
        sl = self._slots[st["slot_key"]]
        if sl["dense"] is None:
            sl["dense"], sl["label"] = torch.empty_like(dense, device=self.device), torch.empty_like(label, device=self.device)
            sl["sparse"] = torch.zeros(dense.shape[0], st["rm"]["widths"][0], dtype=torch.float32, device=self.device, requires_grad=True)
        sl["dense"].copy_(dense, non_blocking=True)
        sl["label"].copy_(label, non_blocking=True)
        if self.overlap_collectives:
            segs = (self._seg0_rw, self._seg0b, self._seg1a, self._seg1b, self._seg2a, self._seg2b)
            colls = (self._coll0_issue, self._coll0_wait, self._coll1a, self._coll1b, self._coll2a, None)
        else:
            segs, colls = (self._seg0, self._seg1, self._seg2), (self._coll0, self._coll1, None)
        capture = False
        if self.use_graph and self._can_graph() and sl["graph"] is None:
            sl["seen"] += 1
            capture = sl["seen"] > self.warmup_iters  # before: eager (lazy inits, GEMM tuning, RCCL channel setup)
            if capture and self.cuda and torch.cuda.current_stream(self.device) == torch.cuda.default_stream(self.device):
                raise RuntimeError("ShardedTrainStep captures on the current stream: run the training loop under a "
                                   "non-default stream (torch.cuda.set_stream)")
        if capture and self._use_native_driver():
            sl["st"] = st
            if self._capture_native(st, sl):
                self.graph_steps += 1
                if next_kjt is not None and self.prefetch:
                    self._ahead = (next_kjt, self._begin_ahead(next_kjt, t0))
                return sl["loss"]
        if capture:
            sl["st"] = st  # the captured kernels read this state's buffers: keep them alive
            graphs = []
            # one memory pool for the slot's graphs: they replay in capture order, never concurrently -- and the autograd
            # graph of the bottom MLP is built in one capture (G0b) and walked backwards in the next (G1a), which is the
            # arrangement of torch.cuda.make_graphed_callables (forward and backward graphs of one pool)
            pool = sl.setdefault("pool", torch.cuda.graph_pool_handle()) if self.cuda else None
        prog = sl.get("program") if not capture else None
        if prog is not None:
            # steady state: the NEXT batch's input dist first (one graph on the side stream, behind the start of this step:
            # its overflow word is what the host needs first when the next step begins), then one native call queues this
            # step's graphs and the all-reduces between them
            if next_kjt is not None and self.prefetch:
                self._ahead = (next_kjt, self._begin_ahead(next_kjt, t0))
            prog.run(self._raw_stream())
            self.graph_steps += 1
            self.native_steps += 1
            return sl["loss"]
        ahead_done = False
        for i, (seg, coll) in enumerate(zip(segs, colls)):
            if i == 2 and not capture and next_kjt is not None and self.prefetch:
                # the NEXT batch's input dist is queued (on the side stream, behind the START of this step) as soon as this
                # step's first graphs are out -- not at the end: its overflow word is read when the next step begins, and
                # queued last it had the host wait there for the whole chain it had just queued (63 us per step, the largest
                # single item of a step whose host time is its duration: profiles/r04r)
                self._ahead = (next_kjt, self._begin_ahead(next_kjt, t0))
                ahead_done = True
            if capture:
                g = self._capture(lambda seg=seg: seg(st, sl), pool=pool)
                graphs.append(g)
                g.replay()
            elif sl["graph"] is not None:
                sl["graph"][i].replay()
            else:
                seg(st, sl)
            if coll is not None:
                coll(sl.get("st", st) if sl["graph"] is not None else st, sl)
        if capture:
            sl["graph"] = graphs
        self.graph_steps += 1
        if next_kjt is not None and self.prefetch and not ahead_done:
            self._ahead = (next_kjt, self._begin_ahead(next_kjt, t0))
        return sl["loss"]

    # -- native step driver -------------------------------------------------------------------------------
    def _use_native_driver(self) -> bool:
        if not (self._can_graph() and self.overlap_collectives):
            return False
        if self.native_driver is None:
            from . import native_step

            # TZR_NATIVE_DRIVER=0: the six-graph form with torch.distributed's collectives between the graphs (the escape hatch
            # should RCCL kernels inside a hipGraph misbehave on some stack; measured here on ROCm 7.2 / RCCL 2.26.6)
            self.native_driver = os.environ.get("TZR_NATIVE_DRIVER", "1") != "0" and native_step.available()
            if self.native_driver and _consumes_rng(self.model):
                # the driver launches torch's captured graphs with a raw hipGraphLaunch: `CUDAGraph.replay()`'s prologue, which
                # advances the philox offset the graph's random kernels read, does not run -- a dropout mask would repeat every
                # step.  Such models keep the six-graph form (replayed through torch).
                self.native_driver = False
        elif self.native_driver and _consumes_rng(self.model):
            raise ValueError("native_driver=True: the model holds an active Dropout; graphs launched by the native driver do not "
                             "advance torch's random state (use native_driver=None / False)")
        return bool(self.native_driver)

    def _native_comm(self):
        if self._comm is None:
            from .native_step import NativeComm

            self._comm = NativeComm(self.model.pg, self.device)
        return self._comm

    def _native_comm_in(self):
        """a second communicator for the input dist: it runs on the side stream, one batch ahead of the step's collectives"""
        if getattr(self, "_comm_in", None) is None:
            from .native_step import NativeComm

            self._comm_in = NativeComm(self.model.ebc.input_dist_group or self.model.pg, self.device)
        return self._comm_in

    def _capture_native(self, st: dict, sl: dict) -> bool:
        """The step of a slot for the native driver.  RCCL calls on the library's own communicator CAN be captured into a
        hipGraph when they sit on the capturing stream itself (scripts/r05/rccl_own_capture_probe.py: `inline` replays; the
        fork / join form and child graphs crash in hipStreamEndCapture on this stack) -- so all four collectives live INSIDE
        the graph, and the step is ONE graph, one launch:

            owners' row gather | rows all-to-all | replicas' lookup + bottom MLP | pooled gather, dense forward + backward,
            per-id gradient rows | gradient all-to-all | replicas' row sums, dense gradients packed | owners' sort + fused
            optimizer | all-reduce of the replicas' row sums | all-reduce of the dense gradients | replicas' dense row
            update, Adam

        In stream order the collectives hide behind nothing -- but every boundary between two graphs of the six-graph order
        is ~20 us of launch latency on this part (profiles/r05q/timeline.txt), four of them per step, and what they bought
        was at most the replicas' 25 us of row sums under the gradient all-to-all and the owners' 26 us update under the
        all-reduces.  This call is the slot's capture step AND a training step: the graph is replayed behind its capture."""
        from .native_step import StepProgram

        ebc = self.model.ebc
        rm = st["rm"]
        has_rw = "rw_n" in rm
        train_rw = ebc.fused_optimizer is not None and has_rw
        train_dp = ebc.fused_optimizer is not None and "dp_n" in rm
        sp = self._raw_stream()

        def body():
            self._seg0_rw(st, sl)
            if has_rw:
                rows_in, _ = ebc._recv_rows_buffer(st["N_pad"], rm["rw_n"])
                comm.all_to_all(st["rows_out"], rows_in[:st["N_pad"]], stream=sp)
            self._seg0b(st, sl)
            self._seg1a(st, sl)
            if train_rw:
                st["grecv"] = ebc._slot(st["slot"], "grecv", (st["n_recv"], ebc.dim), torch.float32)
                comm.all_to_all(st["grow"], st["grecv"], stream=sp)
            self._seg1b(st, sl)
            self._seg2a(st, sl)
            if train_dp:
                comm.all_reduce(ebc._dp_acc, stream=sp)
            comm.all_reduce(sl["flat"], average=True, stream=sp)
            self._seg2b(st, sl)

        try:
            comm = self._native_comm()
            pool = sl.setdefault("pool", torch.cuda.graph_pool_handle()) if self.cuda else None
            g = self._capture(body, pool=pool)
        except Exception as e:
            # RCCL calls that do not capture on this stack (or a communicator that cannot be made): the step keeps working in
            # the six-graph form with torch.distributed's collectives between the graphs -- said once, loudly, not silently
            self._native_failed("step capture", e)
            return False
        g.replay()
        P = StepProgram()
        P.add_graph(g)
        sl["graph"], sl["program"] = [g], P
        return True

    def _native_failed(self, where: str, err: Exception) -> None:
        import warnings

        self.native_driver = False
        self.native_error = f"{where}: {type(err).__name__}: {err}"
        warnings.warn(f"ShardedTrainStep: native step driver given up ({self.native_error}); continuing with the six-graph form "
                      "(torch.distributed collectives between the graphs)", RuntimeWarning)
        if self.cuda:
            torch.cuda.synchronize(self.device)
