#!/usr/bin/env python3
"""DLRM-Criteo training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full training step of examples/dlrm_criteo.config (26 tables x dim 16, 204M rows,
fused sparse Adagrad in backward, bottom/top MLPs, dot interaction, BCE, dense Adam) over one
synthetic Criteo-shaped batch.  Headline workload: GLOBAL batch 65536, i.e. 65536/N
per rank (8192 at N = 8 = `batch_size: 8192` of the config) -- strong scaling; at N > 1 the weak
reading at 8192 per rank is timed as well and reported under `secondary` (`--scaling weak` makes
--global-batch the per-rank batch instead).  Inputs are resident in HBM.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the north-star quantity: HBM roofline of the pooled embedding forward + backward
                (gather kernel, the backward index plan, the fused-optimizer reduce kernel):
                achieved = algorithmic bytes of fwd + bwd / the sum of the launch
                durations, HIP events recorded around each C-ABI call on the launching stream; peak
                8.0 TB/s.  `kernels` lists the three stages with their own bytes, time and fraction
                (the forward gather and the reduce kernel are the two HBM-bound ones).
  e2e           the same step driven through TrainPipeline.progress from pinned HOST batches (H2D of
                the next batch on a copy stream under the current step): the PCIe-inclusive rate.
  cpu_baseline  the CPU oracle ("port") timed on this host on a bounded sample of the workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); 6.29e12 measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="untimed graph replays between the capture and the --warmup / --steps the line reports (one GPU, graph replay): "
                         "the device's clocks after the idle seconds of a capture")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", "--batch", dest="global_batch", type=int, default=65536,
                    help="batch 65536: global under strong scaling (default), per rank under --scaling weak")
    # "samples/s for DLRM-Criteo at GLOBAL batch 65536 on 1/2/4/8 MI355X (per-rank
    # batch 65536/W, i.e. 8192 at W = 8 -- matches batch_size: 8192 of examples/dlrm_criteo.config;
    # also report weak scaling at 8192/rank)".  At N = 1 both readings are the same run.
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--weak-per-rank-batch", type=int, default=8192,
                    help="per-rank batch of the secondary (weak-scaling) reading at N > 1")
    ap.add_argument("--no-e2e", action="store_true", help="skip the TrainPipeline / pinned-host-batch reading")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the secondary readings (config 2 at batch 8192, Zipf ids, row-wise Adagrad)")
    ap.add_argument("--e2e-steps", type=int, default=40)
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--optimizer", choices=["adagrad", "rowwise_adagrad"], default="adagrad")
    ap.add_argument("--row-layout", choices=["interleaved", "split"], default="interleaved")
    ap.add_argument("--rows-cap", type=int, default=0, help="debug: cap table rows (0 = real 40M tables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--n-batches", type=int, default=8, help="distinct synthetic batches cycled through")
    ap.add_argument("--replicate-small", action="store_true",
                    help="with --force-sharded: replicate small tables even at world 1 (exercise that path)")
    ap.add_argument("--layerwise-loss", action="store_true",
                    help="A/B: logits then tzr_bce_logits as separate autograd nodes instead of DLRM.forward_loss")
    ap.add_argument("--owned-wgrad", action="store_true",
                    help="weight gradient of the layer behind the interaction by tzr_dot_interaction_top_wgrad at every batch size (default: up to 16 384 samples)")
    ap.add_argument("--gemm-wgrad", action="store_true",
                    help="weight gradient of the layer behind the interaction by the GEMM library over a stored z (A/B against tzr_dot_interaction_top_wgrad)")
    ap.add_argument("--torch-bce", action="store_true", help="loss: torch BCE-with-logits instead of tzr_bce_logits")
    ap.add_argument("--torch-adam", action="store_true", help="dense optimizer: torch.optim.Adam(fused) instead of tzr_dense_adam")
    ap.add_argument("--secondary-global-batch", type=int, default=None,
                    help="sharded runs also time this global batch after the main loop (default at N > 1: the other "
                         "scaling regime of --global-batch; 0 = off)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded runs: op-by-op autograd step instead of ShardedTrainStep")
    ap.add_argument("--no-plan-ahead", action="store_true",
                    help="sharded runs: keep the backward index plans (K6) on the main stream instead of one batch ahead")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="sharded runs: do not run the next batch's input dist ahead on the side stream")
    ap.add_argument("--exchange", choices=["auto", "exact", "capacity"], default="auto",
                    help="sharded runs: 'capacity' = fixed-size message slices (one ids all-to-all, no counts through the "
                         "host); a batch that overflows is redone through the exact exchange.  'auto': capacity + "
                         "--step-graph when the per-rank batch is <= 65536 (1-rank proxy with the native step driver: 0.26-0.28 vs "
                         "0.47 ms at 8192, 0.77 vs 0.86 ms at 65536 -- profiles/r05s, r05v), else exact")
    ap.add_argument("--capacity-factor", type=float, default=1.25)
    ap.add_argument("--step-graph", action="store_true",
                    help="sharded runs with --exchange capacity: everything after the input dist (lookups, dense segment, "
                         "sparse + dense optimizers) replayed from ONE hipGraph per pipeline slot with the RCCL calls captured inside (native step driver; "
                         "--no-native-driver: six graphs, RCCL calls between them)")
    ap.add_argument("--overlap-collectives", choices=["auto", "on", "off"], default="auto",
                    help="--step-graph runs: five graphs with the gradient all-to-all / all-reduces issued async between them "
                         "(auto = on; off: three graphs, every collective waited for where it is issued)")
    ap.add_argument("--no-graph-input-dist", action="store_true",
                    help="--step-graph runs: launch the input dist's kernels one by one instead of replaying its two kernel runs from "
                         "hipGraphs (the step is host-bound: 2 replays instead of ~8 launches and their Python, profiles/r04r)")
    ap.add_argument("--no-native-driver", action="store_true",
                    help="--step-graph runs: issue the slot's graphs and collectives from Python through torch's ProcessGroup "
                         "(the round-4 path) instead of one native call on the library's own RCCL communicator (native_step.py)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N=1 debugging: run the row-wise sharded module over a 1-rank RCCL group")
    ap.add_argument("--no-graph", action="store_true", help="N=1: launch every step eagerly instead of hipGraph replay")
    ap.add_argument("--async-plan", action="store_true", help="run the backward index plan on a side stream")
    ap.add_argument("--no-tunable-gemm", action="store_true",
                    help="leave the MLP GEMMs on PyTorch's default hipBLASLt heuristics")
    ap.add_argument("--delta-tracker", action="store_true",
                    help="experiment: record every lookup in the delta-embedding tracker (tzr_delta_mark inside the step); off by default")
    ap.add_argument("--tune", action="append", default=[], help="name=value passed to tzr_tune")
    ap.add_argument("--lib", default=None, help="experiments: file name of a variant build under torcheasyrec_amd/ instead of the product library")
    ap.add_argument("--emulator", action="store_true",
                    help="CPU plumbing check of the N-rank launch path: kernels through the lane emulator (tests/emu), gloo "
                         "instead of RCCL, tables capped (--rows-cap, default 2000), no graphs / e2e / secondary readings.  "
                         "Never a performance number: the line says so in `data`")
    ap.add_argument("--n1-ms", type=float, default=0.0,
                    help="sharded runs: ms per GLOBAL-batch step of the N = 1 line, for the `projection` object's scaling ratios")
    ap.add_argument("--projection-world", type=int, default=0,
                    help="sharded runs: world size the `projection` is made for (default: the job's, or 65536 / per-rank batch at N = 1)")
    ap.add_argument("--no-config-models", action="store_true",
                    help="N = 1 default line: skip the DeepFM / multi_tower_din / MMoE + ZCH step readings")
    ap.add_argument("--no-sharded-proxy", action="store_true",
                    help="N = 1 default line: skip the 1-rank RCCL proxy of the 8192-per-rank sharded step (a child process)")
    ap.add_argument("--dp-max-rows", type=int, default=0,
                    help="sharded runs: tables of at most this many rows are replicated (default: planner.pick_dp_max_rows -- the "
                         "threshold with the least modelled wire + kernel time at the run's (or the projection's) world size; 500 under "
                         "--emulator, so that capped tables still take the row-wise exchange; 65536 = the round-4 constant)")
    ap.add_argument("--no-fuse-finish", action="store_true",
                    help="dense optimizer: the gradients of the bottom MLP and of the first top-MLP layer as finished tensors (their own "
                         "finishing launches) instead of partial sums added up inside the optimizer's launch")
    ap.add_argument("--forms", choices=["auto", "ab", "one_graph", "overlapped"], default="auto",
                    help="sharded --step-graph runs: which form of the step is timed.  one_graph = the native driver (the step ONE "
                         "hipGraph with its RCCL collectives inline), overlapped = six hipGraphs with torch.distributed's collectives "
                         "between them.  ab = both are warmed up and timed (>= 50 steps each, reported as `sharded_forms`), the faster "
                         "one gives `value`; a one-graph form that raises or does not come back within --form-timeout falls back to "
                         "the overlapped form and the line says so.  auto = ab on more than one rank, else whatever the library picks")
    ap.add_argument("--form-timeout", type=float, default=240.0,
                    help="--forms ab: seconds the one-graph form gets for its warm-up + trial + timed steps before the run gives it up")
    ap.add_argument("--form-trial-steps", type=int, default=50, help="--forms ab: steps of each form's trial (at least --steps)")
    ap.add_argument("--no-spawn", action="store_true",
                    help="--gpus N > 1 outside torch.distributed.run: fail instead of re-launching under it")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` (the driver's form) outside a launcher: start the N ranks ourselves --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` -- and hand its exit code back.  Rank 0 of that job prints the JSON line (its `ranks_seen` is an
    all-reduce of ones over the process group: proof that N ranks took part)."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TZR_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


SECONDARY_REPLAYS = 200  # secondary readings are timed over at least this many graph replays / kernel iterations


def config_model_steps(dev, work_stream, steps: int = 20, only=None):
    """Train-step timing of the other BASELINE.json model families at their config's batch size (8192), built from a
    pipeline config TEXT through the same seam a tzrec user has (config.load_pipeline_spec -> rank_model.build_rank_model):
    DeepFM-Criteo (configs[0]'s model on the GPU), multi_tower_din on the Taobao features with a 100-step click sequence
    (configs[3]), MMoE with the user id behind a 200 M-row zero-collision hash, LFU (configs[4]).  One GPU, inputs resident,
    fused sparse Adagrad + dense Adam, nothing skipped.  Eager launches (`ms_per_step`), and the same step replayed from a
    hipGraph (`graph_ms_per_step`; the ZCH step in ring mode: device iteration counter + candidate ring, zch.py)."""
    from torcheasyrec_amd import example_configs as ec
    from torcheasyrec_amd.config import load_pipeline_spec
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, _backward_of_losses, _losses_and_predictions
    from torcheasyrec_amd.rank_model import build_rank_model
    from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor

    def batches(spec, B, n, seed, raw_id_feature=None):
        rng = np.random.default_rng(seed)
        sparse = [f for f in spec.features if f.is_sparse]
        dense = [f for f in spec.features if not f.is_sparse]
        out = []
        for _ in range(n):
            vals, lens = [], []
            seq_len = None
            for f in sparse:
                if f.is_sequence:  # the sub-features of one sequence share their lengths: histories of 10 .. sequence_length clicks
                    if seq_len is None:
                        seq_len = rng.integers(10, f.sequence_length + 1, size=B).astype(np.int32)
                    ln = seq_len
                else:
                    ln = np.ones(B, np.int32)
                lens.append(ln)
                if f.name == raw_id_feature:  # raw 64-bit ids in front of the zero-collision hash: Zipf over 2^34 users
                    u = np.minimum(rng.zipf(1.05, size=int(ln.sum())), 1 << 34).astype(np.int64)
                    vals.append((u * 2654435761 + (1 << 40)) & ((1 << 62) - 1))
                else:
                    vals.append(rng.integers(0, f.num_embeddings, size=int(ln.sum())))
            uni = all(not f.is_sequence for f in sparse)
            kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(np.concatenate(vals).astype(np.int64)),
                                    torch.from_numpy(np.concatenate(lens)), **({"uniform_length": 1} if uni else {}))
            kt = KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                             torch.from_numpy(rng.random((B, max(1, sum(f.value_dim for f in dense))), dtype=np.float32)[:, :sum(f.value_dim for f in dense)]))
            labels = {l: torch.from_numpy((rng.random(B) < 0.25).astype(np.int64)) for l in spec.label_fields}
            out.append(Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt}, labels).to(dev))
        return out

    def run(name, text, config, raw_id_feature=None, capturable=True):
        spec = load_pipeline_spec(text)
        B = spec.batch_size
        torch.manual_seed(7)
        model = build_rank_model(spec, device=dev)
        mc_ = getattr(model.embedding_group, "mc", None)
        if mc_ is not None and capturable:  # ring mode: device iteration counter + candidate ring, the step can be captured (zch.py)
            mc_.device_profile = True
        # (fuse_finish: the Linear + ReLU layers' bias gradients go into the optimizer's launch as the mask kernels' partial rows --
        # no finishing launch per layer; TZR_MODELS_FUSE_FINISH=0: the A/B switch)
        opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr, fuse_finish=os.environ.get("TZR_MODELS_FUSE_FINISH", "1") != "0")
        bs = batches(spec, B, 4, 11, raw_id_feature)
        n_ids = int(np.mean([b.sparse_features[BASE_DATA_GROUP].values().numel() for b in bs]))

        def step(b):
            opt.zero_grad(set_to_none=True)
            losses, _ = _losses_and_predictions(model, model.loss, b)
            _backward_of_losses(losses)
            opt.step()
            return losses

        for i in range(4):
            losses = step(bs[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            losses = step(bs[i % 4])
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        eager = time.perf_counter() - t0
        out = {"config": config, "batch": B, "ids_per_step": n_ids, "ms_per_step": eager / steps * 1e3, "host_queue_ms_per_step": host / steps * 1e3,
               "value": B * steps / eager, "unit": "samples/s", "launch": "eager",
               "loss": {k: float(v.detach()) for k, v in losses.items()}}
        if capturable:
            eg = getattr(model, "embedding_group", None)
            if eg is not None and getattr(eg, "_seq_info", None):
                # sequence groups padded to the configured sequence_length instead of the batch's longest sequence (read back from
                # the device, which a capture cannot do): same results, the padded positions are masked
                eg.static_sequence_padding = True
                out["graph_static_sequence_padding"] = True
            try:
                gs, pool = [], None
                for b in bs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, stream=work_stream):
                        step(b)
                    pool = g.pool()
                    gs.append(g)
                from torcheasyrec_amd.embedding_group import after_graph_replay

                for i in range(4 + 40):  # (every graph launched once + ~20-100 ms of replays: the device's clocks after the idle capture)
                    gs[i % 4].replay()
                    after_graph_replay(model)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n_rep = max(steps, SECONDARY_REPLAYS)
                for i in range(n_rep):
                    gs[i % 4].replay()
                    after_graph_replay(model)  # (a zero-collision hash counts the step; its rounds run between replays when due)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                out.update(graph_ms_per_step=el / n_rep * 1e3, graph_value=B * n_rep / el, graph_replays=n_rep)
                del gs
            except Exception as e:  # a step that is not capturable stays with its eager reading
                out["graph_error"] = repr(e)[:160]
                torch.cuda.synchronize()
        mc = getattr(model.embedding_group, "mc", None)
        if mc is not None:  # one admission / eviction round of the zero-collision hash, timed on its own
            name_t = next(iter(mc.modules_by_table))
            mod = mc.modules_by_table[name_t]
            cand = mc.pending_candidates(name_t)
            # a throwaway module (half as many rows as there are candidates: it admits AND evicts) runs two rounds first: the FIRST use of torch.sort / unique / nonzero and of the selection
            # kernels in a process loads their code objects -- 40-90 ms each, 300 of the 302 ms of a first round and nothing of
            # the second (profiles/r05z/zch_round_profile.txt) -- a cost of the process, not of a round every 1 000 steps
            from torcheasyrec_amd.zch import ManagedCollisionModule, ZchConfig

            n_c = max(int(cand.numel()), 6000)  # (same candidate count: torch picks its sort / select kernels by size)
            toy = ManagedCollisionModule(ZchConfig(max(4096, n_c // 2), 1, mod.cfg.policy, mod.cfg.decay_exponent), dev)
            for it in (1, 2):
                toy.update_and_evict(torch.randint(0, 1 << 40, (n_c,), device=dev), it)
            del toy
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            changed = mod.update_and_evict(cand, mc._iter)
            torch.cuda.synchronize()
            ev = (time.perf_counter() - t0) * 1e3
            out["zch"] = {"table": name_t, "zch_size": mod.cfg.zch_size, "policy": mod.cfg.policy, "eviction_interval": mod.cfg.eviction_interval,
                          "round_ms": ev, "candidates": int(cand.numel()), "rows_changed": int(changed.numel()),
                          "amortised_ms_per_step": ev / max(mod.cfg.eviction_interval, 1)}
        del model, opt, bs
        torch.cuda.empty_cache()
        return out

    res = {}
    for key, mk, config, kw in (
            ("deepfm_criteo_b8192", ec.deepfm_criteo, "DeepFM on the Criteo features (examples/deepfm_criteo.config: 52 tables wide + deep, FM, "
             "deep {512,256,128}, final {64}), batch 8192", {}),
            ("din_taobao_b8192", ec.multi_tower_din_taobao, "multi_tower_din on the Taobao features (examples/multi_tower_din_taobao.config: 100-step "
             "click sequence, DIN attention {256,64}), batch 8192, histories of 10..100 clicks", {}),
            ("mmoe_zch_b8192", ec.mmoe_taobao_zch, "MMoE (3 experts, ctr + cvr towers) with user_id behind a 200M-row zero-collision hash, LFU "
             "(BASELINE.json configs[4]), batch 8192, raw 64-bit Zipf user ids", {"raw_id_feature": "user_id"})):
        if only is not None and key not in only:
            continue
        try:
            res[key] = run(key, mk(), config, **kw)
        except Exception as e:
            res[key] = {"config": config, "error": repr(e)[:300]}
            torch.cuda.empty_cache()
    return res


def sharded_proxy(args) -> dict:
    """`python bench.py --force-sharded --replicate-small --global-batch 8192` as a child: the row-wise sharded step of ONE
    rank at 8192 samples per rank (the per-rank work of the 8-GPU job at global batch 65536) on a 1-rank RCCL group --
    every kernel and RCCL call of the step runs, the collectives are self copies -- and the scaling projection it
    implies.  A process of its own, so that nothing of it (a second set of tables, the process group) can touch the
    parent's line."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-sharded", "--replicate-small", "--global-batch", "8192",
           "--steps", str(max(args.steps, SECONDARY_REPLAYS)), "--warmup", "12", "--no-cpu-baseline", "--no-e2e", "--projection-world", "8"]
    try:
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=420,
                            env=dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300)))
        child = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
        return {"config": "row-wise sharded step of ONE rank at 8192 samples per rank on a 1-rank RCCL group (the per-rank work of "
                          "the 8-GPU job at global batch 65536); collectives are self copies; measured in a child process before "
                          "this process touched the GPU",
                "ms_per_step": child["ms_per_step"], "host_queue_ms_per_step": child.get("host_queue_ms_per_step"),
                "host_flag_wait_ms_per_step": child.get("host_flag_wait_ms_per_step"), "host_busy_ms_per_step": child.get("host_busy_ms_per_step"),
                "launch": child["launch"], "parallelism": child["config"]["parallelism"],
                "exchange": child.get("exchange"), "projection": child.get("projection"), "collectives": child.get("collectives")}
    except Exception as e:
        return {"error": repr(e)[:300]}


def xgmi_projection(model, B_local: int, world_target: int, step_ms: float, n1_ms_per_global_step=None, capacity_factor=1.25,
                    collective_latency_us=None, in_step_collectives=4, input_dist_on_main=True):
    """The scaling arithmetic of the sharded step: what `step_ms` -- this run's time for ONE rank's
    share of the step at `B_local` samples per rank -- means at `world_target` ranks, with the bytes SURVEY 8(d) counts
    for the exchange priced at 153 GB/s per xGMI link (7 links per GPU, one per peer: a full-mesh all-to-all uses all
    of them at once; the ring all-reduce is bound by ONE link's rate).  Wire time is reported both as hidden (overlap
    perfect) and as exposed (no overlap at all): the truth on hardware lies between."""
    W = world_target
    link = 153e9
    D = model.dim
    e = model.ebc
    F_rw = len(e._rw)
    dp_rows = int(getattr(e, "_dp_rows", 0))
    dense_bytes = sum(p.numel() for p in model.dense_parameters()) * 4
    ids_msg = 8.0 * capacity_factor * F_rw * B_local / W          # bytes to ONE peer (capacity-bounded slices)
    rows_msg = 4.0 * D * capacity_factor * F_rw * B_local / W      # rows back / gradient rows out, per peer
    a2a_us = lambda b: b / link * 1e6                               # noqa: E731 -- all peers' links run concurrently
    ring = lambda nbytes: 2.0 * (W - 1) / W * nbytes / link * 1e6  # noqa: E731 -- ring all-reduce, per-link bound
    wire = {"ids_all_to_all_us": a2a_us(ids_msg), "rows_all_to_all_us": a2a_us(rows_msg), "grad_rows_all_to_all_us": a2a_us(rows_msg),
            "replica_row_sums_all_reduce_us": ring(dp_rows * D * 4.0), "dense_grads_all_reduce_us": ring(dense_bytes)}
    wire_total = sum(wire.values())
    # Bytes are not the whole price of a collective: each one is a launch + a rendezvous of W ranks' kernels.  Until a multi-rank
    # run measures it, the planner's own constant (planner.Topology.collective_latency, 20 us) per collective: the step has
    # `in_step_collectives` of them in stream order (rows and gradient all-to-all, the two all-reduces: nothing hides them in
    # the one-graph form) plus the input dist's ids all-to-all when that runs on the step's stream (the ordering edge of
    # ShardedTrainStep._input_dist_on_main) -- the 1-rank proxy's step_ms holds their self-copy kernels, not this.
    if collective_latency_us is None:
        from torcheasyrec_amd.planner import Topology

        collective_latency_us = Topology(W).collective_latency * 1e6
    n_coll = in_step_collectives + (1 if input_dist_on_main else 0)
    latency_total = n_coll * collective_latency_us
    # the critical path cannot hide the rows all-to-all (nothing but the bottom MLP is independent of it) nor the gradient
    # all-to-all's tail; the all-reduces fly under the owners' update
    exposed_min = wire["rows_all_to_all_us"] + wire["grad_rows_all_to_all_us"]
    out = {"per_rank_batch": B_local, "world": W, "proxy_ms_per_step": step_ms,
           "xgmi_link_GBps": link / 1e9, "wire_us": wire, "wire_total_us": wire_total,
           "bytes_per_rank_per_step": {"ids_out": ids_msg * (W - 1), "rows_in": rows_msg * (W - 1), "grad_rows_out": rows_msg * (W - 1),
                                       "replica_row_sums": dp_rows * D * 4.0, "dense_grads": float(dense_bytes)},
           "samples_per_s_if_wire_hidden": W * B_local / (step_ms * 1e-3),
           "samples_per_s_if_a2a_exposed": W * B_local / (step_ms * 1e-3 + exposed_min * 1e-6),
           "samples_per_s_if_wire_exposed": W * B_local / (step_ms * 1e-3 + wire_total * 1e-6),
           "collective_latency_us": collective_latency_us, "collectives_in_stream_order": n_coll, "latency_total_us": latency_total,
           "latency_source": "planner.Topology.collective_latency (assumed, not measured: no multi-rank run from this side)",
           "samples_per_s_with_latency": {
               "wire_hidden": W * B_local / (step_ms * 1e-3 + latency_total * 1e-6),
               "a2a_exposed": W * B_local / (step_ms * 1e-3 + (exposed_min + latency_total) * 1e-6),
               "wire_exposed": W * B_local / (step_ms * 1e-3 + (wire_total + latency_total) * 1e-6)}}
    if n1_ms_per_global_step:
        n1 = W * B_local / (n1_ms_per_global_step * 1e-3)
        out["n1_samples_per_s"] = n1
        out["scaling_vs_n1"] = {"wire_hidden": out["samples_per_s_if_wire_hidden"] / n1,
                                "a2a_exposed": out["samples_per_s_if_a2a_exposed"] / n1,
                                "wire_exposed": out["samples_per_s_if_wire_exposed"] / n1}
        out["scaling_vs_n1_with_latency"] = {k: v / n1 for k, v in out["samples_per_s_with_latency"].items()}
        out["step_ms_needed_for_6x"] = n1_ms_per_global_step / 6.0
    return out


class _Timers:
    """HIP events around launches on torch's current stream (the stream the kernels run on)."""

    def __init__(self):
        self.pairs = {}

    def start(self, name):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.setdefault(name, []).append((e0, e1))
        return e1

    def mean_ms(self, name):
        ps = self.pairs.get(name, [])
        return float(np.mean([a.elapsed_time(b) for a, b in ps])) if ps else None

    def median_ms(self, name):
        ps = self.pairs.get(name, [])
        return float(np.median([a.elapsed_time(b) for a, b in ps])) if ps else None


def enable_tunable_gemm():
    """The MLPs stay on PyTorch.  Its default hipBLASLt heuristic picks poor
    kernels for the skinny fp32 GEMMs of DLRM (N = 64/32/16): PyTorch TunableOp selects among the
    rocBLAS / hipBLASLt solutions instead (fp32 in, fp32 accumulate: numerics unchanged).  A tuning
    file for the B=65536 shapes ships with the package; other shapes are tuned during warm-up."""
    import shutil
    import torch.cuda.tunable as tn

    src = os.path.join(ROOT, "torcheasyrec_amd", "tunableop_gfx950.csv")
    # TZR_TUNABLE_SAVE=<path>: TunableOp writes its table there at exit (shipped entries + whatever
    # this run tuned) -- how torcheasyrec_amd/tunableop_gfx950.csv is refreshed
    dst = os.environ.get("TZR_TUNABLE_SAVE") or f"/tmp/tzr_tunableop_{os.getpid()}.csv"
    if os.path.exists(src):
        shutil.copy(src, dst)
    tn.enable(True)
    # profiling runs set TZR_TUNABLE_TUNING=0: use the shipped selections, never launch candidates
    tn.tuning_enable(os.environ.get("TZR_TUNABLE_TUNING", "1") != "0")
    tn.set_filename(dst, insert_device_ordinal=False)
    if os.path.exists(dst):
        tn.read_file(dst)


def interaction_top_rooflines(dev, B):
    """tzr_dot_interaction_top_fwd / _bwd / _wgrad at the DLRM-Criteo shape (27 rows of 16, first top layer 783 -> 64), as the
    step launches them (the forward keeps no z), alone: median of 200 launches, HIP events on the launching stream.  The
    weight gradient's flops are the product's 2 x 783 x 64 per sample: its rebuilt pair blocks (2 x 3 x 16 x 16 x 16 more) and
    its second launch (the sum over the batch slices) are overhead, inside `launch_ms`."""
    from torcheasyrec_amd import _lib

    L = _lib.lib()
    D, F, H = 16, 26, 64
    n = F + 1
    P = n * (n - 1) // 2
    width = P + D * n
    st = _lib.stream_ptr(dev)
    dense, sparse = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev)
    W1, b1, g1 = torch.randn(H, width, device=dev) * 0.05, torch.randn(H, device=dev), torch.randn(B, H, device=dev)
    y1 = torch.empty(B, H, device=dev)
    gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
    dW = torch.empty(H, width, device=dev)
    ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), dev)

    def fwd():
        _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1),
                                                 H, 1, None, 0, _lib.ptr(y1), H, st), "tzr_dot_interaction_top_fwd")

    def wgrad():
        _lib.check(L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, None,
                                                   _lib.ptr(dW), width, _lib.ptr(ws), ws.numel(), st), "tzr_dot_interaction_top_wgrad")

    def bwd():
        _lib.check(L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1),
                                                 width, None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st), "tzr_dot_interaction_top_bwd")

    out = {"peak": 157.3, "unit": "TFLOP/s", "bound": "mfma", "dtype": "f32 (v_mfma_f32_16x16x4_f32, exact)"}
    kernels = {"forward": "tzr_ia_top_fwd_kernel", "backward": "tzr_ia_top_bwd_kernel",
               "weight_gradient": "tzr_ia_wgrad_kernel + tzr_ia_wgrad_reduce_kernel"}
    for name, fn, flop in (("forward", fwd, B * (2.0 * width * H + 2.0 * P * D)),
                           ("backward", bwd, B * (2.0 * width * H + 2.0 * n * n * D)),
                           ("weight_gradient", wgrad, B * 2.0 * width * H)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(1e7))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(SECONDARY_REPLAYS)]
        for a_, b_ in ev:
            a_.record()
            fn()
            b_.record()
        torch.cuda.synchronize()
        ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)[len(ev) // 2]
        out[name] = {"kernel": kernels[name], "launch_ms": ms, "algorithmic_flop": flop,
                     "achieved": flop / (ms * 1e-3) / 1e12, "frac": flop / (ms * 1e-3) / 157.3e12}
    return out


def _emb_calls(ebc, kjt, gbuf):
    """The C-ABI calls of the embedding path as the training step makes them: the forward (its launch carries the backward's
    index plan when the batch takes the cells plan: tzr_pooled_fwd_cells_plan), the plan if it did not, the apply."""
    ebc._launch_forward(kjt, ("sparse",), with_plan=True)
    if getattr(kjt, "_tzr_plan", None) is None:
        ebc.plan_backward(kjt, ("sparse",))
    ebc._launch_backward(kjt, ("sparse",), [gbuf])


def pmc_traffic(args, B_local):
    """HBM bytes per step of the six embedding launches from the rocprofv3 PMC passes kept under profiles/
    (FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes, scripts/pmc_summary.py).  Counters cannot be read
    inside this process, so the figure comes from a file -- but only from one recorded on THIS library: the
    summary carries the digest of the kernel sources it was measured on (torcheasyrec_amd/_build._digest), and a
    file of another digest (a kernel changed since) is refused.  Returns (bytes | None, why)."""
    import glob

    from torcheasyrec_amd import _build

    if not (B_local == 65536 and args.dist == "uniform" and not args.rows_cap and args.optimizer == "adagrad"):
        return None, "recorded for B = 65536, uniform ids, adagrad only"
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))  # newest round last
    dig = _build._digest()
    for p in reversed(found):
        try:
            d = json.load(open(p))
            if d.get("lib_digest") == dig:
                return float(d["kernels"]["__embedding_fwd_bwd__"]["traffic_corrected"]), os.path.relpath(p, ROOT)
        except Exception:
            continue
    return None, f"no profiles/r*/pmc_traffic.json recorded on library digest {dig[:16]} (kernels changed since the last PMC pass)"


def cpu_baseline(seconds: float):
    """CPU oracle train step (oracle/tzrec_oracle.py, kind "port") on a bounded sample:
    per-rank batch 8192, the five 40M-row tables scaled to 4M rows so the host holds them."""
    from oracle import tzrec_oracle as orc
    from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, synthetic_batch

    torch.set_num_threads(os.cpu_count() or 1)
    rows = [min(r, 4_000_000) for r in CRITEO_ROWS]
    B = 8192
    g = torch.Generator().manual_seed(0)
    W = [(torch.rand(r, 16, generator=g) * 2 - 1) * (1.0 / r) ** 0.5 for r in rows]
    M = [np.zeros((r, 16), np.float32) for r in rows]

    def mk(i, o):
        return [(torch.randn(b, a, generator=g) * (1.0 / a) ** 0.5).requires_grad_(True) for a, b in zip(i, o)]

    dw, fw = mk([NUM_DENSE, 64], [64, 16]), mk([783, 64], [64, 32])
    db = [torch.zeros(64, requires_grad=True), torch.zeros(16, requires_grad=True)]
    fb = [torch.zeros(64, requires_grad=True), torch.zeros(32, requires_grad=True)]
    ow, ob = torch.randn(1, 32, generator=g).requires_grad_(True), torch.zeros(1, requires_grad=True)
    params = dw + db + fw + fb + [ow, ob]
    adam = torch.optim.Adam(params, lr=1e-3)
    p = {"dim": 16, "dense_mlp": list(zip(dw, db)), "final_mlp": list(zip(fw, fb)), "output": (ow, ob),
         "arch_with_sparse": True}
    opt = orc.SparseOptim(kind="adagrad", lr=1e-3)
    batches = [synthetic_batch(s, B, rows) for s in range(4)]

    def step(i):
        dense, kjt, label = batches[i % len(batches)]
        blocks = [b.requires_grad_(True) for b in orc.pooled_lookup(W, ["sum"] * 26, kjt.values(), kjt.lengths(), B)]
        loss = orc.bce_with_logits(orc.dlrm_forward(dense, torch.cat(blocks, dim=1), p), label)
        adam.zero_grad()
        grads = torch.autograd.grad(loss, blocks + params)
        for q, gq in zip(params, grads[26:]):
            q.grad = gq
        adam.step()
        vals = kjt.values().numpy()
        for t in range(26):
            w = W[t].numpy()
            orc.sparse_update(w, M[t], vals[t * B:(t + 1) * B], grads[t].numpy(), opt)

    # thread count: the ops are small (8192-row bags, 64-wide GEMMs); all cores of a big host
    # thrash.  Try a few settings for one step each and keep the fastest.
    step(0)
    best_nt, best_t = None, None
    for nt in sorted({8, 16, 32, min(64, os.cpu_count() or 8)}):
        if nt > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(nt)
        step(0)
        ts = time.perf_counter()
        step(1)
        dt = time.perf_counter() - ts
        if best_t is None or dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    n, t0 = 0, time.perf_counter()
    while True:
        step(n + 1)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": n * B / el, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of the CPU oracle at batch {B}, 40M-row tables scaled to 4M rows, {el:.1f} s"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ and not args.no_spawn \
                and not os.environ.get("TZR_BENCH_SPAWNED"):
            raise SystemExit(spawn_ranks(args))  # `python bench.py --gpus N`: the N ranks are started here
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (or plainly as `python bench.py --gpus {args.gpus}`)")
    emu = args.emulator
    if args.gemm_wgrad or args.owned_wgrad:
        import torcheasyrec_amd.dense as _dense

        _dense.OWNED_WGRAD = bool(args.owned_wgrad)
    # N = 1 default line: the 8192-per-rank sharded step on a 1-rank RCCL group, in a child process that runs BEFORE this
    # process touches the GPU (two processes on one GPU time-slice its queues: 0.50 ms measured next to an idle parent
    # against 0.36 ms alone, profiles/r04l)
    proxy_child = None
    if (world == 1 and args.gpus == 1 and not emu and not args.force_sharded and not args.no_secondary and not args.no_sharded_proxy
            and args.optimizer == "adagrad" and args.dist == "uniform" and not args.rows_cap and args.global_batch == 65536
            and args.scaling == "strong"):
        proxy_child = sharded_proxy(args)
    if emu:
        args.rows_cap = args.rows_cap or 2000
        args.no_graph = args.no_e2e = args.no_secondary = True
        dev = torch.device("cpu")
        work_stream = None
    else:
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        # everything (warm-up, capture, replay, instrumented steps) runs on ONE non-default stream, so
        # autograd's AccumulateGrad nodes and the captured graphs agree on it
        work_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(work_stream)

    def sync():
        if not emu:
            torch.cuda.synchronize()

    sharded = world > 1 or args.force_sharded
    ranks_seen, coll_lib = 1, None
    if sharded:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the process group's flight recorder is what ShardedTrainStep asks whether the watchdog still lists a collective
        # before it opens a hipGraph capture (sharded_step._quiesce_process_group)
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            coll_lib = "gloo (CPU plumbing run)"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            try:
                coll_lib = "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                coll_lib = "RCCL"
        # proof that `world` ranks are in the job: every rank contributes a one
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit(f"process group saw {ranks_seen} ranks, expected {world}")

    from torcheasyrec_amd import _build, _lib
    from torcheasyrec_amd.criteo import (CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, algorithmic_bytes,
                                         criteo_tables, synthetic_batch)
    from torcheasyrec_amd.dlrm import DLRM, bce_with_logits
    if args.torch_bce:
        def bce_with_logits(logits, labels):  # noqa: F811 - A/B switch
            return torch.nn.functional.binary_cross_entropy_with_logits(logits, labels.float())
    from torcheasyrec_amd.dense import root_loss, unit_gradient
    from torcheasyrec_amd.embedding import SparseOptimizerConfig

    unit_gradient(torch.zeros((), dtype=torch.float32, device=dev))  # (the backward's root gradient exists before any capture)

    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu.build_emu import build as build_emu

        _lib.use_library(build_emu())
        assert _lib.backend() == "emu"
    else:
        # (--lib: a variant build for same-box A/B, scripts/r06/build_variant.py; its numbers are experiments, never the contract line)
        _lib.use_library(os.path.join(ROOT, "torcheasyrec_amd", args.lib) if args.lib else _build.build())
        assert _lib.backend() == "hip-gfx950"
    if not args.no_tunable_gemm and not emu:
        enable_tunable_gemm()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.lib().tzr_tune(k.encode(), int(v)), f"tzr_tune {kv}")

    rows = [min(r, args.rows_cap) for r in CRITEO_ROWS] if args.rows_cap else list(CRITEO_ROWS)
    B_global = args.global_batch if args.scaling == "strong" else args.global_batch * world
    B_local = B_global // world
    if args.exchange == "auto":
        small = sharded and not args.no_pipeline and (not args.no_graph or emu) and B_local <= 65536  # (profiles/r05v: 0.77 vs 0.86 ms at 65 536 per rank)
        args.exchange = "capacity" if small else "exact"
        args.step_graph = args.step_graph or small
    if args.step_graph and args.exchange != "capacity":
        raise SystemExit("--step-graph needs --exchange capacity")
    torch.manual_seed(1234)
    sopt = SparseOptimizerConfig(kind=args.optimizer, lr=1e-3)
    if not sharded:
        model = DLRM(criteo_tables(rows), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=sopt,
                     row_layout=args.row_layout)
        parallelism = "single GPU, all tables local (table-wise on one rank)"
    else:
        from torcheasyrec_amd.sharding import ShardedDLRM

        # which tables are replicated: the planner's wire + kernel arithmetic for the world the run stands for (its own, or the
        # one a 1-rank proxy projects to) -- planner.pick_dp_max_rows; `--dp-max-rows N` pins it (65536: the round-4 constant)
        from torcheasyrec_amd.planner import pick_dp_max_rows

        W_plan = world if world > 1 else (args.projection_world or max(2, 65536 // max(B_local, 1)))
        dp_pick, dp_costs = pick_dp_max_rows(rows, 16, W_plan, B_local, capacity_factor=args.capacity_factor)
        dp_max_rows = args.dp_max_rows or (500 if emu else dp_pick)
        dp_choice = {"world": W_plan, "picked": dp_pick, "used": dp_max_rows,
                     "candidates": [c for c in dp_costs if c["dp_max_rows"] in (dp_pick, dp_max_rows, 0, 39060) or c["dp_max_rows"] == max(r for r in rows if r <= 65536)]}
        model = ShardedDLRM(criteo_tables(rows), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=sopt,
                            row_layout=args.row_layout, replicate_at_world1=args.replicate_small,
                            exchange=args.exchange, capacity_factor=args.capacity_factor,
                            dp_max_rows=dp_max_rows)
        parallelism = model.describe() + (f"; exchange: {args.exchange}" + (f" x{args.capacity_factor}" if args.exchange == "capacity" else ""))
    delta_tracker = None
    if args.delta_tracker:  # what train_config.delta_embedding_dump_config adds to a step (never part of the default line)
        from torcheasyrec_amd.delta_embedding_dump import ModelDeltaTracker

        delta_tracker = ModelDeltaTracker(model)
    use_graph = not sharded and not args.no_graph and not emu
    # capturable: the dense Adam step lives inside the captured hipGraph
    if args.torch_adam:
        dense_opt = torch.optim.Adam(list(model.dense_parameters()), lr=1e-3, fused=True, capturable=use_graph)
    else:  # same update, two launches for all ten tensors (torcheasyrec_amd/dense.py)
        from torcheasyrec_amd.dense import FusedDenseAdam

        # (one GPU: the bottom MLP's and the first top-MLP layer's gradients stay partial sums until the optimizer's own launch adds
        # them up -- dense.FUSE_FINISH; the sharded step packs its dense gradients for the all-reduce: finished tensors there)
        dense_opt = FusedDenseAdam(list(model.dense_parameters()), lr=1e-3, fuse_finish=not args.no_fuse_finish)

    # synthetic batches, resident in HBM before the timed region
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    nb = max(1, min(args.n_batches, args.warmup + args.steps))

    def make_batches(Bg, seed0=0):
        """nb batches of global size Bg; every rank keeps its contiguous slice."""
        Bl = Bg // world
        out, hv = [], []
        for s in range(nb):
            dense, kjt, label = synthetic_batch(seed0 + s, Bg, rows, dist=args.dist)
            if world > 1 or Bl != Bg:
                sl = slice(rank * Bl, (rank + 1) * Bl)
                v = kjt.values().view(len(rows), Bg)[:, sl].reshape(-1).contiguous()
                kjt = KeyedJaggedTensor(kjt.keys(), v, torch.ones(len(rows) * Bl, dtype=torch.int32), uniform_length=1)
                dense, label = dense[sl].contiguous(), label[sl].contiguous()
            hv.append(kjt.values().numpy())
            out.append((dense.to(dev), kjt.to(dev), label.to(dev)))
        return out, hv

    batches, host_vals = make_batches(B_global)
    sync()

    timers = _Timers()
    ebc = model.ebc if not sharded else None
    if ebc is not None:
        ebc.async_plan = args.async_plan

    train_step = None
    graph_factory = None
    if emu and sharded and args.step_graph:
        # plumbing runs on the CPU suite: the native driver over the emulator's recorded host functions and the shared-memory RCCL
        # stand-in (tests/emu), so that --forms ab and its fall-back are exercised without a GPU
        from emu.build_emu import RCCL_STUB
        from emu.graphs import EmuGraph

        if os.path.exists(RCCL_STUB):
            os.environ.setdefault("TZR_RCCL_PATH", RCCL_STUB)
            graph_factory = EmuGraph

    def make_train_step(native, input_dist_stream=None):
        from torcheasyrec_amd.sharded_step import ShardedTrainStep

        return ShardedTrainStep(model, dense_opt, use_graph=(not args.no_graph) or graph_factory is not None, prefetch=not args.no_prefetch,
                                plan_ahead=not args.no_plan_ahead, step_graph=args.step_graph,
                                graph_input_dist=args.step_graph and not args.no_graph_input_dist,
                                overlap_collectives={"auto": None, "on": True, "off": False}[args.overlap_collectives],
                                native_driver=native, graph_factory=graph_factory, input_dist_stream=input_dist_stream,
                                **({"warmup_iters": 0} if graph_factory is not None else {}))

    # (--forms ab: the run starts on the six-graph form; the one-graph form is built later, under its deadline)
    forms_ab = (sharded and not args.no_pipeline and args.step_graph and not args.no_native_driver
                and (args.forms == "ab" or (args.forms == "auto" and world > 1)))
    if sharded and not args.no_pipeline:
        train_step = make_train_step(False if (args.no_native_driver or args.forms == "overlapped" or forms_ab) else None)

    def step_body(dense, kjt, label, next_kjt=None):
        if train_step is not None:
            return train_step.step(dense, kjt, label, next_kjt=next_kjt)
        if not sharded and not args.torch_bce and not args.layerwise_loss:
            # the training forward of a one-label rank model = logits + loss (TrainWrapper.forward, tzrec/models/model.py:271-297):
            # the bottom MLP and the top MLP's tail + loss + their backward run as whole-stack kernels (csrc/mlp_ops.hip)
            loss, _ = model.forward_loss(dense, kjt, label)
        else:
            logits = model(dense, kjt)
            loss = bce_with_logits(logits, label)
        with root_loss():  # the unscaled loss is the root of the backward pass
            loss.backward(gradient=unit_gradient(loss))
        if sharded:
            model.allreduce_dense_grads()
        dense_opt.step()
        dense_opt.zero_grad(set_to_none=True)
        return loss

    # eager warm-up (also where TunableOp tunes unseen GEMM shapes).  Graph capture needs lazy
    # initialisation and tuning out of the way, so at least two eager steps run even for --warmup 0/1
    # (untimed, like the requested ones)
    loss = None
    # 3: the pipelined step captures on its 3rd call; 10: the two pipeline slots of --step-graph capture on their 3rd (step) and 4th (input dist) visits
    for i in range(max(args.warmup, (10 if args.step_graph else 3) if train_step is not None else (2 if use_graph else 0))):
        loss = step_body(*batches[i % nb])
    sync()

    graphs, replayed_graphs, settle_done = None, False, 0
    if use_graph:
        # One hipGraph per distinct batch (inputs already in HBM, nothing is copied per step); all
        # graphs share one memory pool since they never run concurrently.  A step = one replay:
        # forward, backward with the fused sparse optimizer, dense Adam -- nothing is skipped.
        graphs, pool, losses = [], None, []
        del loss  # drop the eager autograd graph before capturing
        for bi in range(nb):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=work_stream):
                losses.append(step_body(*batches[bi]))
            pool = g.pool()
            graphs.append(g)
        sync()
        # a hipGraph's FIRST launch uploads it to the device (~75 us each here: with the clock started on never-launched graphs the
        # timed region carried 8 uploads -- 0.6 ms, 6 % of a 20-step reading, profiles/r06ah); every graph is launched once, untimed,
        # like the warm-up steps before it (nb more training steps on the same batches)
        for g in graphs:
            g.replay()
        sync()
        # ... and the device is brought to its running state: capture leaves it idle for seconds, and its clocks come back over the next
        # ~10 ms of work -- the first 24 replays after an idle period ran 0.526, 0.510, ... 0.478 ms (profiles/r06ah), so a 20-step
        # reading right behind the capture measured the ramp, not the step.  --settle-steps more untimed replays (reported as `settle_steps`)
        for i in range(args.settle_steps):
            graphs[i % nb].replay()
        sync()
        settle_done = len(graphs) + args.settle_steps

    def run_step(i, last=False):
        if graphs is not None:
            graphs[i % nb].replay()
            return losses[i % nb]
        return step_body(*batches[i % nb], next_kjt=None if last else batches[(i + 1) % nb][1])

    def timed_block(n_steps, first=0):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; the MAX over ranks.  -> (elapsed, host seconds
        until the steps were queued, of which waiting for overflow words, last loss)"""
        if world > 1:
            dist.barrier()
            sync()
        fw0 = getattr(getattr(model, "ebc", None), "flag_wait_s", 0.0) if sharded else 0.0
        evs = [] if os.environ.get("TZR_BENCH_STEP_TIMES") and not emu else None  # (diagnostic: where inside the region the time goes)
        t0 = time.perf_counter()
        loss_ = None
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        for i in range(n_steps):
            loss_ = run_step(first + i)
            if evs is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
        host_el = time.perf_counter() - t0  # time until the HOST had queued the steps (sharded capacity exchange: includes its waits
        #                                      for the batches' overflow words -- `host_flag_wait_ms_per_step`, the host AHEAD of the device)
        host_fw = (getattr(getattr(model, "ebc", None), "flag_wait_s", 0.0) - fw0) if sharded else 0.0
        sync()
        if world > 1:
            dist.barrier()
            sync()
        el = time.perf_counter() - t0
        if evs is not None:
            d = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
            print("[step times, ms] host queued in %.3f, wall %.3f, device span %.3f: %s" % (
                host_el * 1e3, el * 1e3, evs[0].elapsed_time(evs[-1]), " ".join("%.3f" % x for x in d)), file=sys.stderr, flush=True)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, host_el, host_fw, loss_

    # ---- N > 1: which form of the sharded step?  (VERDICT r5 #1c)  The one-graph form -- the native driver, RCCL collectives captured
    # inline -- is the faster one on the 1-rank proxy and has never run on more than one rank from this side; the six-graph form
    # overlaps its collectives with the neighbouring graphs and uses torch.distributed only.  Both are warmed up and timed here,
    # the faster gives `value`; the one-graph form runs under a deadline and inside try / except: an RCCL or capture error, or a
    # step that does not come back, falls back to the six-graph measurement (already taken) and the line says which and why.
    sharded_forms = None
    official = None
    if forms_ab:
        import threading

        n_trial = max(args.form_trial_steps, args.steps)
        n_warm = max(args.warmup, 10)

        def measure(ts_):
            nonlocal train_step
            train_step = ts_
            for i in range(n_warm):
                run_step(i, last=i == n_warm - 1)
            sync()
            trial = timed_block(n_trial)
            return trial[0] / n_trial * 1e3, timed_block(args.steps, first=args.warmup)

        descr = {"overlapped": "six hipGraphs per step, torch.distributed (RCCL) collectives issued async between them; input dist on a side stream",
                 "one_graph": "ONE hipGraph per step with both all-to-alls and both all-reduces captured inline on the library's own "
                              "communicator, queued by tzr_step_run; the input dist (its own communicator) on the STEP's stream: the "
                              "two communicators are never concurrent",
                 "one_graph_side": "the same with the input dist on a side stream next to the step (its ~30 us hidden; the two "
                                   "communicators' collectives may be in flight together)"}
        ts_six = train_step if train_step.native_driver is False else make_train_step(False)
        six_ms, six_official = measure(ts_six)
        sharded_forms = {"overlapped_ms": six_ms, "one_graph_ms": None, "one_graph_side_ms": None, "picked": "overlapped",
                         "trial_steps": n_trial, "forms": descr}
        best = {"name": "overlapped", "ms": six_ms, "ts": ts_six, "official": six_official}

        def fallback_line(reason, form):
            el = best["official"][0]
            return {"metric": f"samples/sec DLRM-Criteo (examples/dlrm_criteo.config) training, batch {args.global_batch} "
                              + ("per GPU" if args.scaling == "weak" else "global"),
                    "value": B_global * args.steps / el, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "collectives": coll_lib,
                    "config": {"workload": f"dlrm_criteo: 26 tables x dim 16 (204.2M rows, fp32), fused sparse {args.optimizer} + dense Adam, "
                                           f"ids {args.dist}", "global_batch": B_global, "per_rank_batch": B_local, "parallelism": parallelism},
                    "sharded_forms": dict(sharded_forms, picked=best["name"], **{form + "_error": reason}),
                    "launch": f"form `{best['name']}` ({descr[best['name']]}); form `{form}` did not come back: this line was written by "
                              "the deadline, from the measurements taken before it"}

        state = {"form": None}

        def deadline():
            # a collective that never completes cannot be cancelled: every rank's own timer ends its process; rank 0 leaves the line
            if rank == 0:
                print(json.dumps(fallback_line(f"no answer within {args.form_timeout:.0f} s (--form-timeout)", state["form"])), flush=True)
            sys.stdout.flush()
            os._exit(0)

        # the one-graph forms, each under its own deadline and inside try / except.  (`one_graph_side` only on more than one rank --
        # or when asked for by --forms ab on one: there the two differ by the input dist's ~30 us and nothing else)
        candidates = [("one_graph", "main")] + ([("one_graph_side", "side")] if (world > 1 or args.forms == "ab") else [])
        for form, in_stream in candidates:
            ts_one, err, one_ms, one_official = None, None, None, None
            state["form"] = form
            timer = threading.Timer(args.form_timeout, deadline)
            timer.daemon = True
            timer.start()
            try:
                sim = os.environ.get("TZR_BENCH_SIMULATE", "")
                if sim == "hang" or sim == "hang:" + form:
                    time.sleep(10 * args.form_timeout)
                if sim == "raise" or sim == "raise:" + form:
                    raise RuntimeError("simulated RCCL failure (TZR_BENCH_SIMULATE=raise)")
                ts_one = make_train_step(True if graph_factory is not None else None, input_dist_stream=in_stream)
                if not ts_one._use_native_driver():
                    err = "the library cannot reach RCCL (native_step.available() is false)"
                else:
                    one_ms, one_official = measure(ts_one)
                    if ts_one.native_error is not None or not ts_one.native_steps:
                        err = ts_one.native_error or "no step went through the native driver"
            except Exception as e:  # noqa: BLE001 -- any failure of this form is an answer, not the end of the run
                err = f"{type(e).__name__}: {e}"
            # every rank takes the same decision: one failure anywhere gives the form up everywhere (still under the deadline: a
            # rank whose peers are stuck inside a collective never gets this answer)
            okf = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            timer.cancel()
            if float(okf.item()) == 0.0:
                sharded_forms[form + "_error"] = err or "failed on another rank"
                if err and not emu:
                    sync()
                if form == "one_graph":
                    break  # (the side-stream variant is the same machinery plus a hazard: not tried when the plain one failed)
                continue
            sharded_forms[form + "_ms"] = one_ms
            if one_ms <= best["ms"]:
                best = {"name": form, "ms": one_ms, "ts": ts_one, "official": one_official}
        sharded_forms["picked"] = best["name"]
        train_step, official = best["ts"], best["official"]

    elapsed, host_elapsed, host_flag_wait, loss = official if official is not None else timed_block(args.steps, first=args.warmup)
    final_loss = float(loss.item())

    # Secondary reading for N > 1: the OTHER scaling regime on the same model (BASELINE.md quotes the
    # headline both ways: 65 536 per rank = tzrec's per-rank batch_size, and 65 536 global).  Same
    # protocol (untimed warm-up, barrier + synchronize on both sides, max over ranks).
    secondary = None
    B2 = args.secondary_global_batch
    if B2 is None:  # strong headline -> weak at 8192 per rank; weak headline -> the strong reading
        B2 = (args.global_batch if args.scaling == "weak" else args.weak_per_rank_batch * world) if world > 1 else 0
        if emu or args.no_secondary:
            B2 = 0
    if B2 and train_step is not None and B2 != B_global and B2 % world == 0:
        b2, _ = make_batches(B2, seed0=1000)
        for i in range(max(args.warmup, 10 if args.step_graph else 4)):
            step_body(*b2[i % nb], next_kjt=b2[(i + 1) % nb][1])
        sync()
        if world > 1:
            dist.barrier()
            sync()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step_body(*b2[i % nb], next_kjt=b2[(i + 1) % nb][1])
        sync()
        if world > 1:
            dist.barrier()
            sync()
        e2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        secondary = {"scaling": "strong" if args.scaling == "weak" else "weak", "global_batch": B2, "per_rank_batch": B2 // world,
                     "value": B2 * args.steps / e2, "unit": "samples/s", "ms_per_step": e2 / args.steps * 1e3}

    # The same step through the reference's pipeline seam (pipeline.progress(iterator),
    # tzrec/main.py:523 -> utils/dist_util.py:221-303): batches start in PINNED HOST memory, the H2D
    # copy of batch i+1 runs on the copy stream under the step of batch i (372 B/sample = 24 MB per
    # 65536-sample step).  Eager launches (no hipGraph), so this is also the launch-bound reading.
    e2e = None
    if not sharded and not args.no_e2e and args.e2e_steps > 0:
        from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, GraphTrainPipeline, TrainPipeline
        from torcheasyrec_amd.sparse import KeyedTensor

        class _BatchModel(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, b):
                return self.m(b.dense_features[BASE_DATA_GROUP].values(), b.sparse_features[BASE_DATA_GROUP])

            def loss_and_predictions(self, b):  # what TrainWrapper.forward returns: losses + predictions in one call
                loss, logits = self.m.forward_loss(b.dense_features[BASE_DATA_GROUP].values(), b.sparse_features[BASE_DATA_GROUP],
                                                   b.labels["label"])
                return {"bce": loss}, logits

        host = []
        for s_ in range(nb):
            d_, k_, l_ = synthetic_batch(2000 + s_, B_local, rows, dist=args.dist)
            # ids cross PCIe as int32 (Batch.narrow_ids: every Criteo table has fewer than 2^31 rows) and are widened on the
            # device behind the copy: 10.7 instead of 17.5 MB per 65 536-sample step
            host.append(Batch({BASE_DATA_GROUP: KeyedTensor([f"int_{i}" for i in range(NUM_DENSE)], [1] * NUM_DENSE, d_)},
                              {BASE_DATA_GROUP: k_}, {"label": l_}).narrow_ids().pin_memory())
        n_e2e = args.e2e_steps
        loss_of = lambda pred, b: {"bce": bce_with_logits(pred, b.labels["label"])}  # noqa: E731
        # (the lengths of one-id-per-bag keys are constant: GraphTrainPipeline keeps them in its device slots, the eager
        # pipeline moves the whole Batch)
        from torcheasyrec_amd.embedding_group import _batch_tensors

        h2d_graph = sum(t.numel() * t.element_size() for t in _batch_tensors(host[0], skip_constant=True))
        h2d_eager = sum(t.numel() * t.element_size() for t in _batch_tensors(host[0], skip_constant=False))

        def timed(pipe, n_warm):
            it = iter([host[i % nb] for i in range(n_e2e + n_warm + 1)])
            for _ in range(n_warm):
                pipe.progress(it)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_e2e):
                pipe.progress(it)
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        # (a) hipGraph replay per device slot, the next batch's H2D under it (GraphTrainPipeline); (b) the eager
        # TrainPipeline (one Python-launched kernel sequence per step): the launch-bound reading
        # (graph pipeline timed before and after the eager one: the first pipeline of a process also pays for the first
        # touches of the pinned batches)
        # (the first pipeline of a process also pays for the first touches of the pinned batches: one untimed pass first)
        g_runs = []
        W_E2E = 8 + 100  # (the slots' captures + ~50 ms of steps: the device's clocks after the idle captures, as for the headline)
        if not args.torch_adam:
            timed(GraphTrainPipeline(_BatchModel(model), dense_opt, dev, loss_of), 8)
            g_runs.append(timed(GraphTrainPipeline(_BatchModel(model), dense_opt, dev, loss_of), W_E2E))
        e_eager = timed(TrainPipeline(_BatchModel(model), dense_opt, dev, loss_of), 5 + 100)
        if not args.torch_adam:
            for _ in range(2):
                g_runs.append(timed(GraphTrainPipeline(_BatchModel(model), dense_opt, dev, loss_of), W_E2E))
        # the reading is the MEDIAN of the graph pipeline's three timed runs (the product's default pipeline); the eager
        # pipeline is reported next to it
        # ... unless the eager pipeline is the faster one on this box (a graph launch costs ~20 us of device idle in front of
        # every step; eager launches queue without it as long as the host keeps ahead): then that is the reading, and says so
        g_med = sorted(g_runs)[len(g_runs) // 2] if g_runs else None
        use_graph_pipe = g_med is not None and g_med <= e_eager
        e1 = g_med if use_graph_pipe else e_eager
        e2e = {"value": B_local * n_e2e / e1, "unit": "samples/s", "ms_per_step": e1 / n_e2e * 1e3, "steps": n_e2e,
               "statistic": "median of 3 timed runs" if use_graph_pipe else "one timed run (the eager pipeline: faster than the graph pipeline's median here)",
               "h2d_bytes_per_step": h2d_graph if use_graph_pipe else h2d_eager, "id_wire_dtype": "int32 (widened on the device)",
               "launch": ("hipGraph replay per device slot, pinned host batches, H2D of the next batch on a copy stream "
                          "(GraphTrainPipeline.progress)" if use_graph_pipe else
                          "eager, TrainPipeline.progress, pinned host batches, H2D on a copy stream"),
               "eager_ms_per_step": e_eager / n_e2e * 1e3,
               "graph_ms_per_step_runs": [g / n_e2e * 1e3 for g in g_runs]}

    # per-kernel HIP-event timing: instrumented eager steps on the same batches (events cannot sit
    def embedding_stages(ebc_, kjts, host_values, Bl, optimizer, iters=10):
        """HIP events around the three C-ABI calls of the embedding path on the launching stream (a GPU-side sleep
        first, so the host is not in the gaps) -> stage times + the roofline fraction of SURVEY 8(d)'s bytes."""
        tm = _Timers()
        gbuf = torch.randn(Bl, 26 * 16, device=dev) * 1e-3
        for i in range(max(2, int(24e-3 / (2.5e-9 * max(Bl, 1)))) if not emu else 2):  # (~25 ms of the same work first: the device's clocks)
            _emb_calls(ebc_, kjts[i % len(kjts)], gbuf)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2.0e6))
        ebc_._timers = tm
        for i in range(iters):
            k = kjts[i % len(kjts)]
            _emb_calls(ebc_, k, gbuf)
        torch.cuda.synchronize()
        ebc_._timers = None
        ab = [algorithmic_bytes(hv, Bl, rows, optimizer=optimizer) for hv in host_values]
        nbytes = float(np.mean([a["fwd"] + a["bwd"] for a in ab]))
        stat = tm.mean_ms if iters <= 10 else tm.median_ms  # (long secondary runs: the median, so that a host gap late in the queue does not enter)
        f, p_, a_ = stat("fwd"), stat("plan") or 0.0, stat("apply")  # (no plan launch: tzr_pooled_bwd_direct)
        fp_ = stat("fwd+plan")  # the forward's launch carried the plan
        if fp_:
            f, p_ = fp_, 0.0
        return {"fwd_ms": f, "bwd_plan_ms": p_, "bwd_apply_ms": a_, "algorithmic_bytes": nbytes,
                "forward": "tzr_pooled_fwd_cells_plan (the backward's index plan in the forward's launch)" if fp_ else "tzr_pooled_fwd",
                "backward": "tzr_pooled_bwd_cells_apply (plan: see forward)" if fp_ else "tzr_pooled_bwd_direct (one launch: no index plan)" if not p_ else
                            ("tzr_pooled_bwd_cells_plan (1 launch) + tzr_pooled_bwd_cells_apply" if ebc_.backward_form(kjts[0], ("sparse",)) == "cells"
                             else "tzr_pooled_bwd_plan (4 launches) + tzr_pooled_bwd_apply"),
                "iterations": iters, "fwd_bwd_GBps": nbytes / ((f + p_ + a_) * 1e-3) / 1e9, "frac_of_8TBps": nbytes / ((f + p_ + a_) * 1e-3) / HBM_PEAK}

    # N = 1: the other readings BASELINE.json / the north star ask for, on the same box in the same process:
    # config 2 (examples/dlrm_criteo.config at its own batch_size 8192: whole step + embedding stages), the
    # Criteo-like Zipf ids, and the north star's optimizer (row-wise Adagrad; its own tables: 13 GB + 0.8 GB state)
    if rank == 0 and world == 1 and not sharded and not args.no_secondary and args.optimizer == "adagrad" \
            and args.dist == "uniform" and not args.rows_cap and B_global == 65536:
        secondary = {}
        # (a) config 2: batch 8192 on one GPU, hipGraph replay like the headline
        B2 = 8192
        b2 = []
        h2 = []
        for s_ in range(4):
            d_, k_, l_ = synthetic_batch(2000 + s_, B2, rows)
            h2.append(k_.values().numpy())
            b2.append((d_.to(dev), k_.to(dev), l_.to(dev)))
        for i in range(3):
            step_body(*b2[i % 4])
        torch.cuda.synchronize()
        g2, pool2 = [], None
        for bi in range(4):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, pool=pool2, stream=work_stream):
                step_body(*b2[bi])
            pool2 = g_.pool()
            g2.append(g_)
        torch.cuda.synchronize()
        for i in range(4 + 150):  # (every graph launched once + ~20 ms of replays: the device's clocks after the idle capture)
            g2[i % 4].replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n2 = max(args.steps, SECONDARY_REPLAYS)
        for i in range(n2):
            g2[i % 4].replay()
        torch.cuda.synchronize()
        e2_ = time.perf_counter() - t1
        secondary["config2_batch8192"] = {"config": "examples/dlrm_criteo.config, 1 GPU, batch 8192 (BASELINE.json configs[1])",
                                          "value": B2 * n2 / e2_, "unit": "samples/s", "ms_per_step": e2_ / n2 * 1e3, "replays": n2,
                                          "embedding": embedding_stages(ebc, [b[1] for b in b2], h2, B2, "adagrad", iters=SECONDARY_REPLAYS)}
        del g2
        # (b) Zipf(1.05) ids at the headline batch: embedding stages
        bz, hz = [], []
        for s_ in range(4):
            _, k_, _ = synthetic_batch(3000 + s_, B_global, rows, dist="zipf")
            hz.append(k_.values().numpy())
            bz.append(k_.to(dev))
        secondary["zipf_ids_batch65536"] = {"config": "ids Zipf(1.05) clipped to the table (SURVEY 8d secondary distribution)",
                                            "embedding": embedding_stages(ebc, bz, hz, B_global, "adagrad", iters=SECONDARY_REPLAYS)}
        ebc.reset_plan_mode()  # (Zipf ids sent the collection to the exact index plan for good; the headline's ids are evenly drawn)
        del bz
        # (c) row-wise Adagrad, the optimizer the north star names: its own collection (weights [rows, 16] + [rows] state)
        try:
            from torcheasyrec_amd.embedding import EmbeddingBagCollection

            ebc_rw = EmbeddingBagCollection(criteo_tables(rows), device=dev, optimizer=SparseOptimizerConfig(kind="rowwise_adagrad", lr=1e-3),
                                            groups={"sparse": SPARSE_KEYS})
            secondary["rowwise_adagrad_batch65536"] = {
                "config": "fused row-wise Adagrad (protos/optimizer.proto:133-139), uniform ids",
                "embedding": embedding_stages(ebc_rw, [b[1] for b in batches[:4]], host_vals[:4], B_global, "rowwise_adagrad", iters=SECONDARY_REPLAYS)}
            del ebc_rw
        except Exception as e:  # e.g. not enough free HBM next to the main model
            secondary["rowwise_adagrad_batch65536"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # (d) the MFMA-bound kernels of the step: dot interaction fused with the first top-MLP layer, timed alone with
        # HIP events; flops = the ALGORITHMIC ones (layer 2 x 783 x 64 per sample; pairwise products / dX = (G + G^T) X),
        # against the dense fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
        try:
            secondary["interaction_first_layer_mfma"] = interaction_top_rooflines(dev, B_global)
        except Exception as e:
            secondary["interaction_first_layer_mfma"] = {"error": repr(e)[:200]}

    # (f) the other model families of BASELINE.json, built from their configs: DeepFM-Criteo, multi_tower_din (configs[3]),
    # MMoE + zero-collision hash (configs[4]) -- a train step each at batch 8192
    if isinstance(secondary, dict) and "config2_batch8192" in secondary and not args.no_config_models:
        replayed_graphs = graphs is not None
        del graphs  # (their memory pool goes back to the allocator: the 200 M-row ZCH table needs room)
        graphs = None
        torch.cuda.empty_cache()
        secondary.update(config_model_steps(dev, work_stream, steps=max(args.steps, 20)))

    # (e) the sharded proxy measured at the start of this process (see sharded_proxy): ratios against THIS run's N = 1 step
    if isinstance(secondary, dict) and proxy_child is not None:
        if "projection" in proxy_child:
            pj = proxy_child["projection"]
            n1_ms = elapsed / args.steps * 1e3
            n1 = B_global / (n1_ms * 1e-3)
            pj["n1_samples_per_s"] = n1
            pj["scaling_vs_n1"] = {k2: pj[k1] / n1 for k1, k2 in (("samples_per_s_if_wire_hidden", "wire_hidden"),
                                                                   ("samples_per_s_if_a2a_exposed", "a2a_exposed"),
                                                                   ("samples_per_s_if_wire_exposed", "wire_exposed"))}
            pj["step_ms_needed_for_6x"] = n1_ms / 6.0
        secondary["sharded_w1_proxy_b8192"] = proxy_child

    # inside a captured graph); the kernels and inputs are the ones of the timed region
    if ebc is not None and not emu:
        # The three C-ABI calls of the embedding path (pooled forward, backward plan, backward
        # apply) are launched on the batches of the timed region with HIP events recorded right
        # before/after each call on the launching stream.  A GPU-side sleep first lets the host
        # queue everything ahead, so the events bracket the launches only (no host gaps).
        ebc.async_plan = False
        gbuf = torch.randn(B_local, 26 * 16, device=dev) * 1e-3
        # (the same three calls untimed first: the device has been idle under the secondary readings / their child processes, and its
        # clocks come back over ~10 ms of work -- ten timed iterations right away read 5 % long, profiles/r06ah)
        for i in range(150):
            kjt_i = batches[i % nb][1]
            _emb_calls(ebc, kjt_i, gbuf)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2.0e6))
        ebc._timers = timers
        for i in range(max(min(args.steps, 40), 20)):
            kjt_i = batches[i % nb][1]
            _emb_calls(ebc, kjt_i, gbuf)
        torch.cuda.synchronize()
        ebc._timers = None
    ms_per_step = elapsed / args.steps * 1e3
    value = B_global * args.steps / elapsed

    if graphs is not None or replayed_graphs:
        launch_desc = "hipGraph replay"
    elif train_step is None:
        launch_desc = "eager"
    elif not args.step_graph:
        launch_desc = "pipelined: input dist one batch ahead + hipGraph dense segment"
    elif getattr(train_step, "native_steps", 0):
        launch_desc = ("pipelined, native step driver: input dist one batch ahead as ONE hipGraph (ids all-to-all inside), the rest of the step ONE "
                       "hipGraph (both all-to-alls and both all-reduces inside, on the library's own RCCL communicator), queued by tzr_step_run")
    else:
        launch_desc = (f"pipelined: input dist one batch ahead + {'six' if getattr(train_step, 'overlap_collectives', False) else 'three'} hipGraphs "
                       "for the rest of the step, RCCL calls between them (async, waited for on the stream)")
    out = {
        "metric": f"samples/sec DLRM-Criteo (examples/dlrm_criteo.config) training, batch {args.global_batch} "
                  + ("per GPU" if args.scaling == "weak" else "global"),
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "host_queue_ms_per_step": host_elapsed / args.steps * 1e3,
        **({"settle_steps": settle_done,
            "settle_note": "untimed replays between the capture and the timed steps: every hipGraph launched once (its first launch "
                           "uploads it) + --settle-steps more (the device's clocks after the idle seconds of a capture)"}
           if settle_done else {}),
        **({"host_flag_wait_ms_per_step": host_flag_wait / args.steps * 1e3,
            "host_busy_ms_per_step": (host_elapsed - host_flag_wait) / args.steps * 1e3} if sharded else {}),
        "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if not emu else "synthetic -- CPU LANE EMULATOR + gloo, tables capped: launch-path plumbing check, not a measurement",
        "ranks_seen": ranks_seen, "collectives": coll_lib,
        "config": {"workload": "dlrm_criteo: 26 tables x dim 16 (204.2M rows, fp32), fused sparse "
                               f"{args.optimizer} + dense Adam, ids {args.dist}",
                   "global_batch": B_global, "per_rank_batch": B_local, "parallelism": parallelism,
                   "row_layout": args.row_layout, "rows_cap": args.rows_cap or None},
        "final_loss": final_loss,
        "secondary": secondary,
        **({"delta_tracker": True} if delta_tracker is not None else {}),
        **({"sharded_forms": sharded_forms} if sharded_forms is not None else {}),
        "launch": launch_desc,
    }
    if sharded:
        out["exchange"] = dict(model.ebc.exchange_stats, kind=args.exchange)
        if train_step is not None and args.step_graph:
            out["exchange"].update(graph_steps=train_step.graph_steps, eager_steps=train_step.eager_steps,
                                   native_driver_steps=train_step.native_steps)
        # the scaling arithmetic (what this step time means for the >= 6x target): at N = 1 the run is the proxy of one
        # rank of a `65536 / B_local`-rank job, at N > 1 the measured job itself
        Wp = args.projection_world or (world if world > 1 else max(2, 65536 // max(B_local, 1)))
        out["projection"] = xgmi_projection(model, B_local, Wp, ms_per_step, args.n1_ms or None, args.capacity_factor,
                                            input_dist_on_main=(train_step._input_dist_on_main() if world > 1 else True)
                                            if train_step is not None and args.step_graph else False)
        out["projection"]["dp_max_rows_choice"] = dp_choice
        out["projection"]["measured_on"] = (f"{world} rank(s); " + ("every collective is a self copy: wire time NOT in proxy_ms_per_step"
                                                                    if world == 1 else "wire time included in proxy_ms_per_step"))
    if rank == 0 and world == 1:
        ab = [algorithmic_bytes(hv, B_local, rows, optimizer=args.optimizer) for hv in host_vals]
        fwd_b = float(np.mean([a["fwd"] for a in ab]))
        bwd_b = float(np.mean([a["bwd"] for a in ab]))
        t_fwd, t_plan, t_apply = timers.mean_ms("fwd"), timers.mean_ms("plan") or 0.0, timers.mean_ms("apply")
        t_fp = timers.mean_ms("fwd+plan")  # the forward's launch carried the backward's index plan (tzr_pooled_fwd_cells_plan)
        if t_fp:
            t_fwd, t_plan = t_fp, 0.0
        if t_fwd and t_apply:
            # north star: "HBM-bandwidth roofline on the pooled embedding forward+backward".  Bytes are
            # SURVEY.md 8(d)'s algorithmic figures for THESE batches; the plan moves no algorithmic
            # bytes (its time counts against the aggregate in full).
            tot = (t_fwd + t_plan + t_apply) * 1e-3
            ach = (fwd_b + bwd_b) / tot
            traffic, traffic_src = pmc_traffic(args, B_local)

            def stage(name, kernels, nbytes, ms):
                return {"stage": name, "kernels": kernels, "launch_ms": ms, "algorithmic_bytes": nbytes,
                        "GBps": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / HBM_PEAK}

            direct = not t_plan and not t_fp  # no plan launch: the one-launch backward of small batches (tzr_pooled_bwd_direct)
            cells = (not direct) and model.ebc.backward_form(batches[0][1], ("sparse",)) == "cells"  # the one-launch index plan (round 6)
            kind_k = {"adagrad": "adagrad", "rowwise_adagrad": "rowwise", "sgd": "sgd"}.get(args.optimizer)
            apply_k = f"tzr_bwd_reduce_fast_{kind_k}_kernel" if kind_k else "tzr_bwd_reduce_kernel"   # (the optimizer kind is a template parameter)
            cells_k = f"tzr_bwd_cells_apply_{kind_k}_kernel" if kind_k else "tzr_bwd_cells_apply_adam_kernel"
            direct_k = f"tzr_bwd_direct_{kind_k}_kernel" if kind_k else "tzr_bwd_direct_adam_kernel"
            fwd_k = "tzr_pooled_fwd_u1_kernel" if B_local >= 32768 else "tzr_pooled_fwd_kernel"
            if t_fp:
                fwd_k = "tzr_pooled_fwd_u1_cells_plan_kernel"
            stages = [stage("forward" + (" + backward plan (one launch: the plan's workgroups -- every chunk of lookups ordered by bucket in "
                                         "place -- between the forward's)" if t_fp else ""), [fwd_k], fwd_b, t_fwd)]
            if direct:
                stages.append(stage("backward (index sort + fused optimizer, one launch)", [direct_k], bwd_b, t_apply))
            elif t_fp:
                stages.append(stage("backward apply (units gather their cells, sort them in LDS, reduce, update)", [cells_k], bwd_b, t_apply))
            elif False:
                stages.append(stage("backward (index sort + fused optimizer, one launch)", [direct_k], bwd_b, t_apply))
            elif cells:
                stages += [{"stage": "backward plan (one launch: every chunk of lookups ordered by bucket in place)",
                            "kernels": ["tzr_bwd_cells_partition_kernel"], "launch_ms": t_plan, "algorithmic_bytes": 0.0, "GBps": 0.0, "frac": 0.0},
                           stage("backward apply (units gather their cells, sort them in LDS, reduce, update)", [cells_k], bwd_b, t_apply)]
            else:
                stages += [{"stage": "backward plan", "kernels": ["tzr_bwd_hist_kernel", "tzr_bwd_scan_kernel", "tzr_bwd_scatter_kernel",
                                                                 "tzr_bwd_sort_kernel"], "launch_ms": t_plan, "algorithmic_bytes": 0.0,
                            "GBps": 0.0, "frac": 0.0},
                           stage("backward apply (+ the LDS sort of every unit of a table without heavy buckets)", [apply_k], bwd_b, t_apply)]
            out["roofline"] = {
                "bound": "hbm", "kernel": ("pooled embedding forward + backward (" + ("2 launches: " + fwd_k + "; " + direct_k if direct else
                                           "2 launches: " + fwd_k + "; " + cells_k if t_fp else
                                           ("3 launches: " + fwd_k + "; tzr_bwd_cells_partition_kernel; " + cells_k if cells else
                                            "6 launches: " + fwd_k + "; tzr_bwd_hist/scan/scatter/sort_kernel; " + apply_k)) + ")"),
                "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                "traffic": traffic, "launch_ms": t_fwd + t_plan + t_apply,
                "algorithmic_bytes": fwd_b + bwd_b,
                "unique_rows": float(np.mean([a["U"] for a in ab])),
                "kernels": stages,
                "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command on this library "
                                   f"(digest-matched), summed over the launches named in `kernel`: {traffic_src}") if traffic is not None
                                  else f"null: {traffic_src}"}
            # kept for continuity with round 1's line
            out["embedding"] = {"fwd_ms": t_fwd, "bwd_plan_ms": t_plan, "bwd_apply_ms": t_apply,
                                "fwd_bwd_GBps": ach / 1e9, "frac_of_8TBps": ach / HBM_PEAK,
                                # (a builder-side reference point, not the contract's peak: the part's random 64-byte gather probe, profiles/r01c)
                                "random_64B_gather_probe_GBps": 3970.0, "frac_of_that_probe": ach / 3.97e12}
        if e2e is not None:
            out["e2e"] = e2e
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    # RCCL prints its version banner through C stdio, which a pipe buffers until exit: every rank
    # flushes it now, then a barrier, so rank 0's JSON line is the LAST line of the job's stdout
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if sharded:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded:
        # regular teardown.  (Round 3 left through os._exit here: the process used to abort now and then inside
        # destroy_process_group next to captured step graphs -- the watchdog's exception of sharded_step._quiesce_process_group's
        # docstring, gone since every collective runs on RCCL's own stream: profiles/r04c, 560 captures + a normal destroy.)
        sys.stdout.flush()
        sys.stderr.flush()
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
