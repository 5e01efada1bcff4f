"""`python bench.py --gpus N` -- the form the driver uses -- must start the N ranks itself.

CPU plumbing run of the launch path: kernels through the lane emulator, gloo instead of RCCL, tables capped.  What is
checked is the launcher and the line's contract (one JSON line from rank 0, `n_gpus`, `ranks_seen` = an all-reduce of
ones over the group, the `projection` object of a sharded line), not a number."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TZR_BENCH_SPAWNED"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, env=env,
                       cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + "\n---- stderr ----\n" + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line, from rank 0
    assert p.stdout.strip().splitlines()[-1] == lines[0]  # ... and it is the last line of the job's stdout
    return json.loads(lines[0])


def test_bench_gpus2_spawns_its_ranks(emu_path):
    d = _run("--gpus", "2", "--emulator", "--global-batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 64 and d["config"]["per_rank_batch"] == 32
    assert "row-wise" in d["config"]["parallelism"] and d["exchange"]["kind"] == "capacity"
    assert d["exchange"]["graph_steps"] > 0 and d["exchange"]["overflow_retries"] == 0  # the whole-step (slot) path ran
    assert "EMULATOR" in d["data"]  # never mistaken for a measurement
    pr = d["projection"]
    assert pr["world"] == 2 and pr["per_rank_batch"] == 32 and pr["wire_total_us"] > 0
    assert pr["samples_per_s_if_wire_exposed"] <= pr["samples_per_s_if_a2a_exposed"] <= pr["samples_per_s_if_wire_hidden"]


def test_bench_gpus2_times_both_forms_of_the_sharded_step(emu_path):
    """`--gpus 2` (forms auto = ab on more than one rank): the six-graph form and the native driver's one-graph form (here: the
    unchanged step driver over the emulator's recorded graphs and the shared-memory RCCL stand-in) are both warmed up and timed,
    the line carries `sharded_forms`, the picked form gave `value`, and the projection prices a latency per collective."""
    d = _run("--gpus", "2", "--emulator", "--global-batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--form-trial-steps", "6")
    sf = d["sharded_forms"]
    assert sf["overlapped_ms"] > 0 and sf["one_graph_ms"] and sf["one_graph_ms"] > 0 and "one_graph_error" not in sf, sf
    assert sf["one_graph_side_ms"] and sf["one_graph_side_ms"] > 0 and "one_graph_side_error" not in sf, sf
    times = {"overlapped": sf["overlapped_ms"], "one_graph": sf["one_graph_ms"], "one_graph_side": sf["one_graph_side_ms"]}
    assert times[sf["picked"]] == min(times.values()) and sf["trial_steps"] == 6
    assert (d["exchange"]["native_driver_steps"] > 0) == (sf["picked"] != "overlapped")
    pr = d["projection"]
    n_coll = 4 + (sf["picked"] == "one_graph")  # (the one-graph form's input dist runs on the step's stream: its all-to-all counts too)
    assert pr["collective_latency_us"] == 20.0 and pr["collectives_in_stream_order"] == n_coll and pr["latency_total_us"] == 20.0 * n_coll
    wl = pr["samples_per_s_with_latency"]
    assert wl["wire_exposed"] < pr["samples_per_s_if_wire_exposed"] and wl["wire_hidden"] < pr["samples_per_s_if_wire_hidden"]


def test_bench_falls_back_when_the_one_graph_form_fails_or_hangs(emu_path):
    """the one-graph form raising on every rank -> the six-graph form gives the line and `one_graph_error` says why; the
    one-graph form never coming back -> every rank's deadline ends its process, rank 0 leaves the line of the six-graph
    measurement taken before (exit code 0, one JSON line)."""
    base = ("--gpus", "2", "--emulator", "--global-batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--form-trial-steps", "4")
    os.environ["TZR_BENCH_SIMULATE"] = "raise"
    try:
        d = _run(*base)
    finally:
        del os.environ["TZR_BENCH_SIMULATE"]
    sf = d["sharded_forms"]
    assert sf["picked"] == "overlapped" and sf["one_graph_ms"] is None and "simulated RCCL failure" in sf["one_graph_error"]
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["exchange"]["native_driver_steps"] == 0
    os.environ["TZR_BENCH_SIMULATE"] = "hang"
    try:
        d = _run(*base, "--form-timeout", "5")
    finally:
        del os.environ["TZR_BENCH_SIMULATE"]
    sf = d["sharded_forms"]
    assert sf["picked"] == "overlapped" and "no answer within 5 s" in sf["one_graph_error"]
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["steps"] == 2 and d["ms_per_step"] > 0
    # only the side-stream variant hangs: the line is the best of the two forms measured before it
    os.environ["TZR_BENCH_SIMULATE"] = "hang:one_graph_side"
    try:
        d = _run(*base, "--form-timeout", "20")
    finally:
        del os.environ["TZR_BENCH_SIMULATE"]
    sf = d["sharded_forms"]
    assert sf["picked"] in ("overlapped", "one_graph") and sf["one_graph_ms"] > 0 and "no answer within 20 s" in sf["one_graph_side_error"]
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_refuses_a_world_it_was_not_launched_with(emu_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulator"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
