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


def test_bench_refuses_a_world_it_was_not_launched_with(emu_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulator"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
