"""Compile torcheasyrec_amd/csrc/*.hip into libtzrec_hip.so for gfx950 (in-tree, hipcc)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libtzrec_hip.so")
_STAMP = os.path.join(_HERE, ".libtzrec_hip.stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-I", CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    deps = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(_HERE), "include", "tzrec_hip.h")]
    for p in deps:
        with open(p, "rb") as f:
            h.update(f.read())
    # (the flags without the absolute include path: the same sources must give the same digest wherever the tree lies --
    # the GPU box runs from a scratch copy, and bench.py matches profiles/*/pmc_traffic.json by this digest)
    h.update(" ".join(f for f in FLAGS if f != CSRC).encode())
    return h.hexdigest()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(_HERE), "include", "tzrec_hip.h")]


def _object_digest(src: str, extra=()) -> str:
    """what one object depends on: its source, every header (no per-file dependency scan: the headers are few), the flags"""
    h = hashlib.sha256()
    for p in [src] + _headers():
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join([f for f in FLAGS if f != CSRC] + list(extra)).encode())
    return h.hexdigest()


def build_objects(hipcc: str, out_dir: str, extra=(), verbose: bool = False, jobs: int = 0):
    """one object per source, recompiled only when its digest changed, `jobs` compilers at a time; returns the object paths.
    (No -fgpu-rdc: every object carries its own code object, no device symbol crosses a file.)"""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(out_dir, exist_ok=True)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        dig = _object_digest(src, extra)
        st = obj + ".stamp"
        if not (os.path.exists(obj) and os.path.exists(st) and open(st).read() == dig):
            todo.append((src, obj, st, dig))

    def one(job):
        src, obj, st, dig = job
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(st, "w") as f:
            f.write(dig)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(one, todo))
    return objs


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(_STAMP) and open(_STAMP).read() == dig:
        return OUT
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):
            return OUT  # GPU box without a toolchain: use the prebuilt library that travelled with the repo
        raise RuntimeError("hipcc not found and no prebuilt libtzrec_hip.so")
    obj_dir = os.path.join(_HERE, "_obj")
    if force:
        shutil.rmtree(obj_dir, ignore_errors=True)
    objs = build_objects(hipcc, obj_dir, verbose=verbose)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)  # (never a half-written library under a process that has it mapped)
    with open(_STAMP, "w") as f:
        f.write(dig)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
