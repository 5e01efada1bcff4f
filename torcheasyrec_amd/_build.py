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


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(_STAMP) and open(_STAMP).read() == dig:
        return OUT
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):
            return OUT  # GPU box without a toolchain: use the prebuilt library that travelled with the repo
        raise RuntimeError("hipcc not found and no prebuilt libtzrec_hip.so")
    cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-shared", *sources(), "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(_STAMP, "w") as f:
        f.write(dig)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
