"""Build tests/emu/_build/libtzrec_emu.so: the UNCHANGED torcheasyrec_amd/csrc/*.hip kernels compiled
for the host CPU against the lane emulator (tests/emu/hip/hip_runtime.h).  Test infrastructure
only -- lets the kernel logic be checked against the oracle in a container with no GPU."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "torcheasyrec_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libtzrec_emu.so")
RCCL_STUB = os.path.join(OUT_DIR, "librccl_stub.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _sources():
    srcs = sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and f != "abi.hip"  # (the version symbol: abi_emu.cpp's own)
    )
    return srcs + [os.path.join(HERE, "abi_emu.cpp")]


def _digest():
    h = hashlib.sha256()
    deps = _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "tzrec_hip.h"),
        os.path.join(HERE, "hip", "hip_runtime.h"),
        os.path.join(HERE, "tzr_gfx950.h"),
        os.path.join(HERE, "rccl", "rccl.h"),
        os.path.join(HERE, "rccl_stub.cpp"),
    ]
    for p in deps:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _object_digest(src):
    h = hashlib.sha256()
    deps = [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "tzrec_hip.h"),
        os.path.join(HERE, "hip", "hip_runtime.h"),
        os.path.join(HERE, "tzr_gfx950.h"),
        os.path.join(HERE, "rccl", "rccl.h"),
    ]
    for p in deps:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OUT_DIR, exist_ok=True)
    stamp = os.path.join(OUT_DIR, "stamp")
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    cc = CLANG if os.path.exists(CLANG) else "clang++"
    objs, todo = [], []
    for src in _sources():
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        od = _object_digest(src)
        st = obj + ".stamp"
        if force or not (os.path.exists(obj) and os.path.exists(st) and open(st).read() == od):
            todo.append((src, obj, st, od))

    def one(job):
        src, obj, st, od = job
        cmd = [cc, "-x", "c++", "-std=c++20", "-O1", "-g", "-fPIC", "-pthread", "-Wno-unused-value",
               "-I", HERE, "-I", CSRC, "-c", src, "-o", obj]
        subprocess.check_call(cmd)
        with open(st, "w") as f:
            f.write(od)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(one, todo))
    subprocess.check_call([cc, "-shared", "-pthread", "-o", OUT + ".tmp"] + objs + ["-ldl"])
    os.replace(OUT + ".tmp", OUT)
    # the RCCL stand-in of the CPU suite (shared memory between the ranks' processes): a library of its own, reached by path
    subprocess.check_call([cc, "-x", "c++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-I", HERE,
                           os.path.join(HERE, "rccl_stub.cpp"), "-o", RCCL_STUB + ".tmp", "-lrt"])
    os.replace(RCCL_STUB + ".tmp", RCCL_STUB)
    with open(stamp, "w") as f:
        f.write(dig)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
