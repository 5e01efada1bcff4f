#!/usr/bin/env python3
"""A variant build of the library for same-box A/B timing: python scripts/r06/build_variant.py NAME [-DFLAG ...]
-> torcheasyrec_amd/libtzrec_hip_NAME.so (travels to the GPU box; `scripts/emb_ab.py --lib` loads it)."""
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _build  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
objs = _build.build_objects(hipcc, os.path.join(os.path.dirname(_build.OUT), "_obj_" + name), extra=extra)
out = os.path.join(os.path.dirname(_build.OUT), f"libtzrec_hip_{name}.so")
subprocess.check_call([hipcc, f"--offload-arch={_build.ARCH}", "-fPIC", "-shared", *objs, "-o", out])
print(out)
