"""pytest wiring.

* ``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol checks, and the kernel
  *logic* run through the CPU lane emulator build of the unchanged .hip sources (tests/emu).
* ``-m gpu``: the parity tests proper -- same test bodies, real gfx950 library, HIP tensors.

The ``dev`` fixture selects the library: param "emu" -> tests/emu/_build/libtzrec_emu.so + cpu
tensors; param "hip" (marked gpu) -> torcheasyrec_amd/libtzrec_hip.so + cuda tensors.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def emu_path():
    from emu.build_emu import build

    return build()


@pytest.fixture(autouse=True)
def _default_knobs():
    """tzr_tune knobs are process-wide (ShardedTrainStep sets bwd_one_wg_heavy for its side-stream plans): every
    test starts from the defaults of whichever library is loaded."""
    yield
    from torcheasyrec_amd import _lib

    if _lib._lib is not None:
        for name in (b"bwd_apply_waves", b"bwd_apply_fast", b"bwd_no_fuse_sort", b"bwd_scan_slices", b"fwd_tile_b", b"fwd_variant", b"bwd_debug", b"bwd_ch", b"bwd_direct_ch", b"bwd_direct", b"bwd_direct_debug", b"bwd_force_prep", b"linear_bwd_wg", b"bwd_one_wg_heavy", b"ia_bwd_plain", b"ia_bwd_wgs", b"ia_fwd_wgs", b"ia_gen_wgs", b"it_wgs", b"it_stagger", b"mlp_mfma", b"wg_debug", b"it_fwd_stagger"):
            _lib.lib().tzr_tune(name, 0)
        _lib.apply_env_tune()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    from torcheasyrec_amd import _lib

    if request.param == "emu":
        _lib.use_library(request.getfixturevalue("emu_path"))
        assert _lib.backend() == "emu"
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no HIP device is visible")
    _lib.use_native()
    assert _lib.backend().startswith("hip"), "GPU tests must run on the native gfx950 library"
    return torch.device("cuda", 0)


def emu_heavy(dev=None):
    """Historical: marked test variants that were slow on the first lane emulator (one OS thread per lane).  The
    fiber emulator (tests/emu/hip/hip_runtime.h) runs the whole CPU suite in a few minutes: nothing is skipped."""
    return


@pytest.fixture(params=["planned", "cells", "direct"])
def bwd_path(request, dev, monkeypatch):
    """The fused backward in each of its forms on the same test body: "planned" = tzr_pooled_bwd_plan + _apply (four-launch
    index plan, any ids), "cells" = tzr_pooled_bwd_cells_plan + _apply (one-launch plan; batches of one id per bag -- others
    still take the exact plan), "direct" = tzr_pooled_bwd_direct (one launch, small batches) wherever the library takes the
    shape (ragged pooled bags still go through the plan).  By default the library picks by size and the collection by id
    statistics; tests are small, so without this fixture they would only ever see the direct kernel."""
    from torcheasyrec_amd import _lib

    L = _lib.lib()
    assert L.tzr_tune(b"bwd_direct", 1 if request.param == "direct" else -1) == 0
    monkeypatch.setenv("TZR_BWD_PLAN", "cells" if request.param == "cells" else "exact")  # (read by EmbeddingBagCollection.__init__)
    calls = {"direct": 0, "cells": 0, "exact": 0, "path": request.param}
    orig = {n: getattr(L, n) for n in ("tzr_pooled_bwd_direct", "tzr_pooled_bwd_cells_apply", "tzr_pooled_bwd_apply")}

    def counted(name, key):
        def f(*a):
            calls[key] += 1
            return orig[name](*a)
        return f

    L.tzr_pooled_bwd_direct = counted("tzr_pooled_bwd_direct", "direct")
    L.tzr_pooled_bwd_cells_apply = counted("tzr_pooled_bwd_cells_apply", "cells")
    L.tzr_pooled_bwd_apply = counted("tzr_pooled_bwd_apply", "exact")
    try:
        yield calls
    finally:
        for n, f in orig.items():
            setattr(L, n, f)
        L.tzr_tune(b"bwd_direct", 0)
    if request.param == "planned":
        assert calls["direct"] == 0 and calls["cells"] == 0
    if request.param == "cells":
        assert calls["direct"] == 0
