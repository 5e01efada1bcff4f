#!/usr/bin/env python3
"""tzr_dot_interaction_top_fwd (no z) / _bwd / _wgrad alone, a few launches each at the DLRM-Criteo shape (for rocprofv3 counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402

_lib.use_library(_build.build())
L = _lib.lib()
dev = torch.device("cuda", 0)
D, F, H, B = 16, 26, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = F + 1
width = n * (n - 1) // 2 + D * n
st = _lib.stream_ptr(dev)
dense, sparse, g1 = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev), torch.randn(B, H, device=dev)
W1, b1 = torch.randn(H, width, device=dev) * 0.05, torch.randn(H, device=dev)
y1 = torch.empty(B, H, device=dev)
gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), dev)
dW = torch.empty(H, width, device=dev)
for _ in range(5):
    L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1, None, 0,
                                  _lib.ptr(y1), H, st)
    L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width, None,
                                  _lib.ptr(gd), D, _lib.ptr(gs), F * D, st)
    L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, None, _lib.ptr(dW), width,
                                    _lib.ptr(ws), ws.numel(), st)
torch.cuda.synchronize()
print("done")
