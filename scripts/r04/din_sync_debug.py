#!/usr/bin/env python3
"""Where the multi_tower_din train step still synchronises with the device (torch.cuda.set_sync_debug_mode), then a capture attempt."""
import os
import sys
import traceback
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from torcheasyrec_amd import example_configs as ec  # noqa: E402
from torcheasyrec_amd.config import load_pipeline_spec  # noqa: E402
from torcheasyrec_amd.dense import FusedDenseAdam  # noqa: E402
from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, _backward_of_losses, _losses_and_predictions  # noqa: E402
from torcheasyrec_amd.rank_model import build_rank_model  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor  # noqa: E402

bench.enable_tunable_gemm()
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
spec = load_pipeline_spec(ec.multi_tower_din_taobao(batch_size=int(sys.argv[1]) if len(sys.argv) > 1 else 2048))
B = spec.batch_size
torch.manual_seed(7)
model = build_rank_model(spec, device=dev)
opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr)
rng = np.random.default_rng(3)
sparse = [f for f in spec.features if f.is_sparse]
dense = [f for f in spec.features if not f.is_sparse]


def batch():
    vals, lens = [], []
    for f in sparse:
        ln = rng.integers(0, f.sequence_length + 1, size=B).astype(np.int32) if f.is_sequence else np.ones(B, np.int32)
        lens.append(ln)
    seq = [i for i, f in enumerate(sparse) if f.is_sequence]
    for i in seq[1:]:
        lens[i] = lens[seq[0]]
    for f, ln in zip(sparse, lens):
        vals.append(rng.integers(0, f.num_embeddings, size=int(ln.sum())))
    kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(np.concatenate(vals).astype(np.int64)), torch.from_numpy(np.concatenate(lens)))
    feats = {BASE_DATA_GROUP: kjt}
    d = {}
    if dense:
        d = {BASE_DATA_GROUP: KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                                          torch.from_numpy(rng.random((B, sum(f.value_dim for f in dense)), dtype=np.float32)))}
    return Batch(d, feats, {"clk": torch.from_numpy((rng.random(B) < 0.3).astype(np.int64))}).to(dev)


def step(b):
    opt.zero_grad(set_to_none=True)
    losses, _ = _losses_and_predictions(model, model.loss, b)
    _backward_of_losses(losses)
    opt.step()
    return losses


bs = [batch() for _ in range(3)]
model.embedding_group.static_sequence_padding = True
for b in bs:
    step(b)
torch.cuda.synchronize()


def show(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", str(message)[:120])
    for fr in traceback.extract_stack()[:-1]:
        if "torcheasyrec_amd" in fr.filename or "scripts/r04" in fr.filename:
            print("    ", os.path.basename(fr.filename), fr.lineno, fr.line)


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step(bs[0])
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print("---- capture attempt")
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=ws):
        step(bs[1])
    g.replay()
    torch.cuda.synchronize()
    print("captured and replayed")
except Exception:
    traceback.print_exc()
