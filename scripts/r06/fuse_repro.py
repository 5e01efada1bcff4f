import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torcheasyrec_amd import _build, _lib, dense
_lib.use_library(_build.build())
from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
from torcheasyrec_amd.dense import FusedDenseAdam
from torcheasyrec_amd.dlrm import DLRM
from torcheasyrec_amd.embedding import SparseOptimizerConfig
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
rows = [min(r, 100000) for r in CRITEO_ROWS]
model = DLRM(criteo_tables(rows), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3))
opt = FusedDenseAdam(list(model.dense_parameters()), lr=1e-3, fuse_finish=True)
dense.unit_gradient(torch.zeros((), device=dev))
_cache = {}
def batch(B, s):
    if (B, s) not in _cache:
        d, k, y = synthetic_batch(s, B, rows)
        _cache[(B, s)] = (d.to(dev), k.to(dev), y.to(dev))
    return _cache[(B, s)]
def step(B, s):
    d, k, y = batch(B, s)
    loss, _ = model.forward_loss(d, k, y)
    with dense.root_loss():
        loss.backward(gradient=dense.unit_gradient(loss))
    print(B, "pending", [(hex(kk), v[0], v[2] if v[0] == "rows" else "") for kk, v in dense._PENDING.items()], flush=True)
    print("   grads", [(hex(p.grad.data_ptr()), p.grad.numel()) for p in opt.params if p.grad is not None], flush=True)
    opt.step(); opt.zero_grad(set_to_none=True)
    return loss
for s in range(3): step(65536, s)
torch.cuda.synchronize()
batch(65536, 5); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    l = step(65536, 5)
g.replay(); torch.cuda.synchronize()
for s in range(3): step(8192, s)
torch.cuda.synchronize(); print("ok")
