"""Isolate hipGraph capture problems: capture progressively larger parts of the DLRM step."""
import faulthandler, os, sys, torch
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcheasyrec_amd import _build, _lib
from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
from torcheasyrec_amd.dlrm import DLRM, bce_with_logits
from torcheasyrec_amd.embedding import SparseOptimizerConfig
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
rows = [min(r, 200000) for r in CRITEO_ROWS]; B = 8192
model = DLRM(criteo_tables(rows), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3))
model.ebc.async_plan = False
opt = torch.optim.Adam(list(model.dense_parameters()), lr=1e-3, fused=True, capturable=True)
dense, kjt, label = synthetic_batch(0, B, rows); dense, kjt, label = dense.to(dev), kjt.to(dev), label.to(dev)
which = sys.argv[1]
def body():
    if which == "ebc":
        out = model.ebc.forward_grouped(kjt)["sparse"]; out.sum().backward(); return out.sum()
    if which == "inter":
        from torcheasyrec_amd.interaction import dot_interaction
        d = torch.ones(B, 16, device=dev, requires_grad=True); sp = torch.ones(B, 416, device=dev, requires_grad=True)
        y = dot_interaction(d, sp, 16); y.sum().backward(); return y.sum()
    loss = bce_with_logits(model(dense, kjt), label); loss.backward()
    if which == "full": opt.step(); opt.zero_grad(set_to_none=True)
    return loss
for _ in range(3): l = body()
torch.cuda.synchronize(); del l
print(which, "eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=ws):
    l = body()
print(which, "captured", flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize()
print(which, "replayed ok", float(l.item()), flush=True)
