"""Debug: capture the training forward and backward of NGCF as two hipGraphs by hand (what torch.cuda.make_graphed_callables does)."""
import os, sys, faulthandler, torch
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg
from seoul_tourism_recommendation_ngcf_amd.NGCF import _TrainCore
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "drop"
slices = pkg.graphs.seoul_standin(dev, seed=6, n_user=600, n_item=30)
lap = [pkg.graphs.to_sparse_coo(s) for s in slices]
U, I, B = 600, 30, 96
nd = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
torch.manual_seed(4)
if variant == "nodrop":
    model = pkg.NGCF(65, [65, 65, 65], None, None, 1.0, lap, nd, B, dev).to(dev)
else:
    model = pkg.NGCF(65, [65, 65, 65], 0.3, [0.1, 0.1, 0.1], 1.0, lap, nd, B, dev).to(dev)
model.train()
model.node_dropout_mode = model.mess_dropout_mode = "device"
model.auto_train_graph = False
g = torch.Generator().manual_seed(31)
r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)
args = (r(U), r(76), r(2), r(13), r(32), r(7), r(I), r(I))
core = _TrainCore(model, 0, variant != "nodrop", True)
params = [p for p in model.parameters() if p.requires_grad]
print("eager steps", flush=True)
for _ in range(2):
    outs = core(*args)
    gi = torch.autograd.grad(outs, params, grad_outputs=[torch.ones_like(o) for o in outs], allow_unused=True)
torch.cuda.synchronize()
print("warm-up on a side stream", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        outs = core(*args)
        gi = torch.autograd.grad(outs, params, grad_outputs=[torch.empty_like(o) for o in outs], allow_unused=True)
    del outs, gi
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
pool = torch.cuda.graph_pool_handle()
fg, bg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
print("capture forward", flush=True)
with torch.cuda.graph(fg, pool=pool):
    outs = core(*args)
print("forward captured", flush=True)
go = [torch.empty_like(o) for o in outs]
if variant == "bwd_same_graph":
    pass
print("capture backward", flush=True)
with torch.cuda.graph(bg, pool=pool):
    gi = torch.autograd.grad(outs, params, grad_outputs=go, allow_unused=True)
print("backward captured", flush=True)
fg.replay(); bg.replay(); torch.cuda.synchronize()
print("replayed", [None if x is None else float(x.abs().sum()) for x in gi][:4], flush=True)
