"""Does the dense (MFMA) kernel of the user rows hide under the SpMM of the item rows on a second stream?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402

eng = pkg.engine
dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, 128
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
csr_u = eng.LaplacianCSR.from_coo(rows[:nu], cols[:nu], vals[:nu], U, N)
csr_i = eng.LaplacianCSR.from_coo(rows[nu:] - U, cols[nu:], vals[nu:], I, N)
csr_all = eng.LaplacianCSR.from_coo(rows, cols, vals, N, N)
E = torch.randn((N, d), device=dev)
W1, W2 = (torch.randn((d, d), device=dev) * 0.05 for _ in range(2))
b1, b2 = (torch.randn((d,), device=dev) * 0.05 for _ in range(2))
LE = torch.empty((N, d), device=dev)
carry = torch.empty((N, d), device=dev)
norm = torch.empty((N, d), device=dev)
ws1, ws2, ws3 = eng.Workspace(), eng.Workspace(), eng.Workspace()
s2 = torch.cuda.Stream()


def seq_all():
    eng.spmm(csr_all, E, out=LE, ws=ws1)
    eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws2)


def seq_split():
    eng.spmm(csr_u, E, out=LE[:U], ws=ws1)
    eng.spmm(csr_i, E, out=LE[U:], ws=ws1)
    eng.layer_dense(LE[:U], E[:U], W1, b1, W2, b2, carry[:U], norm[:U], ws2)
    eng.layer_dense(LE[U:], E[U:], W1, b1, W2, b2, carry[U:], norm[U:], ws2)


def overlapped():
    s1 = torch.cuda.current_stream()
    eng.spmm(csr_u, E, out=LE[:U], ws=ws1)
    ev = torch.cuda.Event()
    ev.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        eng.layer_dense(LE[:U], E[:U], W1, b1, W2, b2, carry[:U], norm[:U], ws3)
        ev2 = torch.cuda.Event()
        ev2.record(s2)
    eng.spmm(csr_i, E, out=LE[U:], ws=ws2)
    eng.layer_dense(LE[U:], E[U:], W1, b1, W2, b2, carry[U:], norm[U:], ws2)
    s1.wait_event(ev2)


def overlapped_items_first():
    """item rows first, then user-row SpMM overlapped with the (small) item dense; user dense exposed"""
    s1 = torch.cuda.current_stream()
    eng.spmm(csr_i, E, out=LE[U:], ws=ws1)
    ev = torch.cuda.Event()
    ev.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        eng.layer_dense(LE[U:], E[U:], W1, b1, W2, b2, carry[U:], norm[U:], ws3)
        ev2 = torch.cuda.Event()
        ev2.record(s2)
    eng.spmm(csr_u, E, out=LE[:U], ws=ws2)
    eng.layer_dense(LE[:U], E[:U], W1, b1, W2, b2, carry[:U], norm[:U], ws2)
    s1.wait_event(ev2)


for name, fn in (("one launch each", seq_all), ("split, sequential", seq_split), ("split, dense_u || spmm_i", overlapped),
                 ("items first", overlapped_items_first)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s}: {e0.elapsed_time(e1) / 10:7.3f} ms per layer", flush=True)
ref = norm.clone()
seq_all()
torch.cuda.synchronize()
print("overlap result equals sequential:", torch.equal(ref, norm))
