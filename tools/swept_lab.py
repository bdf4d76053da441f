"""Times the row-wise vs the L2-swept SpMM on the two halves of the C3 graph, sweeping the column-block size."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
eng = pkg.engine
dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, int(os.environ.get("LAB_D", "128"))
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
parts = {"user rows": (rows[:nu], cols[:nu], vals[:nu], U), "item rows": (rows[nu:] - U, cols[nu:], vals[nu:], I)}
E = torch.randn((N, d), device=dev)
ws = eng.Workspace()
blocks = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,512,1024,2048".split(","))]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, (r, c, v, nr) in parts.items():
    csr = eng.LaplacianCSR.from_coo(r, c, v, nr, N)
    csr.set_mode(1)
    out = torch.empty((nr, d), device=dev)
    ms = timeit(lambda: eng.spmm(csr, E, out=out, ws=ws))
    ref = out.clone()
    print(f"{name}: row-wise            {ms:7.3f} ms  gather {v.numel() * d * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    for kb in blocks:
        os.environ["NGCF_SWEPT_BLOCK_KB"] = str(kb)
        csr.set_mode(1)
        t0 = time.time()
        csr.set_mode(2)
        tb = time.time() - t0
        ms = timeit(lambda: eng.spmm(csr, E, out=out, ws=ws))
        err = float((out - ref).abs().max())
        print(f"{name}: swept block {kb:5d} KiB {ms:7.3f} ms  gather {v.numel() * d * 4 / ms / 1e9:6.2f} TB/s  "
              f"(plan {tb:.1f} s, max diff {err:.1e})", flush=True)
