"""Can the persistent L2-swept SpMM share the GPU with a COPY-ENGINE transfer?  (VERDICT r2 #1c; the counterpart of coexist_lab.py,
which put a streaming KERNEL beside it and saw the sweep lose 3.4-4x.)  The swept product of the full C3 graph and of one rank's
user rows (W = 8) runs alone, beside pinned-host <-> device copies on a second stream (SDMA engines: no kernel appears for them in a
kernel trace), and beside device-to-device copies on the same device (a blit KERNEL on this one-GPU box; across xGMI the same call
is an SDMA transfer).  Run under `rocprofv3 --kernel-trace --memory-copy-trace --stats` to see which is which."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import dist as nd  # noqa: E402
eng = pkg.engine
dev = torch.device("cuda:0")
U, I, d = 1_000_000, 100_000, 128
u, i, w = pkg.graphs.synthetic_interactions(U, I, 50_000_000, seed=2603, device=dev)
coo = pkg.graphs._normalise(u, i, w, U, I)
N = U + I
full = eng.LaplacianCSR.from_coo(coo["rows"], coo["cols"], coo["vals"], N, N)
full.set_mode(3)
v, deg_u, deg_i = nd.laplacian_values(u, i, w, U, I)
cnt = torch.cat([deg_u, deg_i]).cpu()
ub = nd.balanced_bounds(cnt, 0, U, 8)
(ur, uc, uv), _ = nd.cut_slabs(u, i, v, U, ub[4], ub[5], 0, 0)
slab = eng.LaplacianCSR.from_coo(ur - ub[4], uc - U, uv, ub[5] - ub[4], I)
slab.set_mode(3)
E = torch.randn((N, d), device=dev) * 0.1
Ei = torch.randn((I, d), device=dev) * 0.1
ws = eng.Workspace()
side = torch.cuda.Stream(device=dev)
MB = 64
host = torch.empty(MB << 18, dtype=torch.float32).pin_memory()
dbuf = torch.empty(MB << 18, dtype=torch.float32, device=dev)
dbuf2 = torch.empty_like(dbuf)


def timed(fn, copier, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_copies = 0
    if copier is not None:                       # keep the side stream busy for the whole measurement
        with torch.cuda.stream(side):
            for _ in range(40):
                copier()
                n_copies += 1
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


copiers = {
    "alone": None,
    f"beside pinned host -> device copies of {MB} MiB (SDMA)": lambda: dbuf.copy_(host, non_blocking=True),
    f"beside device -> pinned host copies of {MB} MiB (SDMA)": lambda: host.copy_(dbuf, non_blocking=True),
    f"beside device -> device copies of {MB} MiB on this GPU (blit kernel here; SDMA across xGMI)": lambda: dbuf2.copy_(dbuf, non_blocking=True),
}
for name, csr, tab in (("full C3 product", full, E), ("user rows of one rank of 8", slab, Ei)):
    for mode, label in ((3, "swept"), (0, "row-wise")):
        csr.set_mode(mode)
        for cname, cp in copiers.items():
            t0 = time.perf_counter()
            ms = timed(lambda: eng.spmm(csr, tab, ws=ws), cp)
            print(f"{name}: {label:8s} (swept rows {csr.swept_rows}) {cname}: {ms:.3f} ms", flush=True)
    csr.set_mode(3)
