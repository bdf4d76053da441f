"""r04 lab: the row-sparse transposed product of the last layer's backward (ngcf_spmm_t_rows_f32) on C3's L^T with R = 3 072 rows:
one wave per row with a slot-table gather per stored entry, against 32-row units scanned with the membership bitmap in LDS."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import _lib  # noqa: E402

eng = pkg.engine
dev = torch.device("cuda:0")
U, I, d = 1_000_000, 100_000, 128
coo = pkg.graphs.synthetic_bipartite(U, I, 50_000_000, seed=2603, device=dev)
N = U + I
order = torch.sort(coo["cols"], stable=True).indices
Lt = eng.LaplacianCSR.from_coo(coo["cols"][order], coo["rows"][order], coo["vals"][order], N, N)
lt_cols = coo["rows"].long()
del coo, order
g = torch.Generator().manual_seed(1)
ws = eng.Workspace()
out = torch.empty((N, d), device=dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for n_u, n_i in [(1024, 2048), (1024, 0), (100, 200), (1, 0)][:int(os.environ.get("LAB_CASES", "4"))]:      # (1, 0): next to no hits - the scan and the zero rows alone
    rows = torch.unique(torch.cat([torch.randint(0, U, (n_u,), generator=g), U + torch.randint(0, I, (n_i,), generator=g)])).to(dev)
    R = rows.numel()
    X, init = (torch.randn((R, d), generator=g).to(dev) for _ in range(2))
    slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
    slot[rows] = torch.arange(R, dtype=torch.int32, device=dev)
    hits = int((slot[lt_cols] >= 0).sum())
    for drop in (None, ([11, 12, 13], 0.3)):
        res = []
        for bm in (0, 1):
            _lib.set_option("t_rows_bitmap", bm)
            res.append(timeit(lambda: eng.spmm_t_rows(Lt, slot, X, init, out, ws, drop)))
        print(f"R={R} ({n_u} users, {n_i} items; {hits} hits), edge dropout {'on' if drop else 'off'}: slot-table kernel {res[0]:.1f} us, "
              f"16-row units + bitmap in LDS {res[1]:.1f} us", flush=True)
