"""Times the SpMM of ONE rank's slab of C5 (W = 8; see tests/test_c5_slab_gpu.py) under the row-wise kernels, with the
d-slicing threshold at its default (48 MiB of 128-byte slices) and raised to Infinity-Cache size (NGCF_SLICE_MAX_MB)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import dist as nd  # noqa: E402


def _reload_options():
    """the library reads its NGCF_* variables once; re-read them after changing os.environ"""
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _lib.options_from_env()

eng = pkg.engine
dev = torch.device("cuda:0")
U, I, W, rank, d = 10_000_000, 1_000_000, 8, 3, 256
u, i, w = pkg.graphs.synthetic_interactions(U, I, 500_000_000, seed=2605, device=dev)
v, deg_u, deg_i = nd.laplacian_values(u, i, w, U, I)
cnt = torch.cat([deg_u, deg_i]).cpu()
ub, ib = nd.balanced_bounds(cnt, 0, U, W), nd.balanced_bounds(cnt, U, U + I, W)
lay = nd.ShardLayout(U, I, ub, ib)
(ur, uc, uv), (ir, ic, iv) = nd.cut_slabs(u, i, v, U, ub[rank], ub[rank + 1], ib[rank] - U, ib[rank + 1] - U)
del u, i, v, w
slabs = {"user": (ur - ub[rank], lay.to_padded(uc), uv, lay.n_users_of(rank)), "item": (ir - ib[rank], lay.to_padded(ic), iv, lay.n_items_of(rank))}
X = torch.randn((lay.P, d), device=dev)
ws = eng.Workspace()


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


for mb in ("48", "160", "300"):
    os.environ["NGCF_SLICE_MAX_MB"] = mb
    _reload_options()
    for name, (r, c, vals, n_rows) in slabs.items():
        csr = eng.LaplacianCSR.from_coo(r, c, vals, n_rows, lay.P)
        for mode in (0, 1, 2):
            try:
                csr.set_mode(mode)
            except RuntimeError as e:
                print(name, "mode", mode, "plan failed:", str(e)[:80])
                continue
            if mode == 2 and csr.swept_rows == 0:
                print(f"{name} rows slice<= {mb} MiB mode 2: plan not applicable")
                continue
            out = torch.empty((n_rows, d), device=dev)
            ms = timeit(lambda: eng.spmm(csr, X, out=out, ws=ws))
            nnz = int(r.numel())
            a = nnz * 8 + (n_rows + 1) * 8 + lay.P * d * 4 + n_rows * d * 4
            print(f"{name} rows ({n_rows} x {lay.P}, {nnz} entries, d={d}) slice<= {mb:>3s} MiB mode {mode} swept_rows {csr.swept_rows}: {ms:7.3f} ms  "
                  f"gather {nnz * d * 4 / ms / 1e9:6.2f} TB/s  model-A {a / ms / 1e6 / 8000:.3f}", flush=True)
        del csr
