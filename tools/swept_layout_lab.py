"""Does the stride of the gathered table matter to the L2-swept kernel?  One 64-float slice gathered from a contiguous
[N, 64] table vs from the left half of a [N, 128] table (every other 256 B of the address space), C3 halves."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402


def _reload_options():
    """the library reads its NGCF_* variables once; re-read them after changing os.environ"""
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _lib.options_from_env()

eng = pkg.engine
dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N = U + I
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
parts = {"user rows": (rows[:nu], cols[:nu], vals[:nu], U), "item rows": (rows[nu:] - U, cols[nu:], vals[nu:], I)}
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


for name, (r, c, v, nr) in parts.items():
    csr = eng.LaplacianCSR.from_coo(r, c, v, nr, N)
    for kb in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2048,4096,8192").split(",")]:
        os.environ["NGCF_SWEPT_WINDOW_KB"] = str(kb)
        _reload_options()
        csr.set_mode(1)
        csr.set_mode(2)
        for ld in (64, 128, 256, 512):
            E = torch.randn((N, ld), device=dev)
            out = torch.empty((nr, 64), device=dev)
            ms = timeit(lambda: eng.spmm(csr, E[:, :64], out=out, ws=ws))
            print(f"{name}: window {kb:5d} KiB, one 64-float slice of a table with ld={ld:4d}: {ms:7.3f} ms  "
                  f"gather {v.numel() * 256 / ms / 1e9:6.2f} TB/s", flush=True)
            del E
