"""Can the persistent L2-swept SpMM share the GPU with a collective's kernel?  A stand-in for the latter (32 / 64 workgroups of 512
threads that stream memory for about a millisecond, on a side stream) runs beside the swept product of one rank's user rows of C3
at W = 8 and beside the full C3 product; times with and without it, and the row-wise kernels for comparison."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "coexist_lab.so")
if os.environ.get("NGCF_NO_BUILD_LAB") != "1" and os.environ.get("NGCF_NO_BUILD") != "1":   # never spawn a compiler under a profiler
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "coexist_lab.hip")])
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import dist as nd  # noqa: E402
hog = C.CDLL(so)
eng = pkg.engine
dev = torch.device("cuda:0")
U, I, d = 1_000_000, 100_000, 128
u, i, w = pkg.graphs.synthetic_interactions(U, I, 50_000_000, seed=2603, device=dev)
v, deg_u, deg_i = nd.laplacian_values(u, i, w, U, I)
side = torch.cuda.Stream(device=dev)
src = torch.randn(64 << 20, device=dev)          # 256 MB
dst = torch.empty_like(src)
ws = eng.Workspace()


def timed(fn, wgs):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if wgs:
        with torch.cuda.stream(side):            # ~ several ms of streaming copies by `wgs` workgroups
            assert hog.hog_launch(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_long(src.numel() // 4), 40, wgs,
                                  C.c_void_p(side.cuda_stream)) == 0
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


cases = {}
# full C3 product
coo = pkg.graphs._normalise(u, i, w, U, I)
N = U + I
csr = eng.LaplacianCSR.from_coo(coo["rows"], coo["cols"], coo["vals"], N, N)
X = torch.randn((N, d), device=dev)
out = torch.empty((N, d), device=dev)
cases["full C3 product"] = (csr, X, out)
# one rank's user rows at W = 8 in the bipartite layout (125 K users x 100 K items, 6.2 M entries)
lo, hi = nd.even_bounds(0, U, 8)[4:6]
(ur, uc, uv), _ = nd.cut_slabs(u, i, v, U, lo, hi, 0, 0)
csr_u = eng.LaplacianCSR.from_coo(ur - lo, uc - U, uv, hi - lo, I)
Xi = torch.randn((I, d), device=dev)
out_u = torch.empty((hi - lo, d), device=dev)
cases["user rows of one rank of 8"] = (csr_u, Xi, out_u)
for name, (c, x, o) in cases.items():
    for mode, label in ((3, "swept"), (0, "row-wise")):
        c.set_mode(mode)
        for wgs in (0, 32, 64, 128):
            ms = timed(lambda: eng.spmm(c, x, out=o, ws=ws), wgs)
            print(f"{name}: {label:8s} (swept rows {c.swept_rows}) beside {wgs:3d} streaming workgroups: {ms:.3f} ms", flush=True)
