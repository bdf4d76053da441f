"""Upper bound for an L2-swept gather: units execute in list order, unit k gathers 64 random rows from a table
window that advances with k (what perfectly synchronised column-block sweeping would look like)."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "spmm_lab.so")
if os.environ.get("NGCF_NO_BUILD") != "1":      # never spawn a compiler under a profiler: build first
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "spmm_lab.hip")])
lab = C.CDLL(so)
dev = torch.device("cuda:0")
N, d, nnz, per = 1_000_000, 128, 50_000_000 // 64 * 64, 64
E = torch.randn((N, d), device=dev)
n_units = nnz // per
ub = torch.arange(n_units, device=dev, dtype=torch.int64) * per
ue = ub + per
dst = torch.arange(n_units, device=dev, dtype=torch.int64) % 100_000
out = torch.empty((100_001, d), device=dev)
vals = torch.rand(nnz, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
for variant, rowbytes in ((0, 512), (3, 256)):
    for win_rows in (N, 65536, 16384, 8192, 4096, 2048, 1024):
        pos = torch.arange(nnz, device=dev, dtype=torch.float64) / nnz          # sweep position of each edge
        base = (pos * (N - win_rows)).to(torch.int64)
        cols = (base + torch.randint(0, win_rows, (nnz,), generator=g, device=dev)).to(torch.int32)
        def run():
            assert lab.lab_launch(variant, p(ub), p(ue), p(dst), C.c_int64(n_units), p(cols), p(vals), p(E), C.c_int64(d),
                                  C.c_int(d), p(out), C.c_int64(d), stream) == 0
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"variant {variant} ({rowbytes} B per gathered row-slice): window {win_rows:8d} rows = {win_rows * rowbytes / 2**20:8.2f} MiB: "
              f"{ms:6.3f} ms  {nnz * d * 4 / ms / 1e9:6.2f} TB/s", flush=True)
