"""Item-row half as one launch per column block (d-sliced, slice-major inside a launch), accumulating into LE."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
so = os.path.join(ROOT, "tools", "spmm_lab.so")
if os.environ.get("NGCF_NO_BUILD") != "1":      # never spawn a compiler under a profiler: build first
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "spmm_lab.hip")])
lab = C.CDLL(so)
dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, 128
rows, cols32, vals = coo["rows"], coo["cols"].to(torch.int32), coo["vals"]
rowptr = torch.searchsorted(rows, torch.arange(N + 1, device=dev))
E = torch.randn((N, d), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rb, re_ = rowptr[U:N], rowptr[U + 1:N + 1]            # item rows
# reference result
eng = pkg.engine
nu = int(rowptr[U])
csr_i = eng.LaplacianCSR.from_coo(rows[nu:] - U, coo["cols"][nu:], vals[nu:], I, N)
csr_i.set_mode(1)
ref = eng.spmm(csr_i, E)
key = rows * (1 << 21) + coo["cols"]                 # (row, col) sortable key; cols < 2^21
for nblk in (8, 16, 31, 61):
    brows = (U + nblk - 1) // nblk
    # boundaries of each block inside each item row: searchsorted on the combined key
    bounds = torch.arange(nblk + 1, device=dev) * brows
    q = (torch.arange(U, N, device=dev)[:, None] * (1 << 21) + bounds[None, :].clamp(max=(1 << 21) - 1)).reshape(-1)
    pos = torch.searchsorted(key, q).reshape(I, nblk + 1)
    for variant in (0, 1, 3, 4, 6):
        out = torch.empty((I, d), device=dev)
        def run():
            for b in range(nblk):
                ub, ue = pos[:, b].contiguous(), pos[:, b + 1].contiguous()
                rc = lab.lab_launch_acc(variant, p(ub), p(ue), C.c_int64(I), p(cols32), p(vals), p(E), C.c_int64(d), C.c_int(d),
                                        p(out), C.c_int64(d), C.c_int(1 if b else 0), stream)
                assert rc == 0
        ubs = [(pos[:, b].contiguous(), pos[:, b + 1].contiguous()) for b in range(nblk)]
        def run2():
            for b, (ub, ue) in enumerate(ubs):
                lab.lab_launch_acc(variant, p(ub), p(ue), C.c_int64(I), p(cols32), p(vals), p(E), C.c_int64(d), C.c_int(d),
                                   p(out), C.c_int64(d), C.c_int(1 if b else 0), stream)
        for _ in range(2):
            run2()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run2()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        err = float((out - ref).abs().max())
        print(f"{nblk:3d} column blocks ({brows} users), variant {variant}: {ms:6.3f} ms  gather {(N*0+int(rowptr[N]-rowptr[U])) * d * 4 / ms / 1e9:5.2f} TB/s  max diff {err:.1e}", flush=True)
