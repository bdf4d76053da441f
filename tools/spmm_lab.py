"""Drives tools/spmm_lab.hip on the C3 graph (GPU box only): times SpMM variants on user rows, item rows, both."""
import ctypes as C
import os
import subprocess
import sys

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
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 512
scale = sys.argv[3] if len(sys.argv) > 3 else "c3"
U, I, M = (1_000_000, 100_000, 50_000_000) if scale == "c3" else (100_000, 10_000, 2_000_000)
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N = U + I
d = 128
rows, cols, vals = coo["rows"], coo["cols"].to(torch.int32), coo["vals"]
rowptr = torch.searchsorted(rows, torch.arange(N + 1, device=dev))


def units_for(lo, hi):
    """work units (begin, end, dst row) for rows [lo,hi): rows longer than seg are cut; segment partials -> scratch rows"""
    b, e = rowptr[lo:hi], rowptr[lo + 1:hi + 1]
    ln = e - b
    nseg = torch.clamp((ln + seg - 1) // seg, min=1)
    first = torch.cumsum(nseg, 0) - nseg
    total = int(nseg.sum())
    owner = torch.repeat_interleave(torch.arange(hi - lo, device=dev), nseg)
    k = torch.arange(total, device=dev) - first[owner]
    ub = b[owner] + k * seg
    ue = torch.minimum(ub + seg, e[owner])
    heavy = nseg[owner] > 1
    dst = torch.where(heavy, N + torch.cumsum(heavy.long(), 0) - 1, owner + lo)
    order = torch.argsort((~heavy).long(), stable=True)          # segments first
    return ub[order].contiguous(), ue[order].contiguous(), dst[order].contiguous(), int(heavy.sum())


E = torch.randn((N, d), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, (lo, hi) in {"user rows": (0, U), "item rows": (U, N), "all rows": (0, N)}.items():
    ub, ue, dst, nheavy = units_for(lo, hi)
    out = torch.empty((N + nheavy + 1, d), device=dev)
    nnz = int((ue - ub).sum())
    for v in variants:
        def run():
            rc = lab.lab_launch(v, p(ub), p(ue), p(dst), C.c_int64(ub.numel()), p(cols), p(vals), p(E), C.c_int64(d),
                                C.c_int(d), p(out), C.c_int64(d), stream)
            assert rc == 0, rc
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print(f"{name:10s} variant {v}: {ms:7.3f} ms  units {ub.numel()}  gather {nnz * d * 4 / ms / 1e9:7.2f} TB/s"
              f"  edges/s {nnz / ms / 1e6:8.2f} G", flush=True)
    if name == "all rows" and 0 in variants:
        # sanity: variant 0 vs torch on a few light rows
        lab.lab_launch(0, p(ub), p(ue), p(dst), C.c_int64(ub.numel()), p(cols), p(vals), p(E), C.c_int64(d), C.c_int(d),
                       p(out), C.c_int64(d), stream)
        r = 12345
        a, b_ = int(rowptr[r]), int(rowptr[r + 1])
        want = (vals[a:b_, None] * E[cols[a:b_].long()]).sum(0)
        print("check row", r, float((out[r] - want).abs().max()))
