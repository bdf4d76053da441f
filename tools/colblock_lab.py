"""Item-row half with column blocking: does keeping the gathered user block Infinity-Cache resident pay?"""
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
N, d, seg = U + I, 128, 512
rows, cols32, vals = coo["rows"], coo["cols"].to(torch.int32), coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
E = torch.randn((N, d), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ir, ic = rows[nu:], coo["cols"][nu:]
for nblk in (1, 16, 32, 64, 128):
    brows = (U + nblk - 1) // nblk
    key = ir * nblk + ic // brows                      # (row, block) id, non-decreasing within a row
    change = torch.ones_like(key, dtype=torch.bool)
    change[1:] = key[1:] != key[:-1]
    ub = change.nonzero().flatten() + nu               # unit begins (global entry index)
    ue = torch.cat([ub[1:], torch.tensor([rows.numel()], device=dev)])
    blk = (ic // brows)[ub - nu]
    # cut long units at seg
    ln = ue - ub
    nseg = torch.clamp((ln + seg - 1) // seg, min=1)
    owner = torch.repeat_interleave(torch.arange(ub.numel(), device=dev), nseg)
    first = torch.cumsum(nseg, 0) - nseg
    k = torch.arange(owner.numel(), device=dev) - first[owner]
    b2 = ub[owner] + k * seg
    e2 = torch.minimum(b2 + seg, ue[owner])
    order = torch.sort(blk[owner], stable=True).indices        # block-major execution order
    b2, e2 = b2[order].contiguous(), e2[order].contiguous()
    dst = torch.arange(b2.numel(), device=dev)
    out = torch.empty((b2.numel() + 1, d), device=dev)
    def run():
        rc = lab.lab_launch(0, p(b2), p(e2), p(dst), C.c_int64(b2.numel()), p(cols32), p(vals), p(E), C.c_int64(d),
                            C.c_int(d), p(out), C.c_int64(d), stream)
        assert rc == 0
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
    nnz = rows.numel() - nu
    print(f"item rows, {nblk:2d} column blocks ({brows * d * 4 / 2**20:6.0f} MiB each): {ms:6.3f} ms, units {b2.numel()}, "
          f"gather {nnz * d * 4 / ms / 1e9:5.2f} TB/s, partial rows {b2.numel() * d * 4 / 2**20:5.0f} MiB", flush=True)
