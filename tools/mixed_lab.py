"""One launch that interleaves d-sliced user-row blocks with unsliced item-row blocks (C3, d=128)."""
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
rowptr = torch.searchsorted(rows, torch.arange(N + 1, device=dev))
E = torch.randn((N, d), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
# item units (segments of <= 512 entries; partials to scratch rows)
b, e = rowptr[U:N], rowptr[U + 1:N + 1]
nseg = torch.clamp((e - b + seg - 1) // seg, min=1)
owner = torch.repeat_interleave(torch.arange(I, device=dev), nseg)
first = torch.cumsum(nseg, 0) - nseg
k = torch.arange(owner.numel(), device=dev) - first[owner]
ub = b[owner] + k * seg
ue = torch.minimum(ub + seg, e[owner])
heavy = nseg[owner] > 1
dst = torch.where(heavy, N + torch.cumsum(heavy.long(), 0) - 1, owner + U)
order = torch.argsort((~heavy).long(), stable=True)
ub, ue, dst = ub[order].contiguous(), ue[order].contiguous(), dst[order].contiguous()
n_units = ub.numel()
item_blocks = (n_units + 3) // 4
urb = (U + 3) // 4
user_blocks = urb * 4
out = torch.empty((N + int(heavy.sum()) + 1, d), device=dev)


def timeit(sched):
    def run():
        assert lab.lab_launch_mixed(p(sched), C.c_int64(sched.numel()), p(rowptr), C.c_int64(U), C.c_int64(urb), p(ub), p(ue), p(dst),
                                    C.c_int64(n_units), p(cols32), p(vals), p(E), C.c_int64(d), p(out), C.c_int64(d), stream) == 0
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


users = torch.arange(user_blocks, device=dev, dtype=torch.int32)
items = -1 - torch.arange(item_blocks, device=dev, dtype=torch.int32)
print(f"user blocks {user_blocks}, item blocks {item_blocks}")
print(f"items first, then users (what the product does): {timeit(torch.cat([items, users])):.3f} ms")
print(f"users first, then items                        : {timeit(torch.cat([users, items])):.3f} ms")
print(f"users only {timeit(users):.3f} ms ; items only {timeit(items):.3f} ms")
# proportional interleave: positions of the item blocks spread evenly over the launch
total = user_blocks + item_blocks
pos_items = (torch.arange(item_blocks, device=dev, dtype=torch.float64) * (total / item_blocks)).long()
sched = torch.empty(total, dtype=torch.int32, device=dev)
mask = torch.zeros(total, dtype=torch.bool, device=dev)
mask[pos_items] = True
sched[mask] = items
sched[~mask] = users
print(f"evenly interleaved                              : {timeit(sched):.3f} ms")
# item blocks in runs of 32 (keeps groups of item waves together on the CUs)
for run in (8, 64, 512):
    n_runs = (item_blocks + run - 1) // run
    starts = (torch.arange(n_runs, device=dev, dtype=torch.float64) * ((total - run) / max(n_runs - 1, 1))).long()
    mask = torch.zeros(total, dtype=torch.bool, device=dev)
    idx = (starts[:, None] + torch.arange(run, device=dev)[None, :]).reshape(-1)[:item_blocks]
    mask[idx] = True
    if int(mask.sum()) != item_blocks:
        continue
    sched[mask] = items
    sched[~mask] = users
    print(f"item blocks in runs of {run:4d}                      : {timeit(sched):.3f} ms")
