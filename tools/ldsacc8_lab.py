"""Experiment: 8 vs 16 lanes per entry (32- vs 64-float slices) for the L2-swept SpMM (tools/ldsacc8_lab.hip).
The plan (row pieces -> wave tasks, entries sorted by (task, column window)) is built with torch ops here."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
eng = pkg.engine
so = os.path.join(ROOT, "tools", "ldsacc8_lab.so")
if os.environ.get("NGCF_NO_BUILD") != "1":      # never spawn a compiler under a profiler: build first
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "ldsacc8_lab.hip")])
lab = C.CDLL(so)
DRY = not torch.cuda.is_available()            # CPU dry run of the plan code on a small graph
dev = torch.device("cpu" if DRY else "cuda:0")
U, I, M = (20_000, 3_000, 300_000) if DRY else (1_000_000, 100_000, 50_000_000)
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, 128
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
which = os.environ.get("LAB_HALF", "item")
if which == "item":
    r, c, v, nr = rows[nu:] - U, cols[nu:], vals[nu:], I
else:
    r, c, v, nr = rows[:nu], cols[:nu], vals[:nu], U
E = torch.randn((N, d), device=dev)
ws = eng.Workspace()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = None if DRY else C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


if not DRY:
    csr = eng.LaplacianCSR.from_coo(r, c, v, nr, N)
    csr.set_mode(1)
    ref = torch.empty((nr, d), device=dev)
    ms = timeit(lambda: eng.spmm(csr, E, out=ref, ws=ws))
    print(f"{which} rows: row-wise {ms:7.3f} ms  gather {v.numel() * d * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    del csr

nnz = r.numel()
cnt = torch.bincount(r, minlength=nr)
rowstart = torch.cumsum(cnt, 0) - cnt
rank = torch.arange(nnz, device=dev) - rowstart[r]
c_lo = int(c.min())


def plan(lpe, RW, win_kb, waves):
    EPR = 64 // lpe
    n_waves = 256 * waves
    n_rowpass = max(1, -(-nr // (n_waves * RW)))
    while True:
        n_tasks = n_rowpass * n_waves
        T = max(64, -(-nnz // n_tasks))
        Tp = max(64, T // (4 if EPR == 4 else 16))                   # no row may hold more than 1/EPR of a bucket
        k = torch.clamp((cnt + Tp - 1) // Tp, min=1)                 # pieces per row
        n_pieces = int(k.sum())
        if n_pieces <= n_tasks * RW:
            break
        n_rowpass += 1
    pbase = torch.cumsum(k, 0) - k
    piece = pbase[r] + rank % k[r]
    pn = torch.bincount(piece, minlength=n_pieces)
    order = torch.sort(pn, descending=True, stable=True).indices   # heaviest first; level by level, heaviest piece -> lightest task
    task = torch.empty(n_pieces, dtype=torch.int64, device=dev)
    lrow = torch.empty(n_pieces, dtype=torch.int64, device=dev)
    load = torch.zeros(n_tasks, dtype=torch.int64, device=dev)
    for L in range(-(-n_pieces // n_tasks)):
        idx = order[L * n_tasks:(L + 1) * n_tasks]
        tk = torch.sort(load, stable=True).indices[:idx.numel()]
        task[idx] = tk
        lrow[idx] = L
        load[tk] += pn[idx]
    assert int(lrow.max()) < RW, (int(lrow.max()), RW)
    # piece -> destination row: whole rows write `out`, pieces of cut rows write partial rows after it
    prow = torch.repeat_interleave(torch.arange(nr, device=dev), k)
    cut = k[prow] > 1
    dst_piece = torch.where(cut, nr + torch.cumsum(cut.long(), 0) - 1, prow)
    n_partial = int(cut.sum())
    dst = torch.full((n_tasks * RW,), -1, dtype=torch.int32, device=dev)
    dst[task * RW + lrow] = dst_piece.to(torch.int32)
    win_cols = win_kb * 1024 // (lpe * 16)
    n_win = max(1, -(-(int(c.max()) - c_lo + 1) // win_cols))
    NB = n_tasks * n_win
    bucket = task[piece] * n_win + (c - c_lo) // win_cols
    key2 = bucket * 256 + lrow[piece]
    o2 = torch.sort(key2, stable=True).indices
    total_b = torch.bincount(bucket, minlength=NB)
    maxrow_b = torch.bincount(key2, minlength=NB * 256).view(NB, 256).max(1).values
    # rounds of a bucket: entries of one row never share a round (plain LDS read-modify-write in the kernel)
    R_b = torch.maximum(maxrow_b, (total_b + EPR - 1) // EPR)
    size_b = R_b * EPR
    base_b = torch.cumsum(size_b, 0) - size_b
    bstart = torch.cumsum(total_b, 0) - total_b
    bs = bucket[o2]
    t = torch.arange(nnz, device=dev) - bstart[bs]
    newpos = base_b[bs] + (t % R_b[bs]) * EPR + t // R_b[bs]
    n_slots = int(size_b.sum())
    e_pack = torch.full((n_slots,), -1, dtype=torch.int64, device=dev)
    e_val = torch.zeros(n_slots, dtype=torch.float32, device=dev)
    e_pack[newpos] = (lrow[piece[o2]] << 24) | c[o2]
    e_val[newpos] = v[o2]
    pad = torch.nonzero(e_pack < 0).flatten()                       # empty slots: spare row RW, first column of their bucket
    pb = torch.searchsorted(torch.cumsum(size_b, 0), pad, right=True)
    e_pack[pad] = (RW << 24) | (e_pack[base_b[pb]] & 0xffffff)
    e_pack = e_pack.to(torch.int32)
    tptr = torch.zeros(NB + 1, dtype=torch.int64, device=dev)
    tptr[1:] = torch.cumsum(size_b, 0)
    per_task = tptr[::n_win][1:] - tptr[::n_win][:-1]
    print(f"  plan lpe {lpe} waves {waves} RW {RW} window {win_kb} KiB = {win_cols} cols x {n_win}; row passes {n_rowpass}, T {T}, "
          f"pieces {n_pieces} (partial rows {n_partial}), slots {n_slots} (+{(n_slots - nnz) / nnz * 100:.1f}% padding), "
          f"slots/task min {int(per_task.min())} max {int(per_task.max())}", flush=True)
    return dict(lpe=lpe, RW=RW, n_rowpass=n_rowpass, n_win=n_win, tptr=tptr, e_pack=e_pack, e_val=e_val, dst=dst,
                n_partial=n_partial, prow=prow, cut=cut, EPR=EPR)


bar = torch.zeros(256, dtype=torch.int32, device=dev)
for lpe, RW, waves in ((8, 72, 16), (8, 144, 8), (16, 72, 8)):
    for win_kb in [int(x) for x in os.environ.get("LAB_WIN_KB", "2048,4096").split(",")]:
        P = plan(lpe, RW, win_kb, waves)
        out = torch.zeros((nr + P["n_partial"], d), device=dev)
        if DRY:
            lr = (P["e_pack"].long() >> 24) & 255
            ok = lr < RW
            t_of = torch.bucketize(torch.arange(P["e_pack"].numel()), P["tptr"][::P["n_win"]][1:].contiguous(), right=True)
            drow = P["dst"][(t_of * RW + lr)[ok]].long()
            res = torch.zeros((nr + P["n_partial"], d)).index_add_(0, drow, (P["e_val"][:, None] * E[(P["e_pack"] & 0xffffff).long()])[ok])
            full = res[:nr].index_add_(0, P["prow"][P["cut"]], res[nr:])
            want = torch.zeros((nr, d)).index_add_(0, r, v[:, None] * E[c])
            rel = torch.arange(P["e_pack"].numel()) - P["tptr"][::P["n_win"]][:-1][t_of]
            rid = (t_of * (1 << 40) + (rel // P["EPR"]) * 256 + lr)[ok]
            print("  dry-run plan check, max diff", float((full - want).abs().max()), "round conflicts",
                  int(rid.numel() - torch.unique(rid).numel()), "pads", int((~ok).sum()))
            continue
        for spin, lead in ((500, 1), (500, 2), (0, -1)):
            def run():
                rc = lab.ldsacc8_launch(lpe, waves, p(P["tptr"]), p(P["e_pack"]), p(P["e_val"]), p(P["dst"]), P["n_rowpass"],
                                        P["n_win"], d, p(E), C.c_int64(d), p(out), C.c_int64(d), p(bar), spin, lead, stream)
                assert rc == 0, rc
            ms = timeit(run)
            res = out[:nr].clone()
            if P["n_partial"]:
                res.index_add_(0, P["prow"][P["cut"]], out[nr:])
            err = float((res - ref).abs().max())
            print(f"  lpe {lpe} waves {waves} window {win_kb:5d} KiB lead {lead:2d}: {ms:7.3f} ms  "
                  f"gather {nnz * d * 4 / ms / 1e9:6.2f} TB/s  max diff {err:.1e}", flush=True)
        del P, out
