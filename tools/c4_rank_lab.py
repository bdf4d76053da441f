"""Compute time of ONE rank of the row-partitioned C3 graph (BASELINE config 4) at W = 1, 2, 4, 8, on one GPU: the rank's item
slab and user chunks through the real kernels of dist._propagate_allgather, the exchange left out.  This is the compute side of
the scaling question only (an upper bound on the speed-up if the all-gather cost nothing); the exchange itself needs more than
one GPU to measure.  Also prints the bytes a rank receives per layer."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import dist as nd  # noqa: E402
eng = pkg.engine
dev = torch.device("cuda:0")
U, I, d, n_layer = 1_000_000, 100_000, 128, 3
u, i, w = pkg.graphs.synthetic_interactions(U, I, 50_000_000, seed=2603, device=dev)
v, deg_u, deg_i = nd.laplacian_values(u, i, w, U, I)
cnt = torch.cat([deg_u, deg_i]).cpu()
g = torch.Generator().manual_seed(1)
W1, W2 = ((torch.rand((d, d), generator=g) - 0.5).to(dev) * 0.2 for _ in range(2))
b1, b2 = ((torch.rand((d,), generator=g) - 0.5).to(dev) * 0.1 for _ in range(2))
ws = eng.Workspace()
base = None
SWEPT_CHUNKS = os.environ.get("LAB_SWEPT_CHUNKS") == "1"      # plan the user chunks for the swept kernel too (mode 3)
for W in (1, 2, 4, 8):
    C = int(os.environ.get("LAB_CHUNKS", "4")) if W > 1 else 1
    ub, ib = nd.balanced_bounds(cnt, 0, U, W), nd.balanced_bounds(cnt, U, U + I, W)
    lay = nd.ShardLayout(U, I, ub, ib, nd.chunk_bounds(cnt, ub, C))
    r = W // 2
    (ur, uc, uv), (ir, ic, iv) = nd.cut_slabs(u, i, v, U, ub[r], ub[r + 1], ib[r] - U, ib[r + 1] - U)
    chunks = []
    for j in range(C):
        lo, hi = lay.chunk_range(r, j)
        cr, cc, cv = nd.slab_coo(ur, uc, uv, lo, hi)
        c = eng.LaplacianCSR.from_coo(cr, lay.to_padded(cc), cv, hi - lo, lay.P)
        if W == 1 or SWEPT_CHUNKS:
            c.set_mode(3)
        chunks.append(c)
    csr_i = eng.LaplacianCSR.from_coo(ir - ib[r], lay.to_padded(ic), iv, lay.n_items_of(r), lay.P)
    csr_i.set_mode(3)
    full = torch.randn((lay.P, d), device=dev) * 0.1
    nxt = torch.empty((lay.P, d), device=dev)
    out_u = torch.empty((lay.n_users_of(r), d), device=dev)
    out_i = torch.empty((lay.n_items_of(r), d), device=dev)

    def layer():
        ip = lay.item_pos(r)
        eng.layer_fused(csr_i, full, full[ip:ip + lay.n_items_of(r)], W1, b1, W2, b2, nxt[ip:ip + lay.n_items_of(r)], out_i, ws)
        for j, c in enumerate(chunks):
            lo, hi = lay.chunk_range(r, j)
            p0 = lay.user_pos(r, j)
            eng.layer_fused(c, full, full[p0:p0 + hi - lo], W1, b1, W2, b2, nxt[p0:p0 + hi - lo], out_u[lo - ub[r]:hi - ub[r]], ws)
    for _ in range(3):
        layer()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        layer()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    base = base or ms
    recv = (W - 1) / W * (U + I) * d * 4
    print(f"W={W}: rank {r} holds {sum(c.nnz for c in chunks) + csr_i.nnz} stored entries, swept rows {[csr_i.swept_rows] + [c.swept_rows for c in chunks]}; "
          f"compute {ms:.3f} ms per layer ({n_layer * ms:.2f} ms per 3-layer step, {base / ms:.2f}x the W=1 rank); "
          f"all-gather: {recv / 1e6:.0f} MB received per rank and layer = {recv / 153e9 * 1e3 / max(W - 1, 1) * (W - 1) / 7 if W > 1 else 0:.2f} ms if all 7 links "
          f"of 153 GB/s were busy, {recv / 153e9 * 1e3:.2f} ms over one link", flush=True)

# ---- the bipartite scheme (r03 form, dist._propagate_bipartite): users partitioned by stored entries, item carry replicated in the
# owner-major padded numbering; per rank and layer: item partial sums over the local users (small table, high re-use), the local
# user rows (gather from the item replica), the sum of the W partial-sum slices of the OWNED items and their dense half (I/W rows)
print("bipartite scheme (reduce-scatter of the item partial sums to their owners, sharded item dense, all-gather of the owned carry "
      "rows), one rank's compute, exchange left out:", flush=True)
import ctypes as C  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import _lib  # noqa: E402
lib = _lib.load()
base = None
for W in (1, 2, 4, 8):
    eb = nd.balanced_bounds(cnt, 0, U, W)
    ob = nd.even_bounds(0, I, W)
    mi = max(ob[q + 1] - ob[q] for q in range(W))
    PI = W * mi
    r = W // 2
    lo, hi = eb[r], eb[r + 1]
    (ur, uc, uv), _ = nd.cut_slabs(u, i, v, U, lo, hi, 0, 0)
    pos = nd.padded_item_pos(uc - U, ob, mi)
    csr_u = eng.LaplacianCSR.from_coo(ur - lo, pos, uv, hi - lo, PI)
    order = torch.sort(pos, stable=True).indices
    csr_it = eng.LaplacianCSR.from_coo(pos[order], ur[order] - lo, uv[order], PI, hi - lo)
    csr_it.set_mode(3)
    csr_u.set_mode(0 if os.environ.get("LAB_ROWWISE_USERS") == "1" else 3)     # swept: what runs over the CU-free p2p exchange
    eu = torch.randn((hi - lo, d), device=dev) * 0.1
    ei = torch.randn((PI, d), device=dev) * 0.1
    part = torch.empty((PI, d), device=dev)
    slots = torch.randn((W, mi, d), device=dev) * 0.1
    le_own = torch.empty((mi, d), device=dev)
    cu, nu_ = torch.empty((hi - lo, d), device=dev), torch.empty((hi - lo, d), device=dev)
    ci, ni_ = torch.empty((mi, d), device=dev), torch.empty((mi, d), device=dev)

    def layer_b():
        eng.spmm(csr_it, eu, out=part, ws=ws)
        eng.layer_fused(csr_u, ei, eu, W1, b1, W2, b2, cu, nu_, ws)
        _lib.check(lib.ngcf_sum_slots_f32(C.c_void_p(slots.data_ptr()), mi * d, W, mi * d, C.c_void_p(le_own.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        eng.layer_dense(le_own, ei[r * mi:(r + 1) * mi], W1, b1, W2, b2, ci, ni_, ws)
    for _ in range(3):
        layer_b()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        layer_b()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    base = base or ms

    def _t(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(n):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a_.elapsed_time(b_) / n
    t_a = _t(lambda: eng.spmm(csr_it, eu, out=part, ws=ws))
    t_bs = _t(lambda: eng.spmm(csr_u, ei, ws=ws))
    t_b = _t(lambda: eng.layer_fused(csr_u, ei, eu, W1, b1, W2, b2, cu, nu_, ws))
    t_c = _t(lambda: eng.layer_dense(le_own, ei[r * mi:(r + 1) * mi], W1, b1, W2, b2, ci, ni_, ws))
    print(f"    parts at W={W}: A (item partial sums, {csr_it.nnz} entries -> {PI} rows from {hi - lo} local users) {t_a:.3f} ms; B product alone "
          f"({csr_u.nnz} entries, {hi - lo} rows from the {PI}-row replica) {t_bs:.3f} ms, B with its dense half {t_b:.3f} ms; C (dense half of {mi} owned items) {t_c:.3f} ms; "
          f"a perfect 1/W of the one-GPU product would be {2.96 / W / 2:.3f} ms per half", flush=True)
    sent = 2 * (W - 1) * mi * d * 4
    print(f"W={W}: rank {r}: {csr_u.nnz + csr_it.nnz} stored entries, swept rows [{csr_it.swept_rows}, {csr_u.swept_rows}]; compute {ms:.3f} ms per layer "
          f"({n_layer * ms:.2f} ms per step, {base / ms:.2f}x the W=1 rank); exchange: {sent / 1e6:.1f} MB received per rank and layer "
          f"(partial-sum slices + carry rows) = {sent / max(W - 1, 1) / 153e9 * 1e3:.3f} ms with every peer on its own 153 GB/s link", flush=True)

# ---- r04: what does NOT shrink with W - the per-pass floor of the W = 8 rank (the last loop's W, r, eb, ob still hold): the E0
# copy of the local rows (skipped from the second pass on while the tables are unchanged: dist._propagate_bipartite `keep`), the
# served row gathers over the exchange (owned rows -> exchange buffer, W blocks pulled, one select kernel) and BPR
B, D = 1024, 512
nu, ni = hi - lo, ob[r + 1] - ob[r]
allE = torch.randn((nu + ni, D), device=dev) * 0.1                 # one result buffer, users first (dist._propagate_bipartite)
allE_u, allE_i = allE[:nu], allE[nu:]
uw, iw = torch.randn((U, d), device=dev), torch.randn((I, d), device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
gg = torch.Generator().manual_seed(3)
ids = [torch.randint(0, hi_, (B,), generator=gg).to(dev) for hi_ in (U, I, I)]
mine = torch.empty((3 * B, D), device=dev)
slots = torch.empty((W, 3 * B, D), device=dev)
own_idx = torch.cat([(torch.searchsorted(torch.tensor(eb, device=dev), ids[0], right=True) - 1).clamp(0, W - 1)] +
                    [(torch.searchsorted(torch.tensor(ob, device=dev), t, right=True) - 1).clamp(0, W - 1) for t in ids[1:]]) * (3 * B) + torch.arange(3 * B, device=dev)
crit = pkg.BPR(0.025, B)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def e0_copy():
    eng.copy_rows(uw[lo:hi], allE_u[:, :d])
    eng.copy_rows(iw[ob[r]:ob[r + 1]], allE_i[:, :d])


loc_idx = [(ids[0] - lo).contiguous(), (ids[1] - ob[r]).contiguous(), (ids[2] - ob[r]).contiguous()]   # cached per index tensor in the product


def owned():                                                    # ONE launch: the three gathers of the rows this rank owns
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.ngcf_gather_rows3_f32(p(allE), D, D, p(loc_idx[0]), B, 0, nu, p(mine[:B]), p(loc_idx[1]), B, nu, ni, p(mine[B:2 * B]),
                                         p(loc_idx[2]), B, nu, ni, p(mine[2 * B:]), D, p(status), C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def pulls_local():
    for q in range(W):
        slots[q].copy_(mine)


def select_and_bpr():
    out = eng.gather_rows(slots.view(W * 3 * B, D), own_idx, status)
    return crit(out[:B], out[B:2 * B], out[2 * B:])


t_e0, t_own, t_pull, t_sel = timeit(e0_copy), timeit(owned), timeit(pulls_local), timeit(select_and_bpr)
layer_ms = ms
print(f"W={W} rank {r} per-pass floor (us): E0 copy of the local rows {t_e0:.1f} (first pass only: retained afterwards), owned-row gathers into the "
      f"exchange buffer {t_own:.1f}, select kernel + BPR {t_sel:.1f}; the {W} block pulls of {3 * B * D * 4 / 1e6:.1f} MB as LOCAL device-to-device copies "
      f"{t_pull:.1f} (a stand-in: across xGMI they are {W - 1} SDMA copies on {W - 1} links at once, {3 * B * D * 4 / 153e9 * 1e6:.0f} us each at 153 GB/s) + "
      f"one publish host function (24 us, profiles/r04_memops_lab.txt)", flush=True)
floor_us = t_own + t_sel + 24 + 3 * B * D * 4 / 153e9 * 1e6 + 10
print(f"step estimate at W={W}: 3 x {layer_ms:.3f} ms + floor {floor_us / 1e3:.3f} ms = {3 * layer_ms + floor_us / 1e3:.3f} ms, if the per-layer exchange hides under "
      f"the products it overlaps (DESIGN.md 6.3); the one-GPU step is the driver's BENCH line", flush=True)
