"""Forward + backward (+ Adam) step time on the C3 graph (d0 = 130 -> [128,128,128] so that the injection runs)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg


def _reload_options():
    """the library reads its NGCF_* variables once; re-read them after changing os.environ"""
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _lib.options_from_env()

dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
torch.manual_seed(0)
model = pkg.NGCF(130, [128, 128, 128], 0.3, [0.1, 0.1, 0.1], 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, 1024, dev).to(dev)
model.check_indices = False
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = pkg.BPR(0.025, 1024)
g = torch.Generator().manual_seed(1)
B = 1024
batch = dict(year=torch.full((B,), 18), u_id=torch.randint(0, U, (B,), generator=g), age=torch.randint(0, 76, (B,), generator=g),
             sex=torch.randint(0, 2, (B,), generator=g), month=torch.randint(0, 13, (B,), generator=g),
             day=torch.randint(0, 32, (B,), generator=g), dow=torch.randint(0, 7, (B,), generator=g),
             pos_item=torch.randint(0, I, (B,), generator=g), neg_item=torch.randint(0, I, (B,), generator=g))
batch = {k: v.to(dev) for k, v in batch.items()}
model.eval()
for split in ("1", ""):
    if split:
        os.environ["NGCF_NO_PANEL_SPLIT"] = split
        _reload_options()
    else:
        os.environ.pop("NGCF_NO_PANEL_SPLIT")
    with torch.no_grad():
        for it in range(5):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            model(node_flag=False, **batch)
        torch.cuda.synchronize()
    print(f"inference forward at 130 -> [128,128,128] (reference-legal width), panel split {'off' if split else 'on'}: "
          f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
for mode, node_flag, train in (("eval-mode grads, no dropout", False, False), ("train mode, device node dropout + message dropout", True, True)):
    model.train(train)
    model.node_dropout_mode = "device"
    model.mess_dropout_mode = "device"
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        u, p, n = model(node_flag=node_flag, **batch)
        opt.zero_grad()
        loss = crit(u, p, n)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per training step, loss {float(loss):.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
