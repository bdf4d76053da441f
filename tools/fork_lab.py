"""Under hipGraph capture the two halves of a row-wise product become parallel branches when nnz x d >= 2e8 (csrc/spmm.hip,
launch_spmm).  This lab replays the Seoul-shaped forward at several widths with the fork forced on / off (NGCF_FORK_MIN, NGCF_NO_FORK)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seoul_tourism_recommendation_ngcf_amd as pkg
dev = torch.device("cuda:0")
nu, ni, B = 5840, 100, 1024
coo = pkg.graphs.seoul_standin(dev, seed=1801, n_user=nu, n_item=ni)[0]
nd = {"user": nu, "item": ni, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
g = torch.Generator().manual_seed(2)
ids = {k: torch.randint(0, c, (B,), generator=g).to(dev) for k, c in (("u_id", nu), ("pos_item", ni), ("neg_item", ni), ("age", 76),
                                                                     ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
year = torch.full((B,), 18, device=dev)
for e0, d in ((65, 64), (130, 128), (260, 256), (385, 384), (515, 512)):
    res = {}
    for label, env in (("fork", {"NGCF_FORK_MIN": "0"}), ("one branch", {"NGCF_NO_FORK": "1"})):
        for k in ("NGCF_FORK_MIN", "NGCF_NO_FORK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        torch.manual_seed(1)
        m = pkg.NGCF(e0, [d, d], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], nd, B, dev).to(dev).eval()
        m.check_indices = False
        fwd = pkg.GraphedForward(m, B, 0)
        fwd(year=year, node_flag=False, **ids)
        for _ in range(20):
            fwd.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fwd.replay()
        torch.cuda.synchronize()
        res[label] = (time.perf_counter() - t0) / 300 * 1e3
        del fwd, m
    print(f"embed {e0} -> [{d}, {d}] (nnz x d = {coo['nnz'] * d:.2e}): fork {res['fork']:.4f} ms, one branch {res['one branch']:.4f} ms", flush=True)
