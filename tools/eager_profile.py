"""Host-side profile (cProfile) of the eager NGCF.forward + BPR on the Seoul-shaped C1: where the Python time of a launch-bound
forward goes."""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seoul_tourism_recommendation_ngcf_amd as pkg
dev = torch.device("cuda:0")
nu, ni = 5840, 100
coo = pkg.graphs.seoul_standin(dev, seed=1801, n_user=nu, n_item=ni)[0]
nd = {"user": nu, "item": ni, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
m = pkg.NGCF(65, [64, 64], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], nd, 1024, dev).to(dev).eval()
m.check_indices = False
g = torch.Generator().manual_seed(2)
B = 1024
ids = {k: torch.randint(0, c, (B,), generator=g).to(dev) for k, c in (("u_id", nu), ("pos_item", ni), ("neg_item", ni), ("age", 76),
                                                                     ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
year = torch.full((B,), 18, device=dev)
crit = pkg.BPR(0.025, B)
with torch.no_grad():
    for _ in range(50):
        crit(*m(year=year, node_flag=False, **ids))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        crit(*m(year=year, node_flag=False, **ids))
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("LAB_TOP", "18")))
