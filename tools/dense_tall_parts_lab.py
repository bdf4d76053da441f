"""layer_dense_tall_kernel taken apart (library built with -DNGCF_LAB; NGCF_DENSE_IL_LAB: 1 no barriers, 2 no row staging, 4 no weight
loads in the loop, sums) at the Seoul row count - timing only, the results of the lab variants are wrong."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg
from seoul_tourism_recommendation_ngcf_amd import _lib
eng = pkg.engine
dev = torch.device("cuda:0")
n, d = 5940, 512
LE, E = (torch.randn((n, d), device=dev) * 0.05 for _ in range(2))
W1, W2 = (torch.randn((d, d), device=dev) * 0.05 for _ in range(2))
b1, b2 = (torch.randn((d,), device=dev) * 0.05 for _ in range(2))
carry, norm = (torch.empty((n, d), device=dev) for _ in range(2))
ws = eng.Workspace()
_lib.set_option("dense_tall", 2)
for lab, what in ((0, "complete"), (1, "no barriers"), (2, "no row staging"), (4, "no weight loads"), (3, "no barriers, no staging"), (7, "MFMAs + LDS reads only"), (0, "complete")):
    _lib.set_option("dense_il_lab", lab)
    f = lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{what:28s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us (pack 5 + kernel + row scale 6)", flush=True)
