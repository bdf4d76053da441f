"""layer_dense_resident_il_kernel taken apart (library built with -DNGCF_LAB): NGCF_DENSE_IL_LAB = 1 no stores, 2 no loads, 3 neither."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg
from seoul_tourism_recommendation_ngcf_amd import _lib
eng = pkg.engine
dev = torch.device("cuda:0")
n, d = 1_100_000, 128
LE, E = (torch.randn((n, d), device=dev) * 0.05 for _ in range(2))
W1, W2 = (torch.randn((d, d), device=dev) * 0.05 for _ in range(2))
b1, b2 = (torch.randn((d,), device=dev) * 0.05 for _ in range(2))
carry, norm = (torch.empty((n, d), device=dev) for _ in range(2))
ws = eng.Workspace()
for resident, lab, what in ((1, 0, "resident kernel (product)"), (2, 0, "il complete"), (2, 1, "il without stores"), (2, 2, "il without loads"),
                            (2, 3, "il MFMAs + activation only"), (2, 6, "il no loads, stores hit L2"), (2, 10, "il no loads, staged not stored"), (1, 0, "resident kernel (product)"), (2, 0, "il complete")):
    _lib.set_option("dense_resident", resident)
    _lib.set_option("dense_il_lab", lab)
    f = lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{what:28s} {e0.elapsed_time(e1) / 20:7.4f} ms", flush=True)
