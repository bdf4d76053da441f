"""r04 lab: from how many rows does the weights-resident dense kernel (persistent: 2 048 waves x 32 rows = 65 536 rows per round) beat
the staged one?  r02 set the switch at two full rounds (131 072 rows); a rank's user slab of C3 at W = 8 is 125 015 rows."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import _lib  # noqa: E402

eng = pkg.engine
dev = torch.device("cuda:0")
d = 128
g = torch.Generator().manual_seed(1)
W1, W2 = ((torch.rand((d, d), generator=g) - 0.5).to(dev) * 0.2 for _ in range(2))
b1, b2 = ((torch.rand((d,), generator=g) - 0.5).to(dev) * 0.1 for _ in range(2))
ws = eng.Workspace()


def timeit(fn, n=30):
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


for N in (32768, 49152, 65536, 81920, 98304, 114688, 125015, 131072, 163840, 196608, 250062):
    E = (torch.rand((N, d), generator=g) - 0.5).to(dev) * 0.3
    LE = (torch.rand((N, d), generator=g) - 0.5).to(dev) * 0.3
    carry, norm = torch.empty((N, d), device=dev), torch.empty((N, d), device=dev)
    res = []
    for min_rows in (1 << 30, 1):
        _lib.set_option("dense_resident_min_rows", min_rows)
        res.append(timeit(lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws)))
    print(f"N={N}: staged {res[0]:.1f} us, weights-resident {res[1]:.1f} us", flush=True)
