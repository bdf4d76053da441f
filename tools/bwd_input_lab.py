"""r04 lab: the input-gradient kernels at C3's size (1.1 M rows, K = 128): staged (`bwd_input_resident = 0`) against the
weights-resident kernel (+ the narrow remainder kernel at d_in = 130), and the weight-gradient kernels beside them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402
from seoul_tourism_recommendation_ngcf_amd import _lib, autograd as ag  # noqa: E402

eng = pkg.engine
dev = torch.device("cuda:0")
N = int(os.environ.get("LAB_ROWS", "1100000"))
g = torch.Generator().manual_seed(1)
ws = eng.Workspace()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for d_in in (128, 130):
    d_out = 128
    dM = torch.empty((N, 128), device=dev).uniform_(-0.3, 0.3)
    ld = (d_in + 31) // 32 * 32
    LE = torch.empty((N, ld), device=dev).uniform_(-0.3, 0.3)[:, :d_in]
    E = torch.empty((N, ld), device=dev).uniform_(-0.3, 0.3)[:, :d_in]
    W1, W2 = ((torch.rand((d_out, d_in), generator=g) - 0.5).to(dev) * 0.2 for _ in range(2))
    for res in (0, 1):
        _lib.set_option("bwd_input_resident", res)
        t = timeit(lambda: ag._bwd_input(dM, W1, W2, LE, E, ws))
        print(f"d_in={d_in}: input gradients, bwd_input_resident={res}: {t * 1e3:.1f} us  ({4 * N * d_in * d_out / t / 1e9:.1f} TFLOP/s)", flush=True)
    t = timeit(lambda: ag._bwd_weight(dM, LE, E, ws))
    print(f"d_in={d_in}: weight gradients: {t * 1e3:.1f} us", flush=True)
    del dM, LE, E
