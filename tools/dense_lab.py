"""Times the dense layer kernel at several widths (1 M rows)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg
eng = pkg.engine
dev = torch.device("cuda:0")
n = 1_000_000
ws = eng.Workspace()
for d_in, d_out in ((16, 128), (32, 128), (64, 128), (128, 128), (256, 128)):
    LE, E = (torch.randn((n, d_in), device=dev) for _ in range(2))
    W1, W2 = (torch.randn((d_out, d_in), device=dev) * 0.05 for _ in range(2))
    b1, b2 = (torch.randn((d_out,), device=dev) * 0.05 for _ in range(2))
    carry, norm = (torch.empty((n, d_out), device=dev) for _ in range(2))
    f = lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * n * d_in * d_out
    by = n * (2 * d_in + 2 * d_out) * 4
    print(f"d_in {d_in:4d} d_out {d_out:4d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  {by / ms / 1e9:6.2f} TB/s", flush=True)
