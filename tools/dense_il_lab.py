"""The two weights-resident dense kernels side by side at C3's row count (dense_resident = 1: two waves per SIMD, a tile's stores in
one burst; 2: one wave per SIMD, the stores under the next tile's K loop).  LAB_SCALE: standard deviation of the operands (the bench's
embeddings are small; unit-variance data makes the matrix pipe draw more power)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg
from seoul_tourism_recommendation_ngcf_amd import _lib
eng = pkg.engine
dev = torch.device("cuda:0")
n = int(os.environ.get("LAB_ROWS", 1_100_000))
scale = float(os.environ.get("LAB_SCALE", 0.05))
ws = eng.Workspace()
for d_in, d_out, with_carry in ((128, 128, True), (128, 128, False), (130, 128, True)):
    ld = (d_in + 31) // 32 * 32
    LE, E = ((torch.randn((n, ld), device=dev) * scale)[:, :d_in] for _ in range(2))
    W1, W2 = (torch.randn((d_out, d_in), device=dev) * 0.05 for _ in range(2))
    b1, b2 = (torch.randn((d_out,), device=dev) * 0.05 for _ in range(2))
    carry = torch.empty((n, d_out), device=dev) if with_carry else None
    norm = torch.empty((n, d_out), device=dev)
    ref = None
    for resident in tuple(int(x) for x in os.environ.get("LAB_VARIANTS", "1,2,1,2").split(",")):   # 3: two tiles per wave (lab)
        _lib.set_option("dense_resident", resident)
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
        ms = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = (norm.clone(), None if carry is None else carry.clone())
        same = torch.equal(norm, ref[0]) and (carry is None or torch.equal(carry, ref[1]))
        fl = 4.0 * n * d_in * d_out
        print(f"d_in {d_in} d_out {d_out} carry {with_carry} resident {resident}: {ms:7.4f} ms (with the 5 us pack)  {fl / ms / 1e9:6.1f} TFLOP/s  "
              f"{fl / ms / 1e9 / 157.3 * 100:5.1f} % of 157.3  bit-identical {same}", flush=True)
