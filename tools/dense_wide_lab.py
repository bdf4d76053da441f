"""Wide dense layers (256 / 512 output columns): layer_dense_tall_kernel (+ row_scale_kernel) and layer_dense_direct_kernel against
the staged layer_dense_kernel at Seoul-sized and large row counts."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seoul_tourism_recommendation_ngcf_amd as pkg  # noqa: E402


def _reload_options():
    """the library reads its NGCF_* variables once; re-read them after changing os.environ"""
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _lib.options_from_env()

eng = pkg.engine
dev = torch.device("cuda:0")


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


SHAPES = ((5940, 515, 512), (5940, 512, 512), (5940, 256, 256), (8192, 512, 512), (12288, 512, 512), (16384, 512, 512), (32768, 512, 512),
          (100_000, 512, 512), (100_000, 256, 256), (1_100_000, 256, 256), (1_100_000, 512, 512))
if os.environ.get("LAB_ROWS"):          # e.g. LAB_ROWS=8192,16384,32768: where the two kernels cross
    SHAPES = tuple((int(r), di, do) for r in os.environ["LAB_ROWS"].split(",") for di, do in ((512, 512), (256, 256)))
for n, d_in, d_out in SHAPES:
    ld = (d_in + 31) // 32 * 32
    sc = float(os.environ.get("LAB_SCALE", "1.0"))       # operand scale: small values keep the clock up (DVFS)
    LE = (torch.randn((n, ld), device=dev) * sc)[:, :d_in]
    E = (torch.randn((n, ld), device=dev) * sc)[:, :d_in]
    W1, W2 = (torch.randn((d_out, d_in), device=dev) * 0.1 for _ in range(2))
    b1, b2 = (torch.randn((d_out,), device=dev) * 0.1 for _ in range(2))
    carry = torch.empty((n, d_out), device=dev)
    norm = torch.empty((n, d_out), device=dev)
    ws = eng.Workspace()
    res = {}
    for name, direct, tall in (("tall", "0", "2"), ("direct", "2", "0"), ("staged", "0", "0")):
        os.environ["NGCF_DENSE_DIRECT"], os.environ["NGCF_DENSE_TALL"] = direct, tall
        _reload_options()
        res[name] = t(lambda: eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, ws))
    fl = 4.0 * n * d_in * d_out
    print(f"n={n} {d_in}->{d_out}: " + ", ".join(f"{k} {v*1e3:.1f} us ({fl/v/1e9:.1f} TF, {fl/v/1e9/157.3*100:.0f} %)" for k, v in res.items())
          + "  (each incl. the weight pack; tall incl. row_scale_kernel)", flush=True)
    del LE, E, carry, norm
