"""Times the row-wise / d-sliced kernels against the L2-swept kernel on the two halves of the C3 graph, sweeping the
column-window size (plan time) and the allowed lead (launch time)."""
import os, sys, time
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
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, int(os.environ.get("LAB_D", "128"))
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
parts = {"user rows": (rows[:nu], cols[:nu], vals[:nu], U), "item rows": (rows[nu:] - U, cols[nu:], vals[nu:], I)}
E = torch.randn((N, d), device=dev)
ws = eng.Workspace()
# windows as "KiB", "KiB:K" or "KiB:K:order" (K windows per synchronised sweep step; order = cols | rows inside a window)
_w = [x.split(":") for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2048,3072,4096,6144".split(","))]
windows = [(int(x[0]), x[1] if len(x) > 1 else "1", x[2] if len(x) > 2 else "cols") for x in _w]
leads = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2".split(","))]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, (r, c, v, nr) in parts.items():
    csr = eng.LaplacianCSR.from_coo(r, c, v, nr, N)
    out = torch.empty((nr, d), device=dev)
    for mode, label in ((1, "row-wise"), (0, "row-wise/d-sliced")):
        csr.set_mode(mode)
        ms = timeit(lambda: eng.spmm(csr, E, out=out, ws=ws))
        print(f"{name}: {label:18s} {ms:7.3f} ms  gather {v.numel() * d * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    ref = out.clone()
    for (kb, every, order), waves, cut in [(k_, w_, c_) for k_ in windows for w_ in (os.environ.get("LAB_WAVES", "0").split(","))
                                    for c_ in os.environ.get("LAB_CUT", "4").split(",")]:
        os.environ["NGCF_SWEPT_WINDOW_KB"] = str(kb)
        _reload_options()
        os.environ["NGCF_SWEPT_SYNC_EVERY"] = every
        _reload_options()
        os.environ["NGCF_SWEPT_ORDER"] = order
        _reload_options()
        os.environ["NGCF_SWEPT_WAVES"] = waves
        _reload_options()
        os.environ["NGCF_SWEPT_CUT"] = cut
        _reload_options()
        csr.set_mode(1)
        t0 = time.time()
        csr.set_mode(2)
        tb = time.time() - t0
        for lead in leads:
            os.environ["NGCF_SWEPT_LEAD"] = str(lead)
            _reload_options()
            # launch-time knobs of the wave priorities (LAB_PRIO: lag thresholds in KiB, LAB_GRADED: 0 / 1)
            for prio in os.environ.get("LAB_PRIO", os.environ.get("NGCF_SWEPT_PRIO_KB", "512")).split(","):
                for graded in os.environ.get("LAB_GRADED", "0").split(","):
                    os.environ["NGCF_SWEPT_PRIO_KB"], os.environ["NGCF_SWEPT_PRIO_GRADED"] = prio, graded
                    _reload_options()
                    ms = timeit(lambda: eng.spmm(csr, E, out=out, ws=ws))
                    err = float((out - ref).abs().max())
                    print(f"{name}: swept window {kb:5d} KiB x{every} {order} waves {waves:>2s} cut T/{cut} lead {lead:2d} prio {prio:>4s} KiB "
                          f"graded {graded} {ms:7.3f} ms  gather {v.numel() * d * 4 / ms / 1e9:6.2f} TB/s  (plan {tb:.1f} s, max diff {err:.1e})",
                          flush=True)
