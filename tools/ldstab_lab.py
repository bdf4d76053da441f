"""spmm_ldstab_kernel (table slice in LDS) against the d-sliced row-wise kernel (NGCF_NO_LDSTAB=1) by table size: rows that gather
from n_tab table rows, 75 entries each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seoul_tourism_recommendation_ngcf_amd as pkg


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


g = torch.Generator(device=dev).manual_seed(1)
for n_rows in (5120, 200_000):
    for n_tab in (100, 256, 512):
        for d in (64, 128, 512):
            deg = 75
            rows = torch.arange(n_rows, device=dev).repeat_interleave(deg)
            cols = torch.randint(0, n_tab, (n_rows * deg,), generator=g, device=dev)
            vals = torch.randn((n_rows * deg,), generator=g, device=dev)
            csr = eng.LaplacianCSR.from_coo(rows, cols, vals, n_rows, n_tab)
            X = torch.randn((n_tab, d), device=dev)
            ws = eng.Workspace()
            r = {}
            for flag in ("", "1"):
                if flag:
                    os.environ["NGCF_NO_LDSTAB"] = flag
                    _reload_options()
                else:
                    os.environ.pop("NGCF_NO_LDSTAB", None)
                r[flag] = t(lambda: eng.spmm(csr, X, ws=ws))
            os.environ.pop("NGCF_NO_LDSTAB", None)
            print(f"rows={n_rows} n_tab={n_tab} d={d}: table in LDS {r['']*1e3:.1f} us, d-sliced from L2 {r['1']*1e3:.1f} us", flush=True)
