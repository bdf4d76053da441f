"""Where are the waves of one XCD in the table while the L2-swept kernel runs?  

Runs the swept kernel's DBG instantiation (NGCF_SWEPT_TRACE) on the two halves of the C3 graph: every wave stamps
(s_memrealtime, column it is gathering) every 2 chunks.  For the first sweep (row pass 0, slice 0) the script interpolates
every wave's column at common time points and prints, per XCD, how far apart the waves are in bytes of the 256-byte table
slice - the L2 is 4 MiB, so a spread well above that means a table row is fetched again for the late waves.
usage: python tools/swept_trace_lab.py [half=item|user|both]
"""
import os
import sys

import numpy as np
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
N, d = U + I, 128
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
parts = {"user": (rows[:nu], cols[:nu], vals[:nu], U), "item": (rows[nu:] - U, cols[nu:], vals[nu:], I)}
which = sys.argv[1] if len(sys.argv) > 1 else "both"
E = torch.randn((N, d), device=dev)
ws = eng.Workspace()
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)


def analyse(path, label):
    raw = np.fromfile(path, dtype=np.uint64)
    n_wg, waves, n_samp, win_cols, n_win, col_lo = (int(x) for x in raw[:6].view(np.int64))
    rec = raw[6:].reshape(n_wg * waves, 2 * n_samp + 2)
    xcc, cnt = rec[:, 0].astype(int), rec[:, 1].astype(int)
    t = rec[:, 2::2].astype(np.float64)
    tag = rec[:, 3::2]
    sweep, col = (tag >> np.uint64(32)).astype(int), (tag & np.uint64(0xFFFFFFFF)).astype(np.float64)
    print(f"== {label}: {n_wg} workgroups x {waves} waves, {n_win} windows of {win_cols} columns ({win_cols * 256 / 2**20:.1f} MiB), "
          f"samples per wave {cnt.min()}..{cnt.max()}")
    t0 = min(t[w, 0] for w in range(len(t)) if cnt[w] > 0)
    for x in range(8):
        ws_ = [w for w in range(len(t)) if xcc[w] == x and cnt[w] > 4]
        if not ws_:
            continue
        # first sweep only: samples with sweep index 0
        series = []
        for w in ws_:
            m = (np.arange(n_samp) < cnt[w]) & (sweep[w] == 0)
            if m.sum() > 3:
                series.append((t[w][m] - t0, col[w][m]))
        if not series:
            continue
        lo = max(s[0][0] for s in series)
        hi = min(s[0][-1] for s in series)
        if hi <= lo:
            print(f"  xcd {x}: no common time range")
            continue
        grid = np.linspace(lo, hi, 24)
        pos = np.stack([np.interp(grid, s[0], s[1]) for s in series]) * 256 / 2**20     # MiB into the slice, [wave, time]
        spread90 = np.percentile(pos, 95, axis=0) - np.percentile(pos, 5, axis=0)
        spread_all = pos.max(axis=0) - pos.min(axis=0)
        med = np.median(pos, axis=0)
        near = np.mean(np.abs(pos - med[None, :]) <= 1.5, axis=0)        # share of waves within +-1.5 MiB of the median
        dur_us = (hi - lo) / 100.0                                          # s_memrealtime ticks at 100 MHz
        if x == 0:      # decomposition on one XCD: between workgroups vs inside a workgroup, and who the stragglers are
            wg_of = np.array([w // waves for w in ws_ if ((np.arange(n_samp) < cnt[w]) & (sweep[w] == 0)).sum() > 3])
            mid = pos[:, len(grid) // 2]
            wgs = np.unique(wg_of)
            wg_med = np.array([np.median(mid[wg_of == g]) for g in wgs])
            med_of = {g: np.median(mid[wg_of == g]) for g in wgs}
            inside = mid - np.array([med_of[g] for g in wg_of])
            print(f"  xcd 0 at mid-sweep: std of workgroup medians {wg_med.std():.2f} MiB (min..max {wg_med.min() - np.median(mid):+.1f} .. "
                  f"{wg_med.max() - np.median(mid):+.1f}); std inside a workgroup {inside.std():.2f} MiB "
                  f"(p1 {np.percentile(inside, 1):+.1f}, p99 {np.percentile(inside, 99):+.1f})")
            widx = np.array([w % waves for w in ws_ if ((np.arange(n_samp) < cnt[w]) & (sweep[w] == 0)).sum() > 3])
            print("  xcd 0: mean offset from the workgroup median by wave number: " +
                  " ".join(f"{np.mean(inside[widx == q]):+.2f}" for q in range(waves)))
            # are the late waves late throughout?  rank correlation of the offset from the median at 1/4 and 3/4 of the sweep
            q1, q3 = pos[:, len(grid) // 4] - np.median(pos[:, len(grid) // 4]), pos[:, 3 * len(grid) // 4] - np.median(pos[:, 3 * len(grid) // 4])
            r1, r3 = np.argsort(np.argsort(q1)), np.argsort(np.argsort(q3))
            print(f"  xcd 0: rank correlation of a wave's offset at 1/4 and at 3/4 of the sweep: {np.corrcoef(r1, r3)[0, 1]:.2f}; "
                  f"offset p1/p5/p50/p95/p99 at mid-sweep: " + " ".join(f"{np.percentile(mid - np.median(mid), q):+.1f}" for q in (1, 5, 50, 95, 99)))
        print(f"  xcd {x}: {len(series):4d} waves, common span {dur_us:7.1f} us; spread p5..p95 median {np.median(spread90):6.1f} MiB "
              f"(max {spread90.max():6.1f}), min..max median {np.median(spread_all):6.1f} MiB; waves within +-1.5 MiB of the "
              f"median position: {100 * np.median(near):4.1f} %")
    # start skew of the workgroups and of the waves
    starts = np.array([t[w, 0] for w in range(len(t)) if cnt[w] > 0]) - t0
    ends = np.array([t[w, cnt[w] - 1] for w in range(len(t)) if cnt[w] > 0]) - t0
    print(f"  first stamp of the waves: {np.percentile(starts, 50) / 100:.1f} us median, {starts.max() / 100:.1f} us max; "
          f"last stamps {np.percentile(ends, 5) / 100:.1f} .. {ends.max() / 100:.1f} us")


for name, (r, c, v, nr) in parts.items():
    if which not in ("both", name):
        continue
    csr = eng.LaplacianCSR.from_coo(r, c, v, nr, N)
    csr.set_mode(2)
    out = torch.empty((nr, d), device=dev)
    for _ in range(3):
        eng.spmm(csr, E, out=out, ws=ws)
    torch.cuda.synchronize()
    trace = os.path.join(out_dir, f"swept_trace_{name}")
    os.environ["NGCF_SWEPT_TRACE"] = trace
    _reload_options()
    eng.spmm(csr, E, out=out, ws=ws)
    torch.cuda.synchronize()
    del os.environ["NGCF_SWEPT_TRACE"]
    _reload_options()
    analyse(trace + ".part0", f"{name} rows")
    os.remove(trace + ".part0")
