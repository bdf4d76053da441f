"""Diagnostic build of the library (cycle stamps in spmm_swept_kernel): where does a block iteration go?"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seoul_tourism_recommendation_ngcf_amd import _build
_build.LIB = os.path.join(ROOT, "tools", "ngcf_stamped.so")      # load the stamped build instead
_build.needs_build = lambda: False
import seoul_tourism_recommendation_ngcf_amd as pkg
from seoul_tourism_recommendation_ngcf_amd import _lib
eng = pkg.engine
lib = _lib.load()
lib.lab_swept_debug_ptr.restype = C.c_void_p
lib.lab_swept_debug_ptr.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
U, I, M = 1_000_000, 100_000, 50_000_000
coo = pkg.graphs.synthetic_bipartite(U, I, M, seed=2603, device=dev)
N, d = U + I, 128
rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
nu = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
E = torch.randn((N, d), device=dev)
ws = eng.Workspace()
for kb in [int(x) for x in sys.argv[1].split(",")]:
    os.environ["NGCF_SWEPT_BLOCK_KB"] = str(kb)
    csr = eng.LaplacianCSR.from_coo(rows[nu:] - U, cols[nu:], vals[nu:], I, N)
    csr.set_mode(2)
    out = torch.empty((I, d), device=dev)
    for _ in range(2):
        eng.spmm(csr, E, out=out, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.spmm(csr, E, out=out, ws=ws); e1.record(); torch.cuda.synchronize()
    ptr = lib.lab_swept_debug_ptr(csr._h)
    buf = (C.c_uint64 * (256 * 8))()
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(buf, C.c_void_p(ptr), 256 * 8 * 8, 2)
    t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
    m = t.mean(0)
    print(f"block {kb} KiB: {e0.elapsed_time(e1):.2f} ms; mean cycles per WG(thread0): blkptr {m[0]:.3g} entries {m[1]:.3g} "
          f"gather {m[2]:.3g} accum {m[3]:.3g} barrier {m[4]:.3g} total {m[5]:.3g}; xcc counts "
          f"{torch.bincount(t[:,6].long(), minlength=8).tolist()} max rank {int(t[:,7].max())}", flush=True)
    tot = t[:, 5]
    order = torch.argsort(tot, descending=True)
    print("   total cycles per WG: min %.3g median %.3g max %.3g; slowest WGs %s" % (tot.min(), tot.median(), tot.max(), order[:8].tolist()))
    print("   work (entries+gather+accum) of slowest: %s ; of median WG: %.3g" % ((t[order[:4], 1:4].sum(1)).tolist(), t[:, 1:4].sum(1).median()))
    print("   per-phase of slowest WG:", t[order[0]].tolist())
