"""GPU tests of ngcf_csr_filter / ngcf_csr_filter_remap: the reference's per-layer `sparse_dropout` (NGCF.py:93-100,124-126: a COO
tensor rebuilt from `indices[:, mask]`) as a device compaction of the CSR, and of its transpose through an entry map.
Integer / index work: bit-exact against torch indexing on the same mask."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _pkg():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    return pkg


def _arrays(csr, nnz=None):
    """(rowptr, colidx, vals) of a CSR handle as host tensors (colidx / vals cut to the stored entries)."""
    from seoul_tourism_recommendation_ngcf_amd import _lib
    lib = _lib.load()
    n = csr.n_rows
    rp = torch.empty(n + 1, dtype=torch.int64)
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(rp.data_ptr()), C.c_void_p(lib.ngcf_csr_rowptr(csr._h)), C.c_size_t(8 * (n + 1)), 2)
    m = int(rp[-1]) if nnz is None else nnz
    ci = torch.empty(max(m, 1), dtype=torch.int32)
    va = torch.empty(max(m, 1), dtype=torch.float32)
    if m:
        hip.hipMemcpy(C.c_void_p(ci.data_ptr()), C.c_void_p(lib.ngcf_csr_colidx(csr._h)), C.c_size_t(4 * m), 2)
        hip.hipMemcpy(C.c_void_p(va.data_ptr()), C.c_void_p(lib.ngcf_csr_vals(csr._h)), C.c_size_t(4 * m), 2)
    return rp, ci[:m], va[:m]


def _random_coo(n_rows, n_cols, nnz, heavy, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randint(0, n_rows, (nnz,), generator=g)
    for r, k in heavy:
        rows = torch.cat([rows, torch.full((k,), r)])
    rows = torch.sort(rows).values
    cols = torch.randint(0, n_cols, (rows.numel(),), generator=g)
    vals = torch.randn(rows.numel(), generator=g)
    return rows, cols, vals


@pytest.mark.parametrize("n_rows,n_cols,nnz,heavy,p", [(1, 1, 1, (), 0.5), (50, 7, 0, (), 0.3), (3000, 200, 40000, ((5, 9000), (2999, 700)), 0.3),
                                                        (5940, 5940, 300000, ((5900, 5000), (5901, 4100), (5902, 258)), 0.3),
                                                        (100, 100000, 250000, (), 0.99), (20000, 300, 100000, (), 0.0)])
def test_filter_equals_torch_masking(n_rows, n_cols, nnz, heavy, p, dev):
    eng = _pkg().engine
    rows, cols, vals = _random_coo(n_rows, n_cols, nnz, heavy, n_rows + nnz)
    csr = eng.LaplacianCSR.from_coo(rows.to(dev), cols.to(dev), vals.to(dev), n_rows, n_cols)
    g = torch.Generator().manual_seed(3)
    out = None
    E = torch.randn((n_cols, 68), generator=g).to(dev)
    for rep in range(3):                                       # the destination is re-used: same object, new mask
        keep = torch.rand(rows.numel(), generator=g) >= p
        out = csr.filtered(keep.to(dev), None, int(keep.sum()), reuse=out)
        assert out.nnz == int(keep.sum()) and out.n_rows == n_rows and out.n_cols == n_cols
        rp, ci, va = _arrays(out)
        want_rp = torch.zeros(n_rows + 1, dtype=torch.int64)
        want_rp[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n_rows), 0)
        assert torch.equal(rp, want_rp)
        assert torch.equal(ci, cols[keep].to(torch.int32)) and torch.equal(va, vals[keep])
        # and the product on the thinned copy (segments shared with the source, some now empty or short)
        ref = eng.LaplacianCSR.from_coo(rows[keep].to(dev), cols[keep].to(dev), vals[keep].to(dev), n_rows, n_cols)
        for d in (68, 64, 5):
            got, want = eng.spmm(out, E[:, :d]), eng.spmm(ref, E[:, :d])
            scale = max(float(want.abs().max()), 1.0) if want.numel() else 1.0
            assert float((got - want).abs().max()) <= 2e-6 * scale if want.numel() else True
    # without the count the object reports an upper bound and still multiplies correctly
    keep = torch.rand(rows.numel(), generator=g) >= p
    out = csr.filtered(keep.to(dev), None, -1, reuse=out)
    assert out.nnz == csr.nnz
    rp, ci, va = _arrays(out)
    assert int(rp[-1]) == int(keep.sum()) and torch.equal(ci, cols[keep].to(torch.int32))


def test_filter_chain_with_transpose_maps(dev):
    """Three cumulative thinnings (NGCF.py:126) of L and of L^T: L^T is thinned with the flags drawn for L through the entry map,
    and the map is carried to the next layer by ngcf_csr_filter_remap; every thinned transpose equals the transpose of the thinned L."""
    pkg = _pkg()
    eng = pkg.engine
    from seoul_tourism_recommendation_ngcf_amd import _lib
    lib = _lib.load()
    coo = pkg.graphs.synthetic_bipartite(3000, 150, 60000, seed=5, device=dev)
    N = 3150
    rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
    L = eng.LaplacianCSR.from_coo(rows, cols, vals, N, N)
    order = torch.sort(cols, stable=True).indices
    Lt = eng.LaplacianCSR.from_coo(cols[order], rows[order], vals[order], N, N)
    emap = order.to(torch.int32)
    g = torch.Generator().manual_seed(11)
    src, src_t = L, Lt
    r, c, v = rows.cpu(), cols.cpu(), vals.cpu()
    for k in range(3):
        keep = torch.rand(src.nnz, generator=g) >= 0.3
        kd = keep.to(dev)
        n_src_t = src_t.nnz
        nxt = src.filtered(kd, None, int(keep.sum()))
        nxt_t = src_t.filtered(kd, emap, int(keep.sum()))
        r, c, v = r[keep], c[keep], v[keep]
        o = torch.sort(c, stable=True).indices
        rp, ci, va = _arrays(nxt_t)
        want_rp = torch.zeros(N + 1, dtype=torch.int64)
        want_rp[1:] = torch.cumsum(torch.bincount(c, minlength=N), 0)
        assert torch.equal(rp, want_rp) and torch.equal(ci, r[o].to(torch.int32)) and torch.equal(va, v[o])
        new_map = torch.empty(max(nxt.nnz, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.ngcf_csr_filter_remap(nxt_t._h, C.c_void_p(kd.data_ptr()), C.c_void_p(emap.data_ptr()), n_src_t,
                                             C.c_void_p(nxt.filter_pos), C.c_void_p(new_map.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.equal(new_map[:nxt.nnz].cpu(), o.to(torch.int32))
        src, src_t, emap = nxt, nxt_t, new_map[:nxt.nnz].contiguous()


def test_filter_argument_errors(dev):
    eng = _pkg().engine
    csr = eng.LaplacianCSR.from_coo(torch.tensor([0, 1], device=dev), torch.tensor([1, 0], device=dev), torch.ones(2, device=dev), 2, 2)
    with pytest.raises(RuntimeError, match="keep flags"):
        csr.filtered(torch.ones(3, dtype=torch.bool, device=dev))
    with pytest.raises(RuntimeError, match="ROCm device"):
        csr.filtered(torch.ones(2, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="nnz_kept"):
        csr.filtered(torch.ones(2, dtype=torch.bool, device=dev), None, 5)
