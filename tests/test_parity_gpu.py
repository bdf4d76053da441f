"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Tolerances (calibrated on the reference itself):
  * all_E / propagation outputs vs the reference CPU forward: atol=2e-5, rtol=2e-3 (fp32; the summation
    order inside a row and the fused `(LE+E).W1` differ from the reference's op order);
  * "HIP error vs fp64 <= 4x the reference's own fp32 error vs fp64";
  * gathers u/p/n: BIT-EXACT copies of the engine's own all_E rows;
  * feature injection: bit-exact; BPR loss: rtol=1e-5.
Nothing here reads /root/reference: expected values come from tests/golden/*.npz and oracle/.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import ngcf_oracle as orc
from conftest import FWD_CASES, load_golden
from golden_util import batch_of, ctor_args, lap_list_of, layer_params, sd_of

pytestmark = pytest.mark.gpu

ATOL, RTOL = 2e-5, 2e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _pkg():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    return pkg


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def c_spmm(oracle_clib, rowptr, col32, vals, E):
    out = np.empty((len(rowptr) - 1, E.shape[1]), np.float32)
    fn = oracle_clib.ngcf_oracle_spmm_csr_f32
    fn.restype = None
    E = np.ascontiguousarray(E)
    fn(_p(rowptr), _p(col32), _p(vals), C.c_int64(len(rowptr) - 1), _p(E), C.c_int64(E.shape[1]),
       C.c_int(E.shape[1]), _p(out), C.c_int64(E.shape[1]))
    return out


def random_csr(rng, n_rows, n_cols, mean_deg, heavy=(), empty=()):
    deg = rng.poisson(mean_deg, n_rows)
    for r, k in heavy:
        deg[r] = k
    for r in empty:
        deg[r] = 0
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    cols = rng.integers(0, n_cols, nnz).astype(np.int64)
    vals = rng.normal(0, 0.3, nnz).astype(np.float32)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), deg)
    return rowptr, rows, cols, vals


# --------------------------------------------------------------------------------------------
# golden fixtures through the nn.Module surface
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FWD_CASES)
@pytest.mark.parametrize("lap_on", ["cpu", "cuda"])
def test_module_forward_matches_golden(name, lap_on, dev):
    pkg = _pkg()
    g = load_golden(name)
    sd, b = sd_of(g), batch_of(g)
    laps = lap_list_of(g, "cpu" if lap_on == "cpu" else dev)          # demo.py keeps lap_list on the CPU
    model = pkg.NGCF(**ctor_args(g, laps, dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()
    n_user = int(g["meta"][0])
    batch = {k: (v.to(dev) if k != "neg_item" or v.numel() else v) for k, v in b.items()}
    u, p, n = model(node_flag=False, **batch)
    torch.cuda.synchronize()
    # in-place mutation of user_embedding.weight: bit-exact
    assert np.array_equal(model.user_embedding.weight.detach().cpu().numpy(), g["out_user_weight_after"])
    all_E = torch.cat((model.all_users_emb, model.all_items_emb), 0).detach().cpu().numpy()
    assert all_E.shape == g["out_all_E"].shape
    assert np.array_equal(all_E[:, :int(g["meta"][2])],
                          np.concatenate([g["out_user_weight_after"], sd["item_embedding.weight"].numpy()]))
    np.testing.assert_allclose(all_E, g["out_all_E"], atol=ATOL, rtol=RTOL)
    # error vs fp64 no worse than 4x the reference's own
    yi = int(g["year_idx"])
    w1, b1, w2, b2 = layer_params(sd, len(g["layers"]))
    f64 = orc.propagate_f64(g[f"lap{yi}_rows"], g[f"lap{yi}_cols"], g[f"lap{yi}_vals"], all_E[:, :int(g["meta"][2])],
                            [w.numpy() for w in w1], [x.numpy() for x in b1], [w.numpy() for w in w2],
                            [x.numpy() for x in b2])
    ref_err = np.abs(g["out_all_E"] - f64).max()
    assert np.abs(all_E - f64).max() <= 4 * ref_err + 1e-7
    # gathers: bit-exact copies of the engine's own rows, and within tolerance of the reference's
    assert torch.equal(u, model.all_users_emb[batch["u_id"]])
    assert torch.equal(p, model.all_items_emb[batch["pos_item"]])
    np.testing.assert_allclose(u.detach().cpu().numpy(), g["out_u"], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(p.detach().cpu().numpy(), g["out_p"], atol=ATOL, rtol=RTOL)
    if g["out_n"].size:
        assert torch.equal(n, model.all_items_emb[batch["neg_item"]])
        np.testing.assert_allclose(n.detach().cpu().numpy(), g["out_n"], atol=ATOL, rtol=RTOL)
    else:
        assert n.numel() == 0 and n.device.type == "cpu"              # `torch.empty(0)`, NGCF.py:153
    assert u.shape == (len(b["u_id"]), all_E.shape[1])
    assert model.all_users_emb.shape[0] == n_user
    # second call: the injection is idempotent for emb_ratio == 1, outputs repeat bit for bit
    if float(g["meta"][3]) == 1.0:
        u2, _, _ = model(node_flag=False, **batch)
        assert torch.equal(u2, u)


def test_module_node_dropout_reference_mode_matches_golden_edge_sets(dev):
    """node_flag=True in eval mode: cumulative, unscaled node dropout drawn from the CPU generator (NGCF.py:93-100)."""
    pkg = _pkg()
    g = load_golden("fwd_train_dropout")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()                                   # message dropout off, node dropout on: experiment.py:72 leftover
    batch = {k: v.to(dev) for k, v in b.items()}
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    model(node_flag=True, **batch)
    got = torch.cat((model.all_users_emb, model.all_items_emb), 0).detach().cpu().numpy()
    # oracle with the same CPU generator state
    user_w = torch.from_numpy(g["out_user_weight_after"])
    w1, b1, w2, b2 = layer_params(sd, 3)
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    want = orc.propagate_torch(lap_list_of(g)[int(g["year_idx"])], user_w, sd["item_embedding.weight"], w1, b1, w2, b2,
                               mess_dropout=None, training=False, node_dropout=float(g["meta"][5]), node_flag=True)
    np.testing.assert_allclose(got, want.numpy(), atol=ATOL, rtol=RTOL)


def test_module_train_mode_forward_matches_golden_masks(dev):
    """Training-mode forward (NGCF.py:123-142): node dropout AND message dropout, both in "reference" mode, drawn from
    torch's default CPU generator in the reference's order (per layer: node mask, then message noise).  With the fixture's
    generator state the zero pattern of every layer block is bit-identical to the reference's and the values agree with
    the reference's train-mode all_E at the forward tolerance.  Two forwards in a row consume the generator exactly as
    two reference forwards do."""
    pkg = _pkg()
    g = load_golden("fwd_train_dropout")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.train()
    assert model.mess_dropout_mode == "reference" and model.node_dropout_mode == "reference"
    batch = {k: v.to(dev) for k, v in b.items()}
    # the oracle on this machine's torch, same generator state, two forwards in a row
    user_w = torch.from_numpy(g["out_user_weight_after"])
    w1, b1, w2, b2 = layer_params(sd, 3)
    L = lap_list_of(g)[int(g["year_idx"])]
    mess, pn = [float(x) for x in g["mess"]], float(g["meta"][5])
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    want = [orc.propagate_torch(L, user_w, sd["item_embedding.weight"], w1, b1, w2, b2, mess_dropout=mess, training=True,
                                node_dropout=pn, node_flag=True, return_carry=True) for _ in range(2)]
    state_after = torch.get_rng_state()
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    with torch.no_grad():
        got = []
        for _ in range(2):
            model(node_flag=True, **batch)
            got.append(torch.cat((model.all_users_emb, model.all_items_emb), 0).detach().cpu().numpy())
    assert torch.equal(torch.get_rng_state(), state_after)            # same number of draws as the reference path
    d0 = int(g["meta"][2])
    for (w_all, w_carries), mine in zip(want, got):
        off = d0
        for c in w_carries:
            blk = mine[:, off:off + c.shape[1]]
            assert np.array_equal(blk == 0, c.numpy() == 0)            # bit-identical message + node masks
            off += c.shape[1]
        np.testing.assert_allclose(mine, w_all.numpy(), atol=ATOL, rtol=RTOL)
    # and against the reference's own outputs stored in the fixture, wherever this host's torch draws the same Bernoulli
    # stream as the machine that generated it (the CPU bernoulli kernel is chosen by build and CPU vendor)
    same_stream = all(np.array_equal(want[0][1][k].numpy() == 0, g[f"carry_{k}"] == 0) for k in range(3))
    if same_stream:
        np.testing.assert_allclose(got[0], g["out_all_E"], atol=ATOL, rtol=RTOL)
        off = d0
        for k in range(3):
            assert np.array_equal(got[0][:, off:off + g[f"carry_{k}"].shape[1]] == 0, g[f"carry_{k}"] == 0)
            off += g[f"carry_{k}"].shape[1]


def test_reference_mode_message_dropout_gradients(dev):
    """Backward through the host-drawn noise tensor: every parameter gradient vs torch autograd through the oracle."""
    pkg = _pkg()
    g = load_golden("fwd_train_dropout")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.train()
    batch = {k: v.to(dev) for k, v in b.items()}
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    u, p, n = model(node_flag=True, **batch)
    loss = pkg.BPR(0.025, len(b["u_id"]))(u, p, n)
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(("w1_list", "w2_list", "item_emb"))}
    uw = torch.from_numpy(g["out_user_weight_after"]).clone().requires_grad_(True)
    w1, b1, w2, b2 = ([leaves[f"{n_}.{k}.{t}"] for k in range(3)]
                      for n_, t in (("w1_list", "weight"), ("w1_list", "bias"), ("w2_list", "weight"), ("w2_list", "bias")))
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    all_E = orc.propagate_torch(lap_list_of(g)[int(g["year_idx"])], uw, leaves["item_embedding.weight"], w1, b1, w2, b2,
                                mess_dropout=[float(x) for x in g["mess"]], training=True, node_dropout=float(g["meta"][5]),
                                node_flag=True)
    ou, op, on = orc.gather_torch(all_E, int(g["meta"][0]), b["u_id"], b["pos_item"], b["neg_item"])
    want_loss = orc.bpr_torch(ou, op, on, 0.025, len(b["u_id"]))
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    named = dict(model.named_parameters())
    for k, leaf in list(leaves.items()) + [("user_embedding.weight", uw)]:
        scale = float(leaf.grad.abs().max())
        np.testing.assert_allclose(named[k].grad.cpu().numpy(), leaf.grad.numpy(), atol=2e-3 * scale + 1e-9, rtol=2e-3, err_msg=k)


# --------------------------------------------------------------------------------------------
# SpMM kernel vs the plain-C oracle
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [4, 16, 64, 65, 96, 128, 130, 200, 256, 512])
def test_spmm_matches_c_oracle(d, dev, oracle_clib):
    eng = _pkg().engine
    rng = np.random.default_rng(d)
    n_rows, n_cols = 777, 901
    rowptr, rows, cols, vals = random_csr(rng, n_rows, n_cols, 9, heavy=[(5, 700), (400, 1500), (776, 64), (3, 65)],
                                          empty=[0, 1, 500])
    E = rng.normal(0, 0.5, (n_cols, d)).astype(np.float32)
    want = c_spmm(oracle_clib, rowptr, cols.astype(np.int32), vals, E)
    csr = eng.LaplacianCSR.from_coo(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev),
                                    torch.from_numpy(vals).to(dev), n_rows, n_cols)
    assert (csr.n_rows, csr.n_cols, csr.nnz) == (n_rows, n_cols, len(vals))
    Ed = torch.from_numpy(E).to(dev)
    for seg in (512, 64):                     # default plan, then force row segmentation of the long rows
        csr.plan(seg)
        assert csr.n_segments == sum(-(-int(k) // seg) for k in np.diff(rowptr) if k > seg)
        got = eng.spmm(csr, Ed).detach().cpu().numpy()
        np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL)
        assert np.all(got[[0, 1, 500]] == 0)                      # empty rows are written as zeros
    if d % 64 == 0:
        # the L2-swept kernel, forced onto this small matrix; rows longer than a quarter of a wave task's share are
        # dealt to several pieces -> partial sums + fix-up
        csr.set_mode(3)
        assert csr.swept_rows == 0                                # mode 3: too small to pay, stays row-wise
        csr.set_mode(2)
        assert csr.swept_rows == n_rows
        got = eng.spmm(csr, Ed).detach().cpu().numpy()
        np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL)
        assert np.all(got[[0, 1, 500]] == 0)
        got2 = eng.spmm(csr, Ed).detach().cpu().numpy()
        assert np.array_equal(got, got2)                          # deterministic: fixed summation order
        big = torch.zeros((n_cols, d + 24), device=dev)           # strided operand through the swept kernel
        big[:, 8:8 + d] = Ed
        got3 = eng.spmm(csr, big[:, 8:8 + d]).detach().cpu().numpy()
        assert np.array_equal(got, got3)
        csr.plan(64)                                              # re-planning keeps the mode and rebuilds the parts
        assert csr.swept_rows == n_rows
        np.testing.assert_allclose(eng.spmm(csr, Ed).detach().cpu().numpy(), want, atol=ATOL, rtol=RTOL)
        csr.set_mode(1)
        assert csr.swept_rows == 0
    if d % 4 != 0:
        # 16-byte aligned rows at a width that is not a multiple of 4 (the padded first-layer input at embed_size 65 /
        # 130): wide panel on the float4 kernels - or the swept kernel - plus a narrow scalar tail panel
        pad = torch.zeros((n_cols, (d + 3) // 4 * 4 + 4), device=dev)
        pad[:, :d] = Ed
        for mode in (0, 2):
            csr.set_mode(mode)
            got = eng.spmm(csr, pad[:, :d]).detach().cpu().numpy()
            np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL)
        csr.set_mode(1)
    # a column-sliced (strided) operand, as the engine uses for all_E blocks
    big = torch.zeros((n_cols, d + 24), device=dev)
    big[:, 8:8 + d] = Ed
    got = eng.spmm(csr, big[:, 8:8 + d]).detach().cpu().numpy()
    np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL)


def test_default_segment_length_follows_the_matrix_size(dev, oracle_clib):
    """A Seoul-shaped product (5 840 user rows of ~75 entries, 100 item rows of ~4 400: BASELINE configs 0-1) is latency-bound on
    a few hundred waves if its long rows are cut every 2 048 entries; the constructors plan with the power of two in
    [64, 2048] that gives >= 4 096 segments (256 at 876 K entries), and the result stays the oracle's."""
    eng = _pkg().engine
    rng = np.random.default_rng(7)
    U, I, per_user = 5840, 100, 75
    u = np.repeat(np.arange(U, dtype=np.int64), per_user)
    i = np.concatenate([rng.choice(I, per_user, replace=False) for _ in range(U)]).astype(np.int64)
    rows = np.concatenate([u, i + U])
    cols = np.concatenate([i + U, u])
    vals = rng.normal(0, 0.3, rows.size).astype(np.float32)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    N = U + I
    csr = eng.LaplacianCSR.from_coo(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev),
                                    torch.from_numpy(vals).to(dev), N, N)
    lens = np.bincount(rows, minlength=N)
    assert csr.nnz == 876_000 and csr.n_segments == sum(-(-int(k) // 256) for k in lens if k > 256)
    rowptr = np.zeros(N + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    # the user rows gather from 100 table rows: their blocks run on the table-in-LDS kernel (spmm_ldstab_kernel), whole and
    # partial 64-float slices; mode 1 keeps everything on the plain row-wise kernel
    for d in (64, 4, 96, 512, 200, 516, 768, 772):
        E = rng.normal(0, 0.5, (N, d)).astype(np.float32)
        want = c_spmm(oracle_clib, rowptr, cols.astype(np.int32), vals, E)
        Ed = torch.from_numpy(E).to(dev)
        got = eng.spmm(csr, Ed).cpu().numpy()
        # rows of ~4 400 terms of size ~0.15 summed in fp32, the oracle sequentially, the engine segment by segment: absolute
        # tolerance for sums of that length (the error of either order against fp64 is ~3e-5 where the terms cancel)
        np.testing.assert_allclose(got, want, atol=1e-4, rtol=RTOL)
        csr.set_mode(1)
        plain = eng.spmm(csr, Ed).cpu().numpy()
        csr.set_mode(0)
        np.testing.assert_allclose(plain, want, atol=1e-4, rtol=RTOL)
        np.testing.assert_allclose(got, plain, atol=2e-6, rtol=2e-5)
    small = eng.LaplacianCSR.from_coo(torch.from_numpy(rows[:5000]).to(dev), torch.from_numpy(cols[:5000]).to(dev),
                                      torch.from_numpy(vals[:5000]).to(dev), N, N)
    assert small.n_segments == sum(-(-int(k) // 64) for k in np.bincount(rows[:5000], minlength=N) if k > 64)


@pytest.mark.parametrize("d,order,lpe", [(64, "cols", 16), (128, "cols", 16), (256, "cols", 16), (128, "rows", 16), (576, "cols", 16),
                                         (128, "cols", 32), (256, "cols", 32), (128, "rows", 32), (384, "cols", 32)])
def test_spmm_sliced_and_swept_kernels_large_matrix(d, order, lpe, dev, lib_options):
    """Bipartite matrix: the user rows (small gathered table) run d-sliced, the item rows unsliced; the L2-swept
    kernel (forced; both layouts of a wave's entry list inside a window) and the plain row-wise kernel must give the
    same product."""
    lib_options(swept_order_rows=int(order == "rows"), swept_lpe=lpe)      # lpe 32: 128-float slices, two entries per round
    pkg = _pkg()
    eng = pkg.engine
    coo = pkg.graphs.synthetic_bipartite(150000, 12000, 2600000, seed=33, device=dev)
    N = coo["n_user"] + coo["n_item"]
    csr = eng.LaplacianCSR.from_coo(coo["rows"], coo["cols"], coo["vals"], N, N)
    X = torch.randn((N, d), generator=torch.Generator(device=dev).manual_seed(d), device=dev)
    auto = eng.spmm(csr, X)                    # mode 0: row groups, sliceable ones d-sliced
    csr.set_mode(2)
    swept = eng.spmm(csr, X)
    csr.set_mode(1)
    rowwise = eng.spmm(csr, X)                 # mode 1: no slicing
    scale = float(rowwise.abs().max())
    assert float((swept - rowwise).abs().max()) <= 2e-6 * max(scale, 1.0)
    assert float((auto - rowwise).abs().max()) <= 2e-6 * max(scale, 1.0)
    if d == 64:
        # the swept kernel gathers with 32-bit byte offsets: a table spanning 2..4 GiB (offsets above 2^31) must
        # still be addressed correctly, and one spanning more than 4 GiB must fall back to the row-wise kernels
        csr.set_mode(2)
        for ld in (3400, 6700):
            wide = torch.zeros((N, ld), device=dev)
            wide[:, 1000:1000 + d] = X
            assert torch.equal(eng.spmm(csr, wide[:, 1000:1000 + d]), swept if ld == 3400 else auto)
            del wide
        csr.set_mode(1)
    rows = torch.cat([torch.randint(0, N, (64,), device=dev), torch.tensor([coo["n_user"], coo["n_user"] + 1], device=dev)])
    rp = torch.searchsorted(coo["rows"], torch.stack([rows, rows + 1]))
    for r, (lo, hi) in zip(rows.tolist(), rp.T.tolist()):
        want = (coo["vals"][lo:hi].double()[:, None] * X[coo["cols"][lo:hi]].double()).sum(0)
        np.testing.assert_allclose(swept[r].detach().cpu().numpy(), want.detach().cpu().numpy(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("lpe,d", [(16, 64), (32, 128)])
def test_spmm_swept_many_rows_several_row_passes(lpe, d, dev, lib_options):
    """More output rows than the chip's LDS holds at once: several row passes (16 waves x 36 rows per workgroup; the 8 x 72
    shape of round 1 is a lab instantiation, compiled only with -DNGCF_LAB).  Also a group with a handful of very long rows
    (cut into strided pieces) and duplicate entries inside a row."""
    pkg = _pkg()
    eng = pkg.engine
    coo = pkg.graphs.synthetic_bipartite(330000, 3000, 2000000, seed=35, device=dev)
    N = coo["n_user"] + coo["n_item"]
    rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
    dup = torch.arange(0, rows.numel(), 1000, device=dev)         # every 1000th entry stored twice
    rows, cols, vals = torch.cat([rows, rows[dup]]), torch.cat([cols, cols[dup]]), torch.cat([vals, vals[dup]])
    order = torch.sort(rows, stable=True).indices
    rows, cols, vals = rows[order], cols[order], vals[order]
    lib_options(swept_lpe=lpe)
    csr = eng.LaplacianCSR.from_coo(rows, cols, vals, N, N)
    X = torch.randn((N, d), generator=torch.Generator(device=dev).manual_seed(5), device=dev)
    rowwise = eng.spmm(csr, X)
    csr.set_mode(2)
    assert csr.swept_rows == N
    swept = eng.spmm(csr, X)
    scale = float(rowwise.abs().max())
    assert float((swept - rowwise).abs().max()) <= 2e-6 * max(scale, 1.0)
    assert torch.equal(swept, eng.spmm(csr, X))


@pytest.mark.parametrize("n_rows,n_cols,deg,heavy", [(1, 1, 1, ()), (5, 3, 40, ()), (2000, 70000, 1, ()),
                                                      (300, 50000, 3, ((0, 120000), (299, 9000))), (40, 17, 0, ((7, 5000),))])
@pytest.mark.parametrize("lpe", [16, 32])
def test_spmm_swept_degenerate_shapes(n_rows, n_cols, deg, heavy, lpe, dev, oracle_clib, lib_options):
    """The swept plan forced onto shapes it was not made for: one entry, a few columns with thousands of duplicates per
    row (rounds = longest row), one entry per row over many windows, a row far longer than everything else together
    (strided pieces + fix-up), everything in one row."""
    eng = _pkg().engine
    rng = np.random.default_rng(n_rows + n_cols)
    rowptr, rows, cols, vals = random_csr(rng, n_rows, n_cols, deg, heavy=heavy)
    E = rng.normal(0, 0.5, (n_cols, 4 * lpe)).astype(np.float32)     # one slice of the part's geometry
    want = c_spmm(oracle_clib, rowptr, cols.astype(np.int32), vals, E)
    lib_options(swept_lpe=lpe)
    csr = eng.LaplacianCSR.from_coo(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev),
                                    torch.from_numpy(vals).to(dev), n_rows, n_cols)
    csr.set_mode(2)
    assert csr.swept_rows == (n_rows if len(vals) else 0)
    got = eng.spmm(csr, torch.from_numpy(E).to(dev)).detach().cpu().numpy()
    scale = max(float(np.abs(want).max()), 1.0)
    assert float(np.abs(got - want).max()) <= 2e-5 * scale           # long rows: summation order differs from the oracle's


def test_spmm_unsorted_and_duplicate_coo(dev, oracle_clib):
    eng = _pkg().engine
    rng = np.random.default_rng(3)
    n = 300
    rowptr, rows, cols, vals = random_csr(rng, n, n, 6)
    # duplicates: repeat the first 50 entries; then shuffle everything (host sort path)
    rows2, cols2, vals2 = (np.concatenate([a, a[:50]]) for a in (rows, cols, vals))
    perm = rng.permutation(len(rows2))
    E = rng.normal(0, 0.5, (n, 64)).astype(np.float32)
    want = orc.spmm_coo_f64(rows2, cols2, vals2, n, E)
    csr = eng.LaplacianCSR.from_coo(*(torch.from_numpy(a[perm]).to(dev) for a in (rows2, cols2, vals2)), n, n)
    assert csr.nnz == len(rows2)                       # duplicates kept as separate entries
    got = eng.spmm(csr, torch.from_numpy(E).to(dev)).detach().cpu().numpy()
    np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL)


def test_csr_errors(dev):
    eng = _pkg().engine
    r = torch.tensor([0, 1, 5], device=dev)
    c = torch.tensor([0, 1, 2], device=dev)
    v = torch.ones(3, device=dev)
    with pytest.raises(IndexError):
        eng.LaplacianCSR.from_coo(r, c, v, 3, 3)          # row id 5 out of range
    with pytest.raises(RuntimeError, match="ROCm device"):
        eng.LaplacianCSR.from_coo(r.cpu(), c.cpu(), v.cpu(), 6, 6)
    csr = eng.LaplacianCSR.from_coo(r, c, v, 6, 6)
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        eng.spmm(csr, torch.zeros((5, 8), device=dev))
    wide = eng.spmm(csr, torch.ones((6, 1030), device=dev))   # wider than one 512-column panel
    assert wide.shape == (6, 1030) and float(wide[0, 1029]) == 1.0 and float(wide[2].abs().sum()) == 0
    empty = eng.LaplacianCSR.from_coo(r[:0], c[:0], v[:0], 4, 4)   # empty matrix
    out = eng.spmm(empty, torch.ones((4, 8), device=dev))
    assert out.shape == (4, 8) and float(out.abs().sum()) == 0


# --------------------------------------------------------------------------------------------
# dense half of the layer (MFMA) vs the torch oracle
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d_in,d_out,n_rows", [(65, 65, 57), (65, 64, 324), (64, 64, 129), (130, 128, 300),
                                               (128, 128, 1000), (96, 96, 33), (16, 32, 5), (256, 256, 200),
                                               (515, 512, 70), (512, 512, 64), (128, 200, 77), (200, 300, 40)])
def test_layer_dense_matches_oracle(d_in, d_out, n_rows, dev):
    eng = _pkg().engine
    g = torch.Generator().manual_seed(d_in * 1000 + d_out)
    LE = torch.randn((n_rows, d_in), generator=g) * 0.3
    E = torch.randn((n_rows, d_in), generator=g) * 0.3
    bound = 1 / d_in ** 0.5
    W1, W2 = ((torch.rand((d_out, d_in), generator=g) * 2 - 1) * bound for _ in range(2))
    b1, b2 = ((torch.rand((d_out,), generator=g) * 2 - 1) * bound for _ in range(2))
    # oracle: NGCF.py:131-146 op for op
    M = torch.nn.functional.linear(LE, W1, b1) + torch.nn.functional.linear(E, W1, b1) \
        + torch.nn.functional.linear(LE * E, W2, b2)
    want_c = torch.nn.functional.leaky_relu(M, 0.2)
    want_n = torch.nn.functional.normalize(want_c, p=2, dim=1)
    carry = torch.full((n_rows, d_out), float("nan"), device=dev)
    allE = torch.full((n_rows, d_out + 9), float("nan"), device=dev)
    ws = eng.Workspace()
    eng.layer_dense(LE.to(dev), E.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), carry, allE[:, 5:5 + d_out], ws)
    np.testing.assert_allclose(carry.detach().cpu().numpy(), want_c.numpy(), atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(allE[:, 5:5 + d_out].detach().cpu().numpy(), want_n.numpy(), atol=ATOL, rtol=RTOL)
    assert torch.isnan(allE[:, :5]).all() and torch.isnan(allE[:, 5 + d_out:]).all()    # nothing outside the block
    # last layer: carry omitted
    allE2 = torch.empty((n_rows, d_out), device=dev)
    eng.layer_dense(LE.to(dev), E.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), None, allE2, ws)
    assert torch.equal(allE2, allE[:, 5:5 + d_out].contiguous())


def test_layer_dense_zero_row_normalises_to_zero(dev):
    """F.normalize of an all-zero row is zeros (eps clamp), NGCF.py:144."""
    eng = _pkg().engine
    d = 64
    LE = torch.zeros((40, d), device=dev)
    E = torch.zeros((40, d), device=dev)
    W = torch.randn((d, d), device=dev) * 0.1
    b0 = torch.zeros(d, device=dev)
    out = torch.empty((40, d), device=dev)
    eng.layer_dense(LE, E, W, b0, W, b0, None, out, eng.Workspace())
    assert float(out.abs().max()) == 0.0 and not torch.isnan(out).any()


def test_message_dropout_statistics_and_determinism(dev):
    eng = _pkg().engine
    n, d, p = 4096, 128, 0.25
    g = torch.Generator().manual_seed(1)
    LE, E = (torch.randn((n, d), generator=g).to(dev) for _ in range(2))
    W1, W2 = (torch.randn((d, d), generator=g).to(dev) * 0.1 for _ in range(2))
    b = torch.ones(d, device=dev)
    ws = eng.Workspace()
    base = torch.empty((n, d), device=dev)
    nb = torch.empty((n, d), device=dev)
    eng.layer_dense(LE, E, W1, b, W2, b, base, nb, ws)
    c1, c2, c3 = (torch.empty((n, d), device=dev) for _ in range(3))
    eng.layer_dense(LE, E, W1, b, W2, b, c1, nb, ws, drop_p=p, drop_seed=11)
    eng.layer_dense(LE, E, W1, b, W2, b, c2, nb, ws, drop_p=p, drop_seed=11)
    eng.layer_dense(LE, E, W1, b, W2, b, c3, nb, ws, drop_p=p, drop_seed=12)
    assert torch.equal(c1, c2) and not torch.equal(c1, c3)
    dropped = (c1 == 0) & (base != 0)
    frac = float(dropped.float().mean())
    assert abs(frac - p) < 0.01
    kept = ~dropped
    np.testing.assert_allclose(c1[kept].detach().cpu().numpy(), (base[kept] / (1 - p)).detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    # normalised block is the normalisation of the dropped carry
    np.testing.assert_allclose(nb.detach().cpu().numpy(), torch.nn.functional.normalize(c3, dim=1).detach().cpu().numpy(), atol=1e-6)


# --------------------------------------------------------------------------------------------
# whole propagation at a moderate size vs the torch-CPU oracle and the fp64 yardstick
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d0,layers", [(128, (128, 128, 128)), (65, (64, 64)), (256, (256,))])
def test_propagate_medium_graph(d0, layers, dev):
    pkg = _pkg()
    coo = pkg.graphs.synthetic_bipartite(20000, 1500, 400000, seed=9, device=dev)
    n_user, n_item = coo["n_user"], coo["n_item"]
    lap = pkg.graphs.to_sparse_coo(coo)
    num_dict = {"user": n_user, "item": n_item, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(4)
    model = pkg.NGCF(d0, list(layers), 0.3, [0.1] * len(layers), 1.0, [lap], num_dict, 1024, dev).to(dev).eval()
    all_E = model.propagate(0).detach().cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    w1, b1, w2, b2 = layer_params(sd, len(layers))
    want = orc.propagate_torch(lap.cpu(), sd["user_embedding.weight"], sd["item_embedding.weight"], w1, b1, w2, b2)
    np.testing.assert_allclose(all_E.numpy(), want.numpy(), atol=ATOL, rtol=RTOL)
    f64 = orc.propagate_f64(coo["rows"].detach().cpu().numpy(), coo["cols"].detach().cpu().numpy(), coo["vals"].detach().cpu().numpy(),
                            want[:, :d0].numpy(), [w.numpy() for w in w1], [x.numpy() for x in b1],
                            [w.numpy() for w in w2], [x.numpy() for x in b2])
    ref_err = np.abs(want.numpy() - f64).max()
    assert np.abs(all_E.numpy() - f64).max() <= 4 * ref_err + 1e-7
    # size-independent properties: every propagated block has unit rows (or zero rows)
    off = d0
    for d in layers:
        nrm = all_E[:, off:off + d].norm(dim=1)
        assert torch.all((nrm - 1).abs() < 1e-5) or torch.all(((nrm - 1).abs() < 1e-5) | (nrm == 0))
        off += d


def test_spmm_linearity_and_permutation_invariance_large(dev):
    """Properties that hold at any size (run at ~4M stored entries, d=128)."""
    pkg = _pkg()
    eng = pkg.engine
    coo = pkg.graphs.synthetic_bipartite(200000, 20000, 2000000, seed=21, device=dev)
    N = coo["n_user"] + coo["n_item"]
    csr = eng.LaplacianCSR.from_coo(coo["rows"], coo["cols"], coo["vals"], N, N)
    assert csr.n_segments > 0                                # popular items are cut into segments
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.randn((N, 128), generator=g, device=dev)
    Y = torch.randn((N, 128), generator=g, device=dev)
    LX, LY = eng.spmm(csr, X), eng.spmm(csr, Y)
    LZ = eng.spmm(csr, 2.0 * X - 0.5 * Y)
    scale = float(LX.abs().max())
    assert float((LZ - (2.0 * LX - 0.5 * LY)).abs().max()) <= 1e-5 * max(scale, 1.0)
    # entry order must not matter beyond rounding: shuffle the COO (host sort path keeps rows grouped)
    perm = torch.randperm(coo["rows"].numel(), device=dev)
    csr2 = eng.LaplacianCSR.from_coo(coo["rows"][perm], coo["cols"][perm], coo["vals"][perm], N, N)
    LX2 = eng.spmm(csr2, X)
    assert float((LX2 - LX).abs().max()) <= 1e-5 * max(scale, 1.0)
    # L is symmetric (matrix.py:49-52): <Y, L X> == <L Y, X>
    a, b = float((Y.double() * LX.double()).sum()), float((LY.double() * X.double()).sum())
    assert abs(a - b) <= 1e-6 * max(abs(a), 1.0) + 1e-3
    # checksum against fp64 on the heaviest row and 2048 random rows
    rows = torch.cat([torch.randint(0, N, (2048,), device=dev), torch.tensor([coo["n_user"]], device=dev)])
    rp = torch.searchsorted(coo["rows"], torch.stack([rows, rows + 1]))
    for r, (lo, hi) in zip(rows.tolist()[-8:], rp.T.tolist()[-8:]):
        want = (coo["vals"][lo:hi].double()[:, None] * X[coo["cols"][lo:hi]].double()).sum(0)
        np.testing.assert_allclose(LX[r].detach().cpu().numpy(), want.detach().cpu().numpy(), atol=ATOL, rtol=RTOL)


# --------------------------------------------------------------------------------------------
# gathers, feature injection, BPR
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [193, 260, 512])
def test_gather_rows_bit_exact_and_bounds(d, dev):
    eng = _pkg().engine
    g = torch.Generator().manual_seed(d)
    table = torch.randn((500, d + 3), generator=g).to(dev)[:, :d]      # strided view like all_items_emb
    idx = torch.randint(0, 400, (77,), generator=g).to(dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = eng.gather_rows(table, idx, status, row_off=100, n_idx_rows=400)
    assert torch.equal(out, table[100:][idx]) and int(status.item()) == 0
    bad = idx.clone()
    bad[3] = 400
    eng.gather_rows(table, bad, status, row_off=100, n_idx_rows=400)
    assert int(status.item()) != 0
    out0 = eng.gather_rows(table, idx[:0], status)
    assert out0.shape == (0, d)
    # the three gathers of a forward in one launch (NGCF.py:151-155): users [0, 100), items [100, 500); no negative items
    status.zero_()
    u_idx = torch.randint(0, 100, (33,), generator=g).to(dev)
    n_idx = torch.randint(0, 400, (5,), generator=g).to(dev)
    u, p, n = eng.gather_rows3(table, ((u_idx, 0, 100), (idx, 100, 400), (n_idx, 100, 400)), status)
    assert torch.equal(u, table[u_idx]) and torch.equal(p, table[100:][idx]) and torch.equal(n, table[100:][n_idx])
    u, p, n = eng.gather_rows3(table, ((u_idx, 0, 100), (idx, 100, 400), (None, 100, 400)), status)
    assert n is None and torch.equal(u, table[u_idx]) and torch.equal(p, table[100:][idx]) and int(status.item()) == 0
    u, p, n = eng.gather_rows3(table, ((u_idx[:0], 0, 100), (idx[:0], 100, 400), (None, 100, 400)), status)
    assert u.shape == (0, d) and p.shape == (0, d)
    eng.gather_rows3(table, ((u_idx + 100, 0, 100), (idx, 100, 400), (n_idx, 100, 400)), status)   # a user id past the users
    assert int(status.item()) != 0


def test_forward_raises_index_error_and_shape_error(dev):
    pkg = _pkg()
    g = load_golden("fwd_sigA_small")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    batch = {k: v.to(dev) for k, v in b.items()}
    bad = dict(batch, pos_item=batch["pos_item"].clone())
    bad["pos_item"][0] = 10 ** 6
    with pytest.raises(IndexError):
        model(node_flag=False, **bad)
    bad = dict(batch, year=torch.tensor([25], device=dev))          # 25 % 18 = 7 >= len(lap_list)
    with pytest.raises(IndexError):
        model(node_flag=False, **bad)
    model(node_flag=False, **batch)                                   # still usable afterwards
    # embed_size not a multiple of 5: RuntimeError like the reference's shape mismatch (NGCF.py:114)
    args = ctor_args(g, lap_list_of(g, dev), dev)
    args["embed_size"] = 64
    m64 = pkg.NGCF(**args).to(dev)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        m64(node_flag=False, **batch)


def test_feature_injection_last_duplicate_wins(dev):
    eng = _pkg().engine
    g = torch.Generator().manual_seed(8)
    U, d0, fw, B = 50, 65, 13, 12
    user_w = torch.randn((U, d0), generator=g)
    cards = [76, 2, 13, 32, 7]
    tables = [torch.randn((c, fw), generator=g) for c in cards]
    idx = [torch.randint(0, c, (B,), generator=g) for c in cards]
    u_id = torch.randint(0, U, (B,), generator=g)
    u_id[7] = u_id[2]
    u_id[11] = u_id[2]
    for ratio in (1.0, 0.7, 0.25):
        want = user_w.clone()
        feats = {"age": tables[0], "sex": tables[1], "month": tables[2], "day": tables[3], "dow": tables[4]}
        # sequential last-writer semantics of the CPU index_put_ (computed from the PRE-update rows)
        rhs = want[u_id] * (1 - ratio) + torch.cat([t[i] for t, i in zip(tables, idx)], 1) * ratio
        for bpos in range(B):
            want[u_id[bpos]] = rhs[bpos]
        ref = orc.feature_inject_torch(user_w.clone(), feats, u_id, *idx, ratio)
        assert torch.equal(ref, want)                                  # the oracle has the same semantics
        got = user_w.clone().to(dev)
        scratch = torch.full((U,), -1, dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        eng.feature_inject(got, [t.to(dev) for t in tables], [i.to(dev) for i in idx], u_id.to(dev), ratio, scratch, status)
        assert torch.equal(got.cpu(), want)
        assert int(status.item()) == 0 and bool((scratch == -1).all())


def test_bpr_matches_golden(dev):
    pkg = _pkg()
    g = load_golden("bpr")
    for tag in ("full", "bcast", "one"):
        u, p, n = (torch.from_numpy(g[f"{tag}_{k}"]).to(dev) for k in "upn")
        wd, bs = (float(x) for x in g[f"{tag}_wd_bs"])
        crit = pkg.BPR(weight_decay=wd, batch_size=int(bs))
        loss = crit(u, p, n)
        assert loss.dim() == 0 and loss.device.type == "cuda"
        ref = float(g[f"{tag}_loss"])
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
        assert float(crit(u, p, n)) == float(loss)                     # deterministic reduction
        total = torch.zeros((), device=dev)
        total += loss                                                  # experiment.py:59,101
    with pytest.raises(RuntimeError):
        pkg.BPR(0.025, 8)(torch.zeros((3, 4), device=dev), torch.zeros((2, 4), device=dev), torch.zeros((3, 4), device=dev))


def test_eval_style_batch_and_topk_consumers(dev):
    """experiment.py:82-111 pattern: neg_item=torch.empty(0), mm + topk on the returned tensors."""
    pkg = _pkg()
    g = load_golden("fwd_sigB_y19")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()
    batch = {k: v.to(dev) for k, v in b.items()}
    batch["neg_item"] = torch.empty(0)
    with torch.no_grad():
        u, p, _ = model(node_flag=False, **batch)
        pred = torch.mm(u, p.T)
        neg = torch.cat((p[1:], p[1:][:1]))
        loss = pkg.BPR(0.025, 25)(u, p[:1], neg)
        _, rank = torch.topk(pred[0], 3)
    assert pred.shape == (len(b["u_id"]), len(b["u_id"])) and rank.numel() == 3 and torch.isfinite(loss)
    scores = torch.mm(u, model.all_items_emb.T)                         # demo.py:233-235
    assert scores.shape == (len(b["u_id"]), int(g["meta"][1]))


# --------------------------------------------------------------------------------------------
# BASELINE.json configs[0..1]: the Seoul-shaped graph (stand-in: the real lap_list.pkl is a missing blob)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("embed,layers", [(65, (64, 64)), (515, (512, 512))])
def test_seoul_standin_forward_matches_oracle(embed, layers, dev):
    """C1 (65 -> [64,64], the CPU-runnable case) and C2 (2-layer d=512) through the full nn.Module forward."""
    pkg = _pkg()
    slices = pkg.graphs.seoul_standin(dev)
    assert len(slices) == 2 and slices[1]["nnz"] >= slices[0]["nnz"]              # carry-over quirk: slice 1 is a superset
    laps = [pkg.graphs.to_sparse_coo(s) for s in slices]
    U, I = slices[0]["n_user"], slices[0]["n_item"]
    num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(1801)
    model = pkg.NGCF(embed, list(layers), 0.3, [0.1] * len(layers), 1.0, laps, num_dict, 25, dev).to(dev).eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    B = 25                                                                       # eval batch: one user x 25 items
    batch = dict(year=torch.full((B,), 19), u_id=torch.full((B,), 4711), age=torch.full((B,), 30), sex=torch.ones(B, dtype=torch.int64),
                 month=torch.full((B,), 7), day=torch.full((B,), 15), dow=torch.full((B,), 2),
                 pos_item=torch.randperm(I, generator=g)[:B], neg_item=torch.empty(0))
    with torch.no_grad():
        u, p, n = model(node_flag=False, **{k: (v.to(dev) if v.numel() else v) for k, v in batch.items()})
    feats = {"age": sd["age_emb.weight"], "sex": sd["sex_emb.weight"], "month": sd["month_emb.weight"],
             "day": sd["day_emb.weight"], "dow": sd["dow_emb.weight"]}
    uw = sd["user_embedding.weight"].clone()
    uw[4711] = torch.cat([feats["age"][30], feats["sex"][1], feats["month"][7], feats["day"][15], feats["dow"][2]])  # emb_ratio 1
    w1, b1, w2, b2 = layer_params(sd, len(layers))
    want = orc.propagate_torch(laps[1].cpu(), uw, sd["item_embedding.weight"], w1, b1, w2, b2)   # year 19 -> slice 1
    got = torch.cat((model.all_users_emb, model.all_items_emb), 0).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=ATOL, rtol=RTOL)
    assert torch.equal(model.user_embedding.weight.detach().cpu(), uw)
    assert n.numel() == 0 and torch.equal(u.cpu(), got[:U][batch["u_id"]]) and torch.equal(p.cpu(), got[U:][batch["pos_item"]])


# --------------------------------------------------------------------------------------------
# edge cases of the module surface
# --------------------------------------------------------------------------------------------
def test_forward_edge_cases_empty_batch_single_layer_and_device_move(dev):
    pkg = _pkg()
    coo = pkg.graphs.synthetic_bipartite(400, 30, 4000, seed=6, device="cpu")       # lap_list kept on the CPU
    lap = pkg.graphs.to_sparse_coo(coo)
    num_dict = {"user": 400, "item": 30, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(3)
    model = pkg.NGCF(65, [64], 0.3, [0.1], 1.0, [lap], num_dict, 4, torch.device("cpu"))    # one layer
    model = model.to(dev).eval()
    empty = torch.empty(0, dtype=torch.int64, device=dev)
    with torch.no_grad():
        u, p, n = model(year=torch.tensor([18], device=dev), u_id=empty, age=empty, sex=empty, month=empty, day=empty,
                        dow=empty, pos_item=empty, neg_item=torch.empty(0), node_flag=False)
    assert u.shape == (0, 129) and p.shape == (0, 129) and n.numel() == 0
    assert model.all_users_emb.shape == (400, 129) and model.all_items_emb.shape == (30, 129)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = orc.propagate_torch(lap, sd["user_embedding.weight"], sd["item_embedding.weight"], [sd["w1_list.0.weight"]],
                               [sd["w1_list.0.bias"]], [sd["w2_list.0.weight"]], [sd["w2_list.0.bias"]])
    got = torch.cat((model.all_users_emb, model.all_items_emb), 0).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=ATOL, rtol=RTOL)
    # widths the dense kernel does not cover fail loudly (no silent fallback)
    wide = pkg.NGCF(65, [600], None, None, 1.0, [lap], num_dict, 4, dev).to(dev).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="512"):
        wide.propagate(0)
    # moving the module to the CPU makes forward raise again (no fallback), moving back works
    model.cpu()
    with pytest.raises(RuntimeError, match="no CPU"):
        model.propagate(0)
    model.to(dev)
    with torch.no_grad():
        again = model.propagate(0)
    assert torch.equal(again.cpu(), got)


def test_full_size_c3_properties(dev):
    """BASELINE.json configs[2] at FULL size (1 M x 100 K, ~100 M stored entries, d=128): size-independent
    properties - linearity, symmetry of L, fp64 spot rows (incl. the heaviest item row), unit-norm blocks."""
    pkg = _pkg()
    eng = pkg.engine
    U, I = 1_000_000, 100_000
    coo = pkg.graphs.synthetic_bipartite(U, I, 50_000_000, seed=2603, device=dev)
    N = U + I
    assert coo["nnz"] == 2 * coo["interactions"] and coo["nnz"] > 99_000_000
    num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(2603)
    model = pkg.NGCF(128, [128, 128, 128], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, 1024, dev).to(dev).eval()
    csr = model.laplacian_csr(0)
    assert csr.nnz == coo["nnz"] and csr.n_segments > 0
    assert csr.swept_rows == N                  # at this size the cached Laplacian runs on the L2-swept kernel (mode 3)
    g = torch.Generator(device=dev).manual_seed(9)
    X = torch.randn((N, 128), generator=g, device=dev)
    Y = torch.randn((N, 128), generator=g, device=dev)
    LX, LY = eng.spmm(csr, X), eng.spmm(csr, Y)
    LZ = eng.spmm(csr, 3.0 * X + 0.25 * Y)
    scale = float(LX.abs().max())
    assert float((LZ - (3.0 * LX + 0.25 * LY)).abs().max()) <= 2e-5 * max(scale, 1.0)
    csr.set_mode(0)                             # the row-wise / d-sliced kernels give the same product
    assert csr.swept_rows == 0
    assert float((eng.spmm(csr, X) - LX).abs().max()) <= 2e-6 * max(scale, 1.0)
    csr.set_mode(3)
    assert torch.equal(eng.spmm(csr, X), LX)    # rebuilt plan, same bits
    a, b = float((Y.double() * LX.double()).sum()), float((LY.double() * X.double()).sum())
    assert abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1.0) + 1e-2
    rows = torch.cat([torch.randint(0, N, (24,), device=dev), torch.tensor([U, U + 1, 0], device=dev)])
    rp = torch.searchsorted(coo["rows"], torch.stack([rows, rows + 1]))
    for r, (lo, hi) in zip(rows.tolist(), rp.T.tolist()):
        want = (coo["vals"][lo:hi].double()[:, None] * X[coo["cols"][lo:hi]].double()).sum(0)
        np.testing.assert_allclose(LX[r].cpu().numpy(), want.cpu().numpy(), atol=ATOL, rtol=RTOL)
    with torch.no_grad():
        all_E = model.propagate(0)
    assert all_E.shape == (N, 512) and bool(torch.isfinite(all_E).all())
    assert torch.equal(all_E[:U, :128], model.user_embedding.weight.detach())          # block 0 is E0, bit for bit
    for k in range(3):
        nrm = all_E[:, 128 * (k + 1):128 * (k + 2)].norm(dim=1)
        assert float((nrm - 1).abs().max()) < 1e-5
    # layer 1 of a sample of rows against fp64 from the definition (NGCF.py:130-144)
    sd = {k_: v.detach() for k_, v in model.state_dict().items()}
    E0 = all_E[:, :128].double()
    W1, b1, W2, b2 = (sd[f"{n_}.0.{t}"].double() for n_, t in (("w1_list", "weight"), ("w1_list", "bias"), ("w2_list", "weight"), ("w2_list", "bias")))
    for r, (lo, hi) in list(zip(rows.tolist(), rp.T.tolist()))[-6:]:
        le = (coo["vals"][lo:hi].double()[:, None] * E0[coo["cols"][lo:hi]]).sum(0)
        m = (le + E0[r]) @ W1.T + 2 * b1 + (le * E0[r]) @ W2.T + b2
        m = torch.where(m >= 0, m, 0.2 * m)
        np.testing.assert_allclose(all_E[r, 128:256].cpu().numpy(), (m / m.norm()).cpu().numpy(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("U,I,inter,d0,layers,B", [(1, 1, 1, 5, (4,), 3), (3, 2, 0, 10, (8, 8), 4), (17, 300, 900, 15, (33, 7), 64),
                                                    (2000, 3, 2500, 65, (65,), 4096), (64, 64, 4096, 20, (16, 16, 16, 16), 1)])
def test_odd_shapes_against_oracle(U, I, inter, d0, layers, B, dev):
    """Degenerate and lopsided graphs, tiny and odd widths, empty Laplacian, batch larger than the user count."""
    pkg = _pkg()
    g = torch.Generator().manual_seed(U * 31 + I)
    if inter > 0:
        key = torch.unique(torch.randint(0, U * I, (inter,), generator=g))
        u, i = key // I, key % I
        w = torch.rand(key.numel(), generator=g) * 4.5 + 0.5
        coo = pkg.graphs._normalise(u, i, w, U, I)
    else:
        coo = {"rows": torch.empty(0, dtype=torch.int64), "cols": torch.empty(0, dtype=torch.int64), "vals": torch.empty(0)}
    coo.update({"n_user": U, "n_item": I})
    lap = pkg.graphs.to_sparse_coo(coo)
    num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(5)
    model = pkg.NGCF(d0, list(layers), 0.3, [0.1] * len(layers), 0.5, [lap], num_dict, B, dev).to(dev).eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = dict(year=torch.full((B,), 18), u_id=torch.randperm(max(U, B), generator=g)[:B] % U,
                 age=torch.randint(0, 76, (B,), generator=g), sex=torch.randint(0, 2, (B,), generator=g),
                 month=torch.randint(0, 13, (B,), generator=g), day=torch.randint(0, 32, (B,), generator=g),
                 dow=torch.randint(0, 7, (B,), generator=g), pos_item=torch.randint(0, I, (B,), generator=g),
                 neg_item=torch.randint(0, I, (B,), generator=g))
    with torch.no_grad():
        u_e, p_e, n_e = model(node_flag=False, **{k: v.to(dev) for k, v in batch.items()})
        loss = pkg.BPR(0.025, B)(u_e, p_e, n_e)
    # oracle with the engine's duplicate rule (last occurrence wins), RHS from the pre-update rows
    feats = torch.cat([sd["age_emb.weight"][batch["age"]], sd["sex_emb.weight"][batch["sex"]], sd["month_emb.weight"][batch["month"]],
                       sd["day_emb.weight"][batch["day"]], sd["dow_emb.weight"][batch["dow"]]], 1)
    uw = sd["user_embedding.weight"].clone()
    rhs = uw[batch["u_id"]] * (1 - 0.5) + feats * 0.5
    for bpos in range(B):
        uw[batch["u_id"][bpos]] = rhs[bpos]
    assert torch.equal(model.user_embedding.weight.detach().cpu(), uw)
    w1, b1, w2, b2 = layer_params(sd, len(layers))
    want = orc.propagate_torch(lap, uw, sd["item_embedding.weight"], w1, b1, w2, b2)
    got = torch.cat((model.all_users_emb, model.all_items_emb), 0).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=ATOL, rtol=RTOL)
    ou, op, on = orc.gather_torch(want, U, batch["u_id"], batch["pos_item"], batch["neg_item"])
    ref_loss = float(orc.bpr_torch(ou, op, on, 0.025, B))
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss) + 1e-6


@pytest.mark.parametrize("d_in,d_out,mode", [(128, 128, "eval"), (130, 128, "hash"), (64, 100, "mask"), (144, 128, "last"),
                                             (120, 128, "mask"), (128, 128, "last"), (113, 128, "hash")])
def test_resident_dense_kernel_is_bit_identical_to_the_staged_one(d_in, d_out, mode, dev, lib_options):
    """layer_dense_resident_kernel (weights resident in LDS, no barriers; taken from 131 072 rows on at 97..128 output columns)
    and - in a library built with -DNGCF_LAB, otherwise 2 means 1 - layer_dense_resident_il_kernel (dense_resident = 2: one wave per
    SIMD, the finished tile stored under the next tile's K loop; 128 output columns, 8 or 9 chunks) against layer_dense_kernel on the same inputs: the k order of every output element
    is the same, so carry and normalised block must agree bit for bit - in eval mode, with the hash dropout, with a host-drawn
    noise tensor, and without a carry."""
    import os
    eng = _pkg().engine
    n = 140_001                                              # above the kernel's threshold; not a multiple of 32: a partial last tile
    g = torch.Generator().manual_seed(d_in + d_out)
    ld = (d_in + 31) // 32 * 32
    LE = (torch.randn((n, ld), generator=g) * 0.5).to(dev)[:, :d_in]
    E = (torch.randn((n, ld), generator=g) * 0.5).to(dev)[:, :d_in]
    W1, W2 = ((torch.randn((d_out, d_in), generator=g) * 0.1).to(dev) for _ in range(2))
    b1, b2 = ((torch.randn((d_out,), generator=g) * 0.1).to(dev) for _ in range(2))
    mask = (torch.rand((n, d_out), generator=g) > 0.3).float().to(dev) / 0.7 if mode == "mask" else None
    kw = dict(drop_p=0.3 if mode in ("hash", "mask") else 0.0, drop_seed=77 if mode == "hash" else 0, drop_mask=mask)
    outs = []
    for resident in (1, 0, 2):
        lib_options(dense_resident=resident)
        carry = None if mode == "last" else torch.full((n, d_out), 5.0, device=dev)
        norm = torch.full((n, d_out + 3), 7.0, device=dev)[:, :d_out]                # a column slice of a wider matrix
        eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, eng.Workspace(), **kw)
        outs.append((carry, norm.clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[2][1], outs[1][1])
    assert bool((norm.as_strided((n, 3), (d_out + 3, 1), d_out) == 7.0).all())      # nothing written past the slice
    if mode != "last":
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[2][0], outs[1][0])
        if mode == "hash":
            frac = float((outs[0][0] == 0).float().mean())
            assert abs(frac - 0.3) < 0.01
    assert float((outs[0][1].norm(dim=1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("d_in,d_out,n,mode", [(515, 512, 5941, "eval"), (512, 512, 300, "hash"), (256, 256, 4099, "mask"),
                                              (130, 200, 1000, "last"), (260, 300, 33, "eval"), (8, 512, 64, "hash"),
                                              (256, 256, 40_003, "eval")])
def test_direct_dense_kernel_is_bit_identical_to_the_staged_one(d_in, d_out, n, mode, dev, lib_options):
    """layer_dense_direct_kernel (256 / 512 output columns, operands straight from global memory, no LDS staging) and
    layer_dense_tall_kernel + row_scale_kernel (96-row x 128-column workgroups, three row tiles per wave, the row norm in a second
    kernel that adds the squares in the same order) against layer_dense_kernel on the same inputs: same k order per output element, so carry and normalised block agree bit for bit -
    eval mode, hash dropout, host-drawn noise, no carry, output widths below the padded width, partial last tiles."""
    import os
    eng = _pkg().engine
    g = torch.Generator().manual_seed(d_in + d_out + n)
    ld = (d_in + 31) // 32 * 32
    LE = (torch.randn((n, ld), generator=g) * 0.5).to(dev)[:, :d_in]
    E = (torch.randn((n, ld), generator=g) * 0.5).to(dev)[:, :d_in]
    W1, W2 = ((torch.randn((d_out, d_in), generator=g) * 0.1).to(dev) for _ in range(2))
    b1, b2 = ((torch.randn((d_out,), generator=g) * 0.1).to(dev) for _ in range(2))
    mask = (torch.rand((n, d_out), generator=g) > 0.3).float().to(dev) / 0.7 if mode == "mask" else None
    kw = dict(drop_p=0.3 if mode in ("hash", "mask") else 0.0, drop_seed=77 if mode == "hash" else 0, drop_mask=mask)
    outs = []
    for direct, tall in ((2, 0), (0, 0), (0, 2)):           # direct kernel, staged kernel, tall kernel (+ row_scale_kernel)
        lib_options(dense_direct=direct, dense_tall=tall)
        carry = None if mode == "last" else torch.full((n, d_out), 5.0, device=dev)
        norm = torch.full((n, d_out + 3), 7.0, device=dev)[:, :d_out]            # a column slice of a wider matrix
        eng.layer_dense(LE, E, W1, b1, W2, b2, carry, norm, eng.Workspace(), **kw)
        outs.append((carry, norm.clone()))
        assert bool((norm.as_strided((n, 3), (d_out + 3, 1), d_out) == 7.0).all())      # nothing written past the slice
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[2][1], outs[1][1])
    if mode != "last":
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[2][0], outs[1][0])
    assert float((outs[0][1].norm(dim=1) - 1).abs().max()) < 1e-5
    # and against the plain formula (NGCF.py:131-146, eval mode) on the rows of the first tile
    if mode == "eval":
        le, e = LE[:32].double(), E[:32].double()
        m = (le + e) @ W1.double().T + b1.double() + (le * e) @ W2.double().T + b2.double() + b1.double()
        m = torch.nn.functional.leaky_relu(m, 0.2)
        torch.testing.assert_close(outs[0][0][:32].double(), m, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("n_rows,n_tab,deg,d", [(1, 1, 1, 4), (63, 7, 3, 60), (65, 100, 50, 64), (5000, 511, 200, 68),
                                               (70000, 512, 20, 128), (300, 100, 700, 516), (2000, 64, 0, 64),
                                               (4097, 300, 130, 200)])
def test_table_in_lds_kernel_shapes(n_rows, n_tab, deg, d, dev, oracle_clib):
    """spmm_ldstab_kernel at its edges: one table row and the 512-row limit (131 KB of LDS), rows of 0 / 1 / 64 < n <= 128 / more
    than 128 entries (the third-batch path) and rows longer than the segment length inside the group (left to the segment
    kernels), partial last slices, more rows than one workgroup block; the table sits in the MIDDLE of a wider column range
    (col_lo > 0).  Against the C oracle and the plain row-wise kernels."""
    eng = _pkg().engine
    rng = np.random.default_rng(n_rows * 7 + n_tab)
    n_cols, col_lo = 3 * n_tab + 11, n_tab + 5
    lens = rng.poisson(deg, n_rows) if deg else np.zeros(n_rows, np.int64)
    if n_rows > 10 and deg:
        lens[3], lens[n_rows // 2] = 0, 3 * max(deg, 100)              # an empty row and a long one
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    nnz = int(rowptr[-1])
    cols = (col_lo + rng.integers(0, n_tab, nnz)).astype(np.int64)
    vals = rng.normal(0, 0.3, nnz).astype(np.float32)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lens)
    if nnz == 0:
        rows, cols, vals = np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    csr = eng.LaplacianCSR.from_coo(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(vals).to(dev),
                                    n_rows, n_cols)
    E = rng.normal(0, 0.5, (n_cols, d)).astype(np.float32)
    Ed = torch.from_numpy(E).to(dev)
    want = c_spmm(oracle_clib, rowptr, cols.astype(np.int32), vals, E)
    got = eng.spmm(csr, Ed).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=RTOL)
    csr.set_mode(1)
    plain = eng.spmm(csr, Ed).cpu().numpy()
    np.testing.assert_allclose(got, plain, atol=5e-6, rtol=2e-5)
    assert np.all(got[lens == 0] == 0)
