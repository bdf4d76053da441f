"""GPU tests of the backward pass (`loss.backward()`, experiment.py:57): gradients of every parameter from the
HIP path vs torch autograd through the CPU oracle (op-for-op restatement of the reference), rtol 2e-3."""
import numpy as np
import pytest
import torch

import ngcf_oracle as orc
from conftest import load_golden
from golden_util import batch_of, ctor_args, lap_list_of, layer_params, sd_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _pkg():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    return pkg


def _oracle_grads(g, sd, b, node_flag, rng_state=None, wd=0.025, extra=0.0):
    n_layer = len(g["layers"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(("w1_list", "w2_list", "item_emb"))}
    uw = torch.from_numpy(g["out_user_weight_after"]).clone().requires_grad_(True)     # state after the injection
    w1 = [leaves[f"w1_list.{k}.weight"] for k in range(n_layer)]
    b1 = [leaves[f"w1_list.{k}.bias"] for k in range(n_layer)]
    w2 = [leaves[f"w2_list.{k}.weight"] for k in range(n_layer)]
    b2 = [leaves[f"w2_list.{k}.bias"] for k in range(n_layer)]
    L = lap_list_of(g)[int(g["year_idx"])]
    if rng_state is not None:
        torch.set_rng_state(rng_state)
    all_E = orc.propagate_torch(L, uw, leaves["item_embedding.weight"], w1, b1, w2, b2, mess_dropout=None,
                                training=False, node_dropout=float(g["meta"][5]), node_flag=node_flag)
    u, p, n = orc.gather_torch(all_E, int(g["meta"][0]), b["u_id"], b["pos_item"], b["neg_item"])
    loss = orc.bpr_torch(u, p, n, wd, len(b["u_id"]))
    if extra:                                      # a second, dense consumer of all_E (the item block)
        loss = loss + extra * all_E[int(g["meta"][0]):].pow(2).sum()
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    grads["user_embedding.weight"] = uw.grad
    return float(loss), grads


@pytest.mark.parametrize("name,node_flag", [("fwd_sigA_small", False), ("fwd_sigB_y19", False), ("fwd_130_128", False),
                                            ("fwd_train_dropout", True)])
def test_parameter_gradients_match_oracle_autograd(name, node_flag, dev):
    pkg = _pkg()
    g = load_golden(name)
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()                                   # message dropout off; node dropout follows node_flag
    batch = {k: v.to(dev) for k, v in b.items()}
    crit = pkg.BPR(0.025, len(b["u_id"]))
    rng_state = torch.from_numpy(g["rng_state"]) if node_flag else None
    if node_flag:
        torch.set_rng_state(rng_state)
    u, p, n = model(node_flag=node_flag, **batch)
    assert u.requires_grad and p.requires_grad
    loss = crit(u, p, n)
    loss.backward()
    want_loss, want = _oracle_grads(g, sd, b, node_flag, rng_state)
    assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss)
    named = dict(model.named_parameters())
    for k, wg in want.items():
        got = named[k].grad
        assert got is not None, k
        scale = float(wg.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), wg.numpy(), atol=2e-3 * scale + 1e-9, rtol=2e-3, err_msg=k)
    for k in ("age_emb.weight", "sex_emb.weight", "month_emb.weight", "day_emb.weight", "dow_emb.weight"):
        assert named[k].grad is None               # the injection goes through .data (NGCF.py:114): no gradient


def test_gradients_with_a_second_consumer_of_all_E(dev):
    """GatherTriple hands the gradient of all_E over as a row-sparse tensor; when something else also uses all_E (here a penalty on
    `model.all_items_emb`) autograd adds a dense gradient to it and Propagate.backward takes its dense path: same gradients as
    torch autograd through the oracle."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    g = load_golden("fwd_sigB_y19")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()
    batch = {k: v.to(dev) for k, v in b.items()}
    u, p, n = model(node_flag=False, **batch)
    before = ag.sparse_last_layer_calls
    loss = pkg.BPR(0.025, len(b["u_id"]))(u, p, n) + 0.01 * model.all_items_emb.pow(2).sum()
    loss.backward()
    assert ag.sparse_last_layer_calls == before    # a dense gradient arrived: no row-sparse shortcut
    want_loss, want = _oracle_grads(g, sd, b, False, extra=0.01)
    assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss)
    named = dict(model.named_parameters())
    for k, wg in want.items():
        scale = float(wg.abs().max())
        np.testing.assert_allclose(named[k].grad.cpu().numpy(), wg.numpy(), atol=2e-3 * scale + 1e-9, rtol=2e-3, err_msg=k)


def test_training_step_with_adam_runs_and_lowers_the_loss(dev):
    """The experiment.py:45-59 pattern: forward(node_flag=True) -> BPR -> backward -> Adam.step, in train mode."""
    pkg = _pkg()
    g = load_golden("fwd_sigB_y19")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    crit = pkg.BPR(0.025, len(b["u_id"])).to(dev)
    batch = {k: v.to(dev) for k, v in b.items()}
    torch.manual_seed(0)
    losses = []
    total = 0
    for _ in range(6):
        u, p, n = model(node_flag=True, **batch)
        opt.zero_grad()
        loss = crit(u, p, n)
        loss.backward()
        opt.step()
        total += loss
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())


def test_dropout_backward_matches_autograd_on_the_realised_mask(dev):
    """Message dropout: the backward must use exactly the forward's keep mask (recomputed from the seed)."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    eng = pkg.engine
    n, d, p = 300, 96, 0.3
    gen = torch.Generator().manual_seed(2)
    LE, E = (torch.randn((n, d), generator=gen) * 0.4 for _ in range(2))
    W1, W2 = (torch.randn((d, d), generator=gen) * 0.1 for _ in range(2))
    b1, b2 = (torch.randn((d,), generator=gen) * 0.1 for _ in range(2))
    dN, dC = (torch.randn((n, d), generator=gen) for _ in range(2))
    carry = torch.empty((n, d), device=dev)
    nb = torch.empty((n, d), device=dev)
    eng.layer_dense(LE.to(dev), E.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), carry, nb, eng.Workspace(),
                    drop_p=p, drop_seed=99)
    dM = ag._bwd_pre(dN.to(dev), dC.to(dev), carry, 0.2, p, 99).cpu()
    # reference: autograd through leaky -> (mask/(1-p)) -> normalise with the realised mask
    M = (torch.nn.functional.linear(LE, W1, b1) + torch.nn.functional.linear(E, W1, b1)
         + torch.nn.functional.linear(LE * E, W2, b2)).requires_grad_(True)
    A = torch.nn.functional.leaky_relu(M, 0.2)
    mask = (carry.cpu() != 0) | (A.detach() == 0)
    Cc = A * mask / (1 - p)
    Nn = torch.nn.functional.normalize(Cc, p=2, dim=1)
    ((Nn * dN).sum() + (Cc * dC).sum()).backward()
    np.testing.assert_allclose(dM.numpy(), M.grad.numpy(), atol=2e-5, rtol=2e-3)


def test_bpr_backward_matches_golden_grads(dev):
    pkg = _pkg()
    g = load_golden("bpr")
    for tag in ("full", "bcast", "one"):
        u, p, n = (torch.from_numpy(g[f"{tag}_{k}"]).to(dev).requires_grad_(True) for k in "upn")
        wd, bs = (float(x) for x in g[f"{tag}_wd_bs"])
        loss = pkg.BPR(wd, int(bs))(u, p, n)
        (3.0 * loss).backward()                    # non-unit upstream gradient
        for t, k in ((u, "gu"), (p, "gp"), (n, "gn")):
            want = 3.0 * g[f"{tag}_{k}"]
            np.testing.assert_allclose(t.grad.cpu().numpy(), want, atol=1e-7 + 1e-5 * np.abs(want).max(), rtol=1e-4)


@pytest.mark.parametrize("which", ["u", "p", "n", "pn"])
def test_bpr_backward_of_a_broadcast_operand_is_a_sum_in_row_order(which, dev):
    """A [1, D] operand against R rows (bprloss.py broadcasts like torch): its gradient is the sum of the per-row terms - formed by
    one workgroup in row order (r04: no float atomic is left), so two runs are bit-identical and the values are torch autograd's
    through the oracle's loss (R and D beyond one chunk / one column block: 700 rows, 300 columns)."""
    pkg = _pkg()
    R, D, wd, bs = 700, 300, 0.025, 64
    gen = torch.Generator().manual_seed(5)
    base = {k: torch.randn((1 if k in which else R, D), generator=gen) * 0.3 for k in "upn"}
    got = []
    for _ in range(2):
        t = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        (2.0 * pkg.BPR(wd, bs)(t["u"], t["p"], t["n"])).backward()
        got.append({k: v.grad.clone() for k, v in t.items()})
    for k in "upn":
        assert got[0][k].shape == base[k].shape and torch.equal(got[0][k], got[1][k])
    t = {k: v.clone().double().requires_grad_(True) for k, v in base.items()}
    (2.0 * orc.bpr_torch(t["u"], t["p"], t["n"], wd, bs)).backward()
    for k in "upn":
        want = t[k].grad.numpy()
        np.testing.assert_allclose(got[0][k].cpu().numpy(), want, atol=1e-7 + 2e-5 * np.abs(want).max(), rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("n_rows,d_in,d_out,strided", [(0, 16, 8, False), (33, 7, 5, False), (1000, 65, 64, False),
                                                        (9001, 128, 128, False), (70000, 130 - 2, 96, True), (3000, 130, 200, False),
                                                        (777, 515, 64, True),
                                                        # r04: from 65 536 rows up the 1..3 columns beyond a multiple of 128 (130, 515-wide
                                                        # first layers) run on the narrow kernel: 2, 3 and 1 remainder columns, ragged row count
                                                        (70001, 130, 128, False), (66000, 515, 96, True), (65537, 129, 128, False)])
def test_weight_gradient_kernel_matches_fp64(n_rows, d_in, d_out, strided, dev):
    """gW = dM^T . [LE+E | LE*E] and gb = column sums of dM (MFMA kernel, csrc/backward.hip) against fp64 from the definition (NGCF.py:131-136),
    incl. widths that are not multiples of 32 or 4, a row count that is not a multiple of the block, strided operands
    and the empty case.  Summation order differs from any reference GEMM: tolerance, relative to the result's scale."""
    from seoul_tourism_recommendation_ngcf_amd import autograd, engine
    g = torch.Generator(device=dev).manual_seed(n_rows + d_in)
    mk = lambda n, d: torch.randn((n, d + (12 if strided else 0)), generator=g, device=dev)[:, :d]  # noqa: E731
    dM, LE, E = mk(n_rows, d_out), mk(n_rows, d_in), mk(n_rows, d_in)
    ws = engine.Workspace()
    g1, gb1, g2, gb = autograd._bwd_weight(dM, LE, E, ws)
    assert g1.shape == (d_out, d_in) and g2.shape == (d_out, d_in) and gb.shape == (d_out,)
    got = torch.cat([g1, g2], 1)
    assert torch.equal(gb1, 2.0 * gb)
    SP = torch.cat([LE.double() + E.double(), LE.double() * E.double()], 1)
    want = dM.double().t() @ SP
    scale = max(float(want.abs().max()), 1.0) if n_rows else 1.0
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    # the bias gradient (column sums of dM) from the same pass
    want_b = dM.double().sum(0)
    assert float((gb.double() - want_b).abs().max()) <= 2e-5 * max(float(want_b.abs().max()), 1.0)
    a1, _, a2, gb2 = autograd._bwd_weight(dM, LE, E, ws)
    assert torch.equal(got, torch.cat([a1, a2], 1)) and torch.equal(gb, gb2)             # fixed summation order


@pytest.mark.parametrize("n_rows,d_in,d_out,strided", [(1, 4, 4, False), (130, 128, 128, False), (1000, 130, 128, True), (333, 65, 64, True),
                                                        (257, 160, 96, False), (500, 300, 128, False), (64, 515, 512, True)])
def test_fused_input_gradient_kernel_vs_torch(n_rows, d_in, d_out, strided, dev):
    """ngcf_layer_bwd_input_f32: dLE = dM.W1 + (dM.W2)*E, dE = dM.W1 + (dM.W2)*LE, one MFMA kernel per 128/160-column panel."""
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    eng = _pkg().engine
    g = torch.Generator().manual_seed(n_rows + d_in)
    dM = (torch.randn((n_rows, d_out), generator=g) * 0.3).to(dev)
    W1, W2 = ((torch.randn((d_out, d_in), generator=g) * 0.1).to(dev) for _ in range(2))
    if strided:      # LE / E as column slices of wider matrices (all_E blocks, padded LE)
        LE = (torch.randn((n_rows, d_in + 30), generator=g) * 0.5).to(dev)[:, :d_in]
        E = (torch.randn((n_rows, d_in + 7), generator=g) * 0.5).to(dev)[:, 3:3 + d_in]
    else:
        LE, E = ((torch.randn((n_rows, d_in), generator=g) * 0.5).to(dev) for _ in range(2))
    dLE, dE = ag._bwd_input(dM, W1, W2, LE, E, eng.Workspace())
    dS, dP = dM.double() @ W1.double(), dM.double() @ W2.double()
    scale = float(dS.abs().max()) + 1e-12
    np.testing.assert_allclose(dLE.cpu().numpy(), (dS + dP * E.double()).cpu().numpy(), atol=2e-6 * max(scale, 1.0), rtol=2e-5)
    np.testing.assert_allclose(dE.cpu().numpy(), (dS + dP * LE.double()).cpu().numpy(), atol=2e-6 * max(scale, 1.0), rtol=2e-5)


@pytest.mark.parametrize("n_rows,d_in,d_out,strided", [(131072 + 37, 128, 128, False), (140001, 130, 128, True), (133000, 96, 65, False),
                                                        (131072, 515, 64, True), (150000, 131, 100, False)])
def test_resident_input_gradient_kernel_is_bit_identical_to_the_staged_one(n_rows, d_in, d_out, strided, dev, lib_options):
    """r04: from 131 072 rows up (K = d_out <= 128) the input gradients run on `layer_bwd_input_resident_kernel` ([W1 | W2] of a
    128-column panel resident in LDS, dM read once, no barrier after the prologue) plus `layer_bwd_input_narrow_kernel` for the
    1..4 columns beyond the last panel.  Same k order per output element as the staged kernel: the panels are bit-identical to it;
    the narrow columns (plain dot products, another order) within rounding; everything against fp64."""
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    eng = _pkg().engine
    g = torch.Generator().manual_seed(n_rows + d_in)
    ldm = (d_out + 31) // 32 * 32                                      # dM as _bwd_pre hands it over: rows padded to 32 floats, padding NOT zeroed
    dM = torch.full((n_rows, ldm), float("nan"), device=dev)[:, :d_out]
    dM.copy_((torch.randn((n_rows, d_out), generator=g) * 0.3).to(dev))
    W1, W2 = ((torch.randn((d_out, d_in), generator=g) * 0.1).to(dev) for _ in range(2))
    if strided:
        LE = (torch.randn((n_rows, d_in + 30), generator=g) * 0.5).to(dev)[:, :d_in]
        E = (torch.randn((n_rows, d_in + 7), generator=g) * 0.5).to(dev)[:, 3:3 + d_in]
    else:
        LE, E = ((torch.randn((n_rows, d_in), generator=g) * 0.5).to(dev) for _ in range(2))
    ws = eng.Workspace()
    got = ag._bwd_input(dM, W1, W2, LE, E, ws)
    lib_options(bwd_input_resident=0)
    want = ag._bwd_input(dM, W1, W2, LE, E, ws)                         # the staged kernel
    lib_options(bwd_input_resident=1)
    n_panel = d_in if d_in % 128 == 0 or d_in % 128 > 4 else d_in // 128 * 128      # columns that ran on the resident kernel
    for a, b in zip(got, want):
        assert torch.equal(a[:, :n_panel], b[:, :n_panel])
        if n_panel < d_in:
            torch.testing.assert_close(a[:, n_panel:], b[:, n_panel:], rtol=1e-5, atol=1e-6)
    rows = torch.cat([torch.arange(0, 64), torch.randint(0, n_rows, (512,), generator=g), torch.arange(n_rows - 40, n_rows)]).to(dev)
    dS, dP = dM[rows].double() @ W1.double(), dM[rows].double() @ W2.double()
    scale = max(float(dS.abs().max()), 1.0)
    np.testing.assert_allclose(got[0][rows].cpu().numpy(), (dS + dP * E[rows].double()).cpu().numpy(), atol=2e-6 * scale, rtol=2e-5)
    np.testing.assert_allclose(got[1][rows].cpu().numpy(), (dS + dP * LE[rows].double()).cpu().numpy(), atol=2e-6 * scale, rtol=2e-5)


@pytest.mark.parametrize("node_mode", [None, "reference", "device"])
def test_row_sparse_last_layer_backward_equals_the_dense_path(node_mode, dev, monkeypatch):
    """The last layer's backward on the <= 3 B gathered rows only (compacted dense kernels + ngcf_spmm_t_rows_f32 for
    L^T . dLE) against the dense path (full SpMM on the transposed CSR): same gradients up to the summation order;
    with and without node dropout (thinned matrices in both modes) and with device-mode message dropout (hash by matrix row)."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    coo = pkg.graphs.synthetic_bipartite(4000, 300, 60000, seed=8, device=dev)
    num_dict = {"user": 4000, "item": 300, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    B = 200
    g = torch.Generator().manual_seed(4)
    r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)  # noqa: E731
    batch = dict(year=torch.full((B,), 18, device=dev), u_id=r(4000), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                 pos_item=r(300), neg_item=r(300))
    batch["pos_item"][:20] = 0                                     # a heavy item row several times, duplicates in the batch
    grads = []
    monkeypatch.setattr(ag, "DENSE_GRAD_MAX_BYTES", 0)             # (a graph this small would get a dense gradient: force the row-sparse form)
    for sparse in (True, False):
        torch.manual_seed(21)
        model = pkg.NGCF(130, [128, 64], 0.3, [0.2, 0.2], 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, B, dev).to(dev)
        model.train()
        model.mess_dropout_mode = "device"
        if node_mode:
            model.node_dropout_mode = node_mode
        ag.SPARSE_LAST_LAYER = sparse
        before = ag.sparse_last_layer_calls
        torch.manual_seed(5)                                       # same masks / seeds in both runs
        u, p, n = model(node_flag=node_mode is not None, **batch)
        pkg.BPR(0.025, B)(u, p, n).backward()
        assert (ag.sparse_last_layer_calls - before) == (1 if sparse else 0)
        grads.append({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
    ag.SPARSE_LAST_LAYER = True
    assert set(grads[0]) == set(grads[1]) and "w1_list.1.weight" in grads[0] and "user_embedding.weight" in grads[0]
    for k in grads[0]:
        scale = float(grads[1][k].abs().max()) + 1e-12
        np.testing.assert_allclose(grads[0][k].cpu().numpy(), grads[1][k].cpu().numpy(), atol=2e-5 * scale, rtol=1e-4, err_msg=k)


def test_one_layer_model_at_embed_515_trains(dev, monkeypatch):
    """A one-layer model at the reference-legal embed_size 515 (515 -> [64]): the last layer's input is 515 wide, beyond the
    512 columns a lane of the row-sparse L^T product holds (it runs as two column panels) - same gradients as with the
    row-sparse path switched off."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    coo = pkg.graphs.synthetic_bipartite(900, 60, 9000, seed=3, device=dev)
    num_dict = {"user": 900, "item": 60, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    B = 64
    g = torch.Generator().manual_seed(9)
    r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)  # noqa: E731
    batch = dict(year=torch.full((B,), 18, device=dev), u_id=r(900), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                 pos_item=r(60), neg_item=r(60))
    grads = []
    monkeypatch.setattr(ag, "DENSE_GRAD_MAX_BYTES", 0)             # force the row-sparse form on this small graph
    try:
        for sparse in (True, False):
            torch.manual_seed(2)
            model = pkg.NGCF(515, [64], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, B, dev).to(dev)
            ag.SPARSE_LAST_LAYER = sparse
            u, p, n = model(node_flag=False, **batch)
            pkg.BPR(0.025, B)(u, p, n).backward()
            grads.append({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
    finally:
        ag.SPARSE_LAST_LAYER = True
    assert "w1_list.0.weight" in grads[0] and "user_embedding.weight" in grads[0]
    for k in grads[0]:
        scale = float(grads[1][k].abs().max()) + 1e-12
        np.testing.assert_allclose(grads[0][k].cpu().numpy(), grads[1][k].cpu().numpy(), atol=2e-5 * scale, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("d,heavy,drop", [(65, True, 0.0), (128, True, 0.3), (515, False, 0.0), (600, True, 0.0), (4, False, 0.5)])
def test_row_sparse_transposed_product_vs_dense(d, heavy, drop, dev):
    """ngcf_spmm_t_rows_f32: out = init + L^T . X for an X that is non-zero on a few rows only, against the full product on the
    CSR of L^T with X scattered into a dense matrix (same device-side edge dropout); bit-identical when repeated."""
    pkg = _pkg()
    eng = pkg.engine
    coo = pkg.graphs.synthetic_bipartite(6000, 120 if heavy else 3000, 90000, seed=13, device=dev)   # 120 items: rows cut into segments
    N = coo["n_user"] + coo["n_item"]
    order = torch.sort(coo["cols"], stable=True).indices
    Lt = eng.LaplacianCSR.from_coo(coo["cols"][order], coo["rows"][order], coo["vals"][order], N, N)
    assert Lt.n_segments > 0 or not heavy                         # the heavy case has rows cut into segments + fix-up
    g = torch.Generator().manual_seed(d)
    rows = torch.unique(torch.cat([torch.randint(0, N, (300,), generator=g), torch.arange(N - 5, N)])).to(dev)
    R = rows.numel()
    X = torch.randn((R, d), generator=g).to(dev)
    init = torch.randn((R, d), generator=g).to(dev)
    slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
    slot[rows] = torch.arange(R, dtype=torch.int32, device=dev)
    ed = ([11, 12], drop) if drop else None
    out = torch.full((N, d + 3), 7.0, device=dev)[:, :d]
    ws = eng.Workspace()
    eng.spmm_t_rows(Lt, slot, X, init, out, ws, ed)
    Xd = torch.zeros((N, d), device=dev)
    Xd[rows] = X
    want = eng.spmm(Lt, Xd, ws=ws, edge_drop=None if ed is None else (ed[0], ed[1], True))
    want[rows] += init
    scale = max(float(want.abs().max()), 1.0)
    assert float((out - want).abs().max()) <= 2e-5 * scale
    again = torch.empty((N, d), device=dev)
    eng.spmm_t_rows(Lt, slot, X, init, again, ws, ed)
    assert torch.equal(again, out)
    eng.spmm_t_rows(Lt, slot, X, None, again, ws, ed)            # without the direct part
    want[rows] -= init
    assert float((again - want).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("d,n_item,drop", [(128, 20000, 0.0), (130, 600, 0.3), (65, 20000, 0.3), (600, 600, 0.0)])   # 600: two 512-column panels
def test_row_sparse_transposed_product_with_the_bitmap_in_lds_is_bit_identical(d, n_item, drop, dev, lib_options):
    """r04: on a large matrix (>= 2^22 stored entries, N <= 1.13 M) `ngcf_spmm_t_rows_f32` runs `spmm_t_rows_bm_kernel`: 16 consecutive
    rows per wave, the membership of an entry's column in the R rows tested on a bitmap in LDS, hits parked and replayed in entry
    order, units drawn from a two-level counter.  Per row the same entries in the same order with the same fmaf: the results must be
    the one-wave-per-row slot-table kernel's bit for bit (cut rows inside a unit: 600 items; a ragged last unit: 200 600 rows;
    device-side edge dropout; with and without the direct part; rows 0..2 and the last 70 among the R rows)."""
    pkg = _pkg()
    eng = pkg.engine
    coo = pkg.graphs.synthetic_bipartite(200_000, n_item, 2_300_000, seed=17, device=dev)
    N = coo["n_user"] + coo["n_item"]
    order = torch.sort(coo["cols"], stable=True).indices
    Lt = eng.LaplacianCSR.from_coo(coo["cols"][order], coo["rows"][order], coo["vals"][order], N, N)
    assert Lt.nnz >= 1 << 22 and (Lt.n_segments > 0 or n_item != 600)
    g = torch.Generator().manual_seed(d)
    rows = torch.unique(torch.cat([torch.randint(0, N, (3000,), generator=g), torch.arange(N - 70, N), torch.arange(0, 3)])).to(dev)
    R = rows.numel()
    X, init = (torch.randn((R, d), generator=g).to(dev) for _ in range(2))
    slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
    slot[rows] = torch.arange(R, dtype=torch.int32, device=dev)
    ed = ([11, 12], drop) if drop else None
    ws = eng.Workspace()
    outs = []
    for bitmap in (0, 1):
        lib_options(t_rows_bitmap=bitmap)
        a = torch.full((N, d + 3), 7.0, device=dev)[:, :d]
        b = torch.full((N, d), 7.0, device=dev)
        eng.spmm_t_rows(Lt, slot, X, init, a, ws, ed)
        eng.spmm_t_rows(Lt, slot, X, None, b, ws, ed)
        outs.append((a.clone(), b.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0].abs().max()) > 0 and bool((outs[1][0][rows] - outs[1][1][rows] - init).abs().max() < 1e-4)


@pytest.mark.parametrize("node_mode,dense_grad", [(None, True), ("reference", True), ("device", True), (None, False), ("device", False)])
def test_training_gradients_are_bit_identical_from_run_to_run(node_mode, dense_grad, dev, monkeypatch):
    """Two training steps from the same seed on the same batch (duplicate users and items in it) give bit-identical gradients of
    every parameter: the gather backward sums duplicates in batch order, L^T . dLE of the row-sparse last layer runs in entry order,
    the weight gradients add their partials in workgroup order - no float atomics anywhere (the reference on CPU is deterministic too)."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import autograd as ag
    if not dense_grad:                                             # the row-sparse hand-over of larger graphs (one host sync), forced here
        monkeypatch.setattr(ag, "DENSE_GRAD_MAX_BYTES", 0)
    coo = pkg.graphs.synthetic_bipartite(4000, 300, 60000, seed=8, device=dev)
    num_dict = {"user": 4000, "item": 300, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    B = 256
    g = torch.Generator().manual_seed(4)
    r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)  # noqa: E731
    batch = dict(year=torch.full((B,), 18, device=dev), u_id=r(4000), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                 pos_item=r(300), neg_item=r(300))
    batch["u_id"][:40] = batch["u_id"][40:80]                      # duplicate users
    batch["pos_item"][:30] = 0                                     # one heavy item row many times
    grads = []
    for _ in range(2):
        torch.manual_seed(21)
        model = pkg.NGCF(65, [65, 65, 65], 0.3, [0.1, 0.1, 0.1], 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, B, dev).to(dev)
        model.train()
        model.mess_dropout_mode = "device"
        if node_mode:
            model.node_dropout_mode = node_mode
        torch.manual_seed(5)
        u, p, n = model(node_flag=node_mode is not None, **batch)
        pkg.BPR(0.025, B)(u, p, n).backward()
        grads.append({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
    assert set(grads[0]) == set(grads[1]) and "user_embedding.weight" in grads[0] and "w2_list.2.bias" in grads[0]
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("M,N", [(1, 10), (64, 5), (65, 1 << 40), (3072, 5940), (8192, 1_100_000), (5000, 3)])
def test_rows_sort_unique_kernel_vs_torch_unique(M, N, dev):
    """ngcf_rows_sort_unique (one workgroup: bitonic sort in LDS, head flags, scan) against torch.unique + a stable sort: distinct
    rows ascending, positions grouped by row in batch order, group bounds, count - integer work, bit-exact."""
    import ctypes as C
    from seoul_tourism_recommendation_ngcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M)
    idx = torch.randint(0, N, (M,), generator=g, dtype=torch.int64).to(dev)
    buf = torch.full((3 * M + 2,), -7, dtype=torch.int64, device=dev)
    order, rows, segptr, cnt = buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.ngcf_rows_sort_unique(p(idx), M, N - 1, p(order), p(rows), p(segptr), p(cnt), None))     # N < 2^19: 32-bit keys
    torch.cuda.synchronize()
    want_rows, inv, counts = torch.unique(idx, return_inverse=True, return_counts=True)
    R = int(cnt.item())
    assert R == want_rows.numel()
    assert torch.equal(rows[:R], want_rows)
    assert torch.equal(order, torch.sort(inv, stable=True).indices)
    want_ptr = torch.zeros(R + 1, dtype=torch.int64, device=dev)
    want_ptr[1:] = torch.cumsum(counts, 0)
    assert torch.equal(segptr[:R + 1], want_ptr)
