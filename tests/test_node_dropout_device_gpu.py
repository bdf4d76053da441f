"""Device-side node dropout: in-kernel hash mask, cumulative and unscaled like NGCF.py:93-100,
checked against the CPU oracle run on the explicitly thinned matrices (mask recomputed on the host)."""
import numpy as np
import pytest
import torch

import ngcf_oracle as orc
from conftest import load_golden
from golden_util import batch_of, ctor_args, lap_list_of, layer_params, sd_of

pytestmark = pytest.mark.gpu
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xff51afd7ed558ccd)) & M64
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xc4ceb9fe1a85ec53)) & M64
    x ^= x >> np.uint64(33)
    return (x & np.uint64(0xFFFFFFFF)).astype(np.uint64)


def keep_mask(rows, cols, seeds, p):
    """The library's mask (csrc/common.h, edge_keep): keyed by the entry's (row, column) in L."""
    key = (np.asarray(rows).astype(np.uint64) << np.uint64(32)) | (np.asarray(cols).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    nnz = key.size
    e = (key * np.uint64(0x9E3779B97F4A7C15)) & M64
    thr = np.uint64(int(p * 4294967296.0))
    keep = np.ones(nnz, bool)
    for s in seeds:
        keep &= mix32(np.uint64(s) ^ e) >= thr
    return keep


def test_dropout_spmm_matches_host_mask_and_statistics():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    eng = pkg.engine
    dev = torch.device("cuda:0")
    coo = pkg.graphs.synthetic_bipartite(30000, 2000, 500000, seed=4, device=dev)
    N = coo["n_user"] + coo["n_item"]
    csr = eng.LaplacianCSR.from_coo(coo["rows"], coo["cols"], coo["vals"], N, N)
    X = torch.randn((N, 128), generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    seeds, p = [12345, 987654321, 5], 0.3
    for n in (1, 2, 3):
        got = eng.spmm(csr, X, edge_drop=(seeds[:n], p, False))
        keep = torch.from_numpy(keep_mask(coo["rows"].cpu().numpy(), coo["cols"].cpu().numpy(), seeds[:n], p)).to(dev)
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p) ** n) < 0.005                       # cumulative thinning
        thin = eng.LaplacianCSR.from_coo(coo["rows"][keep], coo["cols"][keep], coo["vals"][keep], N, N)
        want = eng.spmm(thin, X)                                        # values are NOT rescaled
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-6 * max(scale, 1.0)


def test_dropout_on_the_swept_kernel_and_on_the_transpose():
    """The same mask whatever walks the matrix: the L2-swept kernel (forced onto a small matrix, rows cut into pieces
    included), the row-wise kernels, and the CSR of L^T with `transposed` set - L^T loses exactly the entries L lost."""
    import seoul_tourism_recommendation_ngcf_amd as pkg
    eng = pkg.engine
    dev = torch.device("cuda:0")
    coo = pkg.graphs.synthetic_bipartite(20000, 300, 400000, seed=9, device=dev)      # 300 item rows of ~1 300 entries: cut rows
    N = coo["n_user"] + coo["n_item"]
    rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
    X = torch.randn((N, 128), generator=torch.Generator(device=dev).manual_seed(2), device=dev)
    seeds, p = [77, 1234567], 0.25
    keep = torch.from_numpy(keep_mask(rows.cpu().numpy(), cols.cpu().numpy(), seeds, p)).to(dev)
    thin = eng.LaplacianCSR.from_coo(rows[keep], cols[keep], vals[keep], N, N)
    want = eng.spmm(thin, X)
    scale = max(float(want.abs().max()), 1.0)
    csr = eng.LaplacianCSR.from_coo(rows, cols, vals, N, N)
    plain = eng.spmm(csr, X, edge_drop=(seeds, p, False))
    assert float((plain - want).abs().max()) <= 2e-6 * scale
    csr.set_mode(2)                                                                    # swept wherever the shape allows
    assert csr.swept_rows == N
    swept = eng.spmm(csr, X, edge_drop=(seeds, p, False))
    assert float((swept - want).abs().max()) <= 2e-6 * scale
    assert torch.equal(swept, eng.spmm(csr, X, edge_drop=(seeds, p, False)))           # deterministic
    # the transpose: rows of L^T are columns of L
    order = torch.sort(cols, stable=True).indices
    csr_t = eng.LaplacianCSR.from_coo(cols[order], rows[order], vals[order], N, N)
    thin_t = eng.LaplacianCSR.from_coo(cols[keep], rows[keep], vals[keep], N, N)       # (from_coo sorts)
    want_t = eng.spmm(thin_t, X)
    for mode in (0, 2):
        csr_t.set_mode(mode)
        got_t = eng.spmm(csr_t, X, edge_drop=(seeds, p, True))
        assert float((got_t - want_t).abs().max()) <= 2e-6 * max(float(want_t.abs().max()), 1.0)


def test_module_device_node_dropout_forward_and_gradients():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    dev = torch.device("cuda:0")
    g = load_golden("fwd_sigB_y19")
    sd, b = sd_of(g), batch_of(g)
    model = pkg.NGCF(**ctor_args(g, lap_list_of(g, dev), dev)).to(dev)
    model.load_state_dict(sd)
    model.eval()
    model.node_dropout_mode = "device"
    batch = {k: v.to(dev) for k, v in b.items()}
    torch.manual_seed(77)
    # what the module will draw: its private generator follows torch.manual_seed without consuming the default stream
    seeds = [int(x) for x in torch.randint(0, 2 ** 62, (3,), dtype=torch.int64,
                                           generator=torch.Generator().manual_seed(77 ^ 0x5DEECE66D))]
    state = torch.get_rng_state()
    u, p, n = model(node_flag=True, **batch)
    assert torch.equal(torch.get_rng_state(), state)                 # device mode leaves the default CPU generator alone
    loss = pkg.BPR(0.025, len(b["u_id"]))(u, p, n)
    loss.backward()
    # oracle on the explicitly thinned matrices
    yi = int(g["year_idx"])
    rows, cols, vals = g[f"lap{yi}_rows"], g[f"lap{yi}_cols"], g[f"lap{yi}_vals"]
    N = int(g["meta"][0] + g["meta"][1])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(("w1_list", "w2_list", "item_emb"))}
    uw = torch.from_numpy(g["out_user_weight_after"]).clone().requires_grad_(True)
    w1, b1, w2, b2 = ([leaves[f"{n_}.{k}.{t}"] for k in range(3)] for n_, t in (("w1_list", "weight"), ("w1_list", "bias"), ("w2_list", "weight"), ("w2_list", "bias")))
    E = torch.cat((uw, leaves["item_embedding.weight"]), 0)
    blocks = [E]
    for k in range(3):
        keep = keep_mask(rows, cols, seeds[:k + 1], float(g["meta"][5]))
        Lk = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows[keep], cols[keep]])), torch.from_numpy(vals[keep]), (N, N))
        LE = torch.mm(Lk, E)
        M = torch.nn.functional.linear(LE, w1[k], b1[k]) + torch.nn.functional.linear(E, w1[k], b1[k]) \
            + torch.nn.functional.linear(LE * E, w2[k], b2[k])
        E = torch.nn.functional.leaky_relu(M, 0.2)
        blocks.append(torch.nn.functional.normalize(E, p=2, dim=1))
    all_E = torch.cat(blocks, 1)
    got = torch.cat((model.all_users_emb, model.all_items_emb), 0).detach().cpu()
    np.testing.assert_allclose(got.numpy(), all_E.detach().numpy(), atol=2e-5, rtol=2e-3)
    ou, op, on = orc.gather_torch(all_E, int(g["meta"][0]), b["u_id"], b["pos_item"], b["neg_item"])
    want_loss = orc.bpr_torch(ou, op, on, 0.025, len(b["u_id"]))
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    named = dict(model.named_parameters())
    for k, leaf in list(leaves.items()) + [("user_embedding.weight", uw)]:
        wg = leaf.grad
        scale = float(wg.abs().max())
        np.testing.assert_allclose(named[k].grad.cpu().numpy(), wg.numpy(), atol=2e-3 * scale + 1e-9, rtol=2e-3, err_msg=k)
