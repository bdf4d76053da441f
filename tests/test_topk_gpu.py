"""Top-k selection kernel (experiment.py:104-111, demo.py:234-235 pattern) vs torch.topk."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rows,n_cols,k", [(25, 100, 3), (25, 100, 25), (4, 100, 100), (37, 5000, 100), (8, 100000, 100),
                                             (3, 1500, 1024), (5, 7, 7), (2, 300, 1)])
def test_topk_rows_matches_torch(n_rows, n_cols, k):
    from seoul_tourism_recommendation_ngcf_amd import engine as eng
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n_cols + k)
    s = torch.randn((n_rows, n_cols + 5), generator=g, device=dev)[:, :n_cols]      # strided rows
    vals, idx = eng.topk_rows(s, k)
    tv, ti = torch.topk(s, k, dim=1)
    assert torch.equal(vals, tv)
    assert torch.equal(torch.gather(s, 1, idx), vals) and idx.dtype == torch.int64
    assert all(len(set(r.tolist())) == k for r in idx)                              # no column twice


def test_topk_ties_negative_and_special_values():
    from seoul_tourism_recommendation_ngcf_amd import engine as eng
    dev = torch.device("cuda:0")
    s = torch.tensor([[1.0, 3.0, 3.0, -2.0, 3.0, 0.0, -0.0, 3.0],
                      [-5.0, -1.0, -1.0, -7.0, float("-inf"), -1.0, -3.0, -1.0],
                      [2.0, float("inf"), 2.0, 2.0, 2.0, 2.0, 2.0, 2.0]], device=dev)
    vals, idx = eng.topk_rows(s, 3)
    assert vals.tolist() == [[3.0, 3.0, 3.0], [-1.0, -1.0, -1.0], [float("inf"), 2.0, 2.0]]
    assert idx.tolist() == [[1, 2, 4], [1, 2, 5], [1, 0, 2]]                         # ties: lowest column first
    with pytest.raises(RuntimeError, match="out of range"):
        eng.topk_rows(s, 9)


def test_recommend_topk_on_module_outputs():
    """demo.py:233-235: scores of all items for the returned user rows, top-k of them - one hand-written launch."""
    import seoul_tourism_recommendation_ngcf_amd as pkg
    dev = torch.device("cuda:0")
    coo = pkg.graphs.synthetic_bipartite(2000, 150, 30000, seed=2, device=dev)
    num_dict = {"user": 2000, "item": 150, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    model = pkg.NGCF(65, [64, 64], 0.3, [0.1, 0.1], 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, 8, dev).to(dev).eval()
    with torch.no_grad():
        model.propagate(0)
        u = model.all_users_emb[torch.tensor([3, 77, 1999], device=dev)]
        vals, idx, scores = pkg.engine.recommend_topk(u, model.all_items_emb, 100, return_scores=True)
        want = torch.mm(u.cpu(), model.all_items_emb.cpu().T).to(dev)       # the reference's own op on the CPU (demo.py:233-234)
    # the score matrix is the reference's CPU torch.mm within fp32 summation-order noise; the selection is exact on it
    torch.testing.assert_close(scores, want, atol=1e-5, rtol=1e-5)
    tv, ti = torch.topk(scores, 100)
    assert torch.equal(vals, tv) and torch.equal(torch.gather(scores, 1, idx), vals)
    assert idx.shape == (3, 100) and idx.dtype == torch.int64 and int(idx.max()) < 150


@pytest.mark.parametrize("B,n_items,D,k", [(1, 25, 193, 3), (1, 100, 260, 100), (7, 100, 514, 20), (64, 5000, 65, 100),
                                           (300, 20000, 128, 50), (1024, 3000, 96, 7), (9, 257, 33, 257)])
def test_recommend_topk_kernel_vs_torch(B, n_items, D, k):
    """experiment.py:93,104-109 (`torch.mm(u_embeds, pos_i_embeds.T)` then topk 3 / ks) and the demo's all-item ranking at
    several shapes, odd widths and strided item tables (rows of all_E) included."""
    from seoul_tourism_recommendation_ngcf_amd import engine as eng
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B * 7 + D)
    u = torch.randn((B, D), generator=g, device=dev)
    items = torch.randn((n_items + 3, D + 5), generator=g, device=dev)[3:, 2:2 + D]      # a strided, offset view
    vals, idx, scores = eng.recommend_topk(u, items, k, return_scores=True)
    want = torch.mm(u.cpu(), items.cpu().T).to(dev)                  # the reference's op on the CPU (the oracle), not rocBLAS
    torch.testing.assert_close(scores, want, atol=2e-4, rtol=2e-5)
    tv, _ = torch.topk(scores, k)
    assert torch.equal(vals, tv) and torch.equal(torch.gather(scores, 1, idx), vals)
    assert all(len(set(r.tolist())) == k for r in idx[:16])
    # the same ranking as the reference's path wherever the k-th and (k+1)-th reference scores are not a near-tie
    rv, ri = torch.topk(want, min(k + 1, n_items))
    if k < n_items:
        clear = (rv[:, k - 1] - rv[:, k]) > 1e-3
        assert bool(clear.any())
        assert all(set(idx[b].tolist()) == set(ri[b, :k].tolist()) for b in clear.nonzero().flatten().tolist()[:32])
    with pytest.raises(RuntimeError, match="out of range"):
        eng.recommend_topk(u, items, n_items + 1)
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        eng.recommend_topk(u, items[:, :D - 1], 1)
