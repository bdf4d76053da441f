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
    """demo.py:233-235: scores of all items for the returned user rows, top-k of them."""
    import seoul_tourism_recommendation_ngcf_amd as pkg
    dev = torch.device("cuda:0")
    coo = pkg.graphs.synthetic_bipartite(2000, 150, 30000, seed=2, device=dev)
    num_dict = {"user": 2000, "item": 150, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    model = pkg.NGCF(65, [64, 64], 0.3, [0.1, 0.1], 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, 8, dev).to(dev).eval()
    with torch.no_grad():
        all_E = model.propagate(0)
        u = model.all_users_emb[torch.tensor([3, 77, 1999], device=dev)]
        vals, idx = pkg.engine.recommend_topk(u, model.all_items_emb, 100)
        tv, _ = torch.topk(torch.mm(u, model.all_items_emb.T), 100)
    assert torch.equal(vals, tv) and idx.shape == (3, 100) and int(idx.max()) < 150
