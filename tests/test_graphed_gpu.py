"""hipGraph replay of the inference forward (seoul_tourism_recommendation_ngcf_amd/graphed.py) against the eager forward."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pkg():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _batch(g, B, U, I, dev):
    r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)  # noqa: E731
    return dict(year=torch.full((B,), 18, device=dev), u_id=r(U), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                pos_item=r(I), neg_item=r(I))


@pytest.mark.parametrize("embed,layers", [(65, [64, 64]), (130, [128])])
def test_graphed_forward_replays_the_eager_forward(embed, layers, dev):
    pkg = _pkg()
    slices = pkg.graphs.seoul_standin(dev, seed=5, n_user=700, n_item=40)
    lap = [pkg.graphs.to_sparse_coo(s) for s in slices]
    U, I, B = 700, 40, 96
    num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(3)
    eager = pkg.NGCF(embed, layers, None, None, 0.3, lap, num_dict, B, dev).to(dev).eval()
    graphed_model = copy.deepcopy(eager)
    graphed_model.lap_list = lap
    before = graphed_model.user_embedding.weight.detach().clone()
    fwd = pkg.GraphedForward(graphed_model, B, year_idx=0)
    assert torch.equal(graphed_model.user_embedding.weight.detach(), before)      # capture leaves the weights alone
    g = torch.Generator().manual_seed(11)
    for _ in range(3):                                                            # the injection accumulates over calls
        batch = _batch(g, B, U, I, dev)
        want = eager(node_flag=False, **batch)
        got = fwd(node_flag=False, **batch)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert torch.equal(graphed_model.user_embedding.weight.detach(), eager.user_embedding.weight.detach())
        assert torch.equal(graphed_model.all_items_emb, eager.all_items_emb)
    # eager calls on the SAME module in between (another year slice, a training step that builds the transposed CSR and
    # grows the module's workspace) must not disturb what the captured graph writes through: it owns its buffers
    other = _batch(g, B, U, I, dev)
    other["year"] = torch.full((B,), 19, device=dev)
    for m in (eager, graphed_model):
        m(node_flag=False, **other)
        m.train()
        u, p, n = m(node_flag=False, **other)
        pkg.BPR(0.025, B)(u, p, n).backward()
        m.zero_grad()
        m.eval()
    batch = _batch(g, B, U, I, dev)
    want, got = eager(node_flag=False, **batch), fwd(node_flag=False, **batch)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    bad = _batch(g, B, U, I, dev)
    bad["pos_item"][5] = I                                                        # out of range -> IndexError, like eager
    with pytest.raises(IndexError):
        fwd(node_flag=False, **bad)
    with pytest.raises(IndexError):
        eager(node_flag=False, **bad)                                             # (both have injected the bad batch)
    with pytest.raises(RuntimeError):
        fwd(node_flag=False, **{k: v[:10] for k, v in bad.items()})               # other batch size than captured
    ok = _batch(g, B, U, I, dev)                                                  # usable again after the error
    want, got = eager(node_flag=False, **ok), fwd(node_flag=False, **ok)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
