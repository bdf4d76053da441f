"""`autograd.E0Cache` (r04): inference forwards keep their all_E and skip the copy of E0 = cat(user table, item table)
(NGCF.py:120-121) while the tables are unchanged and nobody else holds the previous result."""
import pytest
import torch

from seoul_tourism_recommendation_ngcf_amd.autograd import E0Cache


class _Owner:
    pass


def _forward(o, c, padded=False):
    t = torch.zeros(10, 8)[:, :6] if padded else torch.zeros(10, 8)    # the two layouts of autograd._alloc_all_E
    o._all_E, o.all_users_emb, o.all_items_emb = t, t[:6, :], t[6:, :]
    c.all_E = t


@pytest.mark.parametrize("padded", [False, True])
def test_a_result_somebody_else_holds_is_never_reused(padded):
    """The hold detection on plain CPU tensors: module-only references -> reusable; the tensor, one of the module's views, a slice
    of a view or a detached alias held by a caller -> not; copies (advanced indexing, clone) do not count.  Both layouts of all_E:
    a plain [N, D] tensor and (D not a multiple of 4) a [N, D] view of a padded buffer."""
    o, c = _Owner(), E0Cache()
    _forward(o, c, padded)
    assert c.only_the_modules(o)
    held = o.all_items_emb
    assert not c.only_the_modules(o)
    del held
    sl = o.all_items_emb[:2]
    assert not c.only_the_modules(o)
    del sl
    whole = o._all_E
    assert not c.only_the_modules(o)
    del whole
    alias = o.all_users_emb.detach()
    assert not c.only_the_modules(o)
    del alias
    capsule = torch.utils.dlpack.to_dlpack(o.all_items_emb)            # an exported view (no new Python reference, no new storage user)
    assert not c.only_the_modules(o)
    del capsule
    copy_ = o.all_items_emb[torch.tensor([0, 1])]
    clone = o.all_users_emb.clone()
    assert c.only_the_modules(o) and copy_.shape[0] == 2 and clone.shape[0] == 6
    # the module's attributes point at another forward's tensors (a graph replay in between): not this cache's to reuse
    t2 = torch.zeros(10, 8)
    o._all_E, o.all_users_emb, o.all_items_emb = t2, t2[:6], t2[6:]
    assert not c.only_the_modules(o)
    c.invalidate()
    assert c.all_E is None and not c.only_the_modules(o)


def test_touched_rows_are_bounded():
    c = E0Cache()
    c.all_E = torch.zeros(4, 4)
    for _ in range(E0Cache.MAX_TOUCHED):
        c.touch(torch.tensor([1]))
    assert len(c.touched) == E0Cache.MAX_TOUCHED
    c.touch(torch.tensor([1]))                                          # one more: forget everything, the next forward copies in full
    assert c.all_E is None and c.touched == []


@pytest.mark.gpu
@pytest.mark.parametrize("embed,layers", [(65, [64, 64]), (130, [128])])
def test_retained_all_E_is_bit_identical_to_a_fresh_one(embed, layers):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import seoul_tourism_recommendation_ngcf_amd as pkg
    dev = torch.device("cuda:0")
    U, I, B = 3000, 200, 128
    coo = pkg.graphs.synthetic_bipartite(U, I, 40000, seed=1, device=dev)
    lap = [pkg.graphs.to_sparse_coo(coo)]
    num = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    models = []
    for reuse in (True, False):
        torch.manual_seed(7)
        m = pkg.NGCF(embed, layers, None, None, 0.5, lap, num, B, dev).to(dev).eval()      # emb_ratio 0.5: the injected rows move every call
        m.auto_graph = False                                            # the eager inference path (what a big graph takes)
        m.reuse_all_E = reuse
        models.append(m)
    a, b = models
    g = torch.Generator().manual_seed(3)

    def batch():
        r = lambda hi: torch.randint(0, hi, (B,), generator=g).to(dev)  # noqa: E731
        out = dict(year=torch.full((B,), 18, device=dev), u_id=r(U), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                   pos_item=r(I), neg_item=r(I))
        out["u_id"][:4] = out["u_id"][4:8]                              # duplicates
        return out

    def both(bt):
        with torch.no_grad():
            ra, rb = a(node_flag=False, **bt), b(node_flag=False, **bt)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)
        assert torch.equal(a.all_users_emb, b.all_users_emb) and torch.equal(a.all_items_emb, b.all_items_emb)
        assert torch.equal(a.user_embedding.weight, b.user_embedding.weight)
        assert torch.equal(a.all_users_emb[:, :embed], a.user_embedding.weight)      # block 0 IS the table, bit for bit (NGCF.py:120)
    both(batch())
    ptr = a._all_E.data_ptr()
    for _ in range(3):
        both(batch())
        assert a._all_E.data_ptr() == ptr                               # same buffer, E0 not copied again: only the injected rows
    assert b._e0_cache.all_E is None
    # a caller keeps the item block (demo.py:233): its contents survive the next forward, which takes a fresh all_E
    held = a.all_users_emb
    want = held.clone()
    both(batch())                                                       # (no table changed: only the hold keeps the buffer from being re-used)
    assert torch.equal(held, want) and a._all_E.untyped_storage().data_ptr() != held.untyped_storage().data_ptr()
    assert not torch.equal(a.all_users_emb, want)                       # the new forward's injected rows differ
    del held
    # in-place updates of a table (an optimizer step, load_state_dict) are seen through the version counter
    both(batch())
    ptr = a._all_E.data_ptr()
    with torch.no_grad():
        a.user_embedding.weight.add_(0.25)
        b.user_embedding.weight.add_(0.25)
    both(batch())
    sd = {k: v.clone() * 0.5 for k, v in a.state_dict().items()}
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    both(batch())
    # a write through .data from outside is the documented blind spot: invalidate_all_E() is the remedy
    a.item_embedding.weight.data[3] = 1.0
    b.item_embedding.weight.data[3] = 1.0
    a.invalidate_all_E()
    both(batch())
    # propagate() alone (bench.py's step) and a training forward in between
    with torch.no_grad():
        a.propagate(0)
        b.propagate(0)
        a.propagate(0)
    assert torch.equal(a.all_items_emb, b.all_items_emb)
    bt = batch()
    for m in (a, b):
        m.train()
        u, p, n = m(node_flag=False, **bt)
        pkg.BPR(0.025, B)(u, p, n).backward()
        m.eval()
    both(batch())
