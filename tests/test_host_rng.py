"""`ngcf_torch_cpu_bernoulli` (csrc/hostrng.hip, r04): the reference's dropout masks come from torch's default CPU generator
(`nn.Dropout` on CPU tensors, /root/reference/model/NGCF.py:93-100,142).  The routine regenerates that mt19937 stream in vector
loops; it must be torch's own serial kernel bit for bit - flags, noise values, kept count AND the generator state afterwards - at
every word position (odd positions, block boundaries) and size.  Host code: runs without a GPU."""
import ctypes as C  # noqa: I001

import pytest
import torch

import importlib

from seoul_tourism_recommendation_ngcf_amd import _lib

ngcf_mod = importlib.import_module("seoul_tourism_recommendation_ngcf_amd.NGCF")      # (the package re-exports the class under this name)


def _ours(n, keep, want_noise):
    lib = _lib.load()
    st = torch.get_rng_state().clone()
    flags = torch.empty(n, dtype=torch.uint8)
    noise = torch.empty(n, dtype=torch.float32) if want_noise else None
    scale = float(torch.ones(1).div_(keep)) if want_noise else 0.0
    k = C.c_int64()
    _lib.check(lib.ngcf_torch_cpu_bernoulli(st.data_ptr(), st.numel(), n, keep, flags.data_ptr(), noise.data_ptr() if want_noise else None,
                                            scale, C.byref(k)))
    return st, flags, noise, int(k.value)


@pytest.mark.parametrize("pre", [0, 3, 617, 1247, 624 * 3 + 1])
@pytest.mark.parametrize("n", [1, 7, 311, 312, 313, 20_000])
def test_host_draws_are_torchs_own(pre, n):
    for p_drop, dtype in ((0.3, torch.float64), (0.1, torch.float32)):
        torch.manual_seed(100 + pre)
        if pre:
            torch.rand(pre)                              # float draws take ONE word each: odd positions, several blocks in
        st1, flags, noise, kept = _ours(n, 1 - p_drop, dtype == torch.float32)
        want = torch.nn.functional.dropout(torch.ones(n, dtype=dtype), p_drop, True)     # what the reference calls
        assert torch.equal(flags.bool(), want != 0) and kept == int((want != 0).sum())
        assert torch.equal(st1, torch.get_rng_state())                                   # the generator is where torch left its own
        if noise is not None:
            assert torch.equal(noise, want)                                              # 0 or 1/(1-p), rounded as torch rounds it


def test_module_helper_consumes_the_default_generator_like_the_reference():
    """`NGCF._reference_bernoulli` (what the reference-mode forward calls): same masks and the same generator state as two
    `F.dropout` calls in the reference's order - the node mask on float64 ones, then the message noise on float32 ones."""
    assert ngcf_mod._host_rng_ok()
    torch.manual_seed(77)
    want_keep = torch.nn.functional.dropout(torch.ones(5003, dtype=torch.float64), 0.3, True) != 0
    want_noise = torch.nn.functional.dropout(torch.ones((37, 65), dtype=torch.float32), 0.1, True)
    after = torch.get_rng_state()
    torch.manual_seed(77)
    keep, n_kept, _ = ngcf_mod._reference_bernoulli(5003, 0.3)
    _, _, noise = ngcf_mod._reference_bernoulli(37 * 65, 0.1, (37, 65))
    assert torch.equal(keep.bool(), want_keep) and n_kept == int(want_keep.sum()) and torch.equal(noise, want_noise)
    assert torch.equal(torch.get_rng_state(), after)


def test_degenerate_probabilities_do_not_touch_the_generator():
    """`F.dropout` with p = 0 returns its input and with p = 1 zeros, both without a draw: so must the helper."""
    torch.manual_seed(5)
    before = torch.get_rng_state()
    for p in (0.0, 1.0):
        keep, n_kept, _ = ngcf_mod._reference_bernoulli(100, p)
        _, _, noise = ngcf_mod._reference_bernoulli(6 * 7, p, (6, 7))
        want = torch.nn.functional.dropout(torch.ones(100, dtype=torch.float64), p, True)
        assert torch.equal(keep.bool(), want != 0) and n_kept == int((want != 0).sum())
        assert torch.equal(noise, torch.nn.functional.dropout(torch.ones((6, 7)), p, True))
    assert torch.equal(torch.get_rng_state(), before)


def test_bad_state_is_refused():
    lib = _lib.load()
    st = torch.get_rng_state().clone()
    assert lib.ngcf_torch_cpu_bernoulli(st.data_ptr(), 100, 5, 0.5, None, None, 0.0, None) != 0        # too short for the layout
    st[8:12] = 0                                                                                      # left = 0: not a state torch produces
    assert lib.ngcf_torch_cpu_bernoulli(st.data_ptr(), st.numel(), 5, 0.5, None, None, 0.0, None) != 0


def test_draw_ahead_hands_over_exactly_what_the_forward_would_draw():
    """r04: `_DrawAhead` draws the next forward's reference-mode masks on a helper thread from a copy of the generator state.  The
    result is taken only if the default generator is still in that state and the program (entries of L, rates, widths) is the same;
    then flags, kept counts, noise tensors and the generator state afterwards are what drawing in place gives - and after a
    reseed, a foreign draw or another program the helper's work is dropped."""
    import copy
    import importlib
    mod = importlib.import_module("seoul_tourism_recommendation_ngcf_amd.NGCF")
    if not mod._host_rng_ok():
        pytest.skip("this torch build draws differently: masks come from torch itself, nothing is drawn ahead")
    program = (9001, 0.3, (0.1, 0.0, 0.25), 321, (10, 8, 6, 7))       # node dropout + message dropout on two of three layers

    def in_place(seed):
        torch.manual_seed(seed)
        return list(mod._draw_program(program)), torch.get_rng_state()

    def same(a, b):
        for (f1, k1, n1), (f2, k2, n2) in zip(a, b):
            assert k1 == k2 and torch.equal(f1, f2) and ((n1 is None and n2 is None) or torch.equal(n1, n2))

    want, state_after = in_place(5)
    assert want[1][2] is None and want[0][2].shape == (321, 8) and want[1][0].numel() == want[0][1]   # p = 0: no draw; cumulative sizes
    ahead = mod._DrawAhead()
    torch.manual_seed(5)
    ahead.start(program)
    got = ahead.take(program)
    assert got is not None and ahead.hits == 1
    same(got, want)
    assert torch.equal(torch.get_rng_state(), state_after)
    # the generator moved on in between (a reseed; a foreign draw): dropped, the forward draws in place
    torch.manual_seed(5)
    ahead.start(program)
    torch.manual_seed(6)
    assert ahead.take(program) is None and ahead.misses == 1
    torch.manual_seed(5)
    ahead.start(program)
    torch.rand(1)
    before = torch.get_rng_state()
    assert ahead.take(program) is None and torch.equal(torch.get_rng_state(), before)
    # another program (the other year slice, eval mode from epoch 2 on): dropped
    torch.manual_seed(5)
    ahead.start(program)
    assert ahead.take((9001, 0.3, None, 321, (10, 8, 6, 7))) is None
    assert ahead.take(program) is None                                  # nothing in flight
    # masks the size of C3's are not drawn ahead; a copied module starts without a helper
    assert not mod._DrawAhead.wanted((100_000_000, 0.1, None, 1_100_000, (128, 128, 128, 128)))
    assert ahead.misses == 3 and ahead.pause == 32                      # three misses in a row: it stands back (next test)
    ahead = mod._DrawAhead()
    torch.manual_seed(5)
    ahead.start(program)
    assert copy.deepcopy(ahead).job is None
    assert ahead.take(program) is not None


def test_draw_ahead_with_programs_that_do_not_repeat():
    """Year slices (or modes) that alternate from batch to batch: the helper's guess - the same program again - is wrong every time.
    Its buffers are put aside per program and come back (no page-locked allocation per forward), and after three misses in a row
    it stands back for 32 forwards; a hit resets the count."""
    import importlib
    mod = importlib.import_module("seoul_tourism_recommendation_ngcf_amd.NGCF")
    if not mod._host_rng_ok():
        pytest.skip("this torch build draws differently: nothing is drawn ahead")
    A = (5000, 0.3, (0.1,), 100, (8, 8))
    B = (7000, 0.3, (0.1,), 100, (8, 8))
    ahead = mod._DrawAhead()
    torch.manual_seed(1)
    seen = {}
    for i in range(3):                                                  # A is guessed, B comes (and the other way round): three misses
        guess, comes = (A, B) if i % 2 == 0 else (B, A)
        ahead.start(guess)
        key = (guess, ahead.turn)
        ptr = ahead.sets[ahead.turn]["flags"][0].data_ptr()
        assert seen.setdefault(key, ptr) == ptr
        assert ahead.take(comes) is None
    assert ahead.misses == 3 and ahead.pause == 32
    for _ in range(32):
        ahead.start(A)
        assert ahead.job is None and ahead.take(A) is None              # standing back: nothing in flight, nothing counted
    assert ahead.misses == 3 and ahead.pause == 0
    ahead.start(A)
    assert ahead.take(A) is not None and ahead.hits == 1 and ahead.miss_streak == 0
    # the buffers of a program that went away and came back are the ones it had (same set index)
    ahead.turn = 1
    ahead.start(A)
    a_ptr = ahead.sets[0]["flags"][0].data_ptr()
    ahead.take(A)
    ahead.turn = 1
    ahead.start(B)
    ahead.take(B)
    ahead.turn = 1
    ahead.start(A)
    assert ahead.sets[0]["flags"][0].data_ptr() == a_ptr
    ahead.take(A)
