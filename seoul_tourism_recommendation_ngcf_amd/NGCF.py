"""`NGCF` - the reference's nn.Module surface, backed by the HIP propagation engine.

Drop-in for `/root/reference/model/NGCF.py:7-156`: same constructor, same keyword `forward`,
same parameter names/shapes (so the reference's 23 checkpoints load with `strict=True`), same
externally read attributes (`all_users_emb`, `all_items_emb`, demo.py:233), same in-place
mutation of `user_embedding.weight` (NGCF.py:114-115).  What differs is where the work runs:
every tensor op of the forward body is a hand-written gfx950 kernel behind the C ABI in
`include/ngcf_hip.h`; torch only owns the memory.  There is no CPU path: calling `forward`
with the module on the CPU raises.
"""
from __future__ import annotations

import os
import warnings
import weakref
from typing import List, Optional

import torch
import torch.nn as nn

from . import engine as _eng
from .autograd import E0Cache, GatherTriple, propagate_with_grad, static_result_baseline, static_result_counts


class _ReplayTrain(torch.autograd.Function):
    """autograd node of a graph-replayed training forward: forward = replay of the captured forward graph, backward = replay of the
    captured backward graph; the static gradient buffers are handed to the model's parameters."""

    @staticmethod
    def forward(ctx, runner, *params):
        runner.fwd.replay()
        ctx.runner = runner
        # The saved activations live in the graphs' static pool and are shared by every replay: while this node can still run its
        # backward, a second replay would overwrite what that backward needs.  The token dies with the node (the caller dropped
        # the loss without a backward) and is dropped when the backward has run; `_TrainGraphs.outstanding()` reads it.
        ctx.token = _Token()
        runner.pending = weakref.ref(ctx.token)
        fresh = [torch.empty_like(o) for o in runner.outs]             # fresh tensors like the eager path (the graph's own are overwritten
        torch._foreach_copy_(fresh, [o.detach() for o in runner.outs])  # by the next replay): one multi-tensor copy
        return tuple(fresh)

    @staticmethod
    def backward(ctx, *grads):
        r = ctx.runner
        if ctx.token is None or r.pending is None or r.pending() is not ctx.token:
            raise RuntimeError("NGCF: this training forward was replayed from a captured graph and its saved activations have been "
                               "overwritten (a second backward through the same forward, or a replay forced in between); set "
                               "model.auto_train_graph = False for such loops")
        ctx.token, r.pending = None, None
        have = [(s, g) for s, g in zip(r.gouts, grads) if g is not None and g.data_ptr() != s.data_ptr()]
        if have:
            torch._foreach_copy_([s for s, _ in have], [g for _, g in have])      # the incoming gradients into the graph's inputs: one launch
        for s, g in zip(r.gouts, grads):
            if g is None:
                s.zero_()
        # The gradients leave as the graph's own buffers (no copies): with `zero_grad()` between two backward passes (the default
        # sets `.grad` to None) autograd adopts them and the next replay finds nobody holding them.  A `.grad` that still IS one
        # of these buffers - gradient accumulation over several backward calls, or `zero_grad(set_to_none=False)` - gets its own
        # storage first, or the replay would overwrite what it is about to be added to.
        for p, g in zip(r.params, r.gins):
            if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                p.grad = p.grad.clone()
        r.bwd.replay()
        return (None, *[None if g is None else g.detach() for g in r.gins])


class _Token:
    """(weak-referenceable marker of an outstanding replayed forward)"""
    __slots__ = ("__weakref__",)


class _TrainGraphs:
    """`NGCF.auto_train_graph`: the eager training forward of one shape and its backward, captured as two hipGraphs that share a
    memory pool (the pattern of `torch.cuda.make_graphed_callables`, with one difference that matters here: the differentiable
    inputs of the capture are fresh ALIASES of the parameters - same storage, new leaves.  The parameters themselves carry
    AccumulateGrad nodes from earlier eager steps for as long as the caller keeps a loss tensor alive, and those belong to the
    default stream: the captured backward would wait for an event recorded outside the capture, which ends the process on this
    runtime)."""

    def __init__(self, model, year_idx: int, node_flag: bool, has_neg: bool, args):
        self.model, self.year_idx, self.node_flag, self.has_neg = model, year_idx, node_flag, has_neg
        self.flat = torch.cat(args)                                    # static index vectors: views of one buffer (one copy per call)
        self.idx, o = [], 0
        for a in args:
            self.idx.append(self.flat[o:o + a.numel()])
            o += a.numel()
        self.params = model._diff_params()
        self.pending = None
        # Both graphs bake in the addresses of everything their kernels touch.  The workspace is this object's own (the module's
        # grow-only one is re-allocated when an eager call asks for more - the other year slice, a larger batch - and the old block
        # goes back to the allocator); the module's scratch / status / seed-state tensors are long-lived, referenced here so that
        # their memory cannot be handed out again, and compared before every replay (`intact`).
        self.ws = _eng.Workspace()
        uw = model.user_embedding.weight.data
        rows_before = uw[self.idx[0].clamp(0, model.n_user - 1)].clone()   # the warm-up forwards inject for real (NGCF.py:114): undone below
        saved_ws, model._ws = model._ws, self.ws
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                              # warm-up: everything lazily created exists before the capture
                for _ in range(3):
                    al = self._aliases()
                    outs = self._body(al)
                    torch.autograd.grad(outs, al, grad_outputs=[torch.empty_like(o) for o in outs], allow_unused=True)
                    del outs, al
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            al = self._aliases()
            status_host = model._status_host_buf()
            with torch.cuda.graph(self.fwd, pool=pool):
                outs = self._body(al)
                status_host.copy_(model._status, non_blocking=True)    # last node: the status word where the host can see it
            self.outs = tuple(outs)
            self.all_E = model._all_E.detach()                         # the capture's all_E: what the module's attributes alias after a replay
            self.gouts = tuple(torch.empty_like(o) for o in outs)
            with torch.cuda.graph(self.bwd, pool=pool):
                gins = torch.autograd.grad(outs, al, grad_outputs=self.gouts, allow_unused=True)
            self.gins = tuple(gins)
        finally:
            model._ws = saved_ws
        # emb_ratio != 1: the injection blends (w <- w (1-r) + feats r) and is not idempotent - the three warm-up forwards must not
        # count, or the batch's user rows would be blended four times by the time of the first replay
        uw[self.idx[0].clamp(0, model.n_user - 1)] = rows_before
        self._keep = (self.ws.buf, model._scratch, model._status, getattr(model, "_seed_state", None), model._status_host)
        self._baked = self._pointers()
        model._all_E = model.all_users_emb = model.all_items_emb = None     # (they alias the capture's all_E: set again by every replay)
        self._free = static_result_baseline(self.all_E)                      # ... which only this object and its autograd nodes hold now

    def _pointers(self):
        m = self.model
        return tuple(None if t is None else t.data_ptr() for t in (self.ws.buf, m._scratch, m._status, getattr(m, "_seed_state", None), m._status_host))

    def intact(self) -> bool:
        """Every buffer whose address the two graphs bake in is still the tensor it was at capture."""
        return self._pointers() == self._baked

    def outstanding(self) -> bool:
        """A forward replayed from these graphs can still run its backward (its saved activations must not be overwritten)."""
        return self.pending is not None and self.pending() is not None

    def _aliases(self):
        return [p.detach().requires_grad_(True) for p in self.params]

    def _body(self, aliases):
        m = self.model
        saved = (m.check_indices, m._forced_year_idx, m.auto_train_graph, m._alias)
        m.check_indices, m._forced_year_idx, m.auto_train_graph, m._alias = False, self.year_idx, False, aliases   # no host syncs inside
        try:
            u_id, age, sex, month, day, dow, pos = self.idx[:7]
            neg = self.idx[7] if self.has_neg else torch.empty(0, dtype=torch.int64)
            u, p, n = NGCF.forward(m, u_id, u_id, age, sex, month, day, dow, pos, neg, self.node_flag)
        finally:
            m.check_indices, m._forced_year_idx, m.auto_train_graph, m._alias = saved
        return (u, p, n) if self.has_neg else (u, p)

    def __call__(self, *args):
        torch.cat(args, out=self.flat)
        outs = _ReplayTrain.apply(self, *self.params)
        m = self.model
        m._e0_cache.invalidate()                                       # (the replay injected rows through `.data`)
        m._all_E = self.all_E                                          # NGCF.py:148-149: the attributes follow the forward that just ran
        m.all_users_emb, m.all_items_emb = self.all_E[:m.n_user, :], self.all_E[m.n_user:, :]
        return outs


_HOST_RNG = {"ok": None}


def _host_rng_ok() -> bool:
    """Whether `ngcf_torch_cpu_bernoulli` (csrc/hostrng.hip: torch's CPU `bernoulli_` stream regenerated in vector loops) may stand
    in for torch's own serial kernel in this process: decided once, by drawing 4 099 elements both ways from copies of one
    generator state - flags, kept count and the advanced state must all be equal.  (Another torch build may draw differently - an
    MKL stream, another state layout: then the reference-mode masks keep coming from torch itself, at torch's speed.)"""
    if _HOST_RNG["ok"] is None:
        ok = False
        try:
            import ctypes as C
            lib = _eng._lib.load()
            g = torch.Generator(device="cpu").manual_seed(0x5EED)
            torch.empty(3, dtype=torch.float32).uniform_(generator=g)          # an odd word position
            st = g.get_state().clone()
            n = 4099
            flags = torch.empty(n, dtype=torch.uint8)
            kept = C.c_int64()
            rc = lib.ngcf_torch_cpu_bernoulli(st.data_ptr(), st.numel(), n, 0.7, flags.data_ptr(), None, 0.0, C.byref(kept))
            want = torch.empty(n, dtype=torch.float64).bernoulli_(0.7, generator=g)
            ok = (rc == 0 and torch.equal(flags.bool(), want != 0) and int(kept.value) == int((want != 0).sum())
                  and torch.equal(st, g.get_state()))
        except Exception:  # noqa: BLE001
            ok = False
        _HOST_RNG["ok"] = bool(ok) and os.environ.get("NGCF_HOST_RNG", "1") != "0"
    return _HOST_RNG["ok"]


def _bernoulli_from_state(st: torch.Tensor, n: int, keep: float, noise_shape=None):
    """n draws of Bernoulli(keep) from the mt19937 state bytes `st` (a `torch.get_rng_state()` tensor, advanced in place) through
    `ngcf_torch_cpu_bernoulli`: (keep flags uint8[n] or None, kept count, noise float32 `noise_shape` or None).  Touches neither
    torch's generator nor the GPU (ctypes releases the GIL for the call): safe on a helper thread."""
    import ctypes as C
    flags = None if noise_shape is not None else torch.empty(n, dtype=torch.uint8)
    noise = torch.empty(noise_shape, dtype=torch.float32) if noise_shape is not None else None
    scale = float(torch.ones(1, dtype=torch.float32).div_(keep)) if noise is not None else 0.0
    kept = C.c_int64()
    _eng._lib.check(_eng._lib.load().ngcf_torch_cpu_bernoulli(st.data_ptr(), st.numel(), n, keep, None if flags is None else flags.data_ptr(),
                                                              None if noise is None else noise.data_ptr(), scale, C.byref(kept)))
    return flags, int(kept.value), noise


def _reference_bernoulli(n: int, p_drop: float, noise_shape=None, state: Optional[torch.Tensor] = None):
    """`nn.Dropout(p_drop)` in training mode on a CPU tensor of n ones, drawn from torch's DEFAULT CPU generator exactly as the
    reference draws it (NGCF.py:93-100 on float64 ones, NGCF.py:142 on the float32 activations; both take one 64-bit draw per
    element): returns (keep flags uint8[n] or None, kept count, noise float32 `noise_shape` or None) - the flags for the node
    dropout, the noise tensor (0 or 1/(1-p) as torch rounds it) for the message dropout.  `state`: draw from these state bytes
    (advanced in place) instead of the default generator (`_DrawAhead`; needs `_host_rng_ok()`)."""
    keep = 1.0 - float(p_drop)
    if keep >= 1.0 or keep <= 0.0:        # torch's dropout returns its input (p = 0) or zeros (p = 1) WITHOUT touching the generator
        kept_all = keep >= 1.0
        if noise_shape is not None:
            return None, -1, torch.full(noise_shape, 1.0 if kept_all else 0.0, dtype=torch.float32)
        return torch.full((n,), 1 if kept_all else 0, dtype=torch.uint8), n if kept_all else 0, None
    if state is not None:
        return _bernoulli_from_state(state, n, keep, noise_shape)
    if _host_rng_ok():
        st = torch.get_rng_state()
        out = _bernoulli_from_state(st, n, keep, noise_shape)
        torch.set_rng_state(st)
        return out
    if noise_shape is not None:
        return None, -1, torch.nn.functional.dropout(torch.ones(noise_shape, dtype=torch.float32), p=p_drop, training=True)
    flags = torch.nn.functional.dropout(torch.ones(n, dtype=torch.float64), p=p_drop, training=True).type(torch.bool)
    return flags, int(flags.sum()), None


def _draw_program(program, state: Optional[torch.Tensor] = None):
    """The draws of ONE forward in the reference's order (NGCF.py:123-142), as a generator of (keep flags, kept count, noise) per
    layer.  `program` = (stored entries of L, node-dropout p or None, message-dropout p per layer or None, N, widths): the node
    mask of layer k has one flag per entry the earlier layers kept (cumulative, NGCF.py:126), so the sizes follow from the draws
    themselves - nothing here depends on the batch or on the GPU."""
    nnz, p_node, drops, N, widths = program
    n = nnz
    for k in range(len(widths) - 1):
        flags = kept = noise = None
        if p_node is not None:
            flags, kept, _ = _reference_bernoulli(n, p_node, None, state)
            n = kept
        if drops is not None and drops[k] > 0:
            noise = _reference_bernoulli(N * widths[k + 1], drops[k], (N, widths[k + 1]), state)[2]
        yield flags, kept, noise


class _DrawAhead:
    """r04: the NEXT forward's reference-mode masks, drawn on a helper thread while this step's launches, backward and optimizer run.
    The masks of a forward are a function of the generator state alone (`_draw_program`), so when a forward ends the helper starts
    from a COPY of the state the forward left behind and runs the same program again; the next forward takes the result only if the
    default generator is still in exactly that state (nobody seeded it or drew from it in between) and the program is the same
    (same year slice, same dropout rates, same mode) - it then installs the state the helper ended in, as if it had drawn itself.
    Anything else: the result is dropped and the masks are drawn in place.  Bit-identical either way.  Off: NGCF_DRAW_AHEAD=0."""

    MAX_BYTES = 64 << 20                   # (flags + noise of one forward: 7 MB at the Seoul shape; not for C3-sized masks)

    def __init__(self):
        self.thread, self.job = None, None
        self.hits = self.misses = 0
        # the helper writes into PINNED buffers (two sets, used in turn: the masks of one forward may still be on their way to the
        # device - asynchronous copies out of page-locked memory - while the helper already draws the next ones into the other set)
        self.sets = [{}, {}]                # per set: {"program", "flags", "noise", "event"}
        self.turn = 0
        self.spare = {}                     # (program, set index) -> a set put aside when another program came (year slices alternate)
        self.miss_streak = self.pause = 0   # three misses in a row (batches whose programs do not repeat): no helper for 32 forwards

    def __reduce__(self):                  # copy.deepcopy(model), torch.save(model): a copy starts without a helper
        return (_DrawAhead, ())

    @staticmethod
    def wanted(program) -> bool:
        nnz, p_node, drops, N, widths = program
        size = (3 * nnz if p_node is not None else 0) + (4 * N * sum(widths[1:]) if drops is not None else 0)
        return os.environ.get("NGCF_DRAW_AHEAD", "1") != "0" and size <= _DrawAhead.MAX_BYTES and _host_rng_ok()

    def start(self, program):
        import threading
        if not self.wanted(program):
            self.thread = self.job = None
            return
        import ctypes as C
        # every buffer is allocated HERE and the helper makes ONE foreign call for the whole forward (ngcf_torch_cpu_bernoulli_seq):
        # between two calls it would have to wait for the interpreter lock, which the launching thread gives up every 5 ms at worst
        nnz, p_node, drops, N, widths = program
        node_keep = None if p_node is None else 1.0 - p_node
        if node_keep is not None and not 0.0 < node_keep < 1.0:          # p = 0 / p = 1: torch draws nothing; not worth a helper
            self.thread = self.job = None
            return
        if self.pause > 0:                                               # the programs of this caller's batches do not repeat: stand back
            self.pause -= 1
            self.thread = self.job = None
            return
        self.turn ^= 1
        bset = self.sets[self.turn]
        if bset.get("program") != program:                               # (page-locked allocations are slow: once per program and set)
            if bset.get("program") is not None and len(self.spare) < 6:
                self.spare[(bset["program"], self.turn)] = dict(bset)    # another year slice / mode: its buffers wait for its return
            bset.clear()
            bset.update(self.spare.pop((program, self.turn), {}))
        if bset.get("program") != program:
            pin = torch.cuda.is_available()
            bset.update(program=program, event=None,
                        flags=[torch.empty(nnz, dtype=torch.uint8, pin_memory=pin) for _ in widths[1:]] if node_keep is not None else None,
                        noise=[torch.empty((N, w), dtype=torch.float32, pin_memory=pin) if drops is not None and drops[k] > 0 else None
                               for k, w in enumerate(widths[1:])])
        elif bset["event"] is not None:
            bset["event"].synchronize()                                  # the copies that read this set two forwards ago (long done)
        draws, slots = [], []                # draws: (n or -1, keep, flags, noise, scale) as the C routine takes them; slots: per layer
        for k in range(len(widths) - 1):
            fl = nz = i_fl = None
            if node_keep is not None:
                fl = bset["flags"][k]                                    # (a layer keeps at most what layer 0 started from)
                i_fl = len(draws)
                draws.append((nnz if k == 0 else -1, node_keep, fl, None, 0.0))     # -1: as many as the previous layer kept
            if drops is not None and drops[k] > 0:
                pk = 1.0 - drops[k]
                nz = bset["noise"][k]
                if 0.0 < pk < 1.0:
                    draws.append((N * widths[k + 1], pk, None, nz, float(torch.ones(1, dtype=torch.float32).div_(pk))))
                else:
                    nz.fill_(1.0 if pk >= 1.0 else 0.0)
            slots.append((fl, i_fl, nz))
        m = len(draws)
        job = {"program": program, "start": torch.get_rng_state(), "end": None, "rc": None}
        st = job["start"].clone()
        c_n, c_keep = (C.c_int64 * m)(*[d[0] for d in draws]), (C.c_double * m)(*[d[1] for d in draws])
        c_flags = (C.c_void_p * m)(*[None if d[2] is None else d[2].data_ptr() for d in draws])
        c_noise = (C.c_void_p * m)(*[None if d[3] is None else d[3].data_ptr() for d in draws])
        c_scale, c_kept = (C.c_float * m)(*[d[4] for d in draws]), (C.c_int64 * m)()
        fn = _eng._lib.load().ngcf_torch_cpu_bernoulli_seq
        st_ptr, st_len = st.data_ptr(), st.numel()

        def work():
            job["rc"] = fn(st_ptr, st_len, m, c_n, c_keep, c_flags, c_noise, c_scale, c_kept)     # (the GIL is released for the call)

        def finish():                        # on the taking thread: (flags cut to what was drawn, kept count, noise) per layer
            if job["rc"] != 0:
                return None
            out, n_flags = [], nnz
            for fl, i_fl, nz in slots:
                kept = None
                if fl is not None:
                    fl, kept = fl[:n_flags], int(c_kept[i_fl])
                    n_flags = kept
                out.append((fl, kept, nz))
            job["end"] = st
            return out
        job["finish"] = finish
        self.job = job
        # (not a daemon: the interpreter waits for it at exit - a millisecond at most - instead of freeing the buffers under it)
        self.thread = threading.Thread(target=work, name="ngcf-draw-ahead", daemon=False)
        self.thread.start()

    def copied(self):
        """The forward has queued its copies of the masks it took: the set they live in is free once the stream got past them."""
        bset = self.sets[self.turn]
        if torch.cuda.is_available() and bset.get("program") is not None:
            if bset["event"] is None:
                bset["event"] = torch.cuda.Event()
            bset["event"].record()

    def take(self, program):
        """The helper's masks for `program` if they are the ones this forward would draw now, else None."""
        job, thread = self.job, self.thread
        self.job = self.thread = None
        if job is None:
            return None
        thread.join()
        result = job["finish"]() if job["program"] == program and torch.equal(job["start"], torch.get_rng_state()) else None
        if result is None:
            self.misses += 1
            self.miss_streak += 1
            if self.miss_streak >= 3:
                self.miss_streak, self.pause = 0, 32
            return None
        torch.set_rng_state(job["end"])
        self.hits += 1
        self.miss_streak = 0
        return result


def _spmm_mode() -> int:
    """SpMM kernel choice for the cached Laplacians (include/ngcf_hip.h, ngcf_csr_set_mode); NGCF_SPMM_MODE overrides."""
    return int(os.environ.get("NGCF_SPMM_MODE", "3"))


class NGCF(nn.Module):
    def __init__(self,
                 embed_size: int,
                 layer_size: list,
                 node_dropout: float,
                 mess_dropout: list,
                 emb_ratio: float,
                 lap_list: list,
                 num_dict: dict,
                 batch_size: int,
                 device):
        super().__init__()
        self.n_user = num_dict['user']
        self.n_item = num_dict['item']
        self.emb_size = embed_size
        self.weight_size = layer_size
        self.n_layer = len(self.weight_size)
        self.batch_size = batch_size
        self.device = device
        self.node_dropout = node_dropout
        self.mess_dropout = mess_dropout
        self.emb_ratio = emb_ratio

        # registration order fixes the state_dict key order of the reference's checkpoints
        fw = self.emb_size // 5                                     # NGCF.py:39-43
        self.month_emb = nn.Embedding(num_dict['month'], fw)
        self.day_emb = nn.Embedding(num_dict['day'], fw)
        self.sex_emb = nn.Embedding(num_dict['sex'], fw)
        self.age_emb = nn.Embedding(num_dict['age'], fw)
        self.dow_emb = nn.Embedding(num_dict['dayofweek'], fw)
        self.item_embedding = nn.Embedding(self.n_item, self.emb_size)
        self.user_embedding = nn.Embedding(self.n_user, self.emb_size)

        self.lap_list = lap_list                                    # borrowed, NGCF.py:52
        for emb in (self.user_embedding, self.item_embedding, self.age_emb, self.sex_emb,
                    self.month_emb, self.dow_emb, self.day_emb):    # NGCF.py:58-68
            nn.init.kaiming_uniform_(emb.weight)

        # W1_k then W2_k per layer, as set_layers does (NGCF.py:73-78): the same torch.manual_seed then gives the same
        # initial Linear weights as the reference class (tests/test_reference_checkpoints.py)
        dims = [self.emb_size] + list(self.weight_size)
        w1, w2 = [], []
        for k in range(self.n_layer):
            w1.append(nn.Linear(dims[k], dims[k + 1], bias=True))
            w2.append(nn.Linear(dims[k], dims[k + 1], bias=True))
        self.w1_list = nn.Sequential(*w1)
        self.w2_list = nn.Sequential(*w2)
        self.node_dropout_list = nn.Sequential(
            *[nn.Dropout(p=self.node_dropout) for _ in range(self.n_layer if self.node_dropout is not None else 0)])
        self.mess_dropout_list = nn.Sequential(
            *[nn.Dropout(p=self.mess_dropout[k]) for k in range(self.n_layer if self.mess_dropout is not None else 0)])

        # engine state (not part of the state_dict)
        self._csr_cache = {}
        self._filt = {}                  # thinned CSRs of the reference-mode node dropout, re-used from step to step
        self._filt_maps = {}             # entry maps of their transposes
        self._row_sorted = {}            # id(lap_list entry) -> its COO rows are non-decreasing
        self._ws = _eng.Workspace()
        self._carry: List[Optional[torch.Tensor]] = [None, None]
        self._scratch: Optional[torch.Tensor] = None
        self._status: Optional[torch.Tensor] = None
        self._status_host: Optional[torch.Tensor] = None
        self.check_indices = True        # raise IndexError on out-of-range ids (one host sync per forward)
        # where the random masks come from: "reference" = torch's default CPU generator, drawn exactly where the reference
        # draws (bit-identical masks for the same torch.manual_seed, one host round trip per layer); "device" = counter
        # hash evaluated inside the kernels (same distribution and semantics, another stream, no host work)
        self.node_dropout_mode = "reference"
        self.mess_dropout_mode = "reference"
        self.all_users_emb = None
        self.all_items_emb = None
        self._all_E = None
        # Inference calls (eval mode, no autograd, node_flag=False) of launch-bound sizes are captured into a hipGraph on first
        # use and replayed afterwards (`_forward_graphed`): unchanged callers (experiment.py:66-119, demo.py:213-236) get the
        # replay speed without knowing about it.  `auto_graph = False` switches it off.
        self.auto_graph = True
        self.auto_graph_max_bytes = 64 << 20     # all_E above this size: the forward is not launch-bound, nothing to gain
        self.index_check_every = 16              # graph replays: read the (sticky) status word every k-th call; 1 = every call
        self._graphs = {}                        # (sizes, year slice, parameter addresses) -> GraphedForward, most recent last
        self._graph_calls = 0
        self._graph_seen = set()
        self._plist = None
        self._year_tag, self._year_idx = None, 0
        self._forced_year_idx = None
        # Training calls (train mode, autograd on) of launch-bound sizes with both dropouts drawn on the DEVICE: forward and backward
        # are captured as two hipGraphs on the second call of a shape (`_TrainGraphs`, over the eager path) and
        # replayed afterwards; `loss.backward()` then replays the backward graph.  Unchanged training loops (experiment.py:45-58) get
        # a host cost of two graph launches for the model's share of the step.  `auto_train_graph = False` switches it off.
        self.auto_train_graph = True
        self._alias = None                       # (capture only) the parameter aliases propagate() differentiates instead of the parameters
        self._train_graphs = {}                  # key -> graphed core module, most recent last
        self._train_seen = {}                    # key -> the stream its first (eager) call ran on
        self._train_calls = 0
        self._capture_warned = False
        # Inference forwards keep their all_E and skip the copy of E0 into block 0 while both embedding tables are unchanged and
        # nobody else holds the previous result (autograd.E0Cache).  `reuse_all_E = False`: a fresh all_E and a full copy per call.
        self.reuse_all_E = True
        self._e0_cache = E0Cache()

    # ------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------
    def _dev(self) -> torch.device:
        dev = self.user_embedding.weight.device
        if dev.type != "cuda":
            raise RuntimeError(
                "NGCF propagation engine: parameters are on '%s'. This module runs hand-written HIP kernels for "
                "MI355X only and has no CPU fallback; call .to('cuda') first." % dev)
        return dev

    def _status_buf(self, dev):
        if self._status is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._status

    def _status_host_buf(self):
        """Pinned host mirror of `_status`: the captured training forward copies the word there as its last node (r04), so
        `_peek_status` sees an earlier replay's out-of-range id without a host sync."""
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._status_host

    def _peek_status(self):
        """Raise IndexError for an out-of-range id of any graph replay the GPU has finished - no host sync unless there is one."""
        h = self._status_host
        if h is not None and int(h[0]) != 0:
            self.check_indices_now()
        for g in list(self._graphs.values()):
            g.peek_status()

    def _scratch_buf(self, dev):
        if self._scratch is None or self._scratch.device != dev or self._scratch.numel() != self.n_user:
            self._scratch = torch.full((self.n_user,), -1, dtype=torch.int32, device=dev)
        return self._scratch

    def laplacian_csr(self, year_idx: int) -> "_eng.LaplacianCSR":
        """CSR of `lap_list[year_idx]`, built once and cached (replaces the per-call `.to(device)`, NGCF.py:118)."""
        dev = self._dev()
        L = self.lap_list[year_idx]            # IndexError for a bad year_idx, like the reference
        key = (year_idx, id(L), str(dev))
        csr = self._csr_cache.get(key)
        if csr is None:
            N = self.n_user + self.n_item
            if tuple(L.shape) != (N, N):
                raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(L.shape)} and {N}x{self.emb_size})")
            rows, cols, vals = self._sorted_coo(L, dev)
            csr = _eng.LaplacianCSR.from_coo(rows, cols, vals, N, N)
            csr.set_mode(_spmm_mode())         # long-lived matrix: worth the L2-swept plan where that pays
            self._csr_cache = {k: v for k, v in self._csr_cache.items() if year_idx not in k[:2]}
            self._csr_cache[key] = csr
        return csr

    @staticmethod
    def _sorted_coo(L: torch.Tensor, dev):
        """COO triplets of a `lap_list` entry on the device, row-sorted (stable) so that the CSR keeps their order:
        the position of an entry in this order is its entry number (used by the device-side node dropout)."""
        if not L.is_sparse:
            raise RuntimeError("lap_list entries must be torch sparse COO tensors (matrix.py:79-83)")
        idx = L._indices().to(dev)
        val = L._values().to(device=dev, dtype=torch.float32)
        rows, cols = idx[0], idx[1]
        if rows.numel() > 1 and bool((rows[1:] < rows[:-1]).any()):
            order = torch.sort(rows, stable=True).indices
            rows, cols, val = rows[order], cols[order], val[order]
        return rows, cols, val

    def laplacian_csr_t(self, year_idx: int) -> "_eng.LaplacianCSR":
        """CSR of `lap_list[year_idx]` transposed (for `L^T . dLE` in the backward), built on first use."""
        dev = self._dev()
        L = self.lap_list[year_idx]
        key = ("T", year_idx, id(L), str(dev))
        csr = self._csr_cache.get(key)
        if csr is None:
            rows, cols, vals = self._sorted_coo(L, dev)
            csr, order = self._transposed_csr(torch.stack([rows, cols]), vals)
            csr.set_mode(_spmm_mode())
            self._csr_cache[key] = csr
            # entry j of L^T is entry order[j] of L: lets the reference-mode node dropout thin L^T with the flags drawn for L
            self._csr_cache[("Tmap",) + key[1:]] = order.to(torch.int32) if order.numel() < 2 ** 31 - 1 else None
        return csr

    def _transposed_csr(self, idx: torch.Tensor, val: torch.Tensor):
        """CSR of the transpose (the device-mode edge dropout is keyed by (row, column) of L: no entry map is needed) and the
        position in L of every entry of L^T."""
        N = self.n_user + self.n_item
        order = torch.sort(idx[1], stable=True).indices          # by column = row of L^T, original order kept inside
        return _eng.LaplacianCSR.from_coo(idx[1][order], idx[0][order], val[order], N, N), order

    def _diff_params(self):
        """The parameters a training forward differentiates, in the order `_alias` is read: the two embedding tables, then W1, b1,
        W2, b2 of every layer (the feature tables enter through `.data`, NGCF.py:103-115: no gradient)."""
        w1, b1, w2, b2 = self._layer_params()
        return [self.user_embedding.weight, self.item_embedding.weight, *w1, *b1, *w2, *b2]

    def _layer_params(self):
        return ([l.weight for l in self.w1_list], [l.bias for l in self.w1_list],
                [l.weight for l in self.w2_list], [l.bias for l in self.w2_list])

    def _reference_draws(self, year_idx: int, node_ref: bool, drop, mess_ref: bool):
        """Everything the reference draws from torch's default CPU generator during one forward, in its order
        (NGCF.py:123-142): per layer first the node-dropout keep mask (`nn.Dropout(p)` on float64 ones, on the
        matrix already thinned by the earlier layers: cumulative, unscaled, NGCF.py:93-100), then the message-dropout
        noise tensor (`nn.Dropout(p_k)` on the [N, d_k+1] activations, NGCF.py:142; no draw for p == 0, as in torch).
        Same `torch.manual_seed` -> bit-identical kept edge sets and zero patterns as the reference on CPU.
        Returns (per-layer thinned CSRs or None, their transposes' builder or None, per-layer noise tensors or None).
        """
        dev = self._dev()
        N = self.n_user + self.n_item
        widths = [self.emb_size] + list(self.weight_size)
        csrs, flags, masks = [], [], []
        if node_ref:
            L = self.lap_list[year_idx]
            row_sorted = self._row_sorted.get(id(L))
            if row_sorted is None:
                r = L._indices()[0]
                row_sorted = self._row_sorted[id(L)] = bool(r.numel() < 2 or not bool((r[1:] < r[:-1]).any()))
            if not row_sorted:           # entry numbers of the stored COO differ from the CSR's: the rebuild path (matrix.py emits sorted rows)
                return self._reference_draws_rebuild(year_idx, drop, mess_ref)
            src = self.laplacian_csr(year_idx)
        # torch's CPU bernoulli spreads a draw over every thread it may use; on a many-core host (a GPU node's 256 CPUs) that is
        # slower than a few threads by an order of magnitude (26 ms instead of ~3 ms for 0.9 M flags).  The numbers drawn do not
        # depend on the thread count (each chunk skips ahead in one stream), so the draws run on at most 16 threads.
        n_thr = torch.get_num_threads()
        if n_thr > 16:
            torch.set_num_threads(16)
        try:
            return self._reference_draws_body(year_idx, node_ref, drop, mess_ref, src if node_ref else None, dev, N, widths)
        finally:
            if n_thr > 16:
                torch.set_num_threads(n_thr)

    def _reference_draws_body(self, year_idx, node_ref, drop, mess_ref, src, dev, N, widths):
        csrs, flags, masks = [], [], []
        program = (src.nnz if node_ref else 0, float(self.node_dropout) if node_ref else None,
                   tuple(float(x) for x in drop) if mess_ref else None, N, tuple(widths))
        ahead = self.__dict__.get("_draw_ahead")
        if ahead is None:
            ahead = self.__dict__["_draw_ahead"] = _DrawAhead()
        draws = ahead.take(program)                          # drawn by the helper thread during the previous step, or None
        ahead_hit = draws is not None                        # (then the masks sit in page-locked buffers: asynchronous copies)
        for k, (keep, n_kept, noise) in enumerate(draws if ahead_hit else _draw_program(program)):
            if node_ref:
                # one flag per entry of the matrix as the previous layers left it (cumulative, NGCF.py:126); the thinned CSR is a
                # device compaction of the previous one into buffers this module keeps from step to step (ngcf_csr_filter)
                keep = keep.to(dev, non_blocking=ahead_hit)
                src = src.filtered(keep, None, n_kept, reuse=self._filt.pop(("L", year_idx, k), None))
                self._filt[("L", year_idx, k)] = src
                csrs.append(src)
                flags.append(keep)
            masks.append(noise.to(dev, non_blocking=ahead_hit) if noise is not None else None)
        if ahead_hit:
            ahead.copied()
        ahead.start(program)                                 # the next forward's masks, from the state this one leaves behind
        # the thinned matrices are not symmetric: their transposes are built only if a backward needs them
        return (csrs if node_ref else None, (lambda: self._thinned_transposes(year_idx, csrs, flags)) if node_ref else None,
                masks if mess_ref else None)

    def _thinned_transposes(self, year_idx: int, csrs, flags):
        """(thinned L_k)^T for every layer, as device compactions of the cached L^T (and of each other): entry j of the transpose
        is kept iff the flag of its twin in L_k-1 is set, found through an entry map that is carried from layer to layer."""
        dev = self._dev()
        src_t = self.laplacian_csr_t(year_idx)
        emap = self._csr_cache.get(("Tmap", year_idx, id(self.lap_list[year_idx]), str(dev)))
        if emap is None:
            raise RuntimeError("reference-mode node dropout: more than 2^31 stored entries")
        lib = _eng._lib.load()
        out = []
        for k, (csr_k, keep) in enumerate(zip(csrs, flags)):
            n_src = src_t.nnz
            t = src_t.filtered(keep, emap, csr_k.nnz, reuse=self._filt.pop(("T", year_idx, k), None))
            self._filt[("T", year_idx, k)] = t
            out.append(t)
            if k + 1 < len(csrs):
                nxt = self._filt_maps.get((year_idx, k))
                if nxt is None or nxt.numel() < max(csr_k.nnz, 1) or nxt.device != dev:
                    nxt = self._filt_maps[(year_idx, k)] = torch.empty(max(n_src, 1), dtype=torch.int32, device=dev)
                with _eng._on(dev):
                    _eng._lib.check(lib.ngcf_csr_filter_remap(t._h, _eng._ptr(keep), _eng._ptr(emap), n_src,
                                                              _eng.C.c_void_p(csr_k.filter_pos), _eng._ptr(nxt), _eng._stream()))
                emap = nxt[:csr_k.nnz]
            src_t = t
        return out

    def _reference_draws_rebuild(self, year_idx: int, drop, mess_ref: bool):
        """The same draws for a `lap_list` entry whose COO is not row-sorted: the mask numbers entries in stored order, so the
        thinned COO is formed with torch indexing and a CSR is built from it per layer (one-off shapes; `Matrix` emits sorted rows)."""
        dev = self._dev()
        N = self.n_user + self.n_item
        widths = [self.emb_size] + list(self.weight_size)
        csrs, kept, masks = [], [], []
        L = self.lap_list[year_idx]
        idx = L._indices().to(dev)
        val = L._values().to(device=dev, dtype=torch.float32)
        for k in range(self.n_layer):
            mask = _reference_bernoulli(int(val.numel()), self.node_dropout)[0].to(dev).bool()
            idx, val = idx[:, mask], val[mask]
            order = torch.sort(idx[0], stable=True).indices
            csrs.append(_eng.LaplacianCSR.from_coo(idx[0][order], idx[1][order], val[order], N, N))
            kept.append((idx, val))
            if mess_ref and drop[k] > 0:
                masks.append(_reference_bernoulli(N * widths[k + 1], drop[k], (N, widths[k + 1]))[2].to(dev, non_blocking=False))
            else:
                masks.append(None)
        return csrs, (lambda: [self._transposed_csr(i, v)[0] for i, v in kept]), masks if mess_ref else None

    def _private_seeds(self, n: int):
        """64-bit seeds for the device-mode hash streams, from a generator of the module's own: the default CPU stream is
        consumed only where the reference consumes it.  Follows torch.manual_seed (re-seeded when the default seed changes)."""
        src = torch.initial_seed()
        if getattr(self, "_seed_gen", None) is None or self._seed_src != src:
            self._seed_gen = torch.Generator(device="cpu").manual_seed(src ^ 0x5DEECE66D)
            self._seed_src = src
        return [int(x) for x in torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, generator=self._seed_gen)]

    def _device_seeds(self) -> torch.Tensor:
        """The 2 n_layer seed words of ONE training forward, as a fresh device tensor.  A persistent state tensor is seeded from
        the module's private generator (which follows `torch.manual_seed`: the first forward after seeding uses exactly the values
        `_private_seeds` would hand out) and stepped on the device before every later forward (`ngcf_seeds_advance`); the forward
        gets a copy of its own, so a backward that runs after another forward still recomputes its own masks.  No host random
        numbers, no host round trip - and inside a captured hipGraph every replay draws new masks."""
        dev = self._dev()
        n = 2 * self.n_layer
        src = torch.initial_seed()
        st = getattr(self, "_seed_state", None)
        if st is None or st.device != dev or st.numel() != n or self._seed_state_src != src:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("device-mode dropout: run one training forward before capturing (the seed state is created on first use)")
            fresh = torch.tensor(self._private_seeds(n), dtype=torch.int64, device=dev)
            if st is not None and st.device == dev and st.numel() == n:
                st.copy_(fresh)                  # re-seeded (torch.manual_seed): in place - captured graphs bake this tensor's address in
            else:
                self._seed_state = fresh
            self._seed_state_src = src
        else:
            with _eng._on(dev):                  # (the launch stream must be the MODEL's device's, whatever device is current)
                _eng._lib.check(_eng._lib.load().ngcf_seeds_advance(_eng._ptr(self._seed_state), n, _eng._stream()))
        return self._seed_state.clone()

    # ------------------------------------------------------------------------------------
    # propagation (NGCF.py:120-149)
    # ------------------------------------------------------------------------------------
    def propagate(self, year_idx: int = 0, node_flag: bool = False) -> torch.Tensor:
        """all_E = [E0 | norm(E1) | ... | norm(En)]  for the current parameters; sets all_users_emb/all_items_emb."""
        self._dev()
        if self.node_dropout_mode not in ("reference", "device"):
            raise ValueError("node_dropout_mode must be 'reference' or 'device'")
        if self.mess_dropout_mode not in ("reference", "device"):
            raise ValueError("mess_dropout_mode must be 'reference' or 'device'")
        drop = [0.0] * self.n_layer
        if self.training and self.mess_dropout is not None:            # nn.Dropout follows train()/eval(), NGCF.py:142
            drop = [float(p) for p in self.mess_dropout[:self.n_layer]]
        node_ref = bool(node_flag) and self.node_dropout_mode == "reference"
        mess_ref = any(p > 0 for p in drop) and self.mess_dropout_mode == "reference"
        csrs, csrs_t_fn, masks = None, None, None
        if node_ref or mess_ref:
            csrs, csrs_t_fn, masks = self._reference_draws(year_idx, node_ref, drop, mess_ref)
        edge_drops = None
        node_dev = bool(node_flag) and csrs is None                    # node dropout in "device" mode: the cached CSR, thinned in-kernel
        if csrs is None:
            csrs = [self.laplacian_csr(year_idx)] * self.n_layer
            csrs_t_fn = lambda: [self.laplacian_csr_t(year_idx)] * self.n_layer   # noqa: E731
        mess_dev = any(p > 0 for p in drop) and masks is None
        seeds = [0] * self.n_layer
        keep_alive = None
        if node_dev or mess_dev:
            # "device" modes: the seeds of this forward live in device memory and reach the kernels as tagged addresses
            # (include/ngcf_hip.h, NGCF_SEED_PTR_TAG): words 0..n-1 the node-dropout seeds (layer k uses 0..k: cumulative, unscaled -
            # the mask is a counter-based hash of (seed, row, column) evaluated inside the SpMM), words n..2n-1 the message-dropout
            # seeds of the layer epilogues.  The backward recomputes the masks from the same words.
            keep_alive = self._device_seeds()
            tag = lambda i: (0xD5ED << 48) | (keep_alive.data_ptr() + 8 * i)   # noqa: E731
            if node_dev:
                ns = [tag(k) for k in range(self.n_layer)]
                edge_drops = [(ns[:k + 1], float(self.node_dropout)) for k in range(self.n_layer)]
            if mess_dev:
                seeds = [tag(self.n_layer + k) for k in range(self.n_layer)]
        w1, b1, w2, b2 = self._layer_params()
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        if self._alias is not None:                                    # _TrainGraphs: same storage, fresh autograd leaves
            n = self.n_layer
            uw, iw = self._alias[0], self._alias[1]
            w1, b1, w2, b2 = (self._alias[2 + j * n:2 + (j + 1) * n] for j in range(4))
        all_E = propagate_with_grad(self, csrs, csrs_t_fn, uw, iw, w1, b1, w2, b2, drop, seeds, edge_drops, masks, keep_alive)
        self._all_E = all_E
        self.all_users_emb = all_E[:self.n_user, :]                    # NGCF.py:148-149
        self.all_items_emb = all_E[self.n_user:, :]
        return all_E

    # ------------------------------------------------------------------------------------
    # forward (NGCF.py:102-156)
    # ------------------------------------------------------------------------------------
    def _graph_key(self, dev, sizes, year_idx):
        # the parameter tensors' addresses: from a cached list of the Parameter objects (walking the module tree costs 25 us, more
        # than the replay's launch) that is rebuilt after train() / eval(), .to() / .cuda(), load_state_dict() and every 16th call
        self._key_calls = getattr(self, "_key_calls", 0) + 1           # (its own counter: `_graph_calls` only moves on inference replays)
        if self._plist is None or self._key_calls % 16 == 0:
            self._plist = list(self.parameters())
        ptrs = tuple(p.data_ptr() for p in self._plist)
        return (sizes, year_idx, id(self.lap_list[year_idx]), str(dev), ptrs, float(self.emb_ratio))     # (emb_ratio is a kernel argument)

    def train(self, mode: bool = True):
        self._plist = None
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._plist = None
        self._train_graphs.clear()               # (their static buffers live on the old device / dtype)
        self._e0_cache.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def invalidate_all_E(self):
        """Forget the retained all_E (`reuse_all_E`): the next inference forward copies E0 in full.  Needed only after a write to an
        embedding table through `.data` from outside this module (such writes bypass the version counter the cache watches)."""
        self._e0_cache.invalidate()

    def load_state_dict(self, *args, **kwargs):
        self._plist = None
        return super().load_state_dict(*args, **kwargs)

    def _year_index(self, year: torch.Tensor) -> int:
        """`year.unique()[0] % 18` (NGCF.py:117) = the smallest year of the batch, read back from the device once per tensor:
        the same tensor OBJECT passed again unmodified (same version counter) needs no second host sync."""
        if self._forced_year_idx is not None:          # GraphedTrainStep: the slice was fixed when the step was captured
            return self._forced_year_idx
        if not year.numel():
            return 0
        ref, ver = self._year_tag if self._year_tag is not None else (None, -1)
        if ref is None or ref() is not year or ver != year._version:    # identity of the OBJECT (an address alone is re-used by the allocator)
            self._year_idx = int(year.min().item() % 18)
            self._year_tag = (weakref.ref(year), year._version)
        return self._year_idx

    def _forward_graphed(self, dev, year, u_id, age, sex, month, day, dow, pos_item, neg_item):
        """The inference forward as a hipGraph replay (graphed.GraphedForward): captured on first use per (index vector lengths,
        year slice), re-captured when a parameter or `lap_list` entry was replaced (`.to()`, a new tensor assigned); in-place
        updates (`load_state_dict`, optimizer steps) need nothing - the graph reads the parameters when it runs.  Returns fresh
        tensors like the eager path.  Out-of-range ids raise IndexError at the next call after the GPU has run the replay (its last
        node copies the sticky status word to pinned host memory: `_peek_status`, no sync), at the latest `index_check_every` calls
        later (a synchronising read); `check_indices_now()` reads it on demand."""
        from .graphed import GraphedForward
        year_idx = self._year_index(year)
        sizes = (len(u_id), len(pos_item), len(neg_item))
        for v in (age, sex, month, day, dow):
            if len(v) != sizes[0]:
                raise RuntimeError("shape mismatch: feature index vectors and u_id differ in length")
        key = self._graph_key(dev, sizes, year_idx)                    # IndexError for a bad year_idx, like the reference
        g = self._graphs.pop(key, None)
        if g is None:
            if key not in self._graph_seen:                            # first call of a shape runs eagerly: its ids are checked at once,
                if len(self._graph_seen) > 64:                         # and a shape that never comes back is never captured
                    self._graph_seen.clear()
                self._graph_seen.add(key)
                return None
            stale = [k for k in self._graphs if k[:4] == key[:4]]      # same shape, replaced parameters: those graphs are dead
            for k in stale + list(self._graphs)[:max(0, len(self._graphs) - len(stale) - 7)]:
                self._graphs.pop(k, None)
            g = GraphedForward(self, sizes[0], year_idx, with_neg=sizes[2] > 0, pos_size=sizes[1], neg_size=sizes[2] or None)
        self._graphs[key] = g
        if self._static_result_held(g.out[0], g._free):
            return None                                                # a caller kept the last replay's all_*_emb: eager, into fresh tensors
        self._graph_calls += 1
        if self.check_indices:
            self._peek_status()                                        # an earlier replay's bad id, as soon as the GPU got there
        g.load_inputs(u_id=u_id, age=age, sex=sex, month=month, day=day, dow=dow, pos_item=pos_item,
                      neg_item=neg_item if sizes[2] > 0 else None)
        u, p, n = g.replay(check=False)
        if self.check_indices and self._graph_calls % max(1, int(self.index_check_every)) == 0:
            g.check_status()
        outs = [torch.empty_like(u), torch.empty_like(p)] + ([torch.empty_like(n)] if sizes[2] > 0 else [])
        torch._foreach_copy_(outs, [u, p] + ([n] if sizes[2] > 0 else []))         # fresh tensors like the eager path, one launch
        return outs[0], outs[1], (outs[2] if sizes[2] > 0 else torch.empty(0))

    def _static_result_held(self, static_all_E, free_counts) -> bool:
        """Does a caller still hold the previous replay's `all_users_emb` / `all_items_emb` / `_all_E` (or a view of them)?  They are
        views of the graph's static all_E, which the next replay overwrites - the reference hands out a fresh `cat` per call
        (NGCF.py:147-149), so a held tensor must stay what it was: then this forward runs eagerly (r04; r03 documented the
        overwrite as a deviation).  The module's own references are dropped first - both paths set them again."""
        d = self.__dict__
        if d.get("_all_E") is static_all_E or getattr(d.get("all_users_emb"), "_base", None) is static_all_E:
            d["_all_E"] = d["all_users_emb"] = d["all_items_emb"] = None
        return static_result_counts(static_all_E) != free_counts

    def check_indices_now(self):
        """Raise IndexError if any graph-replayed forward since the last check saw an out-of-range id."""
        for g in self._graphs.values():
            g.check_status()
        if self._train_graphs and self._status is not None and int(self._status.item()) != 0:
            self._status.zero_()
            if self._status_host is not None:
                self._status_host.zero_()
            raise IndexError("index out of range in NGCF.forward (u_id / feature ids / pos_item / neg_item)")

    def _train_graph_wanted(self, node_flag, u_id, pos_item, neg_item) -> bool:
        """A differentiable forward of a launch-bound size whose random masks (if any) are drawn on the device.  r04: `self.training`
        is NOT required - the reference's own loop calls `self.model.eval()` at the end of its first epoch and never `train()` again
        (experiment.py:61,72), so from epoch 2 on it trains in eval mode: autograd on, `node_flag=True` (node dropout still drawn,
        NGCF.py:124-126), message dropout off (`nn.Dropout` follows eval, NGCF.py:142)."""
        if not (self.auto_train_graph and torch.is_grad_enabled() and len(u_id) > 0 and len(pos_item) > 0):
            return False
        if len(u_id) + len(pos_item) + len(neg_item) > 8192:            # beyond this GatherTriple.backward sizes its problem on the host
            return False
        if (self.n_user + self.n_item) * (self.emb_size + sum(self.weight_size)) * 4 > min(self.auto_graph_max_bytes, 32 << 20):
            return False                                                # (32 MB: where GatherTriple.backward stays free of host syncs)
        if node_flag and self.node_dropout and self.node_dropout_mode != "device":
            return False                                                # reference-mode masks are drawn on the host: nothing to capture
        if (self.training and self.mess_dropout is not None and any(float(p) > 0 for p in self.mess_dropout[:self.n_layer])
                and self.mess_dropout_mode != "device"):
            return False
        if not self.user_embedding.weight.requires_grad and not any(p.requires_grad for p in self._diff_params()):
            return False                                                # nothing to differentiate: the plain forward
        return not torch.cuda.is_current_stream_capturing()

    def _capture_refusal(self, dev, first_stream) -> Optional[str]:
        """Why the training forward must NOT be captured right now, or None.  A captured backward that depends on anything recorded
        outside the capture ends the PROCESS on this runtime (SIGSEGV inside hipStreamEndCapture - gpurun_out/r03/gputests29b.log,
        DESIGN.md 7 - not an exception that could be caught), so everything that could smuggle such a dependency in, or run foreign
        code inside the capture, keeps the eager path."""
        for p in self._diff_params():
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
                return "a differentiable parameter carries a backward / post-accumulate-grad hook"
        if self._backward_hooks or self._backward_pre_hooks:
            return "the module carries a backward hook"
        if torch.is_anomaly_enabled():
            return "autograd anomaly mode is on"
        if torch._C._autograd._top_saved_tensors_default_hooks(True) is not None:
            return "saved-tensor hooks are active"
        if torch.cuda.current_stream(dev).cuda_stream != first_stream:
            return "the current stream is not the one the first (eager) call of this shape ran on"
        return None

    def _forward_train_graphed(self, dev, year, u_id, age, sex, month, day, dow, pos_item, neg_item, node_flag):
        """The training forward as a hipGraph replay whose autograd node replays the captured backward (module docstring of
        `auto_train_graph`).  The first call of a shape runs eagerly (its ids are checked at once); the second captures: three
        warm-up iterations on a side stream, then the two graphs - the dropout seed state is put back afterwards, so the masks of
        the following steps are exactly those of an eager run.  Parameters are read when the graphs run: optimizer steps and
        `load_state_dict` need nothing; a replaced parameter tensor gives a new key."""
        year_idx = self._year_index(year)
        has_neg = len(neg_item) > 0
        sizes = (len(u_id), len(pos_item), len(neg_item))
        for v in (age, sex, month, day, dow):
            if len(v) != sizes[0]:
                raise RuntimeError("shape mismatch: feature index vectors and u_id differ in length")
        key = self._graph_key(dev, sizes, year_idx) + (      # + everything else a capture bakes into its kernel arguments
            bool(node_flag), self.node_dropout_mode, self.mess_dropout_mode, self.node_dropout,
            None if self.mess_dropout is None else tuple(float(x) for x in self.mess_dropout), bool(self.training))
        to_dev = lambda t: t.to(device=dev, dtype=torch.int64).contiguous()   # noqa: E731
        args = [to_dev(t) for t in (u_id, age, sex, month, day, dow, pos_item)] + ([to_dev(neg_item)] if has_neg else [])
        g = self._train_graphs.pop(key, None)
        if g is not None and not g.intact():                           # a baked buffer was replaced: these graphs are dead
            g = None
        draws = bool((node_flag and self.node_dropout) or
                     (self.training and self.mess_dropout is not None and any(float(p) > 0 for p in self.mess_dropout[:self.n_layer])))
        if draws and getattr(self, "_seed_state", None) is not None and self._seed_state_src != torch.initial_seed():
            if g is not None:
                self._train_graphs[key] = g
            return None                                                # re-seeded since the last draw: this call draws eagerly (and re-seeds in place)
        if g is not None and (g.outstanding() or self._static_result_held(g.all_E, g._free)):
            self._train_graphs[key] = g
            return None                                                # an earlier replay still awaits its backward, or a caller kept its
                                                                       # all_*_emb (views of the graph's static all_E): this forward runs eagerly
        if g is None:
            cur = torch.cuda.current_stream(dev).cuda_stream
            if key not in self._train_seen:
                if len(self._train_seen) > 64:
                    self._train_seen.clear()
                self._train_seen[key] = cur
                return None
            why = self._capture_refusal(dev, self._train_seen[key])
            if why is not None:
                self._train_seen[key] = cur
                if not self._capture_warned:
                    self._capture_warned = True
                    warnings.warn(f"NGCF.auto_train_graph: the training forward is not captured ({why}); it keeps the eager path", stacklevel=3)
                return None
            for k in list(self._train_graphs)[:max(0, len(self._train_graphs) - 3)]:     # a handful of shapes (full batch, last batch)
                self._train_graphs.pop(k, None)
            if getattr(self, "_seed_state", None) is None:
                self._device_seeds()                                   # (created outside the capture; nothing has drawn from it yet)
            self._scratch_buf(dev)
            seed_state = self._seed_state.clone()
            status = self._status_buf(dev)
            with _eng._on(dev):                                        # (streams and graphs belong to the model's device)
                g = _TrainGraphs(self, year_idx, bool(node_flag), has_neg, args)
            self._seed_state.copy_(seed_state)                         # the warm-up forwards stepped it
            if self.check_indices and int(status.item()) != 0:
                status.zero_()
                raise IndexError("index out of range in NGCF.forward (u_id / feature ids / pos_item / neg_item)")
        self._train_graphs[key] = g
        self._train_calls += 1
        if self.check_indices:
            self._peek_status()                                        # an earlier replay's bad id, as soon as the GPU got there
        with _eng._on(dev):
            outs = g(*args)
        if self.check_indices and self._train_calls % max(1, int(self.index_check_every)) == 0:
            self.check_indices_now()
        return outs[0], outs[1], (outs[2] if has_neg else torch.empty(0))      # (fresh tensors: `_ReplayTrain.forward` copies them out)

    def forward(self, year, u_id, age, sex, month, day, dow, pos_item, neg_item, node_flag):
        dev = self._dev()
        status = self._status_buf(dev)
        fw = self.emb_size // 5
        if 5 * fw != self.emb_size:
            # the reference fails at NGCF.py:114 with a shape-mismatch RuntimeError for such widths
            raise RuntimeError(f"shape mismatch: value tensor of shape [{len(u_id)}, {5 * fw}] cannot be broadcast "
                               f"to indexing result of shape [{len(u_id)}, {self.emb_size}]")
        if (self.auto_graph and not self.training and not node_flag and not torch.is_grad_enabled() and len(u_id) > 0 and
                len(pos_item) > 0 and (self.n_user + self.n_item) * (self.emb_size + sum(self.weight_size)) * 4 <= self.auto_graph_max_bytes
                and not torch.cuda.is_current_stream_capturing()):
            out = self._forward_graphed(dev, year, u_id, age, sex, month, day, dow, pos_item, neg_item)
            if out is not None:
                return out
        if self._train_graph_wanted(node_flag, u_id, pos_item, neg_item):
            out = self._forward_train_graphed(dev, year, u_id, age, sex, month, day, dow, pos_item, neg_item, node_flag)
            if out is not None:
                return out
        # feature injection into user_embedding.weight.data, no autograd (NGCF.py:103-115)
        u_idx = u_id.to(device=dev, dtype=torch.int64).contiguous()
        with torch.no_grad():
            keep = _eng.feature_inject(
                self.user_embedding.weight.data,
                (self.age_emb.weight.data, self.sex_emb.weight.data, self.month_emb.weight.data,
                 self.day_emb.weight.data, self.dow_emb.weight.data),
                (age, sex, month, day, dow), u_idx, self.emb_ratio, self._scratch_buf(dev), status)
        self._e0_cache.touch(u_idx)                                    # (a `.data` write: the retained all_E follows these rows)

        year_idx = self._year_index(year)                              # == year.unique()[0] % 18, NGCF.py:117
        self.propagate(year_idx, bool(node_flag))

        p_idx = pos_item.to(device=dev, dtype=torch.int64).contiguous()
        n_idx = neg_item.to(device=dev, dtype=torch.int64).contiguous() if len(neg_item) > 0 else None
        neg_i_embeddings = torch.empty(0)                                       # NGCF.py:153
        if self._all_E.requires_grad:
            outs = GatherTriple.apply(self._all_E, self.n_user, status, u_idx, p_idx, n_idx)
            u_embeddings, pos_i_embeddings = outs[0], outs[1]
            if n_idx is not None:
                neg_i_embeddings = outs[2]
        else:
            u_embeddings, pos_i_embeddings, neg = _eng.gather_rows3(                       # NGCF.py:151-155, one launch
                self._all_E, ((u_idx, 0, self.n_user), (p_idx, self.n_user, self.n_item), (n_idx, self.n_user, self.n_item)), status)
            if neg is not None:
                neg_i_embeddings = neg
        if self.check_indices and not torch.cuda.is_current_stream_capturing():   # (a captured step checks the sticky word after its replay)
            if int(status.item()) != 0:
                status.zero_()
                raise IndexError("index out of range in NGCF.forward (u_id / feature ids / pos_item / neg_item)")
        del keep
        return u_embeddings, pos_i_embeddings, neg_i_embeddings

