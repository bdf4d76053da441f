"""hipGraph replay of the inference forward (NGCF.py:102-156) for launch-bound graphs.

On the Seoul-sized graph (5 940 nodes) one forward is ~25 kernel launches of a few microseconds
each: the step is bound by launch and Python overhead, not by the GPU.  `GraphedForward` captures the device side
of `NGCF.forward` - feature injection, propagation, the three row gathers - once into a hipGraph (through
`torch.cuda.CUDAGraph`: the library launches on torch's current stream, so its kernels are captured like torch's
own) and replays it per batch.  What stays outside the graph is what needs the host: choosing the year slice
(`year.unique()[0] % 18`, NGCF.py:117 - fixed at capture) and raising for out-of-range ids (the status word is
read after the replay).

Opt-in, eval mode only (no dropout draws), fixed batch size.  The returned tensors and `model.all_users_emb /
all_items_emb` are the graph's static buffers: valid until the next replay (the eager forward returns fresh ones).
"""
from __future__ import annotations

import torch

from . import engine as _eng
from .autograd import propagate_forward, static_result_baseline


class _Buffers:
    """What `propagate_forward` keeps on its `owner` between calls (workspace, carry ping-pong, padded E0).  The graph
    bakes their addresses in, so it owns a private set: an eager call on the same module (another year slice, a backward
    pass) may grow or replace the module's own buffers without touching anything a replay writes through."""

    def __init__(self):
        self._ws = _eng.Workspace()
        self._carry = [None, None]


class GraphedForward:
    def __init__(self, model, batch_size: int, year_idx: int = 0, with_neg: bool = True, pos_size: int = None,
                 neg_size: int = None, criterion=None):
        """`batch_size`: length of u_id and of the five feature index vectors; `pos_size` / `neg_size`: lengths of pos_item /
        neg_item when they differ from it (experiment.py:82-91 passes 25 users and 25 candidate items; demo.py one user row per
        query and every item).  `criterion` (a `BPR` module, r04): the loss of the gathered rows is captured too - the whole step
        forward + BPR (experiment.py:45-56 without the backward) is then ONE graph launch; `self.loss` holds it after a replay."""
        if model.training:
            raise RuntimeError("GraphedForward captures the eval-mode forward: call model.eval() first")
        if model.emb_size % 5 != 0:
            raise RuntimeError("embed_size must be a multiple of 5 (NGCF.py:39-43,114)")
        self.model, self.B, self.year_idx, self.with_neg = model, int(batch_size), int(year_idx), bool(with_neg)
        self.Bp = self.B if pos_size is None else int(pos_size)
        self.Bn = (self.B if neg_size is None else int(neg_size)) if self.with_neg else 0
        dev = model._dev()
        self.dev = dev
        z = lambda n: torch.zeros(n, dtype=torch.int64, device=dev)  # noqa: E731
        self.inputs = {k: z(self.B) for k in ("u_id", "age", "sex", "month", "day", "dow")}
        self.inputs["pos_item"] = z(self.Bp)
        self.inputs["neg_item"] = z(max(self.Bn, 1))
        self.csr = model.laplacian_csr(self.year_idx)      # built (and planned) outside the capture; kept alive here
        self.bufs = _Buffers()
        if criterion is not None and not self.with_neg:
            raise RuntimeError("GraphedForward: capturing the loss needs the negative items (with_neg=True)")
        self.criterion, self._loss_ws, self.loss = criterion, _eng.Workspace(), None
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        # r04: a pinned host mirror of the status word, copied by the LAST node of the graph: the next call can see an earlier
        # replay's out-of-range id without a host sync (`peek_status`), instead of up to `index_check_every` calls later
        self.status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.scratch = torch.full((model.n_user,), -1, dtype=torch.int32, device=dev)
        # the weights the injection overwrites must look the same at capture time as before it
        saved = model.user_embedding.weight.data[:1].clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up: every cached buffer exists before the capture
            for _ in range(2):
                out = self._body()
                if criterion is not None:
                    self._loss(out)
            del out
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()
            if criterion is not None:
                self.loss = self._loss(self.out)
            self.status_host.copy_(self.status, non_blocking=True)
        torch.cuda.synchronize(dev)
        model.user_embedding.weight.data[:1].copy_(saved)  # row 0 was injected with the all-zero warm-up batch
        self.status.zero_()
        self.status_host.zero_()
        self._baked = self._baked_pointers()
        self._free = static_result_baseline(self.out[0])     # what the static all_E reads when nobody but this object holds it

    def _body(self):
        m, i = self.model, self.inputs
        with torch.no_grad():
            _eng.feature_inject(m.user_embedding.weight.data,
                                (m.age_emb.weight.data, m.sex_emb.weight.data, m.month_emb.weight.data,
                                 m.day_emb.weight.data, m.dow_emb.weight.data),
                                (i["age"], i["sex"], i["month"], i["day"], i["dow"]), i["u_id"], m.emb_ratio,
                                self.scratch, self.status)
            w1, b1, w2, b2 = m._layer_params()
            all_E = propagate_forward(self.bufs, [self.csr] * m.n_layer, m.user_embedding.weight, m.item_embedding.weight,
                                      w1, b1, w2, b2, [0.0] * m.n_layer, [0] * m.n_layer)
            u, p, n = _eng.gather_rows3(all_E, ((i["u_id"], 0, m.n_user), (i["pos_item"], m.n_user, m.n_item),
                                                (i["neg_item"] if self.with_neg else None, m.n_user, m.n_item)), self.status)
            if n is None:
                n = torch.empty(0)
        return all_E, u, p, n

    def _loss(self, out):
        """bprloss.py:15-22 on the gathered rows, with a workspace of this object's own (the graph bakes its address in)."""
        c = self.criterion
        return _eng.bpr_loss(out[1], out[2], out[3], c.weight_decay, c.batch_size, self._loss_ws)

    def __call__(self, u_id, age, sex, month, day, dow, pos_item, neg_item=None, year=None, node_flag=False, check=True):
        """Same arguments as `NGCF.forward` (keyword calls work); `year` is not inspected - the slice was fixed at
        capture - and `node_flag` must be False."""
        if node_flag:
            raise RuntimeError("GraphedForward replays the node_flag=False forward")
        self.load_inputs(u_id, age, sex, month, day, dow, pos_item, neg_item)
        return self.replay(check)

    def load_inputs(self, u_id, age, sex, month, day, dow, pos_item, neg_item=None):
        """The batch into the graph's static index buffers: one multi-tensor copy when every vector is an int64 tensor on the
        graph's device (the usual case), else one copy each."""
        given = dict(u_id=u_id, age=age, sex=sex, month=month, day=day, dow=dow, pos_item=pos_item)
        if self.with_neg:
            given["neg_item"] = neg_item
        for k, v in given.items():
            want = self.Bp if k == "pos_item" else self.Bn if k == "neg_item" else self.B
            if v is None or int(v.numel()) != want:
                raise RuntimeError(f"GraphedForward was captured for {want} elements of {k}: got "
                                   f"{0 if v is None else int(v.numel())}")
        vals = list(given.values())
        if all(v.dtype == torch.int64 and v.device == self.dev and v.dim() == 1 for v in vals):
            torch._foreach_copy_([self.inputs[k] for k in given], vals)
        else:
            for k, v in given.items():
                self.inputs[k].copy_(v.reshape(-1), non_blocking=True)

    def _baked_pointers(self):
        b = self.bufs
        return tuple(t.data_ptr() for t in (b._ws.buf, b._carry[0], b._carry[1], self.scratch,
                                            self.status, self.status_host, self._loss_ws.buf) if t is not None)

    def replay(self, check: bool = True):
        """Replay on whatever `self.inputs[...]` (the graph's static int64 index buffers) hold: callers that write their
        batches straight into those buffers save the eight small copies of `__call__`.  `check=False` skips reading the
        status word (one host sync); it is sticky, so a later `check_status()` still reports an out-of-range id."""
        if self._baked_pointers() != self._baked:
            raise RuntimeError("GraphedForward: a buffer baked into the captured graph was replaced; capture again")
        self.graph.replay()
        all_E, u, p, n = self.out
        m = self.model
        m._e0_cache.invalidate()                 # (the replay injected rows through `.data` behind the retained all_E's back)
        m._all_E = all_E
        m.all_users_emb, m.all_items_emb = all_E[:m.n_user], all_E[m.n_user:]       # NGCF.py:148-149
        if check and m.check_indices:
            self.check_status()
        return u, p, n

    def check_status(self):
        if int(self.status.item()) != 0:                   # (a host sync: every replay so far has written its mirror)
            self.status.zero_()
            self.status_host.zero_()
            raise IndexError("index out of range in NGCF.forward (u_id / feature ids / pos_item / neg_item)")

    def peek_status(self):
        """No host sync: has a replay that the GPU has FINISHED seen an out-of-range id?  (Its last node copied the sticky status
        word into pinned host memory.)  Then raise like `check_status`."""
        if int(self.status_host[0]) != 0:
            self.check_status()



class GraphedTrainStep:
    """One training step of experiment.py:45-58 - `model(node_flag=...)` -> `optimizer.zero_grad()` -> BPR -> `loss.backward()` ->
    `optimizer.step()` - captured into ONE hipGraph and replayed per batch.  On the Seoul-sized graph the step is ~70 launches of a
    few microseconds; replayed, the host issues one graph launch and the step runs at its kernel time.

    Opt-in (the optimizer is the caller's and must be built with `capturable=True`), for launch-bound sizes (all_E up to
    `autograd.DENSE_GRAD_MAX_BYTES`: the backward then has no host sync), dropout in "device" mode (the seeds live in device memory
    and are stepped inside the graph: every replay draws new masks).  Constructing it runs `warmup` REAL training steps on
    `first_batch` (they count as training, like the warm-up of torch's whole-network capture recipe) and then captures.  A replay
    is bit-identical to the eager step it replaces."""

    KEYS = ("year", "u_id", "age", "sex", "month", "day", "dow", "pos_item", "neg_item")

    def __init__(self, model, criterion, optimizer, first_batch: dict, node_flag: bool = True, warmup: int = 3):
        from . import autograd as _ag
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures the training step: call model.train() first")
        if model.node_dropout_mode != "device" or model.mess_dropout_mode != "device":
            raise RuntimeError("GraphedTrainStep needs node_dropout_mode = mess_dropout_mode = 'device' (the reference modes draw their "
                               "masks on the host every step)")
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise RuntimeError("GraphedTrainStep: build the optimizer with capturable=True (torch keeps its step counters on the device then)")
        dev = model._dev()
        N, D = model.n_user + model.n_item, model.emb_size + sum(model.weight_size)
        if N * D * 4 > _ag.DENSE_GRAD_MAX_BYTES:
            raise RuntimeError("GraphedTrainStep is for launch-bound sizes: the backward of a larger graph reads the number of gathered "
                               "rows back to the host (a step of that size is not launch-bound anyway)")
        self.model, self.criterion, self.optimizer, self.node_flag, self.dev = model, criterion, optimizer, bool(node_flag), dev
        self.inputs = {k: first_batch[k].to(device=dev, dtype=torch.int64).clone() for k in self.KEYS}
        self.year_idx = int(self.inputs["year"].min().item() % 18) if self.inputs["year"].numel() else 0
        self.status = model._status_buf(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        model._forced_year_idx = self.year_idx
        try:
            with torch.cuda.stream(side):                    # warm-up: every cached buffer, plan and optimizer state exists before the capture
                for _ in range(max(int(warmup), 1)):
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            status_host = model._status_host_buf()
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
                status_host.copy_(self.status, non_blocking=True)     # last node: the status word where the host can see it
        finally:
            model._forced_year_idx = None
        torch.cuda.synchronize(dev)
        self._baked = self._baked_pointers()
        self.steps_done = max(int(warmup), 1)                # real steps taken on first_batch so far (the capture itself runs nothing)

    def _body(self):
        m = self.model
        u, p, n = m(node_flag=self.node_flag, **self.inputs)
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.criterion(u, p, n)
        loss.backward()
        self.optimizer.step()
        return loss

    def _baked_pointers(self):
        m = self.model
        return tuple(t.data_ptr() for t in ([m._ws.buf, self.criterion._ws.buf, getattr(m, "_seed_state", None), self.status, m._status_host] + list(m.parameters()))
                     if t is not None)

    def __call__(self, **batch):
        """One training step on `batch` (the nine keyword tensors of NGCF.forward, lengths as captured); returns the loss (a 0-dim
        tensor owned by the graph: valid until the next call)."""
        vals = []
        for k in self.KEYS:
            v = batch[k]
            if int(v.numel()) != int(self.inputs[k].numel()):
                raise RuntimeError(f"GraphedTrainStep was captured for {self.inputs[k].numel()} elements of {k}: got {int(v.numel())}")
            vals.append(v)
        if all(v.dtype == torch.int64 and v.device == self.dev and v.dim() == 1 for v in vals):
            torch._foreach_copy_([self.inputs[k] for k in self.KEYS], vals)
        else:
            for k, v in zip(self.KEYS, vals):
                self.inputs[k].copy_(v.reshape(-1), non_blocking=True)
        if self._baked_pointers() != self._baked:
            raise RuntimeError("GraphedTrainStep: a buffer or parameter baked into the captured step was replaced; capture again")
        if self.model.check_indices:
            self.model._peek_status()                        # an earlier step's out-of-range id, as soon as the GPU got there (no sync)
        self.graph.replay()
        self.steps_done += 1
        m = self.model
        m._e0_cache.invalidate()
        if m.check_indices and self.steps_done % max(1, int(m.index_check_every)) == 0 and int(self.status.item()) != 0:
            self.status.zero_()
            if m._status_host is not None:
                m._status_host.zero_()
            raise IndexError("index out of range in a graph-replayed training step (u_id / feature ids / pos_item / neg_item)")
        return self.loss
