"""Multi-GPU propagation: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference is single-device; this is new design.  Rows of `L`, `E` and
`all_E` are independent given the previous layer's `E`, so the graph is row-partitioned and one exchange
step per layer moves embeddings between ranks.  Two exchange schemes share the same partition helpers:

``allgather`` (the north-star scheme, BASELINE config 4)
    Every rank owns a contiguous user range and a contiguous item range, both cut so that stored entries
    are balanced (`ngcf_shard_plan`).  The user range is cut again into `chunks` row chunks (balanced the same
    way).  Nodes are renumbered into a padded chunk-major / rank-major space (``ShardLayout``) so that ONE
    `all_gather_into_tensor` per chunk drops every rank's freshly computed carry rows straight into the replica
    the next layer gathers from - no unpack copies.  The layer is pipelined: chunk j's kernels are followed at once
    by chunk j's asynchronous all-gather, which runs on RCCL's stream while chunk j+1 computes; the layer waits
    only before the next layer's first read.  Bytes received per rank and layer: (W-1)/W * N * d * 4.

``bipartite`` (~10x fewer exchanged bytes; the default of bench.py at N > 1)
    `L = [[0, R], [R^T, 0]]` (matrix.py:49-52): user rows only read item embeddings and vice versa.
    Users are partitioned (by stored entries), the (small) item carry is replicated.  User rows are then fully local; item
    rows are partial sums over the local users, REDUCE-SCATTERED to the rank that owns the item, whose dense half runs
    there only, followed by an ALL-GATHER of the owned items' carry rows into every rank's item replica - the
    neighbour embeddings of the next layer's user rows.  Both exchanges overlap with SpMM kernels.  fp32 summation order
    differs from the single-GPU engine (tolerance, not bit-exact).

A rank never holds more of the graph than its own slabs: they are cut from the interaction triplets (host or
device tensors, `ShardedPropagation.from_interactions`) or from a row-sorted COO (`from_coo`).

Transport (`NGCF_DIST_COLLECTIVES`): "p2p" (default on device tensors) - the CU-free exchange of include/ngcf_hip.h
(`ngcf_p2p_*`: every rank's producers write into its own IPC-exported exchange buffer, the peers pull with device-to-device
copies on copy-engine streams, host threads wait on shared sequence words), which leaves every CU to the L2-swept SpMM;
"torch" - torch.distributed collectives (RCCL under "nccl"); "cabi" - `ngcf_allgather_rows` on the group's communicator.

Compute is always the HIP engine (`engine.py`); nothing here has a CPU path.  The layout/exchange helpers
are backend-agnostic tensor plumbing, which is what the world_size-2 `gloo` tests exercise on CPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib
from . import engine as _eng
from .autograd import E0Cache

SCHEME_NOTES = {
    "allgather": "row partition (users and items, cut by stored entries) + RCCL all-gather of the carry per layer, user slab "
                 "pipelined in row chunks (BASELINE config 4)",
    "bipartite": "row partition of the users (cut by stored entries), item carry replicated: item rows = partial sums over the local "
                 "users, reduce-scattered to the item's owner (dense half there only), then an all-gather of the owned items' carry "
                 "rows - the neighbour embeddings of the next layer - per layer (~10x fewer exchanged bytes than gathering the users)",
}


# ------------------------------------------------------------------------------------------------
# partition + layout helpers (pure index arithmetic, any device)
# ------------------------------------------------------------------------------------------------
def row_counts(rows: torch.Tensor, n_rows: int) -> torch.Tensor:
    return torch.bincount(rows, minlength=n_rows)


def balanced_bounds(counts: torch.Tensor, begin: int, end: int, world: int) -> List[int]:
    """Contiguous cut of rows [begin, end) into `world` ranges with ~equal stored entries (ngcf_shard_plan)."""
    rp = torch.zeros(counts.numel() + 1, dtype=torch.int64)
    rp[1:] = torch.cumsum(counts.to("cpu", torch.int64), 0)
    return _eng.shard_plan(rp, begin, end, world)


def even_bounds(begin: int, end: int, world: int) -> List[int]:
    n = end - begin
    return [begin + (n * w) // world for w in range(world + 1)]


def chunk_bounds(counts: torch.Tensor, rank_bounds: Sequence[int], chunks: int) -> List[int]:
    """Every rank's range cut again into `chunks` entry-balanced row chunks: W*chunks+1 ascending bounds."""
    out = [rank_bounds[0]]
    for r in range(len(rank_bounds) - 1):
        out += balanced_bounds(counts, rank_bounds[r], rank_bounds[r + 1], chunks)[1:]
    return out


class ShardLayout:
    """Padded numbering of the N = U + I nodes for the all-gather scheme.

    Rank r owns users [ub[r], ub[r+1]) and items [ib[r], ib[r+1]) (global node ids; items are U-based); its user
    range is cut into C chunks [cb[r*C+j], cb[r*C+j+1]).  Padded position of the k-th user of chunk j of rank r:
    j*W*mc + r*mc + k (chunk-major, then rank-major: the region of chunk j is what ONE all-gather of that chunk
    fills); of rank r's k-th item: W*C*mc + r*mi + k, with mc / mi the largest user chunk / item range.  A replica of E
    in this numbering has P = W*(C*mc + mi) rows; padding rows are never referenced by any column index.
    With C = 1 this is the plain rank-major layout.
    """

    def __init__(self, n_user: int, n_item: int, user_bounds: Sequence[int], item_bounds: Sequence[int],
                 user_chunk_bounds: Optional[Sequence[int]] = None):
        assert len(user_bounds) == len(item_bounds) and user_bounds[0] == 0 and user_bounds[-1] == n_user
        assert item_bounds[0] == n_user and item_bounds[-1] == n_user + n_item
        self.n_user, self.n_item = n_user, n_item
        self.world = len(user_bounds) - 1
        self.ub, self.ib = list(user_bounds), list(item_bounds)
        self.cb = list(user_chunk_bounds) if user_chunk_bounds is not None else list(user_bounds)
        assert (len(self.cb) - 1) % self.world == 0
        self.chunks = (len(self.cb) - 1) // self.world
        assert [self.cb[r * self.chunks] for r in range(self.world + 1)] == self.ub, "chunk bounds must refine the rank bounds"
        assert all(a <= b for a, b in zip(self.cb, self.cb[1:]))
        self.mc = max(max(b - a for a, b in zip(self.cb, self.cb[1:])), 1)
        self.mu = self.mc                                         # (name kept for the one-chunk layout)
        self.mi = max(max(self.ib[r + 1] - self.ib[r] for r in range(self.world)), 1)
        self.n_user_pos = self.world * self.chunks * self.mc
        self.P = self.n_user_pos + self.world * self.mi

    def n_users_of(self, r, j=None):
        if j is None:
            return self.ub[r + 1] - self.ub[r]
        return self.cb[r * self.chunks + j + 1] - self.cb[r * self.chunks + j]

    def n_items_of(self, r): return self.ib[r + 1] - self.ib[r]
    def user_pos(self, r, j=0): return (j * self.world + r) * self.mc
    def item_pos(self, r): return self.n_user_pos + r * self.mi
    def chunk_region(self, j): return j * self.world * self.mc, (j + 1) * self.world * self.mc
    def chunk_range(self, r, j): return self.cb[r * self.chunks + j], self.cb[r * self.chunks + j + 1]

    def to_padded(self, node: torch.Tensor) -> torch.Tensor:
        """Global node id -> padded position (vectorised)."""
        dev = node.device
        cb = torch.tensor(self.cb, device=dev)
        ib = torch.tensor(self.ib, device=dev)
        is_item = node >= self.n_user
        # empty chunks repeat a bound: right=True picks the last chunk that starts at or before the node, the one holding it
        ci = (torch.searchsorted(cb, node, right=True) - 1).clamp(0, self.world * self.chunks - 1)
        ri = (torch.searchsorted(ib, node, right=True) - 1).clamp(0, self.world - 1)
        r, j = ci // self.chunks, ci % self.chunks
        pu = (j * self.world + r) * self.mc + (node - cb[ci])
        pi = self.n_user_pos + ri * self.mi + (node - ib[ri])
        return torch.where(is_item, pi, pu)

    def owner_of_user(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        ub = torch.tensor(self.ub, device=u.device)
        r = (torch.searchsorted(ub, u, right=True) - 1).clamp(0, self.world - 1)
        return r, u - ub[r]

    def owner_of_item(self, i: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """`i` is an item index in [0, I) (as in pos_item/neg_item)."""
        ib = torch.tensor(self.ib, device=i.device)
        g = i + self.n_user
        r = (torch.searchsorted(ib, g, right=True) - 1).clamp(0, self.world - 1)
        return r, g - ib[r]


def padded_item_pos(item: torch.Tensor, item_bounds: Sequence[int], mi: int) -> torch.Tensor:
    """bipartite scheme: item index -> position in the owner-major padded numbering: item i of owner q (item_bounds[q] <= i <
    item_bounds[q+1]) sits at q*mi + (i - item_bounds[q]), mi = the largest owner range.  One all-gather of [mi, d] per rank then
    IS the item replica of the next layer, and a reduce-scatter over [W*mi, d] hands every owner its rows."""
    ib = torch.tensor(list(item_bounds), device=item.device)
    q = (torch.searchsorted(ib, item, right=True) - 1).clamp(0, len(item_bounds) - 2)
    return q * mi + (item - ib[q])


def slab_coo(rows, cols, vals, lo: int, hi: int):
    """Entries of rows [lo, hi) of a row-sorted COO, row ids made slab-relative."""
    a, b = (int(x) for x in torch.searchsorted(rows, torch.tensor([lo, hi], device=rows.device)))
    return rows[a:b] - lo, cols[a:b], vals[a:b]


def laplacian_values(u: torch.Tensor, i: torch.Tensor, w: torch.Tensor, n_user: int, n_item: int):
    """Values of the symmetric-normalised Laplacian for the interaction triplets, the way `Matrix.create_matrix` forms
    them (matrix.py:55-62: count degrees, d^-1/2 in float32, product in float64, cast to float32); also the degrees."""
    deg_u = torch.bincount(u, minlength=n_user)
    deg_i = torch.bincount(i, minlength=n_item)
    ds_u = deg_u.to(torch.float32).pow(-0.5)
    ds_i = deg_i.to(torch.float32).pow(-0.5)
    ds_u[torch.isinf(ds_u)] = 0
    ds_i[torch.isinf(ds_i)] = 0
    return (ds_u[u].double() * w.double() * ds_i[i].double()).float(), deg_u, deg_i


def cut_slabs(u: torch.Tensor, i: torch.Tensor, v: torch.Tensor, n_user: int, user_lo: int, user_hi: int,
              item_lo: int, item_hi: int):
    """This rank's part of L = [[0, R], [R^T, 0]] from the interaction triplets sorted by (u, i) with Laplacian values v:
    the user rows [user_lo, user_hi) (a contiguous slice of the triplets) and the item rows [item_lo, item_hi) (item
    INDICES, a masked selection re-sorted by (i, u)).  Returns two (rows, cols, vals) triplets with GLOBAL node ids,
    row-sorted, entries inside a row in ascending column order - the same entries and order as the matching rows of
    the full COO of `graphs._normalise`, without ever forming it."""
    a, b = (int(x) for x in torch.searchsorted(u, torch.tensor([user_lo, user_hi], device=u.device)))
    user_rows = (u[a:b], i[a:b] + n_user, v[a:b])
    sel = (i >= item_lo) & (i < item_hi)
    iu, ii, iv = u[sel], i[sel], v[sel]
    order = torch.sort(ii, stable=True).indices           # (u, i) order -> (i, u) order
    item_rows = (ii[order] + n_user, iu[order], iv[order])
    return user_rows, item_rows


def allgather_rows(full_block: torch.Tensor, send: torch.Tensor, group=None, async_op: bool = False):
    """full_block[W*m, d] <- every rank's send[m, d], rank-major (one collective, no unpack)."""
    assert full_block.is_contiguous() and send.is_contiguous()
    assert full_block.shape[0] == send.shape[0] * dist.get_world_size(group)
    return dist.all_gather_into_tensor(full_block, send, group=group, async_op=async_op)


def owner_rows_sum(local_rows: torch.Tensor, owned: torch.Tensor, group=None) -> torch.Tensor:
    """Rows served by their owning rank (others contribute zeros), summed over ranks: every rank ends up with
    all B rows.  x + 0 is exact, so values are the owners' bits."""
    out = torch.where(owned[:, None], local_rows, torch.zeros((), dtype=local_rows.dtype, device=local_rows.device))
    dist.all_reduce(out, group=group)
    return out


class _CabiAllGather:
    """`ngcf_allgather_rows` (include/ngcf_hip.h): ncclAllGather on a communicator handle, issued by the library on a
    stream of its own.  Opt-in (NGCF_DIST_COLLECTIVES=cabi, backend "nccl" only): the handle is the process group's own
    RCCL communicator.  The default path is `torch.distributed`'s all_gather_into_tensor, which does the same thing."""

    def __init__(self, group, device):
        pg = group if group is not None else dist.group.WORLD
        backend = pg._get_backend(torch.device(device))
        dist.barrier(group=group)                                  # the communicator exists after the first collective
        self.comm = C.c_void_p(int(backend._comm_ptr()))
        self.stream = torch.cuda.Stream(device=device)

    def __call__(self, full_block: torch.Tensor, send: torch.Tensor):
        cur = torch.cuda.current_stream(send.device)
        self.stream.wait_stream(cur)                               # the rows were produced on the compute stream
        _lib.check(_lib.load().ngcf_allgather_rows(self.comm, C.c_void_p(send.data_ptr()), C.c_void_p(full_block.data_ptr()),
                                                   send.shape[0], send.shape[1], C.c_void_p(self.stream.cuda_stream)))
        send.record_stream(self.stream)
        full_block.record_stream(self.stream)
        return self

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


# ------------------------------------------------------------------------------------------------
# differentiable exchange steps and layer pieces: the TRAINING path of the sharded engine (experiment.py:57 across ranks)
# ------------------------------------------------------------------------------------------------
def _reduce_scatter_rows(full: torch.Tensor, group) -> torch.Tensor:
    """own[m, d] = sum over ranks of full[rank*m:(rank+1)*m] (gloo has no reduce-scatter: all-reduce + own rows)."""
    if isinstance(group, P2PCollectives):
        return group.reduce_scatter(full)
    W, r = dist.get_world_size(group), dist.get_rank(group)
    m = full.shape[0] // W
    if dist.get_backend(group) == "nccl":
        out = torch.empty((m, full.shape[1]), dtype=full.dtype, device=full.device)
        dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
        return out
    tmp = full.contiguous().clone()
    dist.all_reduce(tmp, group=group)
    return tmp[r * m:(r + 1) * m].clone()


def _all_gather_rows(send: torch.Tensor, comm) -> torch.Tensor:
    """`comm`: a process group (or None = the default group) for torch.distributed, or a `P2PCollectives` for the CU-free exchange."""
    if isinstance(comm, P2PCollectives):
        return comm.all_gather(send)
    full = torch.empty((send.shape[0] * dist.get_world_size(comm), send.shape[1]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(full, send.contiguous(), group=comm)
    return full


class AllGatherRows(torch.autograd.Function):
    """full[W*m, d] = every rank's send[m, d], rank-major.  Adjoint: reduce-scatter of the gradient (every rank used every row)."""

    @staticmethod
    def forward(ctx, send, comm):
        ctx.comm = comm
        return _all_gather_rows(send, comm)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_rows(g, ctx.comm), None


class ReduceScatterRows(torch.autograd.Function):
    """own[m, d] = sum over ranks of their full[rank*m:(rank+1)*m].  Adjoint: all-gather of the gradient."""

    @staticmethod
    def forward(ctx, full, comm):
        ctx.comm = comm
        return _reduce_scatter_rows(full, comm)

    @staticmethod
    def backward(ctx, g):
        return _all_gather_rows(g, ctx.comm), None


class OwnerRowsSum(torch.autograd.Function):
    """`owner_rows_sum` with its adjoint.  Every rank computes the SAME loss from the same gathered rows, so the gradient of a
    served row is that loss's gradient on the rank that served it and nothing elsewhere - no exchange in the backward."""

    @staticmethod
    def forward(ctx, local_rows, owned, group):
        ctx.save_for_backward(owned)
        return owner_rows_sum(local_rows, owned, group)

    @staticmethod
    def backward(ctx, g):
        (owned,) = ctx.saved_tensors
        return torch.where(owned[:, None], g, torch.zeros((), dtype=g.dtype, device=g.device)), None, None


class SpmmFn(torch.autograd.Function):
    """LE = A . E on the HIP SpMM with A^T given as a second CSR: the backward is the product with that one."""

    @staticmethod
    def forward(ctx, E, csr, csr_t, ws):
        ctx.csr_t, ctx.ws = csr_t, ws
        return _eng.spmm(csr, E.detach(), ws=ws)

    @staticmethod
    def backward(ctx, g):
        d = g.shape[1]
        gp = torch.empty((g.shape[0], (d + 31) // 32 * 32), dtype=torch.float32, device=g.device)[:, :d]
        gp.copy_(g)                                                # 128-byte aligned rows for the gather
        return _eng.spmm(ctx.csr_t, gp, ws=ctx.ws), None, None, None


class DenseLayerFn(torch.autograd.Function):
    """(carry, norm) of one layer's dense half (NGCF.py:131-146, eval-mode: no message dropout) for a slab of rows, with the
    hand-written backward kernels of autograd.py (normalise / LeakyReLU backward, MFMA weight and input gradients)."""

    @staticmethod
    def forward(ctx, LE, E, W1, b1, W2, b2, ws):
        n, d_out = LE.shape[0], W1.shape[0]
        ld = (d_out + 31) // 32 * 32
        carry = torch.empty((n, ld), dtype=torch.float32, device=LE.device)[:, :d_out]
        norm = torch.empty((n, d_out), dtype=torch.float32, device=LE.device)
        if n:
            _eng.layer_dense(LE.detach(), E.detach(), W1.detach(), b1.detach(), W2.detach(), b2.detach(), carry, norm, ws)
        ctx.ws = ws
        ctx.save_for_backward(LE.detach(), E.detach(), carry, W1.detach(), W2.detach())
        return carry, norm

    @staticmethod
    def backward(ctx, dC, dN):
        from . import autograd as ag
        LE, E, carry, W1, W2 = ctx.saved_tensors
        n, d_in = LE.shape
        if n == 0:
            z = torch.zeros_like
            return z(LE), z(E), z(W1), W1.new_zeros(W1.shape[0]), z(W2), W2.new_zeros(W2.shape[0]), None
        dM = ag._bwd_pre(dN.contiguous() if dN is not None else None, dC.contiguous() if dC is not None else None, carry,
                         _eng.LEAKY_SLOPE, 0.0, 0)
        gW1, gb1, gW2, gb2 = ag._bwd_weight(dM, LE, E, ctx.ws)
        dLE, dE = ag._bwd_input(dM, W1, W2, LE, E, ctx.ws)
        return dLE, dE, gW1, gb1, gW2, gb2, None


class _SumGrads(torch.autograd.Function):
    """Identity on the (replicated) parameters; its backward makes every parameter's gradient complete on every rank - once per
    parameter and step, as in data-parallel training.

    `slabs` = {position in `params`: row bounds per rank}: that parameter (the user table) is read by rank r through rows
    [bounds[r], bounds[r+1]) only, so its gradient is non-zero in that slab only and the slabs are disjoint: an ALL-GATHER of the
    slabs (padded to the largest) gives every rank the full gradient - (W-1)/W x |table| received per rank instead of the
    2 (W-1)/W x |table| of a ring all-reduce of W dense [U, d0] matrices that are zero outside one slab each (r04; at C3 / W = 8:
    448 MB instead of 896 MB per rank and step for the 512 MB user table).  Everything else (the item table - every rank's user rows
    gather from all items -, the weights) is summed by an all-reduce."""

    @staticmethod
    def forward(ctx, group, slabs, *params):
        ctx.group, ctx.slabs = group, dict(slabs or {})
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        out = []
        W = dist.get_world_size(ctx.group)
        r = dist.get_rank(ctx.group)
        for i, gr in enumerate(grads):
            if gr is None:
                out.append(None)
                continue
            gr = gr.contiguous()
            bounds = ctx.slabs.get(i)
            if bounds is None or W == 1:
                dist.all_reduce(gr, group=ctx.group)
                out.append(gr)
                continue
            mx = max(max(bounds[q + 1] - bounds[q] for q in range(W)), 1)
            send = gr.new_zeros((mx,) + tuple(gr.shape[1:]))
            send[:bounds[r + 1] - bounds[r]] = gr[bounds[r]:bounds[r + 1]]
            full = gr.new_empty((W * mx,) + tuple(gr.shape[1:]))
            dist.all_gather_into_tensor(full, send, group=ctx.group)
            res = torch.zeros_like(gr)                             # (rows no rank reads keep a zero gradient)
            for q in range(W):
                n_q = bounds[q + 1] - bounds[q]
                if n_q:
                    res[bounds[q]:bounds[q + 1]] = full[q * mx:q * mx + n_q]
            out.append(res)
        return (None, None, *out)


class _DevMem:
    """`__cuda_array_interface__` carrier: a torch tensor over device memory the library allocated (the exchange buffer)."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n_floats),), "typestr": "<f4", "version": 2}


class P2PExchange:
    """The CU-free exchange of include/ngcf_hip.h (`ngcf_p2p_*`): one exchange buffer per rank, peers pull from it with
    device-to-device copies on copy-engine streams, host threads do the waiting.  All ranks of `group` must live on one node
    (IPC handles + POSIX shared memory).  `floats(offset, n)` gives a tensor over this rank's buffer: producers write there."""

    TIMEOUT_MS = float(os.environ.get("NGCF_P2P_TIMEOUT_MS", "60000"))

    def __init__(self, group, device, n_floats: int):
        lib = _lib.load()
        self.group, self.dev = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_floats = int(n_floats)
        tok = [os.urandom(8).hex() if self.rank == 0 else None]
        dist.broadcast_object_list(tok, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.name = f"ngcf_p2p_{tok[0]}"
        # Every rank takes part in every collective of this constructor whatever happened to it locally (a rank that raised
        # between two of them would leave the others waiting in the next one): failures travel with the gathered objects and
        # all ranks raise together.
        h, err = C.c_void_p(), None
        mine = (C.c_char * 64)()
        try:
            with _eng._on(self.dev):
                _lib.check(lib.ngcf_p2p_create(self.rank, self.world, max(self.n_floats, 64) * 4, self.name.encode(), C.byref(h)))
            _lib.check(lib.ngcf_p2p_handle(h, mine))
        except Exception as exc:  # noqa: BLE001
            err = f"rank {self.rank}: {exc!r}"[:300]
        self._h = h
        got = [None] * self.world
        dist.all_gather_object(got, (err, bytes(mine.raw)), group=group)
        errs = [e for e, _ in got if e]
        if not errs:
            try:
                blob = (C.c_char * (64 * self.world)).from_buffer_copy(b"".join(hb for _, hb in got))
                with _eng._on(self.dev):
                    _lib.check(lib.ngcf_p2p_connect(self._h, blob))
            except Exception as exc:  # noqa: BLE001
                err = f"rank {self.rank}: {exc!r}"[:300]
            got2 = [None] * self.world
            dist.all_gather_object(got2, err, group=group)      # (doubles as the barrier: everyone is connected before the first pull)
            errs = [e for e in got2 if e]
        if errs:
            self.close_quietly()
            raise RuntimeError("p2p exchange could not be set up: " + "; ".join(errs))
        self.base = int(lib.ngcf_p2p_local(self._h))
        self._mem = torch.as_tensor(_DevMem(self.base, max(self.n_floats, 64)), device=self.dev)
        assert self._mem.data_ptr() == self.base and self._mem.dtype == torch.float32
        self.seq = {}                      # slot -> last published / expected step

    def floats(self, offset: int, n: int) -> torch.Tensor:
        assert 0 <= offset and offset + n <= self.n_floats
        return self._mem[offset:offset + n]

    def publish(self, slot: int, seq: int):
        _lib.check(_lib.load().ngcf_p2p_publish(self._h, slot, seq, _eng._stream()))

    def pull(self, peer: int, slot: int, seq: int, src_off_floats: int, dst: torch.Tensor):
        """dst (contiguous, this device) <- rank `peer`'s exchange buffer [src_off, src_off + dst.numel()), once it published `seq`."""
        assert dst.is_contiguous() and dst.dtype == torch.float32
        _lib.check(_lib.load().ngcf_p2p_pull(self._h, peer, slot, seq, src_off_floats * 4, C.c_void_p(dst.data_ptr()),
                                             dst.numel() * 4, self.TIMEOUT_MS))

    def ack(self, peer: int, slot: int, seq: int):
        _lib.check(_lib.load().ngcf_p2p_ack(self._h, peer, slot, seq))

    def wait_acks(self, slot: int, seq: int):
        _lib.check(_lib.load().ngcf_p2p_wait_acks(self._h, slot, seq, self.TIMEOUT_MS))

    def fence(self):
        _lib.check(_lib.load().ngcf_p2p_fence(self._h, _eng._stream()))

    def join(self):
        _lib.check(_lib.load().ngcf_p2p_join(self._h, _eng._stream()))

    def stats(self, reset: bool = False):
        """(host milliseconds spent blocked in waits, waits, waits that found their word missing) since creation / the last reset."""
        ms, n, nb = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(_lib.load().ngcf_p2p_stats(self._h, C.byref(ms), C.byref(n), C.byref(nb), 1 if reset else 0))
        return float(ms.value), int(n.value), int(nb.value)

    def peers_from(self, start: int):
        """All ranks, beginning after `start` (every rank starts its pulls at another peer: the links are used evenly)."""
        return [(start + 1 + q) % self.world for q in range(self.world)]

    def selftest(self) -> bool:
        """Every rank writes a pattern, publishes it, pulls every peer's and compares; True on every rank iff it worked everywhere."""
        ok = 1
        try:
            n = 1024
            probe = self.floats(0, n)
            probe.copy_(torch.arange(n, dtype=torch.float32, device=self.dev) + 1000.0 * self.rank)
            self.publish(63, 1)
            got = torch.empty((self.world, n), dtype=torch.float32, device=self.dev)
            # `got` may reuse the block of a temporary whose kernels are still queued on this stream (the allocator hands memory
            # out by stream order): the copies must not land before them.  (Without this fence the self-test failed one run in
            # three with five ranks on one GPU: a pending kernel of the old owner wrote over the pulled rows.)
            self.fence()
            for q in self.peers_from(self.rank):
                self.pull(q, 63, 1, 0, got[q])
                self.ack(q, 63, 1)
            self.join()
            torch.cuda.synchronize(self.dev)
            want = torch.arange(n, dtype=torch.float32, device=self.dev)[None] + 1000.0 * torch.arange(self.world, device=self.dev)[:, None]
            ok = int(torch.equal(got, want))
            self.wait_acks(63, 1)
        except Exception as exc:  # noqa: BLE001
            ok = 0
            self.last_error = f"rank {self.rank}: {exc!r}"[:300]
        if ok == 0 and not getattr(self, "last_error", None):
            self.last_error = f"rank {self.rank}: pulled rows differ from what the peers wrote"
        errs = [None] * self.world                       # every rank learns what failed where (the bench line quotes it)
        dist.all_gather_object(errs, getattr(self, "last_error", None) if ok == 0 else None, group=self.group)
        errs = [e for e in errs if e]
        if errs:
            self.last_error = "; ".join(errs)[:600]
        return not errs

    def close_quietly(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.load().ngcf_p2p_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:  # noqa: BLE001
            pass

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.dev)
            self._mem = None
            _lib.load().ngcf_p2p_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class P2PCollectives:
    """All-gather and reduce-scatter of row blocks over a `P2PExchange` of its own: the exchange steps of the TRAINING path (forward
    and backward) without a collective kernel on the CUs.  Every call uses the next of four regions of the exchange buffer and its
    sequence slot in turn (all ranks issue the same sequence of calls), so a producer overwrites a region only after every peer has
    read what it held four calls earlier.  The host blocks in each call until the peers have published (no pipelining here)."""

    REGIONS = 4

    def __init__(self, group, device, region_floats: int):
        self.region = int(region_floats)
        self.ex = P2PExchange(group, device, self.REGIONS * self.region)
        self.rank, self.world, self.dev = self.ex.rank, self.ex.world, self.ex.dev
        self._op, self._last = 0, {}

    def _stage(self, t: torch.Tensor):
        """The rows of `t` [n, d] into the next region of this rank's exchange buffer (row stride d4 = d rounded up to 4 floats:
        every rank's block is then a whole number of 16-byte pieces), published; returns (slot, seq, region offset, d4)."""
        n, d = t.shape
        d4 = (d + 3) // 4 * 4
        assert t.dtype == torch.float32 and n * d4 <= self.region, "P2PCollectives: block larger than a region"
        reg = self._op % self.REGIONS
        self._op += 1
        slot, seq, off = 32 + reg, self._op, reg * self.region
        self.ex.wait_acks(slot, self._last.get(slot, 0))
        self.ex.floats(off, n * d4).view(n, d4)[:, :d].copy_(t)
        self.ex.publish(slot, seq)
        self._last[slot] = seq
        self.ex.fence()                          # the copies below land in fresh tensors: after whatever used that memory before
        return slot, seq, off, d4

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        m, d = send.shape
        slot, seq, off, d4 = self._stage(send)
        full = torch.empty((self.world * m, d4), dtype=torch.float32, device=self.dev)
        for q in self.ex.peers_from(self.rank):
            self.ex.pull(q, slot, seq, off, full[q * m:(q + 1) * m])
            self.ex.ack(q, slot, seq)
        self.ex.join()
        return full[:, :d]

    def reduce_scatter(self, full: torch.Tensor) -> torch.Tensor:
        m, d = full.shape[0] // self.world, full.shape[1]
        slot, seq, off, d4 = self._stage(full)
        slots = torch.empty((self.world, m, d4), dtype=torch.float32, device=self.dev)
        for q in self.ex.peers_from(self.rank):
            self.ex.pull(q, slot, seq, off + self.rank * m * d4, slots[q])
            self.ex.ack(q, slot, seq)
        self.ex.join()
        own = torch.empty((m, d4), dtype=torch.float32, device=self.dev)
        with _eng._on(self.dev):
            _lib.check(_lib.load().ngcf_sum_slots_f32(_eng._ptr(slots), m * d4, self.world, m * d4, _eng._ptr(own), _eng._stream()))
        return own[:, :d]                        # (the padding columns carry whatever the regions held: never read)


# ------------------------------------------------------------------------------------------------
# sharded propagation on the HIP engine
# ------------------------------------------------------------------------------------------------
class ShardedPropagation:
    """NGCF.py:120-156 over `world` GPUs for one Laplacian slice.

    Build with `from_interactions` (this rank keeps only its slabs; the triplets may live on the host) or `from_coo`
    (a row-sorted COO of the [N, N] Laplacian: tests and small graphs).  Parameters are replicated: pass the same `NGCF`
    module (same seed / same state_dict) on every rank.
    """

    def __init__(self, model, rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor,
                 mode: str = "allgather", group=None, chunks: Optional[int] = None):
        """From the FULL row-sorted COO (every rank builds or loads the same one and keeps only its part)."""
        U, I = model.n_user, model.n_item
        n_ue = int(torch.searchsorted(rows, torch.tensor([U], device=rows.device)))
        if n_ue and (int(cols[:n_ue].min()) < U or (n_ue < rows.numel() and int(cols[n_ue:].max()) >= U)):
            raise RuntimeError("the sharded engine needs L = [[0, R], [R^T, 0]] (matrix.py:49-52)")
        if n_ue * 2 != rows.numel():
            raise RuntimeError("the sharded engine needs both triangles of L stored (matrix.py:51-52)")
        if mode == "bipartite":
            # the item rows of this scheme are formed as the transpose of the user rows: the lower triangle must BE that transpose
            # (the reference's Laplacian is symmetric, matrix.py:51-62; a re-weighted or thinned lower triangle is not supported)
            order = torch.sort(cols[:n_ue], stable=True).indices
            if not (torch.equal(cols[:n_ue][order], rows[n_ue:]) and torch.equal(rows[:n_ue][order], cols[n_ue:])
                    and torch.equal(vals[:n_ue][order], vals[n_ue:])):
                raise RuntimeError("the bipartite exchange scheme needs a symmetric L (R^T stored as the exact transpose of R, "
                                   "matrix.py:51-62); use mode='allgather' for anything else")
        cnt = row_counts(rows, U + I)

        def cut(user_lo, user_hi, item_lo, item_hi):
            a = slab_coo(rows, cols, vals, user_lo, user_hi)
            b = slab_coo(rows, cols, vals, U + item_lo, U + item_hi)
            return (a[0] + user_lo, a[1], a[2]), (b[0] + U + item_lo, b[1], b[2])
        self._setup(model, cnt[:U], cnt[U:], cut, mode, group, chunks, rows.device)

    @classmethod
    def from_coo(cls, model, rows, cols, vals, mode: str = "allgather", group=None, chunks: Optional[int] = None):
        return cls(model, rows, cols, vals, mode, group, chunks)

    @classmethod
    def from_interactions(cls, model, u: torch.Tensor, i: torch.Tensor, w: torch.Tensor, mode: str = "allgather",
                          group=None, chunks: Optional[int] = None, device=None):
        """From the unique interaction triplets (u, i, w) sorted by (u, i), on any device (host tensors included): the
        degrees and the Laplacian values are formed where the triplets live, and only this rank's slabs - about 2/W of
        the stored entries - reach the compute device.  The doubled [N, N] COO is never materialised."""
        self = cls.__new__(cls)
        v, deg_u, deg_i = laplacian_values(u, i, w, model.n_user, model.n_item)
        self._setup(model, deg_u, deg_i, lambda ul, uh, il, ih: cut_slabs(u, i, v, model.n_user, ul, uh, il, ih),
                    mode, group, chunks, device if device is not None else model._dev())
        return self

    def _setup(self, model, deg_u, deg_i, cut, mode, group, chunks, dev):
        assert mode in ("bipartite", "allgather")
        self.model, self.mode, self.group = model, mode, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.U, self.I = model.n_user, model.n_item
        self.N = self.U + self.I
        self.dev = dev = torch.device(dev)
        self.ws = _eng.Workspace()
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        U, I, W, r = self.U, self.I, self.world, self.rank
        swept_mode = int(os.environ.get("NGCF_SPMM_MODE", "3"))
        to_dev = lambda t: t.to(dev, non_blocking=True)            # noqa: E731
        self.backend = self._pick_backend(group, dev)
        if mode == "bipartite":
            cnt = torch.cat([deg_u.to("cpu", torch.int64), deg_i.to("cpu", torch.int64)])
            self.ub = balanced_bounds(cnt, 0, U, W)                # user ranges with ~equal stored entries
            self.ib = even_bounds(0, I, W)                         # item OWNER ranges (item indices): reduce + dense + carry of those rows
            self.mi = mi = max(max(self.ib[q + 1] - self.ib[q] for q in range(W)), 1)
            self.PI = W * mi                                       # padded item numbering: item i of owner q sits at q*mi + (i - ib[q])
            lo, hi = self.ub[r], self.ub[r + 1]
            self.nu, self.ni = hi - lo, self.ib[r + 1] - self.ib[r]
            (ur, uc, uv), _ = cut(lo, hi, 0, 0)
            ur, uc, uv = to_dev(ur), to_dev(uc), to_dev(uv)
            pos = self.item_pos(uc - U)
            # user rows: local users x all items (replicated item carry, padded numbering)
            self.csr_u = _eng.LaplacianCSR.from_coo(ur - lo, pos, uv, self.nu, self.PI)
            # item rows restricted to local user columns -> partial sums: the transpose of the same slice
            order = torch.sort(pos, stable=True).indices
            self.csr_it = _eng.LaplacianCSR.from_coo(pos[order], ur[order] - lo, uv[order], self.PI, max(self.nu, 1))
            # The item partial sums run alone on the GPU: L2-swept kernel where it pays.  The user rows overlap with the exchange:
            # over copy engines (p2p) nothing else occupies a CU and they are swept too; beside an RCCL kernel the persistent
            # one-workgroup-per-CU sweep loses 3.4-4x (profiles/r02_c4_rank_lab.txt), so there they stay on the row-wise kernels.
            self.csr_it.set_mode(swept_mode)
            if self.backend == "p2p" or W == 1:
                self.csr_u.set_mode(swept_mode)
            self.local_nnz = self.csr_u.nnz + self.csr_it.nnz
            widths = [model.emb_size] + list(model.weight_size)
            ld = lambda d: (d + 31) // 32 * 32                     # noqa: E731
            self._ld_in, self._ld_out = max(ld(d) for d in widths[:-1]), max(ld(d) for d in widths[1:])
            if self.backend == "p2p":
                # exchange buffer: per LAYER a region for the partial sums of all items and one for the carry rows of the owned
                # items.  Nothing is overwritten inside a pass, so the peers acknowledge once per pass (not per pull round): a
                # pass starts by waiting for the previous pass's acknowledgements.
                nl = len(widths) - 1
                self._off_part = [k * self.PI * self._ld_in for k in range(nl)]
                self._off_carry = [nl * self.PI * self._ld_in + k * mi * self._ld_out for k in range(nl)]
                self.p2p = self._open_p2p(group, dev, nl * (self.PI * self._ld_in + mi * self._ld_out))
        else:
            cnt = torch.cat([deg_u.to("cpu", torch.int64), deg_i.to("cpu", torch.int64)])
            ub = balanced_bounds(cnt, 0, U, W)
            ib = balanced_bounds(cnt, U, self.N, W)
            if chunks is None:      # pipeline depth of the user slab; one chunk when there is nobody to exchange with
                chunks = int(os.environ.get("NGCF_DIST_CHUNKS", "4")) if W > 1 else 1
            self.chunks = chunks = max(1, int(chunks))
            self.layout = lay = ShardLayout(U, I, ub, ib, chunk_bounds(cnt, ub, chunks))
            (ur, uc, uv), (ir, ic, iv) = cut(ub[r], ub[r + 1], ib[r] - U, ib[r + 1] - U)
            ur, uc, uv, ir, ic, iv = (to_dev(t) for t in (ur, uc, uv, ir, ic, iv))
            self.nu, self.ni = lay.n_users_of(r), lay.n_items_of(r)
            self.csr_u = []                                        # one CSR per row chunk of the user slab
            for j in range(chunks):
                lo, hi = lay.chunk_range(r, j)
                cr, cc, cv = slab_coo(ur, uc, uv, lo, hi)
                self.csr_u.append(_eng.LaplacianCSR.from_coo(cr, lay.to_padded(cc), cv, hi - lo, lay.P))
            self.csr_i = _eng.LaplacianCSR.from_coo(ir - ib[r], lay.to_padded(ic), iv, self.ni, lay.P)
            # the item slab runs with no collective in flight (the layer's gathers have all been waited for): swept kernel
            # where it pays.  With one chunk and one rank nothing overlaps the user slab either.
            self.csr_i.set_mode(swept_mode)
            if W == 1 or self.backend == "p2p":
                for c in self.csr_u:
                    c.set_mode(swept_mode)
            self.local_nnz = sum(c.nnz for c in self.csr_u) + self.csr_i.nnz
            if self.backend == "p2p":
                # exchange buffer: the carry rows this rank produces (item slab + every user chunk), two layers in turn
                widths = [model.emb_size] + list(model.weight_size)
                self._ld_out = max((d + 31) // 32 * 32 for d in widths[1:])
                per_layer = (lay.mi + chunks * lay.mc) * self._ld_out
                self._off_send = [0, per_layer]
                self.p2p = self._open_p2p(group, dev, 2 * per_layer)
        self._bufs = {}
        self._cabi = None
        self._last_pub = {}
        self._calls = 0
        if self.backend == "cabi":
            self._cabi = _CabiAllGather(group, dev)

    # this rank's rows of all_E (NGCF.py:147-149): stored under the attribute names autograd.E0Cache looks at, so that the same
    # hold detection decides whether the previous pass's buffer may be written again
    @property
    def allE_u(self):
        return self.__dict__.get("all_users_emb")

    @allE_u.setter
    def allE_u(self, v):
        self.all_users_emb = v

    @property
    def allE_i(self):
        return self.__dict__.get("all_items_emb")

    @allE_i.setter
    def allE_i(self, v):
        self.all_items_emb = v

    def item_pos(self, item: torch.Tensor) -> torch.Tensor:
        """bipartite scheme: item index -> position in the padded item numbering (owner-major)."""
        return padded_item_pos(item, self.ib, self.mi)

    def _pick_backend(self, group, dev) -> str:
        """How the rows travel: "p2p" = copy engines + host-side waiting (ngcf_p2p_*, one node), "torch" = torch.distributed
        collectives (RCCL under the nccl backend), "cabi" = ngcf_allgather_rows on the process group's communicator.
        NGCF_DIST_COLLECTIVES picks one; the default is p2p on device tensors with more than one rank (it falls back to "torch"
        on every rank together if the exchange cannot be set up or its self-test fails)."""
        want = os.environ.get("NGCF_DIST_COLLECTIVES", "p2p" if self.world > 1 else "torch")
        if want == "cabi" and dist.get_backend(group) != "nccl":
            want = "torch"
        if want not in ("p2p", "torch", "cabi"):
            raise ValueError("NGCF_DIST_COLLECTIVES must be p2p, torch or cabi")
        if want == "p2p" and (self.world == 1 or torch.device(dev).type != "cuda"):
            want = "torch"
        return want

    def _open_p2p(self, group, dev, n_floats):
        ex, ok = None, 1
        try:
            ex = P2PExchange(group, dev, n_floats)
        except Exception as exc:  # noqa: BLE001
            ok = 0
            self.p2p_error = repr(exc)[:300]
        t = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        if int(t.item()) == 1 and not ex.selftest():
            t.zero_()
            self.p2p_error = f"self-test failed ({getattr(ex, 'last_error', None)})"
        if int(t.item()) != 1:                                    # every rank takes the same decision
            if ex is not None:
                ex.close()
            self.backend = "torch"
            for c in ([self.csr_u] if self.mode == "bipartite" else self.csr_u):     # beside RCCL kernels: row-wise again
                c.set_mode(0)
            return None
        return ex

    def _buf(self, name, shape):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape):
            b = torch.empty(shape, dtype=torch.float32, device=self.dev)
            self._bufs[name] = b
        return b

    def _params(self):
        m = self.model
        return ([l.weight.detach() for l in m.w1_list], [l.bias.detach() for l in m.w1_list],
                [l.weight.detach() for l in m.w2_list], [l.bias.detach() for l in m.w2_list])

    def spmm_shapes(self):
        """(nnz, rows, cols) of this rank's SpMM launches of one layer (for the roofline's algorithmic bytes)."""
        if self.mode == "bipartite":
            cs = [self.csr_it, self.csr_u]
        else:
            cs = [self.csr_i] + list(self.csr_u)
        return [(c.nnz, c.n_rows, c.n_cols) for c in cs]

    def swept_rows(self):
        cs = [self.csr_it, self.csr_u] if self.mode == "bipartite" else [self.csr_i] + list(self.csr_u)
        return [c.swept_rows for c in cs]

    # -- propagation ----------------------------------------------------------------------------
    def propagate(self):
        """(all_E rows of this rank's users, all_E rows of its items).  When autograd is recording and a parameter of the model
        requires a gradient and the scheme is `bipartite`, the differentiable path runs: `loss.backward()` then leaves in every
        parameter's `.grad` the gradient summed over all ranks (experiment.py:57 across GPUs)."""
        m = self.model
        if torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters()) and self.mode == "bipartite":
            return self._propagate_bipartite_train()
        # (the all-gather scheme has no differentiable path: its results carry no gradient, like any call under torch.no_grad())
        with torch.no_grad():
            return self._propagate_bipartite() if self.mode == "bipartite" else self._propagate_allgather()

    def _train_comm(self):
        """What the differentiable exchange steps run on: the process group (torch.distributed collectives), or - when this
        object's transport is p2p - a `P2PCollectives` over an exchange buffer of its own, created on first use."""
        if self.backend != "p2p":
            return self.group
        if getattr(self, "_p2p_train", None) is None:
            widths = [self.model.emb_size] + list(self.model.weight_size)
            ldmax = max((d + 31) // 32 * 32 for d in widths)
            self._p2p_train = P2PCollectives(self.group, self.dev, self.PI * ldmax)
        return self._p2p_train

    def _propagate_bipartite_train(self):
        """The bipartite scheme as a composition of differentiable pieces: `SpmmFn` (the two CSRs of a rank are each other's
        transposes), `DenseLayerFn` (hand-written backward kernels), `ReduceScatterRows` / `AllGatherRows` (each the other's
        adjoint).  Eval-mode semantics (no dropout).  The exchange steps run over this object's transport in both directions
        (`_train_comm`: the CU-free p2p collectives, or torch.distributed).
        Gradients: every rank's backward yields the contribution of the rows it owns; `_SumGrads` adds them over the ranks, so
        after `loss.backward()` all ranks hold the same, complete `.grad` for every parameter - as after a single-GPU step."""
        m, g = self.model, self.group
        comm = self._train_comm()
        r, W, mi, PI = self.rank, self.world, self.mi, self.PI
        lo, nu, ni = self.ub[r], self.nu, self.ni
        n_layer = m.n_layer
        # parameters enter through one Function whose backward all-reduces their gradients (once per step)
        params = [m.user_embedding.weight, m.item_embedding.weight] + [l.weight for l in m.w1_list] + [l.bias for l in m.w1_list] + \
                 [l.weight for l in m.w2_list] + [l.bias for l in m.w2_list]
        outs = _SumGrads.apply(g, {0: list(self.ub)}, *params)     # the user table is read through the rank's own slab only
        uw, iw = outs[0], outs[1]
        w1, b1 = outs[2:2 + n_layer], outs[2 + n_layer:2 + 2 * n_layer]
        w2, b2 = outs[2 + 2 * n_layer:2 + 3 * n_layer], outs[2 + 3 * n_layer:]
        eu = uw[lo:lo + nu]
        # item replica in the padded numbering (a gather of the replicated table; padding rows read item 0 and are never used)
        src = torch.zeros(PI, dtype=torch.int64, device=self.dev)
        src[self.item_pos(torch.arange(self.I, device=self.dev))] = torch.arange(self.I, device=self.dev)
        ei = iw[src]
        own = slice(r * mi, r * mi + ni)
        blocks_u, blocks_i = [eu], [ei[own]]
        for k in range(n_layer):
            part = SpmmFn.apply(eu, self.csr_it, self.csr_u, self.ws)               # item partial sums over the local users
            le_own = ReduceScatterRows.apply(part, comm)                             # [mi, d]: the owned items, summed over ranks
            le_u = SpmmFn.apply(ei, self.csr_u, self.csr_it, self.ws)               # user rows, fully local
            cu, nrm_u = DenseLayerFn.apply(le_u, eu, w1[k], b1[k], w2[k], b2[k], self.ws)
            ci, nrm_i = DenseLayerFn.apply(le_own[:ni], ei[own], w1[k], b1[k], w2[k], b2[k], self.ws)
            blocks_u.append(nrm_u)
            blocks_i.append(nrm_i)
            if k + 1 < n_layer:
                send = torch.zeros((mi, ci.shape[1]), dtype=torch.float32, device=self.dev)
                send = torch.cat([ci, send[ni:]], 0) if ni < mi else ci
                ei = AllGatherRows.apply(send, comm)
            eu = cu
        self.allE_u, self.allE_i = torch.cat(blocks_u, 1), torch.cat(blocks_i, 1)
        return self.allE_u, self.allE_i

    def _propagate_bipartite(self):
        """Users partitioned, item block replicated.  Per layer:  A  item partial sums over the local users (swept SpMM on the
        transposed user slab)  ->  reduce-scatter: every rank receives the partial sums of the items it OWNS and adds them in
        rank order  ->  B  user rows (fully local, overlaps with that exchange)  ->  C  dense half for the owned items only  ->
        all-gather of their carry rows into every rank's item replica (overlaps with the next layer's A).  The exchanged bytes
        per rank and layer are 2 x (W-1)/W x I x d x 4 (C3, W = 8: 2 x 45 MB) against (W-1)/W x N x d x 4 (493 MB) for the all-gather
        of users; nothing is computed twice (r02 ran the item dense replicated after an all-reduce)."""
        m = self.model
        w1, b1, w2, b2 = self._params()
        widths = [m.emb_size] + [int(w.shape[0]) for w in w1]
        D, n_layer = sum(widths), len(w1)
        r, W, mi, PI = self.rank, self.world, self.mi, self.PI
        lo, nu, ni = self.ub[r], self.nu, self.ni
        ld = lambda d: (d + 31) // 32 * 32                         # noqa: E731
        d0 = widths[0]
        iw = m.item_embedding.weight.detach()
        uwp, iwp = m.user_embedding.weight, m.item_embedding.weight
        # r04: ONE result buffer [nu + ni, D] (users first; the three served gathers then read one table in one launch), and the
        # previous pass's buffer is written again - its block 0 (E0, NGCF.py:120) NOT copied again - while both tables are the same
        # tensors at the same version and nobody else holds the previous result (autograd.E0Cache.only_the_modules: a caller who
        # kept it gets a fresh buffer, as before).  At W = 8 that is the rank's share of the E0 copy per pass (23 us of a 2 ms step).
        e0_tag = (uwp.data_ptr(), int(uwp._version), iwp.data_ptr(), int(iwp._version), tuple(widths), nu, ni)
        cache = self.__dict__.setdefault("_e0", E0Cache())
        keep = cache.tag == e0_tag and cache.only_the_modules(self)
        if keep:
            allE = cache.all_E
        else:
            cache.invalidate()
            self._all_E = self.all_users_emb = self.all_items_emb = None
            allE = torch.empty((nu + ni, D), dtype=torch.float32, device=self.dev)
            _eng.copy_rows(uwp.detach()[lo:lo + nu], allE[:nu, :d0])
            _eng.copy_rows(iw[self.ib[r]:self.ib[r + 1]], allE[nu:, :d0])
            cache.all_E, cache.tag = allE, e0_tag
        allE_u, allE_i = allE[:nu], allE[nu:]                       # this rank's users / the items this rank owns
        # layer-0 item replica from the replicated parameter table (padded numbering): local copies, no communication - and none at
        # all while the table is the same tensor at the same version as in the previous pass (inference loops)
        ei = self._buf(("ei", 0), (PI, ld(d0)))[:, :d0]
        tag = (iw.data_ptr(), m.item_embedding.weight._version, tuple(iw.shape))
        if getattr(self, "_ei0_tag", None) != tag:
            for q in range(W):
                n_q = self.ib[q + 1] - self.ib[q]
                if n_q:
                    _eng.copy_rows(iw[self.ib[q]:self.ib[q + 1]], ei[q * mi:q * mi + n_q])
            self._ei0_tag = tag
        eu = allE_u[:, :d0]
        if d0 % 4 or D % 4:                                        # rows of all_E are not 16-byte aligned: an aligned copy to gather from
            eu = self._buf(("eu0",), (nu, ld(d0)))[:, :d0]
            if not keep:
                _eng.copy_rows(uwp.detach()[lo:lo + nu], eu)
        ex = self.p2p if self.backend == "p2p" else None
        self._calls += 1
        base = self._calls * (n_layer + 1)
        gloo = dist.get_backend(self.group) != "nccl"
        off = d0
        pending = None                                             # the all-gather of the previous layer's item carry
        ACK = 62                                                   # the one acknowledgement slot: "I have read everything of pass n"
        if ex is not None:
            ex.wait_acks(ACK, self._calls - 1)                     # every peer is done with what the previous pass left in my buffer
        for k in range(n_layer):
            d_in, d_out = widths[k], widths[k + 1]
            last = k == n_layer - 1
            seq = base + k + 1
            sp, sc = 2 * k, 2 * k + 1                              # sequence slots: partial sums / carry of this layer
            if ex is not None:
                ex.fence()                                         # copies of this layer start after the previous layer's readers
                # -- A: item partial sums over the local users, into the exchange buffer
                part = ex.floats(self._off_part[k], PI * ld(d_in)).view(PI, ld(d_in))[:, :d_in]
                _eng.spmm(self.csr_it, eu, out=part, ws=self.ws)
                ex.publish(sp, seq)
                # -- the previous layer's item carry arrives while A runs
                if pending is not None:
                    nxt_ei, pseq, psc, pld = pending
                    for q in ex.peers_from(r):
                        ex.pull(q, psc, pseq, self._off_carry[k - 1], nxt_ei[q * mi:(q + 1) * mi])
                    ex.join()
                    ei = nxt_ei[:, :d_in]
                    pending = None
            else:
                if pending is not None:
                    pending.wait()
                part = self._buf(("part", k), (PI, ld(d_in)))[:, :d_in]
                _eng.spmm(self.csr_it, eu, out=part, ws=self.ws)
            # -- B: user rows, fully local (gathers from the item replica); overlaps with the reduce-scatter
            if ex is None:
                le_own = self._buf(("le_own", k), (mi, ld(d_in)))
                full_part = self._bufs[("part", k)]
                if gloo:                                           # gloo has no reduce-scatter: all-reduce and keep the own rows
                    work = dist.all_reduce(full_part, group=self.group, async_op=True)
                else:
                    work = dist.reduce_scatter_tensor(le_own, full_part, group=self.group, async_op=True)
            cu = None if last else self._buf(("cu", k), (nu, ld(d_out)))[:, :d_out]
            _eng.layer_fused(self.csr_u, ei, eu, w1[k], b1[k], w2[k], b2[k], cu, allE_u[:, off:off + d_out], self.ws)
            # -- reduce-scatter: the partial sums of the owned items from every rank, added in rank order
            if ex is not None:
                slots = self._buf(("slots", k), (W, mi, ld(d_in)))
                for q in ex.peers_from(r):
                    ex.pull(q, sp, seq, self._off_part[k] + r * mi * ld(d_in), slots[q])
                    if last:
                        ex.ack(q, ACK, self._calls)                # the last copy of this pass from rank q is enqueued
                ex.join()
                le_own = self._buf(("le_own", k), (mi, ld(d_in)))
                with _eng._on(self.dev):
                    _lib.check(_lib.load().ngcf_sum_slots_f32(_eng._ptr(slots), mi * ld(d_in), W, mi * ld(d_in), _eng._ptr(le_own),
                                                              _eng._stream()))
            else:
                work.wait()
                if gloo:
                    le_own = full_part[r * mi:(r + 1) * mi]
            # -- C: dense half for the owned items; their carry rows go to every rank's replica of the next layer
            if last:
                ci = None
            elif ex is not None:
                ci = ex.floats(self._off_carry[k], mi * ld(d_out)).view(mi, ld(d_out))
            else:
                ci = self._buf(("ci", k), (mi, ld(d_out)))
            if ni:
                _eng.layer_dense(le_own[:ni, :d_in], ei[r * mi:r * mi + ni], w1[k], b1[k], w2[k], b2[k],
                                 None if last else ci[:ni, :d_out], allE_i[:, off:off + d_out], self.ws)
            if not last:
                nxt_ei = self._buf(("ei", k + 1), (PI, ld(d_out)))
                if ex is not None:
                    ex.publish(sc, seq)
                    pending = (nxt_ei, seq, sc, ld(d_out))         # pulled at the top of the next layer, under its A
                else:
                    pending = dist.all_gather_into_tensor(nxt_ei, ci, group=self.group, async_op=True)
                    # (waited for before the next layer's swept A: an RCCL kernel beside the sweep costs more than the wait)
                    ei = nxt_ei[:, :d_out]
            eu = cu
            off += d_out
        self._all_E, self.all_users_emb, self.all_items_emb = allE, allE_u, allE_i
        return allE_u, allE_i

    def _gather_async(self, region: torch.Tensor, send: torch.Tensor):
        if self._cabi is not None:
            return self._cabi(region, send)
        return allgather_rows(region, send, self.group, async_op=True)

    def _propagate_allgather(self):
        m, lay, r, W, Cn = self.model, self.layout, self.rank, self.world, self.chunks
        w1, b1, w2, b2 = self._params()
        widths = [m.emb_size] + [int(w.shape[0]) for w in w1]
        D, n_layer = sum(widths), len(w1)
        nu, ni = self.nu, self.ni
        allE_u = torch.empty((nu, D), dtype=torch.float32, device=self.dev)
        allE_i = torch.empty((ni, D), dtype=torch.float32, device=self.dev)
        d0 = widths[0]
        uw, iw = m.user_embedding.weight.detach(), m.item_embedding.weight.detach()
        _eng.copy_rows(uw[lay.ub[r]:lay.ub[r + 1]], allE_u[:, :d0])
        _eng.copy_rows(iw[lay.ib[r] - self.U:lay.ib[r + 1] - self.U], allE_i[:, :d0])
        # layer-0 replica from the replicated parameter tables: local copies, no communication
        full = self._buf(("full", 0), (lay.P, d0))
        for q in range(W):
            for j in range(Cn):
                lo, hi = lay.chunk_range(q, j)
                if hi > lo:
                    _eng.copy_rows(uw[lo:hi], full[lay.user_pos(q, j):lay.user_pos(q, j) + hi - lo])
            if lay.n_items_of(q):
                _eng.copy_rows(iw[lay.ib[q] - self.U:lay.ib[q + 1] - self.U],
                               full[lay.item_pos(q):lay.item_pos(q) + lay.n_items_of(q)])
        off = d0
        first_row = [lay.chunk_range(r, j)[0] - lay.ub[r] for j in range(Cn)]      # chunk j's rows inside allE_u
        if self.backend == "p2p":
            return self._allgather_layers_p2p(full, allE_u, allE_i, widths, first_row, w1, b1, w2, b2)
        for k in range(n_layer):
            d_out = widths[k + 1]
            last = k == n_layer - 1
            nxt = None if last else self._buf(("full", (k + 1) % 2 + 1), (lay.P, d_out))
            works = []
            # item slab first, alone on the GPU; its (small) all-gather overlaps with the first user chunk
            si = None if last else self._buf(("si", k % 2), (lay.mi, d_out))
            _eng.layer_fused(self.csr_i, full, full[lay.item_pos(r):lay.item_pos(r) + ni], w1[k], b1[k], w2[k], b2[k],
                             None if last else si[:ni], allE_i[:, off:off + d_out], self.ws)
            if not last:
                works.append(self._gather_async(nxt[lay.n_user_pos:], si))
            # user slab, chunk by chunk: chunk j's all-gather is in flight while chunk j+1 computes
            for j in range(Cn):
                n_j = lay.n_users_of(r, j)
                sj = None if last else self._buf(("su", k % 2, j), (lay.mc, d_out))
                pos = lay.user_pos(r, j)
                _eng.layer_fused(self.csr_u[j], full, full[pos:pos + n_j], w1[k], b1[k], w2[k], b2[k],
                                 None if last else sj[:n_j], allE_u[first_row[j]:first_row[j] + n_j, off:off + d_out], self.ws)
                if not last:
                    a, b = lay.chunk_region(j)
                    works.append(self._gather_async(nxt[a:b], sj))
            for wk in works:                                       # only now: the next layer reads the whole replica
                wk.wait()
            if not last:
                full = nxt
            off += d_out
        self.allE_u, self.allE_i = allE_u, allE_i
        return allE_u, allE_i

    def _allgather_layers_p2p(self, full, allE_u, allE_i, widths, first_row, w1, b1, w2, b2):
        """The layers of the all-gather scheme over the CU-free exchange: every piece of carry rows this rank produces (item slab,
        then the user chunks) is written into its exchange buffer and published; the peers - and this rank itself - pull it
        into their replica of the next layer while the following piece computes.  No kernel of the exchange shares the CUs, so
        every piece runs on the L2-swept kernel where its plan pays."""
        lay, r, W, Cn, ex = self.layout, self.rank, self.world, self.chunks, self.p2p
        n_layer = len(w1)
        nu, ni = self.nu, self.ni
        self._calls += 1
        base = self._calls * (n_layer + 1)
        off = widths[0]
        ldo = lambda d: (d + 31) // 32 * 32                        # noqa: E731
        for k in range(n_layer):
            d_out = widths[k + 1]
            last = k == n_layer - 1
            seq, par = base + k + 1, k % 2
            ld = ldo(d_out)
            nxt = None if last else self._buf(("fullp", (k + 1) % 2 + 1), (lay.P, ld))
            if not last:
                ex.fence()
            # pieces: 0 = item slab, 1 + j = user chunk j; their rows inside the exchange buffer and inside every replica
            send_off = [self._off_send[par]] + [self._off_send[par] + (lay.mi + j * lay.mc) * ld for j in range(Cn)]
            rows = [lay.mi] + [lay.mc] * Cn

            def region(q, piece):
                a = lay.item_pos(q) if piece == 0 else lay.user_pos(q, piece - 1)
                return nxt[a:a + rows[piece]]

            def pull_piece(piece):
                slot = par * (Cn + 1) + piece
                for q in ex.peers_from(r):
                    ex.pull(q, slot, seq, send_off[piece], region(q, piece))
                    ex.ack(q, slot, seq)

            for piece in range(Cn + 1):
                slot = par * (Cn + 1) + piece
                send = None
                if not last:
                    ex.wait_acks(slot, self._last_pub.get(slot, 0))
                    send = ex.floats(send_off[piece], rows[piece] * ld).view(rows[piece], ld)
                if piece == 0:
                    _eng.layer_fused(self.csr_i, full, full[lay.item_pos(r):lay.item_pos(r) + ni], w1[k], b1[k], w2[k], b2[k],
                                     None if last else send[:ni, :d_out], allE_i[:, off:off + d_out], self.ws)
                else:
                    j = piece - 1
                    n_j, pos = lay.n_users_of(r, j), lay.user_pos(r, j)
                    _eng.layer_fused(self.csr_u[j], full, full[pos:pos + n_j], w1[k], b1[k], w2[k], b2[k],
                                     None if last else send[:n_j, :d_out],
                                     allE_u[first_row[j]:first_row[j] + n_j, off:off + d_out], self.ws)
                if not last:
                    ex.publish(slot, seq)
                    self._last_pub[slot] = seq
                    if piece > 0:
                        pull_piece(piece - 1)                      # the previous piece travels while this one computes
            if not last:
                pull_piece(Cn)
                ex.join()                                          # only now: the next layer reads the whole replica
                full = nxt[:, :d_out]
            off += d_out
        self.allE_u, self.allE_i = allE_u, allE_i
        return allE_u, allE_i

    # -- gathers + BPR (NGCF.py:151-156, bprloss.py:15-22) ---------------------------------------
    def p2p_stats(self, reset: bool = False):
        """Host time this rank spent blocked inside the exchanges' waits (propagation + gathers), as a dict; None without p2p."""
        exs = [e for e in (getattr(self, "p2p", None), getattr(self, "_gx", None)) if e is not None]
        if not exs:
            return None
        got = [e.stats(reset) for e in exs]
        return {"host_blocked_ms": sum(g[0] for g in got), "waits": sum(g[1] for g in got), "waits_blocked": sum(g[2] for g in got)}

    def invalidate_e0(self):
        """Forget that block 0 of the retained result buffers and the layer-0 item replica hold the current tables: the next pass
        copies them again.  Needed only after a write to an embedding table through `.data` (invisible to the version counters)."""
        self._ei0_tag = None
        if "_e0" in self.__dict__:
            self._e0.invalidate()

    def _local_index(self, ix: torch.Tensor, lo: int) -> torch.Tensor:
        """ix - lo as a contiguous int64 device tensor, computed once per index tensor OBJECT and version (a loop over fixed
        batches pays no launch for it)."""
        import weakref
        memo = self.__dict__.setdefault("_loc_idx", {})
        key = (id(ix), int(ix._version), int(lo))
        hit = memo.get(key)
        if hit is not None and hit[0]() is ix:
            return hit[1]
        if len(memo) > 16:
            memo.clear()
        loc = (ix.to(device=self.dev, dtype=torch.int64) - lo).contiguous()
        memo[key] = (weakref.ref(ix), loc)
        return loc

    def _owned_rows(self, out, sets):
        """The rows of the batch this rank owns, gathered into their positions of `out` [sum of sizes, D]; positions other ranks own
        are left alone (the gather kernels skip ids outside the range: the status word of that call is a throw-away).  One launch
        when this rank's rows live in one buffer (the bipartite inference path), else one per index vector."""
        lib = _lib.load()
        D = int(out.shape[1])
        scrap = self._buf(("scrap_status",), (1,)).view(torch.int32)
        base = self.__dict__.get("_all_E")
        one = (base is not None and all(t._base is base for t, _, _, _ in sets) and len(sets) <= 3
               and self.__dict__.get("all_users_emb") is sets[0][0])
        with _eng._on(self.dev):
            if one:
                args, at = [], 0
                for table, ix, lo, n_rows in sets:
                    b = int(ix.numel())
                    row_off = 0 if table is self.all_users_emb else int(self.all_users_emb.shape[0])
                    args += [_eng._ptr(self._local_index(ix, lo)) if b and n_rows else None, b if n_rows else 0, row_off, n_rows,
                             _eng._ptr(out[at:at + b]) if b and n_rows else None]
                    at += b
                args += [None, 0, 0, 0, None] * (3 - len(sets))
                _lib.check(lib.ngcf_gather_rows3_f32(_eng._ptr(base), _eng._row_major_ld(base, "all_E"), D, *args, D, _eng._ptr(scrap),
                                                     _eng._stream()))
                return
            at = 0
            for table, ix, lo, n_rows in sets:
                b = int(ix.numel())
                if n_rows and b:
                    loc = self._local_index(ix, lo)
                    _lib.check(lib.ngcf_gather_rows_f32(_eng._ptr(table), _eng._row_major_ld(table, "table"), D, _eng._ptr(loc), b, 0, n_rows,
                                                        _eng._ptr(out[at:at + b]), D, _eng._ptr(scrap), _eng._stream()))
                at += b

    def _owner_index(self, u_id, pos_item, neg_item, Btot):
        """int64[Btot]: position b of the batch is row owner(b) * Btot + b of the [W * Btot, D] block of pulled rows.  Computed once
        per set of index tensors (same objects at the same version: a loop over fixed batches pays nothing)."""
        ids = (u_id, pos_item) + ((neg_item,) if neg_item is not None else ())
        key = tuple((id(t), int(t._version), int(t.numel())) for t in ids)
        hit = getattr(self, "_owner_idx", None)
        if hit is not None and hit[0] == key and all(a() is b for a, b in zip(hit[2], ids)):
            return hit[1]
        import weakref
        dev = self.dev
        if self.mode == "bipartite":
            ub, ib = torch.tensor(self.ub, device=dev), torch.tensor(self.ib, device=dev)
        else:
            ub, ib = torch.tensor(self.layout.ub, device=dev), torch.tensor(self.layout.ib, device=dev) - self.U
        W = self.world
        own = [(torch.searchsorted(ub, u_id, right=True) - 1).clamp(0, W - 1)]
        own += [(torch.searchsorted(ib, t, right=True) - 1).clamp(0, W - 1) for t in ids[1:]]
        idx = (torch.cat(own) * Btot + torch.arange(Btot, device=dev)).contiguous()
        self._owner_idx = (key, idx, [weakref.ref(t) for t in ids])
        return idx

    def _gather_p2p(self, sets, sizes, D, u_id, pos_item, neg_item):
        """The three row gathers over the CU-free exchange (r04; r03 used an all-reduce of a zero-filled [B, D] buffer - an RCCL
        kernel on the CUs at the end of every pass): every rank gathers the rows it OWNS into its exchange buffer and publishes;
        every rank pulls every peer's block (copy engines; W blocks of sum(sizes) x D floats = 6.3 MB each at C3's batch, all links
        at once) and one gather kernel picks position b out of its owner's block.  Bit-exact copies of the owners' rows.  Two
        regions in turn, so that a rank overwrites a block only after every peer has read it (acknowledgements per pass)."""
        Btot = sum(sizes)
        need = Btot * D
        gx = getattr(self, "_gx", None)
        if gx is None or self._gx_region < need:                   # (collective: every rank sees the same sizes in the same call)
            if gx is not None:
                for reg_ in (0, 1):                                # every peer has finished reading the old buffer before it goes
                    last = self._gx_calls - ((self._gx_calls - 1 - reg_) % 2)
                    gx.wait_acks(reg_, max(last, 0))
                gx.close()
            self._gx_region = need
            self._gx, self._gx_calls = None, 0
            try:       # (a failure is a failure on every rank: the constructor and the self-test share their verdicts)
                gx = P2PExchange(self.group, self.dev, max(2 * need, 1024))
                if not gx.selftest():
                    err = getattr(gx, "last_error", "self-test failed")
                    gx.close()
                    raise RuntimeError(err)
                self._gx = gx
            except Exception as exc:  # noqa: BLE001
                self.gather_p2p_error = repr(exc)[:300]           # the gathers keep the all-reduce; bench.py quotes the reason
                return None
        reg = self._gx_calls % 2
        self._gx_calls += 1
        seq = self._gx_calls
        gx.wait_acks(reg, max(seq - 2, 0))                              # the previous contents of this region have been read by everyone
        mine = gx.floats(reg * self._gx_region, need).view(Btot, D)
        self._owned_rows(mine, sets)
        gx.publish(reg, seq)
        slots = self._buf(("gather_slots",), (self.world, Btot, D))
        gx.fence()                                                 # the pulls land after the previous pass's reader of `slots`
        for q in gx.peers_from(self.rank):
            gx.pull(q, reg, seq, reg * self._gx_region, slots[q])
            gx.ack(q, reg, seq)
        gx.join()
        idx = self._owner_index(u_id, pos_item, neg_item, Btot)
        out = _eng.gather_rows(slots.view(self.world * Btot, D), idx, self.status)
        return torch.split(out, sizes)

    def _gather_inference(self, u_id, pos_item, neg_item):
        """The three row gathers with ONE exchange.  p2p transport: `_gather_p2p`.  torch.distributed: every rank copies the rows it
        owns into a zero-filled [Bu + Bp + Bn, D] buffer and one all-reduce adds the buffers - x + 0 is exact, so every rank ends
        up with the owners' bits."""
        r = self.rank
        if self.mode == "bipartite":
            u_lo, u_n, i_lo, i_n = self.ub[r], self.nu, self.ib[r], self.ni
        else:
            u_lo, u_n, i_lo, i_n = self.layout.ub[r], self.nu, self.layout.ib[r] - self.U, self.ni
        sets = [(self.allE_u, u_id, u_lo, u_n), (self.allE_i, pos_item, i_lo, i_n)]
        if neg_item is not None:
            sets.append((self.allE_i, neg_item, i_lo, i_n))
        D = int(self.allE_u.shape[1])
        sizes = [int(ix.numel()) for _, ix, _, _ in sets]
        outs = None
        if (self.backend == "p2p" and sum(sizes) > 0 and os.environ.get("NGCF_DIST_GATHER", "p2p") == "p2p"
                and getattr(self, "gather_p2p_error", None) is None):
            outs = self._gather_p2p(sets, sizes, D, u_id, pos_item, neg_item)
        if outs is None:
            buf = torch.zeros((sum(sizes), D), dtype=torch.float32, device=self.dev)
            self._owned_rows(buf, sets)
            dist.all_reduce(buf, group=self.group)
            outs = torch.split(buf, sizes)
        return outs[0], outs[1], (outs[2] if neg_item is not None else torch.empty(0))

    def gather(self, u_id: torch.Tensor, pos_item: torch.Tensor, neg_item: torch.Tensor):
        """(u, pos, neg) `[B, D]` on every rank.  Rows are served by their owning rank and summed."""
        dev = self.dev
        u_id, pos_item = u_id.to(dev), pos_item.to(dev)
        if not (torch.is_grad_enabled() and (self.allE_u.requires_grad or self.allE_i.requires_grad)):
            return self._gather_inference(u_id, pos_item, neg_item.to(dev) if len(neg_item) > 0 else None)

        def served(table, owner, local, n_rows):
            mine = owner == self.rank
            idx = torch.where(mine, local, torch.zeros_like(local))
            if table.requires_grad:                                # training path: torch indexing carries the gradient
                rows = table[idx] if n_rows else torch.zeros((idx.numel(), table.shape[1]), device=dev)
                return OwnerRowsSum.apply(rows, mine, self.group)
            rows = _eng.gather_rows(table, idx, self.status, 0, max(n_rows, 1)) if n_rows else \
                torch.zeros((idx.numel(), table.shape[1]), device=dev)
            return owner_rows_sum(rows, mine, self.group)

        if self.mode == "bipartite":
            ub, ib = torch.tensor(self.ub, device=dev), torch.tensor(self.ib, device=dev)
            ow = (torch.searchsorted(ub, u_id, right=True) - 1).clamp(0, self.world - 1)
            u = served(self.allE_u, ow, u_id - ub[ow], self.nu)
            ow = (torch.searchsorted(ib, pos_item, right=True) - 1).clamp(0, self.world - 1)
            p = served(self.allE_i, ow, pos_item - ib[ow], self.ni)
            n = torch.empty(0)
            if len(neg_item) > 0:
                neg_item = neg_item.to(dev)
                ow = (torch.searchsorted(ib, neg_item, right=True) - 1).clamp(0, self.world - 1)
                n = served(self.allE_i, ow, neg_item - ib[ow], self.ni)
        else:
            lay = self.layout
            ow, loc = lay.owner_of_user(u_id)
            u = served(self.allE_u, ow, loc, self.nu)
            ow, loc = lay.owner_of_item(pos_item)
            p = served(self.allE_i, ow, loc, self.ni)
            n = torch.empty(0)
            if len(neg_item) > 0:
                ow, loc = lay.owner_of_item(neg_item.to(dev))
                n = served(self.allE_i, ow, loc, self.ni)
        return u, p, n
