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

``bipartite`` (~10x fewer exchanged bytes)
    `L = [[0, R], [R^T, 0]]` (matrix.py:49-52): user rows only read item embeddings and vice versa.
    Users are partitioned, the (small) item block is replicated.  User rows are then fully local; item
    rows are partial sums over the local users followed by ONE `all_reduce` of `[I, d]` per layer, which
    overlaps with the user-row kernels.  fp32 summation order differs from the single-GPU engine (tolerance,
    not bit-exact); all ranks hold bit-identical item rows because the all-reduce result is.

A rank never holds more of the graph than its own slabs: they are cut from the interaction triplets (host or
device tensors, `ShardedPropagation.from_interactions`) or from a row-sorted COO (`from_coo`).

Compute is always the HIP engine (`engine.py`); nothing here has a CPU path.  The layout/exchange helpers
are backend-agnostic tensor plumbing, which is what the world_size-2 `gloo` tests exercise on CPU tensors.
Forward only: training across ranks (a backward through the exchange) is not built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib
from . import engine as _eng

SCHEME_NOTES = {
    "allgather": "row partition (users and items, cut by stored entries) + RCCL all-gather of the carry per layer, user slab "
                 "pipelined in row chunks (BASELINE config 4)",
    "bipartite": "users partitioned, item block replicated; item rows = partial sums over local users + one all-reduce of "
                 "[I, d] per layer (an optimisation beside the north-star scheme: ~10x fewer exchanged bytes)",
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
        if mode == "bipartite":
            self.ub = even_bounds(0, U, W)
            lo, hi = self.ub[r], self.ub[r + 1]
            self.nu = hi - lo
            (ur, uc, uv), _ = cut(lo, hi, 0, 0)
            ur, uc, uv = to_dev(ur), to_dev(uc), to_dev(uv)
            # user rows: local users x all items (columns renumbered to item ids)
            self.csr_u = _eng.LaplacianCSR.from_coo(ur - lo, uc - U, uv, self.nu, I)
            # item rows restricted to local user columns -> partial sums: the transpose of the same slice
            order = torch.sort(uc, stable=True).indices
            self.csr_it = _eng.LaplacianCSR.from_coo(uc[order] - U, ur[order] - lo, uv[order], I, max(self.nu, 1))
            # the item partial sums run alone on the GPU (the all-reduce starts after them): L2-swept kernel where it
            # pays.  The user rows overlap with the all-reduce; a persistent one-workgroup-per-CU kernel must not
            # share the CUs with the collective's kernels, so they stay on the row-wise kernels.
            self.csr_it.set_mode(swept_mode)
            self.local_nnz = self.csr_u.nnz + self.csr_it.nnz
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
            if W == 1:
                for c in self.csr_u:
                    c.set_mode(swept_mode)
            self.local_nnz = sum(c.nnz for c in self.csr_u) + self.csr_i.nnz
        self._bufs = {}
        self._cabi = None
        if os.environ.get("NGCF_DIST_COLLECTIVES") == "cabi" and dist.get_backend(group) == "nccl":
            self._cabi = _CabiAllGather(group, dev)

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
        return self._propagate_bipartite() if self.mode == "bipartite" else self._propagate_allgather()

    def _propagate_bipartite(self):
        m = self.model
        w1, b1, w2, b2 = self._params()
        widths = [m.emb_size] + [int(w.shape[0]) for w in w1]
        D, n_layer = sum(widths), len(w1)
        lo = self.ub[self.rank]
        nu, I = self.nu, self.I
        allE_u = torch.empty((nu, D), dtype=torch.float32, device=self.dev)       # this rank's users
        allE_i = torch.empty((I, D), dtype=torch.float32, device=self.dev)        # all items, replicated
        d0 = widths[0]
        _eng.copy_rows(m.user_embedding.weight.detach()[lo:lo + nu], allE_u[:, :d0])
        _eng.copy_rows(m.item_embedding.weight.detach(), allE_i[:, :d0])
        eu, ei = allE_u[:, :d0], allE_i[:, :d0]
        off = d0
        for k in range(n_layer):
            d_in, d_out = widths[k], widths[k + 1]
            last = k == n_layer - 1
            # 1) item rows: partial sums over the local users, then one all-reduce (async on RCCL's stream)
            part = self._buf(("part", k % 2), (I, d_in))
            _eng.spmm(self.csr_it, eu, out=part, ws=self.ws)
            work = dist.all_reduce(part, group=self.group, async_op=True)
            # 2) user rows: fully local (gathers from the replicated item block) - overlaps with the all-reduce
            cu = None if last else self._buf(("cu", k % 2), (nu, d_out))
            _eng.layer_fused(self.csr_u, ei, eu, w1[k], b1[k], w2[k], b2[k], cu, allE_u[:, off:off + d_out], self.ws)
            # 3) dense half for ALL items (replicated: every rank gets bit-identical rows)
            work.wait()
            ci = None if last else self._buf(("ci", k % 2), (I, d_out))
            _eng.layer_dense(part, ei, w1[k], b1[k], w2[k], b2[k], ci, allE_i[:, off:off + d_out], self.ws)
            eu, ei = cu, ci
            off += d_out
        self.allE_u, self.allE_i = allE_u, allE_i
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

    # -- gathers + BPR (NGCF.py:151-156, bprloss.py:15-22) ---------------------------------------
    def gather(self, u_id: torch.Tensor, pos_item: torch.Tensor, neg_item: torch.Tensor):
        """(u, pos, neg) `[B, D]` on every rank.  Rows are served by their owning rank and summed."""
        dev = self.dev
        u_id, pos_item = u_id.to(dev), pos_item.to(dev)

        def served(table, owner, local, n_rows):
            mine = owner == self.rank
            idx = torch.where(mine, local, torch.zeros_like(local))
            rows = _eng.gather_rows(table, idx, self.status, 0, max(n_rows, 1)) if n_rows else \
                torch.zeros((idx.numel(), table.shape[1]), device=dev)
            return owner_rows_sum(rows, mine, self.group)

        if self.mode == "bipartite":
            ub = torch.tensor(self.ub, device=dev)
            ow = (torch.searchsorted(ub, u_id, right=True) - 1).clamp(0, self.world - 1)
            u = served(self.allE_u, ow, u_id - ub[ow], self.nu)
            p = _eng.gather_rows(self.allE_i, pos_item, self.status)
            n = _eng.gather_rows(self.allE_i, neg_item.to(dev), self.status) if len(neg_item) > 0 else torch.empty(0)
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
