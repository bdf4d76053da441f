"""Multi-GPU propagation: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference is single-device; this is new design.  Rows of `L`, `E` and
`all_E` are independent given the previous layer's `E`, so the graph is row-partitioned and one exchange
step per layer moves embeddings between ranks.  Two exchange schemes share the same partition helpers:

``allgather`` (the north-star scheme)
    Every rank owns a contiguous user range and a contiguous item range, both cut so that stored entries
    are balanced (`ngcf_shard_plan`).  Nodes are renumbered into a padded rank-major space
    (``ShardLayout``) so that ONE `all_gather_into_tensor` per node group drops every rank's freshly
    computed carry rows straight into the replica the next layer gathers from - no unpack copies.
    Bytes received per rank and layer: (W-1)/W * N * d * 4.

``bipartite`` (default: ~10x fewer exchanged bytes)
    `L = [[0, R], [R^T, 0]]` (matrix.py:49-52): user rows only read item embeddings and vice versa.
    Users are partitioned, the (small) item block is replicated.  User rows are then fully local; item
    rows are partial sums over the local users followed by ONE `all_reduce` of `[I, d]` per layer, which
    overlaps with the user-row kernels.  fp32 summation order differs from the single-GPU engine (tolerance,
    not bit-exact); all ranks hold bit-identical item rows because the all-reduce result is.

Compute is always the HIP engine (`engine.py`); nothing here has a CPU path.  The layout/exchange helpers
are backend-agnostic tensor plumbing, which is what the world_size-2 `gloo` tests exercise on CPU tensors.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist

from . import engine as _eng


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


class ShardLayout:
    """Padded rank-major numbering of the N = U + I nodes for the all-gather scheme.

    Rank r owns users [ub[r], ub[r+1]) and items [ib[r], ib[r+1]) (global node ids; items are U-based).
    Padded position of rank r's k-th user: r*mu + k; of its k-th item: W*mu + r*mi + k, with mu / mi the
    largest chunk.  A replica of E in this numbering has P = W*(mu+mi) rows; padding rows are never
    referenced by any column index.
    """

    def __init__(self, n_user: int, n_item: int, user_bounds: Sequence[int], item_bounds: Sequence[int]):
        assert len(user_bounds) == len(item_bounds) and user_bounds[0] == 0 and user_bounds[-1] == n_user
        assert item_bounds[0] == n_user and item_bounds[-1] == n_user + n_item
        self.n_user, self.n_item = n_user, n_item
        self.world = len(user_bounds) - 1
        self.ub, self.ib = list(user_bounds), list(item_bounds)
        self.mu = max(max(self.ub[r + 1] - self.ub[r] for r in range(self.world)), 1)
        self.mi = max(max(self.ib[r + 1] - self.ib[r] for r in range(self.world)), 1)
        self.P = self.world * (self.mu + self.mi)

    def n_users_of(self, r): return self.ub[r + 1] - self.ub[r]
    def n_items_of(self, r): return self.ib[r + 1] - self.ib[r]
    def user_pos(self, r): return r * self.mu
    def item_pos(self, r): return self.world * self.mu + r * self.mi

    def to_padded(self, node: torch.Tensor) -> torch.Tensor:
        """Global node id -> padded position (vectorised)."""
        dev = node.device
        ub = torch.tensor(self.ub, device=dev)
        ib = torch.tensor(self.ib, device=dev)
        is_item = node >= self.n_user
        ru = torch.searchsorted(ub, node, right=True) - 1
        ri = torch.searchsorted(ib, node, right=True) - 1
        ru = ru.clamp(0, self.world - 1)
        ri = ri.clamp(0, self.world - 1)
        pu = ru * self.mu + (node - ub[ru])
        pi = self.world * self.mu + ri * self.mi + (node - ib[ri])
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


def allgather_rows(full_block: torch.Tensor, send: torch.Tensor, group=None):
    """full_block[W*m, d] <- every rank's send[m, d], rank-major (one collective, no unpack)."""
    assert full_block.is_contiguous() and send.is_contiguous()
    assert full_block.shape[0] == send.shape[0] * dist.get_world_size(group)
    dist.all_gather_into_tensor(full_block, send, group=group)


def owner_rows_sum(local_rows: torch.Tensor, owned: torch.Tensor, group=None) -> torch.Tensor:
    """Rows served by their owning rank (others contribute zeros), summed over ranks: every rank ends up with
    all B rows.  x + 0 is exact, so values are the owners' bits."""
    out = torch.where(owned[:, None], local_rows, torch.zeros((), dtype=local_rows.dtype, device=local_rows.device))
    dist.all_reduce(out, group=group)
    return out


# ------------------------------------------------------------------------------------------------
# sharded propagation on the HIP engine
# ------------------------------------------------------------------------------------------------
class ShardedPropagation:
    """NGCF.py:120-156 over `world` GPUs for one Laplacian slice.

    `coo` is the FULL row-sorted COO of the [N, N] Laplacian on this rank's device (every rank builds or
    loads the same one and keeps only its part).  Parameters are replicated: pass the same `NGCF` module
    (same seed / same state_dict) on every rank.
    """

    def __init__(self, model, rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor,
                 mode: str = "bipartite", group=None):
        assert mode in ("bipartite", "allgather")
        self.model, self.mode, self.group = model, mode, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.U, self.I = model.n_user, model.n_item
        self.N = self.U + self.I
        dev = rows.device
        self.dev = dev
        self.ws = _eng.Workspace()
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        U, I, W, r = self.U, self.I, self.world, self.rank
        n_user_entries = int(torch.searchsorted(rows, torch.tensor([U], device=dev)))
        if mode == "bipartite":
            if n_user_entries and (int(cols[:n_user_entries].min()) < U or int(cols[n_user_entries:].max()) >= U):
                raise RuntimeError("bipartite exchange needs L = [[0, R], [R^T, 0]] (matrix.py:49-52)")
            self.ub = even_bounds(0, U, W)
            lo, hi = self.ub[r], self.ub[r + 1]
            self.nu = hi - lo
            # user rows: local users x all items (columns renumbered to item ids)
            ur, uc, uv = slab_coo(rows, cols, vals, lo, hi)
            self.csr_u = _eng.LaplacianCSR.from_coo(ur, uc - U, uv, self.nu, I)
            # item rows restricted to local user columns -> partial sums
            ir, ic, iv = rows[n_user_entries:] - U, cols[n_user_entries:], vals[n_user_entries:]
            sel = (ic >= lo) & (ic < hi)
            self.csr_it = _eng.LaplacianCSR.from_coo(ir[sel], ic[sel] - lo, iv[sel], I, max(self.nu, 1))
            # the item partial sums run alone on the GPU (the all-reduce starts after them): L2-swept kernel where it
            # pays.  The user rows overlap with the all-reduce; a persistent one-workgroup-per-CU kernel must not
            # share the CUs with the collective's kernels, so they stay on the row-wise kernels.
            self.csr_it.set_mode(int(os.environ.get("NGCF_SPMM_MODE", "3")))
            self.local_nnz = self.csr_u.nnz + self.csr_it.nnz
        else:
            cnt = row_counts(rows, self.N)
            ub = balanced_bounds(cnt, 0, U, W)
            ib = balanced_bounds(cnt, U, self.N, W)
            self.layout = ShardLayout(U, I, ub, ib)
            lay = self.layout
            pc = lay.to_padded(cols)
            ur, uc, uv = slab_coo(rows, pc, vals, ub[r], ub[r + 1])
            ir, ic, iv = slab_coo(rows, pc, vals, ib[r], ib[r + 1])
            self.nu, self.ni = lay.n_users_of(r), lay.n_items_of(r)
            self.csr_u = _eng.LaplacianCSR.from_coo(ur, uc, uv, self.nu, lay.P)
            self.csr_i = _eng.LaplacianCSR.from_coo(ir, ic, iv, self.ni, lay.P)
            self.csr_i.set_mode(int(os.environ.get("NGCF_SPMM_MODE", "3")))   # runs with no collective in flight (see above)
            self.local_nnz = self.csr_u.nnz + self.csr_i.nnz
        self._bufs = {}

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

    def _propagate_allgather(self):
        m, lay, r, W = self.model, self.layout, self.rank, self.world
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
            if lay.n_users_of(q):
                _eng.copy_rows(uw[lay.ub[q]:lay.ub[q + 1]], full[lay.user_pos(q):lay.user_pos(q) + lay.n_users_of(q)])
            if lay.n_items_of(q):
                _eng.copy_rows(iw[lay.ib[q] - self.U:lay.ib[q + 1] - self.U],
                               full[lay.item_pos(q):lay.item_pos(q) + lay.n_items_of(q)])
        off = d0
        for k in range(n_layer):
            d_out = widths[k + 1]
            last = k == n_layer - 1
            su = None if last else self._buf(("su", k % 2), (lay.mu, d_out))
            si = None if last else self._buf(("si", k % 2), (lay.mi, d_out))
            e_u = full[lay.user_pos(r):lay.user_pos(r) + nu]
            e_i = full[lay.item_pos(r):lay.item_pos(r) + ni]
            # item slab first: its (small) all-gather then overlaps with the user slab's kernels
            _eng.layer_fused(self.csr_i, full, e_i, w1[k], b1[k], w2[k], b2[k], None if last else si[:ni],
                             allE_i[:, off:off + d_out], self.ws)
            nxt = None
            wi = None
            if not last:
                nxt = self._buf(("full", (k + 1) % 2 + 1), (lay.P, d_out))
                wi = dist.all_gather_into_tensor(nxt[W * lay.mu:], si, group=self.group, async_op=True)
            _eng.layer_fused(self.csr_u, full, e_u, w1[k], b1[k], w2[k], b2[k], None if last else su[:nu],
                             allE_u[:, off:off + d_out], self.ws)
            if not last:
                wu = dist.all_gather_into_tensor(nxt[:W * lay.mu], su, group=self.group, async_op=True)
                wi.wait()
                wu.wait()
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
