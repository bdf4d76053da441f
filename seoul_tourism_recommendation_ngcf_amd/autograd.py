"""Forward drivers of the propagation engine and the autograd seam for training.

`propagate_forward` issues the HIP layer kernels for NGCF.py:120-147 (inference path, nothing saved).
`Propagate` / `GatherTriple` / `BPRLoss` are `torch.autograd.Function`s so that `loss.backward()`
(experiment.py:57) reaches every parameter: their backward runs the HIP kernels of the "backward pass" section
of include/ngcf_hip.h - weight and input gradients on the fp32 matrix cores, `L^T . dLE` on the SpMM kernels; no library
GEMM at any width.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import List, Optional, Sequence

import torch

from . import _lib
from . import engine as _eng
from .engine import _ptr, _row_major_ld, _stream


SPARSE_LAST_LAYER = True        # row-sparse backward of the last layer (Propagate.backward); False: always the dense path
DENSE_GRAD_MAX_BYTES = 32 << 20  # all_E up to this size: GatherTriple hands over a dense gradient (no host sync in the backward)
sparse_last_layer_calls = 0     # how often the row-sparse path ran (tests)


def _padded_rows(n: int, d: int, dev) -> torch.Tensor:
    """[n, d] fp32 whose rows start on 128-byte lines (leading dimension rounded up to 32 floats): the float4 / L2-swept SpMM and
    the branch-free dense path then apply at the reference's own widths too (65 -> 96)."""
    return torch.empty((n, (d + 31) // 32 * 32), dtype=torch.float32, device=dev)[:, :d]


def _final_rows(n: int, d: int, dev) -> torch.Tensor:
    """[n, d] fp32, contiguous: the gradient of the two embedding tables as autograd wants it (see Propagate.backward)."""
    return torch.empty((n, d), dtype=torch.float32, device=dev)


def _alloc_all_E(N: int, widths, dev) -> torch.Tensor:
    """all_E [N, D] for the `cat` of NGCF.py:147.  The reference forces embed_size to a multiple of 5 (NGCF.py:39-43: 65, 130,
    515), so D is usually not a multiple of 4 and contiguous rows would not be 16-byte aligned.  Then the rows are padded to a
    multiple of 32 floats (a [N, D] view of a [N, ld] buffer: every row starts on a 128-byte line), so that block 0 - E0, which the
    first layer gathers from - serves the float4 / L2-swept kernels where it lies.  (r02 wrote a second, aligned copy of E0 for
    that: 1.2 GB more traffic per forward at C3's 130-wide tables.)"""
    D = sum(widths)
    if D % 4 == 0 and widths[0] % 4 == 0:
        return torch.empty((N, D), dtype=torch.float32, device=dev)
    return torch.empty((N, (D + 31) // 32 * 32), dtype=torch.float32, device=dev)[:, :D]


def _write_e0(owner, user_w, item_w, all_E, U, d0):
    """E0 into its column block of all_E (the `cat` of NGCF.py:120 and of NGCF.py:147); returns E0 as the first layer reads it:
    block 0 itself, whose rows are 16-byte aligned in either layout of `_alloc_all_E`."""
    _eng.copy_rows(user_w.detach(), all_E[:U, :d0])
    _eng.copy_rows(item_w.detach(), all_E[U:, :d0])
    return all_E[:, :d0]


class _Probe:
    pass


def static_result_counts(all_E: torch.Tensor):
    """(Python references, C++ references, tensors on the storage) of a graph's STATIC all_E - compared with the same reading taken
    when only the graph runner held it (`GraphedForward`, `NGCF._TrainGraphs`): anything more means a caller still holds the previous
    replay's `model.all_items_emb` / `all_users_emb` / a slice of them, and the next forward must not replay over it (the reference
    allocates a fresh all_E per call, NGCF.py:147-149)."""
    return (sys.getrefcount(all_E), all_E._use_count(), torch._C._storage_Use_Count(all_E.untyped_storage()._cdata))


def static_result_baseline(all_E: torch.Tensor):
    """The reading of `static_result_counts` for a tensor only its runner holds, taken through ONE intermediate frame - the shape of
    the later check (`NGCF._static_result_held(all_E, ...)` -> `static_result_counts(all_E)`): every frame that binds the tensor to a
    parameter is one more Python reference, so the two readings must come up the same call depth."""
    return static_result_counts(all_E)


class E0Cache:
    """The all_E of the previous inference forward, kept so that block 0 - E0 = cat(user table, item table), NGCF.py:120 - need not
    be copied again while the tables are unchanged (r04; SURVEY 2.2 K4: 563 MB through `copy_rows_kernel` per forward at C3, 0.21 ms).

    Re-used only when ALL of this holds, else a fresh all_E is allocated and E0 copied in full, exactly as before:
      * both tables are the same tensors (address, shape) at the same autograd version counter as when block 0 was written
        (optimizer steps, `load_state_dict`, `nn.init` and every other in-place op on the Parameter bump it; `.to()` moves it);
      * nobody but the module holds the old all_E or a view of it: Python reference counts of the tensor and of the module's two
        views (`all_users_emb`, `all_items_emb`) and the use count of their storage are exactly the module's own - a caller who
        kept `model.all_items_emb` (demo.py:233) or a slice of it keeps it intact, as with the reference's fresh `cat` per call
        (copy-on-hold rather than a rotation of buffers).
    The rows the feature injection rewrites through `.data` (NGCF.py:114-115 - invisible to the version counter, but done by this
    module's own kernel) are recorded (`touch`) and copied row by row.  What the cache cannot see: a write through `.data` from
    OUTSIDE the module (`.detach()`-ed aliases share the counter and are seen) - call `model.invalidate_all_E()` after such a
    write, or set `NGCF.reuse_all_E = False`."""

    MAX_TOUCHED = 8            # batches of injected rows remembered; beyond that the next forward copies everything

    def __init__(self):
        self.all_E, self.tag, self.touched = None, None, []

    def invalidate(self):
        self.all_E, self.tag, self.touched = None, None, []

    def touch(self, rows: torch.Tensor):
        """`rows` (int64, device) of the user table were rewritten in place."""
        if self.all_E is None:
            return
        if len(self.touched) >= self.MAX_TOUCHED:
            self.invalidate()
        else:
            self.touched.append(rows)

    @staticmethod
    def tag_of(user_w, item_w, widths, dev):
        return (user_w.data_ptr(), int(user_w._version), tuple(user_w.shape), item_w.data_ptr(), int(item_w._version),
                tuple(item_w.shape), tuple(widths), str(dev))

    def _counts(self, owner):
        """(Python references to all_E and to each of the module's two views, C++ references to the three tensors - a DLPack capsule
        or another extension holding one of them shows up there -, tensors on the storage) - read the same way here and in the
        calibration below, so the constants of this interpreter / torch build cancel out."""
        d = owner.__dict__
        return (sys.getrefcount(self.all_E), sys.getrefcount(d["all_users_emb"]), sys.getrefcount(d["all_items_emb"]),
                self.all_E._use_count(), d["all_users_emb"]._use_count(), d["all_items_emb"]._use_count(),
                torch._C._storage_Use_Count(self.all_E.untyped_storage()._cdata))

    _free = {}

    @classmethod
    def _free_counts(cls, padded: bool):
        """What `_counts` reads when nobody but the module and its cache holds the forward's tensors: measured once per layout of
        all_E (`_alloc_all_E`: a plain [N, D] tensor, or - D not a multiple of 4 - a [N, D] view of a padded buffer) on a toy built
        the way `NGCF.propagate` builds the real thing."""
        if padded not in cls._free:
            o, c = _Probe(), cls()
            t = torch.empty((4, 8))[:, :6] if padded else torch.empty((4, 8))
            o._all_E, o.all_users_emb, o.all_items_emb = t, t[:2, :], t[2:, :]
            c.all_E = t
            del t
            cls._free[padded] = c._counts(o)
        return cls._free[padded]

    def only_the_modules(self, owner) -> bool:
        """True iff the cached all_E and the module's two views of it are referenced by the module (and this cache) alone."""
        d = owner.__dict__
        if self.all_E is None or d.get("_all_E") is not self.all_E:
            return False
        root = self.all_E._base if self.all_E._base is not None else self.all_E
        for name in ("all_users_emb", "all_items_emb"):
            if d.get(name) is None or d[name]._base is not root:
                return False
        del root
        return self._counts(owner) == self._free_counts(self.all_E._base is not None)


def _all_E_with_e0(owner, user_w, item_w, N, widths, dev):
    """(all_E with block 0 = E0 written, E0 as the first layer reads it).  `owner._e0_cache` (NGCF modules; the graph runners
    have none): see `E0Cache`."""
    U, d0 = int(user_w.shape[0]), widths[0]
    cache = getattr(owner, "_e0_cache", None)
    if cache is not None and getattr(owner, "reuse_all_E", True) and not torch.cuda.is_current_stream_capturing():
        tag = E0Cache.tag_of(user_w, item_w, widths, dev)
        if cache.tag == tag and cache.only_the_modules(owner):
            all_E = cache.all_E
            for rows in cache.touched:
                _eng.copy_rows_indexed(user_w.detach(), all_E[:U, :d0], rows)
            cache.touched = []
            return all_E, all_E[:, :d0]
        cache.invalidate()                     # (drops the old block before the new one is allocated)
        all_E = _alloc_all_E(N, widths, dev)
        prev = _write_e0(owner, user_w, item_w, all_E, U, d0)
        cache.all_E, cache.tag = all_E, tag
        return all_E, prev
    all_E = _alloc_all_E(N, widths, dev)
    return all_E, _write_e0(owner, user_w, item_w, all_E, U, d0)


def propagate_forward(owner, csrs: Sequence["_eng.LaplacianCSR"], user_w: torch.Tensor, item_w: torch.Tensor,
                      w1, b1, w2, b2, drop: Sequence[float], seeds: Sequence[int], edge_drops=None, masks=None):
    """all_E [N, D] = [E0 | norm(E1) | ... | norm(En)] (NGCF.py:120-147), inference path.

    E0 is written once into its column block of all_E (this is both the `cat` of NGCF.py:120 and
    the one of NGCF.py:147); each layer reads its input in place and writes its normalised output
    straight into its own column block, the un-normalised carry into a ping-pong buffer.
    """
    dev = user_w.device
    U, I = int(user_w.shape[0]), int(item_w.shape[0])
    N = U + I
    d0 = int(user_w.shape[1])
    n_layer = len(w1)
    widths = [d0] + [int(w.shape[0]) for w in w1]
    all_E, prev = _all_E_with_e0(owner, user_w, item_w, N, widths, dev)
    off = d0
    for k in range(n_layer):
        d_out = widths[k + 1]
        carry = None
        if k < n_layer - 1:
            buf = owner._carry[k % 2]
            if buf is None or buf.device != dev or tuple(buf.shape) != (N, d_out):
                buf = _padded_rows(N, d_out, dev)
                owner._carry[k % 2] = buf
            carry = buf
        mk = None if masks is None else masks[k]
        if edge_drops is None:
            _eng.layer_fused(csrs[k], prev, prev, w1[k].detach(), b1[k].detach(), w2[k].detach(), b2[k].detach(),
                             carry, all_E[:, off:off + d_out], owner._ws, drop[k], seeds[k], mk)
        else:   # device-side node dropout: thinned SpMM, then the dense half
            LE = _eng.spmm(csrs[k], prev, ws=owner._ws, edge_drop=(edge_drops[k][0], edge_drops[k][1], False))
            _eng.layer_dense(LE, prev, w1[k].detach(), b1[k].detach(), w2[k].detach(), b2[k].detach(), carry,
                             all_E[:, off:off + d_out], owner._ws, drop[k], seeds[k], mk)
        prev = carry
        off += d_out
    return all_E


# ------------------------------------------------------------------------------------------------
# thin wrappers of the backward entry points
# ------------------------------------------------------------------------------------------------
def _bwd_pre(dN, dC, Cc, leaky, drop_p, seed, mask=None, row_ids=None):
    lib = _lib.load()
    n_rows, d = Cc.shape
    # rows padded to a multiple of 32 floats: 128-byte aligned rows at the reference's own widths too (65 -> 96), which is what
    # the fused input-gradient kernel reads dM through (16-byte pieces; columns past d are masked there)
    ldm = (d + 31) // 32 * 32
    dM = torch.empty((n_rows, ldm), dtype=torch.float32, device=Cc.device)[:, :d]
    with _eng._on(Cc.device):
        _lib.check(lib.ngcf_layer_bwd_pre_f32(_ptr(dN), 0 if dN is None else _row_major_ld(dN, "dN"), _ptr(dC),
                                              0 if dC is None else _row_major_ld(dC, "dC"), _ptr(Cc),
                                              _row_major_ld(Cc, "C"), n_rows, d, leaky, float(drop_p), int(seed),
                                              _ptr(mask), 0 if mask is None else _row_major_ld(mask, "drop_mask"),
                                              _ptr(row_ids), _ptr(dM), ldm, _stream()))
    return dM


def _bwd_weight(dM, LE, E, ws):
    """(gW1, gb1, gW2, gb2): gW1 = dM^T . (LE + E), gW2 = dM^T . (LE * E) on the fp32 matrix cores, gb2 = column sums of dM and
    gb1 = twice that (b1 enters the layer twice) from the same pass over dM (ngcf_layer_bwd_weight_f32), written straight into the
    four gradient tensors; blocks of at most 128 output rows x 128 input columns (one kernel call each)."""
    lib = _lib.load()
    n_rows, d_out = dM.shape
    d_in = int(LE.shape[1])
    gW1 = torch.empty((d_out, d_in), dtype=torch.float32, device=dM.device)
    gW2 = torch.empty((d_out, d_in), dtype=torch.float32, device=dM.device)
    gb1 = torch.empty((d_out,), dtype=torch.float32, device=dM.device)
    gb2 = torch.empty((d_out,), dtype=torch.float32, device=dM.device)
    w = ws.get(int(lib.ngcf_bwd_weight_workspace_bytes()), dM.device)
    with _eng._on(dM.device):
        for o0 in range(0, d_out, 128):
            o1 = min(d_out, o0 + 128)
            for c0 in range(0, d_in, 128):
                c1 = min(d_in, c0 + 128)
                a, b, c = dM[:, o0:o1], LE[:, c0:c1], E[:, c0:c1]
                _lib.check(lib.ngcf_layer_bwd_weight_f32(_ptr(a), _row_major_ld(a, "dM"), _ptr(b), _row_major_ld(b, "LE"),
                                                         _ptr(c), _row_major_ld(c, "E"), n_rows, c1 - c0, o1 - o0,
                                                         _ptr(gW1[o0:o1, c0:c1]), d_in, _ptr(gW2[o0:o1, c0:c1]), d_in,
                                                         _ptr(gb1[o0:o1]) if c0 == 0 else None, _ptr(gb2[o0:o1]) if c0 == 0 else None,
                                                         _ptr(w), w.numel(), _stream()))
    return gW1, gb1, gW2, gb2


def _bwd_input(dM, w1, w2, LE, E, ws):
    """dLE, dE_direct from dM in one MFMA kernel (ngcf_layer_bwd_input_f32): dM.[W1|W2] and its combination with E / LE."""
    lib = _lib.load()
    n_rows, d_out = dM.shape
    d_in = int(LE.shape[1])
    d4 = (d_in + 31) // 32 * 32   # 128-byte aligned rows: L^T . dLE then runs on the float4 / swept kernels at any width
    dLE = torch.empty((n_rows, d4), dtype=torch.float32, device=LE.device)[:, :d_in]
    dE = torch.empty((n_rows, d4), dtype=torch.float32, device=LE.device)[:, :d_in]
    w1, w2 = w1.contiguous(), w2.contiguous()
    w = ws.get(int(lib.ngcf_layer_bwd_input_workspace_bytes(d_out)), dM.device)
    with _eng._on(dM.device):
        _lib.check(lib.ngcf_layer_bwd_input_f32(_ptr(dM), _row_major_ld(dM, "dM"), n_rows, d_out, _ptr(w1), _ptr(w2), d_in,
                                                _ptr(LE), _row_major_ld(LE, "LE"), _ptr(E), _row_major_ld(E, "E"),
                                                _ptr(dLE), d4, _ptr(dE), d4, _ptr(w), w.numel(), _stream()))
    return dLE, dE


def _add_rows(out, add):
    lib = _lib.load()
    with _eng._on(out.device):
        _lib.check(lib.ngcf_add_rows_f32(_ptr(out), _row_major_ld(out, "out"), _ptr(add), _row_major_ld(add, "add"),
                                         out.shape[0], out.shape[1], _stream()))


class Propagate(torch.autograd.Function):
    """all_E = propagate(E0; W) with a hand-written backward (NGCF.py:120-147)."""

    @staticmethod
    def forward(ctx, owner, csrs, csrs_t, drop, seeds, edge_drops, masks, n_layer, user_w, item_w, *params):
        seeds, ctx.seed_words = seeds if isinstance(seeds, tuple) else (seeds, None)
        w1, b1 = params[:n_layer], params[n_layer:2 * n_layer]
        w2, b2 = params[2 * n_layer:3 * n_layer], params[3 * n_layer:]
        dev = user_w.device
        U, I = int(user_w.shape[0]), int(item_w.shape[0])
        N, d0 = U + I, int(user_w.shape[1])
        widths = [d0] + [int(w.shape[0]) for w in w1]
        if getattr(owner, "_e0_cache", None) is not None:
            owner._e0_cache.invalidate()       # (the module's all_E is about to be replaced by one this node saves for its backward)
        all_E = _alloc_all_E(N, widths, dev)
        prev = _write_e0(owner, user_w, item_w, all_E, U, d0)
        off = d0
        ins, les, carries = [], [], []
        for k in range(n_layer):
            d_out = widths[k + 1]
            ed = None if edge_drops is None else (edge_drops[k][0], edge_drops[k][1], False)
            LE = _eng.spmm(csrs[k], prev, ws=owner._ws, edge_drop=ed)        # saved for the backward
            carry = _padded_rows(N, d_out, dev)
            _eng.layer_dense(LE, prev, w1[k].detach(), b1[k].detach(), w2[k].detach(), b2[k].detach(), carry,
                             all_E[:, off:off + d_out], owner._ws, drop[k], seeds[k], None if masks is None else masks[k])
            ins.append(prev)
            les.append(LE)
            carries.append(carry)
            prev = carry
            off += d_out
        ctx.owner, ctx.csrs, ctx.csrs_t, ctx.drop, ctx.seeds, ctx.n_layer = owner, csrs, csrs_t, drop, seeds, n_layer
        ctx.edge_drops, ctx.masks = edge_drops, masks
        ctx.widths, ctx.U = widths, U
        ctx.save_for_backward(all_E, *les, *carries, *[p.detach() for p in params])
        return all_E

    @staticmethod
    def backward(ctx, g_all):
        n, widths, U = ctx.n_layer, ctx.widths, ctx.U
        saved = ctx.saved_tensors
        all_E = saved[0]
        les, carries = saved[1:1 + n], saved[1 + n:1 + 2 * n]
        params = saved[1 + 2 * n:]
        w1, w2 = params[:n], params[2 * n:3 * n]
        ws = ctx.owner._ws
        gw1, gb1, gw2, gb2 = [None] * n, [None] * n, [None] * n, [None] * n
        dC = None
        offs = [sum(widths[:k + 1]) for k in range(n)]
        # The gradient that reaches all_E from the row gathers is non-zero on at most 3 B rows, and GatherTriple hands it over as a
        # row-sparse tensor (rows + a compact [R, D] block of values; autograd densifies it only if another consumer of all_E
        # adds a dense gradient).  The LAST layer's backward then involves those rows only: normalise/LeakyReLU backward, both
        # weight gradients and the input gradients on a compacted [R, d] problem, and L^T . dLE as one pass over the stored entries
        # of L^T that picks the R non-zero rows of dLE through a slot table (ngcf_spmm_t_rows_f32: fixed summation order) -
        # instead of a full SpMM and four passes over 1.1 M rows.  Earlier layers are dense (their dC is), but the part of their incoming
        # gradient that comes through the normalised all_E block is still confined to the R rows: the dense pass runs with
        # dN = 0 and the R rows are redone with their dN.
        rows = gv = None
        if g_all.is_sparse:
            if SPARSE_LAST_LAYER:
                g = g_all.coalesce()
                rows, gv = g.indices()[0].contiguous(), g.values().contiguous()
            else:
                g_all = g_all.to_dense()
        if rows is None:
            g_all = g_all.contiguous()
        for k in reversed(range(n)):
            d_in, d_out = widths[k], widths[k + 1]
            E_k = all_E[:, :widths[0]] if k == 0 else carries[k - 1]
            LE_k, C_k = les[k], carries[k]
            mask_k = None if ctx.masks is None else ctx.masks[k]
            if dC is None and rows is not None:
                global sparse_last_layer_calls
                sparse_last_layer_calls += 1
                dM = _bwd_pre(gv[:, offs[k]:offs[k] + d_out], None, C_k[rows], _eng.LEAKY_SLOPE, ctx.drop[k], ctx.seeds[k],
                              None if mask_k is None else mask_k[rows], rows)
                LE_c, E_c = LE_k[rows], E_k[rows]
                gw1[k], gb1[k], gw2[k], gb2[k] = _bwd_weight(dM, LE_c, E_c, ws)
                dLE_c, dE_c = _bwd_input(dM, w1[k], w2[k], LE_c, E_c, ws)
                N = int(all_E.shape[0])
                dE = _final_rows(N, d_in, all_E.device) if k == 0 else _padded_rows(N, d_in, all_E.device)
                slot = torch.full((N,), -1, dtype=torch.int32, device=all_E.device)
                slot[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=all_E.device)
                ed = None if ctx.edge_drops is None else (ctx.edge_drops[k][0], ctx.edge_drops[k][1])
                _eng.spmm_t_rows(ctx.csrs_t[k], slot, dLE_c, dE_c, dE, ws, ed)     # dE = dE_direct + (thinned L)^T . dLE, fixed order
                dC = dE
                continue
            if rows is not None:
                dM = _bwd_pre(None, dC, C_k, _eng.LEAKY_SLOPE, ctx.drop[k], ctx.seeds[k], mask_k)
                dM[rows] = _bwd_pre(gv[:, offs[k]:offs[k] + d_out], dC[rows], C_k[rows], _eng.LEAKY_SLOPE, ctx.drop[k], ctx.seeds[k],
                                    None if mask_k is None else mask_k[rows], rows)
            else:
                dM = _bwd_pre(g_all[:, offs[k]:offs[k] + d_out], dC, C_k, _eng.LEAKY_SLOPE, ctx.drop[k], ctx.seeds[k], mask_k)
            gw1[k], gb1[k], gw2[k], gb2[k] = _bwd_weight(dM, LE_k, E_k, ws)      # MFMA kernel, operand formed on the fly; biases too
            dLE, dE = _bwd_input(dM, w1[k], w2[k], LE_k, E_k, ws)                 # one MFMA kernel at any width, dS/dP never stored
            del dM
            ed = None if ctx.edge_drops is None else (ctx.edge_drops[k][0], ctx.edge_drops[k][1], True)
            S = _eng.spmm(ctx.csrs_t[k], dLE, ws=ws, edge_drop=ed)                # (thinned L)^T . dLE
            if k == 0 and dE.stride(0) != d_in:
                # The gradient of the embedding tables leaves as the row blocks dE0[:U], dE0[U:].  At a width that is not a multiple
                # of 32 (65, 130, 515: the reference's own) dE is a view of padded rows, and autograd's AccumulateGrad CLONES a
                # gradient that is not laid out like its parameter (r04 trace: 0.24 ms per step at C3, two copy kernels per step at
                # the Seoul shape); the last sum is therefore written into a contiguous matrix, whose row blocks it takes as they are.
                dE = torch.add(dE, S, out=_final_rows(int(dE.shape[0]), d_in, dE.device))
            else:
                _add_rows(dE, S)                                                  # dE += ...
            del S
            dC = dE
        dE0 = dC
        if rows is not None:
            dE0.index_add_(0, rows, gv[:, :widths[0]])                           # the all_E block of E0 itself, R rows
        else:
            _add_rows(dE0, g_all[:, :widths[0]])
        return (None, None, None, None, None, None, None, None, dE0[:U], dE0[U:], *gw1, *gb1, *gw2, *gb2)


class GatherTriple(torch.autograd.Function):
    """(u, pos, neg) row gathers of NGCF.py:151-155; backward scatters into one dense gradient of all_E."""

    @staticmethod
    def forward(ctx, all_E, n_user, status, u_idx, p_idx, n_idx):
        U = int(n_user)
        n_item = int(all_E.shape[0]) - U
        u, p, n = _eng.gather_rows3(all_E, ((u_idx, 0, U), (p_idx, U, n_item), (n_idx, U, n_item)), status)
        ctx.shape, ctx.U, ctx.has_n = tuple(all_E.shape), U, n_idx is not None
        ctx.save_for_backward(u_idx, p_idx, *([n_idx] if n_idx is not None else []))
        return (u, p, n) if n is not None else (u, p)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        idx = ctx.saved_tensors
        N, D = ctx.shape
        offs = (0, ctx.U, ctx.U)
        live = [(g.contiguous(), ix + off) for g, ix, off in zip(grads, idx, offs) if g is not None]
        dev = idx[0].device
        if not live:
            return (torch.zeros((N, D), dtype=torch.float32, device=dev), None, None, None, None, None)
        # The gradient of all_E is non-zero on the gathered rows only (<= 3 B of N): it is handed over as a row-sparse tensor -
        # the sorted unique rows (one host sync for their count) and a compact [R, D] block the duplicates are added into -
        # instead of a zero-filled [N, D] matrix (2.3 GB at C3).  Propagate.backward works on those rows; any other consumer of
        # all_E gets the dense sum from autograd.
        # Duplicates (the same user or item several times in a batch) are added in batch order, row by row, by one kernel with a
        # fixed summation order (ngcf_segment_sum_rows_f32; r02 used float atomics and the gradients differed from run to run).
        pos_all = live[0][1] if len(live) == 1 else torch.cat([r for _, r in live])
        g_all = live[0][0] if len(live) == 1 else torch.cat([g for g, _ in live], dim=0)
        M = int(pos_all.numel())
        if M <= 8192 and N * D * 4 <= DENSE_GRAD_MAX_BYTES:
            # A SMALL graph (the Seoul data: all_E is 6 MB): a dense gradient costs nothing and the whole backward then runs
            # without a single host sync - the distinct rows, their count and the group bounds stay on the device between the
            # sort and the scatter of the segment sums into a zero-filled [N, D] matrix, and Propagate.backward takes its dense
            # path (no compaction gathers, no index_put: ~15 launches fewer per step than the row-sparse path)
            buf = torch.empty(3 * M + 2, dtype=torch.int64, device=dev)
            order, rows, segptr, cnt = buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:]
            G = torch.zeros((N, D), dtype=torch.float32, device=dev)
            with _eng._on(dev):
                _lib.check(lib.ngcf_rows_sort_unique(_ptr(pos_all.contiguous()), M, N - 1, _ptr(order), _ptr(rows), _ptr(segptr), _ptr(cnt), _stream()))
                _lib.check(lib.ngcf_segment_sum_rows_f32(_ptr(g_all), D, D, _ptr(order), _ptr(segptr), M, _ptr(rows), _ptr(cnt), _ptr(G), D,
                                                         N, _stream()))
            return (G, None, None, None, None, None)
        if M <= 8192 and N < (1 << 50):
            # one launch: sorted distinct rows, the positions grouped by row in batch order, the group bounds (ngcf_rows_sort_unique)
            buf = torch.empty(3 * M + 2, dtype=torch.int64, device=dev)
            order, rows, segptr, cnt = buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:]
            with _eng._on(dev):
                _lib.check(lib.ngcf_rows_sort_unique(_ptr(pos_all.contiguous()), M, N - 1, _ptr(order), _ptr(rows), _ptr(segptr), _ptr(cnt), _stream()))
            R = int(cnt.item())                                    # (the one host sync of the backward: sizes the compacted problem)
            rows, segptr = rows[:R], segptr[:R + 1]
        else:
            rows, inv, counts = torch.unique(pos_all, return_inverse=True, return_counts=True)
            R = int(rows.numel())
            order = torch.sort(inv, stable=True).indices
            segptr = torch.zeros(R + 1, dtype=torch.int64, device=dev)
            torch.cumsum(counts, 0, out=segptr[1:])
        vals = torch.empty((R, D), dtype=torch.float32, device=dev)
        with _eng._on(dev):
            _lib.check(lib.ngcf_segment_sum_rows_f32(_ptr(g_all), D, D, _ptr(order), _ptr(segptr), R, None, None, _ptr(vals), D, 0, _stream()))
        G = torch.sparse_coo_tensor(rows[None], vals, (N, D), is_coalesced=True)
        return (G, None, None, None, None, None)


class BPRLoss(torch.autograd.Function):
    """Fused BPR (bprloss.py:15-22) with its gradient kernel."""

    @staticmethod
    def forward(ctx, u, p, n, weight_decay, batch_size, ws):
        ctx.wd, ctx.bs = float(weight_decay), float(batch_size)
        ctx.save_for_backward(u, p, n)
        return _eng.bpr_loss(u, p, n, weight_decay, batch_size, ws)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        u, p, n = (t.contiguous() for t in ctx.saved_tensors)
        du, dp, dn = torch.empty_like(u), torch.empty_like(p), torch.empty_like(n)
        g = g.to(torch.float32).contiguous()
        with _eng._on(u.device):
            _lib.check(lib.ngcf_bpr_backward_f32(_ptr(u), u.shape[0], _ptr(p), p.shape[0], _ptr(n), n.shape[0], u.shape[1],
                                                 ctx.wd, ctx.bs, _ptr(g), _ptr(du), _ptr(dp), _ptr(dn), _stream()))
        return du, dp, dn, None, None, None


def propagate_with_grad(owner, csrs, csrs_t_fn, user_w, item_w, w1, b1, w2, b2, drop, seeds, edge_drops=None,
                        masks=None, keep_alive=None) -> torch.Tensor:
    """Inference path unless a gradient can flow; then the autograd Function (needs the CSRs of L^T)."""
    params = list(w1) + list(b1) + list(w2) + list(b2)
    need = torch.is_grad_enabled() and any(t.requires_grad for t in [user_w, item_w] + params)
    if not need:
        with torch.no_grad():
            return propagate_forward(owner, csrs, user_w, item_w, w1, b1, w2, b2, drop, seeds, edge_drops, masks)
    # (`keep_alive`: the device words behind tagged seeds - stored on the ctx so that they live as long as the backward can run)
    return Propagate.apply(owner, csrs, csrs_t_fn(), list(drop), (list(seeds), keep_alive), edge_drops, masks, len(w1), user_w, item_w,
                           *params)
