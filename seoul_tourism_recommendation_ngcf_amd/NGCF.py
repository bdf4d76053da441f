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
from typing import List, Optional

import torch
import torch.nn as nn

from . import engine as _eng
from .autograd import GatherTriple, propagate_with_grad


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
        self._ws = _eng.Workspace()
        self._carry: List[Optional[torch.Tensor]] = [None, None]
        self._scratch: Optional[torch.Tensor] = None
        self._status: Optional[torch.Tensor] = None
        self.check_indices = True        # raise IndexError on out-of-range ids (one host sync per forward)
        # where the random masks come from: "reference" = torch's default CPU generator, drawn exactly where the reference
        # draws (bit-identical masks for the same torch.manual_seed, one host round trip per layer); "device" = counter
        # hash evaluated inside the kernels (same distribution and semantics, another stream, no host work)
        self.node_dropout_mode = "reference"
        self.mess_dropout_mode = "reference"
        self.all_users_emb = None
        self.all_items_emb = None
        self._all_E = None

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
            csr = self._transposed_csr(torch.stack([rows, cols]), vals)
            csr.set_mode(_spmm_mode())
            self._csr_cache[key] = csr
        return csr

    def _transposed_csr(self, idx: torch.Tensor, val: torch.Tensor) -> "_eng.LaplacianCSR":
        """CSR of the transpose (the device-mode edge dropout is keyed by (row, column) of L: no entry map is needed)."""
        N = self.n_user + self.n_item
        order = torch.sort(idx[1], stable=True).indices          # by column = row of L^T, original order kept inside
        return _eng.LaplacianCSR.from_coo(idx[1][order], idx[0][order], val[order], N, N)

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
        csrs, kept, masks = [], [], []
        if node_ref:
            L = self.lap_list[year_idx]
            idx = L._indices().to(dev)
            val = L._values().to(device=dev, dtype=torch.float32)
        for k in range(self.n_layer):
            if node_ref:
                mask = torch.nn.functional.dropout(torch.ones(val.numel(), dtype=torch.float64),
                                                   p=self.node_dropout, training=True).type(torch.bool).to(dev)
                idx, val = idx[:, mask], val[mask]
                csrs.append(_eng.LaplacianCSR.from_coo(idx[0], idx[1], val, N, N))
                kept.append((idx, val))
            if mess_ref and drop[k] > 0:
                noise = torch.nn.functional.dropout(torch.ones((N, widths[k + 1]), dtype=torch.float32),
                                                    p=drop[k], training=True)
                masks.append(noise.to(dev, non_blocking=False))
            else:
                masks.append(None)
        # the thinned matrices are not symmetric: their transposes are built only if a backward needs them
        return (csrs if node_ref else None, (lambda: [self._transposed_csr(i, v) for i, v in kept]) if node_ref else None,
                masks if mess_ref else None)

    def _private_seeds(self, n: int):
        """64-bit seeds for the device-mode hash streams, from a generator of the module's own: the default CPU stream is
        consumed only where the reference consumes it.  Follows torch.manual_seed (re-seeded when the default seed changes)."""
        src = torch.initial_seed()
        if getattr(self, "_seed_gen", None) is None or self._seed_src != src:
            self._seed_gen = torch.Generator(device="cpu").manual_seed(src ^ 0x5DEECE66D)
            self._seed_src = src
        return [int(x) for x in torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, generator=self._seed_gen)]

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
        if csrs is None:
            csrs = [self.laplacian_csr(year_idx)] * self.n_layer
            csrs_t_fn = lambda: [self.laplacian_csr_t(year_idx)] * self.n_layer   # noqa: E731
            if node_flag:
                # "device" mode: same semantics (cumulative, unscaled), mask = counter-based hash of the entry number
                # evaluated inside the SpMM kernel; one 64-bit seed per layer
                ns = self._private_seeds(self.n_layer)
                edge_drops = [(ns[:k + 1], float(self.node_dropout)) for k in range(self.n_layer)]
        seeds = [0] * self.n_layer
        if any(p > 0 for p in drop) and masks is None:                 # "device" mode: hash stream in the layer epilogue
            seeds = self._private_seeds(self.n_layer)
        w1, b1, w2, b2 = self._layer_params()
        all_E = propagate_with_grad(self, csrs, csrs_t_fn, self.user_embedding.weight, self.item_embedding.weight,
                                    w1, b1, w2, b2, drop, seeds, edge_drops, masks)
        self._all_E = all_E
        self.all_users_emb = all_E[:self.n_user, :]                    # NGCF.py:148-149
        self.all_items_emb = all_E[self.n_user:, :]
        return all_E

    # ------------------------------------------------------------------------------------
    # forward (NGCF.py:102-156)
    # ------------------------------------------------------------------------------------
    def forward(self, year, u_id, age, sex, month, day, dow, pos_item, neg_item, node_flag):
        dev = self._dev()
        status = self._status_buf(dev)
        fw = self.emb_size // 5
        if 5 * fw != self.emb_size:
            # the reference fails at NGCF.py:114 with a shape-mismatch RuntimeError for such widths
            raise RuntimeError(f"shape mismatch: value tensor of shape [{len(u_id)}, {5 * fw}] cannot be broadcast "
                               f"to indexing result of shape [{len(u_id)}, {self.emb_size}]")
        # feature injection into user_embedding.weight.data, no autograd (NGCF.py:103-115)
        with torch.no_grad():
            keep = _eng.feature_inject(
                self.user_embedding.weight.data,
                (self.age_emb.weight.data, self.sex_emb.weight.data, self.month_emb.weight.data,
                 self.day_emb.weight.data, self.dow_emb.weight.data),
                (age, sex, month, day, dow), u_id, self.emb_ratio, self._scratch_buf(dev), status)

        year_idx = int(year.min().item() % 18) if year.numel() else 0   # == year.unique()[0] % 18, NGCF.py:117
        self.propagate(year_idx, bool(node_flag))

        u_idx = u_id.to(device=dev, dtype=torch.int64).contiguous()
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
        if self.check_indices:
            if int(status.item()) != 0:
                status.zero_()
                raise IndexError("index out of range in NGCF.forward (u_id / feature ids / pos_item / neg_item)")
        del keep
        return u_embeddings, pos_i_embeddings, neg_i_embeddings

