"""Synthetic Laplacians in the reference's `lap_list` layout (matrix.py:41-83), for bench and tests.

The real `lap_list.pkl` and the raw data set are absent from the reference checkout
(.MISSING_LARGE_BLOBS), so BASELINE.json's "Seoul tourism graph" is a same-shape stand-in and the
large configs are synthetic by definition.  All graphs are normalised the way
`Matrix.create_matrix` does: degree = COUNT of stored non-zeros per row (matrix.py:55), values keep
their weights, `d^-1/2` in float32, product in float64, result cast to float32 (matrix.py:56-62,82),
both triangles stored, entries row-major sorted.
"""
from __future__ import annotations

from typing import Dict, List

import torch


def _normalise(u: torch.Tensor, i: torch.Tensor, w: torch.Tensor, n_user: int, n_item: int) -> Dict[str, torch.Tensor]:
    """(u, i, w) unique interaction triplets sorted by (u, i) -> row-sorted COO of the [N, N] Laplacian."""
    deg_u = torch.bincount(u, minlength=n_user)
    deg_i = torch.bincount(i, minlength=n_item)
    ds_u = deg_u.to(torch.float32).pow(-0.5)
    ds_i = deg_i.to(torch.float32).pow(-0.5)
    ds_u[torch.isinf(ds_u)] = 0
    ds_i[torch.isinf(ds_i)] = 0
    v = (ds_u[u].double() * w.double() * ds_i[i].double()).float()
    # item rows: sort the same triplets by (i, u); stable sort of i keeps u ascending
    order = torch.sort(i, stable=True).indices
    rows = torch.cat([u, i[order] + n_user])
    cols = torch.cat([i + n_user, u[order]])
    vals = torch.cat([v, v[order]])
    return {"rows": rows, "cols": cols, "vals": vals}


def synthetic_interactions(n_user: int, n_item: int, n_inter: int, seed: int, device, item_skew: bool = True):
    """SURVEY.md 8d C3/C5 generator: u ~ Uniform, i = floor(I * r^2) (popularity skew), de-duplicated, weights
    U(0.5, 5.0).  Returns the unique interaction triplets (u, i, w) sorted by (u, i)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.randint(0, n_user, (n_inter,), generator=g, device=device, dtype=torch.int64)
    r = torch.rand((n_inter,), generator=g, device=device, dtype=torch.float64)
    if item_skew:
        r = r * r
    i = torch.clamp((r * n_item).to(torch.int64), max=n_item - 1)
    key = torch.unique(u * n_item + i)            # sorted, de-duplicated
    del u, i, r
    u, i = key // n_item, key % n_item
    del key
    w = torch.rand((u.numel(),), generator=g, device=device, dtype=torch.float32) * 4.5 + 0.5
    return u, i, w


def synthetic_bipartite(n_user: int, n_item: int, n_inter: int, seed: int, device, item_skew: bool = True):
    """The same graph as the row-sorted COO of its count-degree normalised Laplacian + sizes (nnz = 2 x interactions)."""
    u, i, w = synthetic_interactions(n_user, n_item, n_inter, seed, device, item_skew)
    coo = _normalise(u, i, w, n_user, n_item)
    coo.update({"n_user": n_user, "n_item": n_item, "interactions": int(u.numel()), "nnz": int(2 * u.numel())})
    return coo


def seoul_standin(device, seed: int = 1801, n_user: int = 5840, n_item: int = 100) -> List[Dict[str, torch.Tensor]]:
    """two year slices; mask Bernoulli(0.75) (per-user bottom quartile zeroed,
    utils.py:117-121), weights U(0.5, 5.0); slice 1 = slice 0 overlaid with a fresh draw (the `R` carry-over
    quirk, matrix.py:33,45).  Generated on the CPU generator (small) and moved to `device`."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    slices = []
    W_prev = None
    for _ in range(2):
        mask = torch.rand((n_user, n_item), generator=g) < 0.75
        W = torch.where(mask, torch.rand((n_user, n_item), generator=g) * 4.5 + 0.5, torch.zeros(()))
        if W_prev is not None:
            W = torch.where(W != 0, W, W_prev)
        W_prev = W
        nz = W.nonzero()
        u, i = nz[:, 0].contiguous(), nz[:, 1].contiguous()
        coo = _normalise(u, i, W[u, i].float(), n_user, n_item)
        coo = {k: v.to(device) for k, v in coo.items()}
        coo.update({"n_user": n_user, "n_item": n_item, "interactions": int(u.numel()), "nnz": int(2 * u.numel())})
        slices.append(coo)
    return slices


def to_sparse_coo(coo: Dict[str, torch.Tensor], device=None) -> torch.Tensor:
    """The `lap_list` element type: an (uncoalesced-flagged) torch sparse COO fp32 [N, N] (matrix.py:79-83)."""
    N = coo["n_user"] + coo["n_item"]
    idx = torch.stack([coo["rows"], coo["cols"]])
    vals = coo["vals"]
    if device is not None:
        idx, vals = idx.to(device), vals.to(device)
    return torch.sparse_coo_tensor(idx, vals, (N, N))
