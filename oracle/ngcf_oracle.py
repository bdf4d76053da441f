"""CPU oracle for the NGCF embedding-propagation hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, op for op, the algorithm of the reference
(`haesungpyun/seoul_tourism_recommendation_NGCF`) for the one hot path this repository
accelerates.  It is the *checker*: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product package
(`seoul_tourism_recommendation_ngcf_amd/`) never imports anything under `oracle/`.

Parity pin: the functions below are checked bit-for-bit (torch path) against the imported
reference in `oracle/make_golden.py` (run in the build container, where `/root/reference`
exists) and against the committed fixtures `tests/golden/*.npz` everywhere else.  The
reference itself has no tests, so these fixtures are the pin.

Two restatements live here:

* ``*_torch``  — the same torch CPU ops the reference issues, in the same order
  (`torch.mm(sparse_coo, dense)`, three `F.linear`, `leaky_relu`, `dropout`, `F.normalize`,
  `cat`, index gathers).  Bit-exact against the reference on CPU.  This is also what
  `bench.py` times as "the reference PyTorch CPU path" (`cpu_baseline.kind == "port"`).
* ``*_f64``    — a numpy float64 restatement of the same mathematics, the accuracy yardstick
  ("HIP error vs fp64 <= 4x the reference's own fp32 error vs fp64").

Reference citations are `file:line` into `/root/reference/model/`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.2          # NGCF.py:140
NORM_EPS = 1e-12           # F.normalize default, NGCF.py:144
FEATURE_ORDER = ("age", "sex", "month", "day", "dow")   # concat order, NGCF.py:110


# --------------------------------------------------------------------------------------
# feature injection                                                      NGCF.py:103-115
# --------------------------------------------------------------------------------------
def feature_inject_torch(user_w: torch.Tensor, feat_tables: dict, u_id, age, sex, month, day, dow,
                         emb_ratio: float) -> torch.Tensor:
    """In-place `user_w[u_id] = user_w[u_id]*(1-r) + cat(feats)*r` (NGCF.py:103-115).

    `feat_tables` maps "age"/"sex"/"month"/"day"/"dow" to their `[card, d0//5]` tables.
    The RHS is evaluated from the pre-update rows; duplicate ids resolve as the CPU
    `index_put_` does (last writer).  Raises RuntimeError when 5*(d0//5) != d0, exactly as
    the reference's shape mismatch does.
    """
    feats = torch.cat((feat_tables["age"][age], feat_tables["sex"][sex], feat_tables["month"][month],
                       feat_tables["day"][day], feat_tables["dow"][dow]), dim=1)
    user_w[u_id] = user_w[u_id] * (1 - emb_ratio) + feats.detach().clone() * emb_ratio
    return user_w


def select_year_index(year: torch.Tensor) -> int:
    """`year.unique()[0] % 18` (NGCF.py:117): the smallest year in the batch, 18->0, 19->1."""
    return int(year.unique()[0] % 18)


# --------------------------------------------------------------------------------------
# node dropout                                                           NGCF.py:93-100
# --------------------------------------------------------------------------------------
def sparse_dropout_torch(L: torch.Tensor, p: float) -> torch.Tensor:
    """Keep each stored nonzero w.p. 1-p, values NOT rescaled (NGCF.py:93-100).

    The mask is `nn.Dropout(p)(float64 ones[nnz]) != 0` drawn from the CPU default
    generator; the module is freshly built so it is always in training mode.
    """
    mask = F.dropout(torch.tensor(np.ones(L._nnz())), p=p, training=True).type(torch.bool)
    i = L._indices()[:, mask]
    v = L._values()[mask]
    return torch.sparse_coo_tensor(i, v, L.shape)


# --------------------------------------------------------------------------------------
# propagation                                                            NGCF.py:120-147
# --------------------------------------------------------------------------------------
def propagate_torch(L: torch.Tensor, user_w: torch.Tensor, item_w: torch.Tensor, w1, b1, w2, b2,
                    mess_dropout=None, training: bool = False, node_dropout=None,
                    node_flag: bool = False, return_carry: bool = False):
    """`all_E = cat([E0, norm(E1), ..., norm(En)], 1)` (NGCF.py:120-147), same torch CPU ops.

    `w1[k]`/`w2[k]` are `[d_{k+1}, d_k]` Linear weights, `b1[k]`/`b2[k]` their biases.
    `W1`'s bias is added twice (NGCF.py:131,133).  The un-normalised `E` carries to the next
    layer (NGCF.py:140-142); the normalised copy is what is concatenated (NGCF.py:144-146).
    """
    E = torch.cat((user_w, item_w), dim=0)                      # NGCF.py:120
    all_E = [E]
    carries = []
    for k in range(len(w1)):
        if node_flag:
            L = sparse_dropout_torch(L, node_dropout)           # NGCF.py:126 (cumulative)
        L_E = torch.mm(L, E)                                    # NGCF.py:130
        L_E_W1 = F.linear(L_E, w1[k], b1[k])                    # NGCF.py:131
        E_W1 = F.linear(E, w1[k], b1[k])                        # NGCF.py:133
        L_E_E = L_E * E                                         # NGCF.py:135
        L_E_E_W2 = F.linear(L_E_E, w2[k], b2[k])                # NGCF.py:136
        message = L_E_W1 + E_W1 + L_E_E_W2                      # NGCF.py:138
        E = F.leaky_relu(message, negative_slope=LEAKY_SLOPE)   # NGCF.py:140
        if mess_dropout is not None:
            E = F.dropout(E, p=mess_dropout[k], training=training)   # NGCF.py:142
        carries.append(E)
        all_E.append(F.normalize(E, p=2, dim=1))                # NGCF.py:144-146
    out = torch.cat(all_E, dim=1)                               # NGCF.py:147
    if return_carry:
        return out, carries
    return out


def gather_torch(all_E: torch.Tensor, n_user: int, u_id, pos_item, neg_item):
    """Row gathers of NGCF.py:148-156 (bit-exact copies; empty `neg_item` -> `torch.empty(0)`)."""
    users, items = all_E[:n_user, :], all_E[n_user:, :]
    u = users[u_id, :]
    p = items[pos_item, :]
    n = torch.empty(0)
    if len(neg_item) > 0:
        n = items[neg_item, :]
    return u, p, n


# --------------------------------------------------------------------------------------
# BPR                                                                    bprloss.py:15-22
# --------------------------------------------------------------------------------------
def bpr_torch(u, p, n, weight_decay: float, batch_size: int) -> torch.Tensor:
    """`(-sum logsigmoid(|u.p| - |u.n|) + wd*(sum|u|^2 + sum|p|^2 + sum|n|^2)) / batch_size`.

    Note the `abs` (bprloss.py:18), the ctor-constant divisor (bprloss.py:22) and that the
    three squared norms run over each tensor's OWN rows (a `[1,D]` broadcast `p` counts once).
    """
    x_upos = torch.mul(u, p).sum(dim=1)
    x_uneg = torch.mul(u, n).sum(dim=1)
    x_upn = torch.abs(x_upos) - torch.abs(x_uneg)
    log_prob = F.logsigmoid(x_upn).sum()
    reg = weight_decay * (torch.linalg.norm(u, dim=1).pow(2).sum()
                          + torch.linalg.norm(p, dim=1).pow(2).sum()
                          + torch.linalg.norm(n, dim=1).pow(2).sum())
    return (-log_prob + reg) / batch_size


# --------------------------------------------------------------------------------------
# float64 yardstick (numpy)
# --------------------------------------------------------------------------------------
def spmm_coo_f64(rows, cols, vals, n_rows: int, E: np.ndarray) -> np.ndarray:
    """`L.E` in float64 from COO triplets (NGCF.py:130), duplicates summed."""
    out = np.zeros((n_rows, E.shape[1]), dtype=np.float64)
    np.add.at(out, np.asarray(rows), np.asarray(vals, dtype=np.float64)[:, None] * E[np.asarray(cols)].astype(np.float64))
    return out


def propagate_f64(rows, cols, vals, E0: np.ndarray, w1, b1, w2, b2) -> np.ndarray:
    """float64 restatement of NGCF.py:120-147 (eval mode, no dropout)."""
    import scipy.sparse as sp
    N = E0.shape[0]
    L = sp.coo_matrix((np.asarray(vals, dtype=np.float64), (np.asarray(rows), np.asarray(cols))), shape=(N, N)).tocsr()
    E = E0.astype(np.float64)
    blocks = [E]
    for k in range(len(w1)):
        W1, W2 = np.asarray(w1[k], np.float64), np.asarray(w2[k], np.float64)
        B1, B2 = np.asarray(b1[k], np.float64), np.asarray(b2[k], np.float64)
        LE = L @ E
        M = LE @ W1.T + B1 + E @ W1.T + B1 + (LE * E) @ W2.T + B2
        E = np.where(M >= 0, M, LEAKY_SLOPE * M)
        nrm = np.maximum(np.sqrt((E * E).sum(1, keepdims=True)), NORM_EPS)
        blocks.append(E / nrm)
    return np.concatenate(blocks, axis=1)


def bpr_f64(u, p, n, weight_decay: float, batch_size: int) -> float:
    u, p, n = (np.asarray(t, np.float64) for t in (u, p, n))
    x = np.abs((u * p).sum(1)) - np.abs((u * n).sum(1))
    logsig = np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))
    reg = weight_decay * ((u * u).sum() + (p * p).sum() + (n * n).sum())
    return float((-logsig.sum() + reg) / batch_size)


# --------------------------------------------------------------------------------------
# Laplacian builder                                                      matrix.py:41-83
# --------------------------------------------------------------------------------------
def build_laplacian_list(year, userid, itemid, rating, n_user: int, n_item: int):
    """Sparse restatement of `Matrix.create_matrix` (matrix.py:41-76) -> list of COO triplets.

    Pinned quirks of the reference: `R` is never cleared between years (matrix.py:33,45) so a
    later slice also contains the earlier years' edges; the degree is the COUNT of stored
    non-zeros per row (matrix.py:55) while the values keep their weights; explicit-zero
    ratings create no edge; `d^-1/2` is float32 (matrix.py:56), the product float64
    (matrix.py:58-62), cast to float32 at the end (matrix.py:82); entries come out row-major
    sorted.  Returns `{year_idx: (rows int64, cols int64, vals float32)}` with
    `year_idx = year % 18` (matrix.py:66).
    """
    year, userid, itemid = (np.asarray(a) for a in (year, userid, itemid))
    rating = np.asarray(rating, dtype=np.float32)
    R = {}                                          # dok semantics: last assignment wins
    out = {}
    N = n_user + n_item
    seen_years = []
    for y in year.tolist():
        if y not in seen_years:
            seen_years.append(y)                    # pandas .unique(): order of appearance
    for y in seen_years:
        sel = np.nonzero(year == y)[0]
        for j in sel:
            key = (int(userid[j]), int(itemid[j]))
            if rating[j] != 0:
                R[key] = rating[j]
            else:
                R.pop(key, None)                    # dok assignment of 0 deletes the entry
        us = np.fromiter((k[0] for k in R), dtype=np.int64, count=len(R))
        its = np.fromiter((k[1] for k in R), dtype=np.int64, count=len(R))
        w = np.fromiter(R.values(), dtype=np.float32, count=len(R))
        rows = np.concatenate([us, its + n_user])
        cols = np.concatenate([its + n_user, us])
        vals = np.concatenate([w, w])
        deg = np.bincount(rows, minlength=N)
        with np.errstate(divide="ignore"):
            d_sqrt = np.power(deg.astype(np.float64)[:, None], -0.5, dtype=np.float32).squeeze()
        d_sqrt[np.isinf(d_sqrt)] = 0.0
        v = (d_sqrt[rows].astype(np.float64) * vals.astype(np.float64)) * d_sqrt[cols].astype(np.float64)
        order = np.lexsort((cols, rows))
        keep = v[order] != 0
        out[int(y) % 18] = (rows[order][keep], cols[order][keep], v[order][keep].astype(np.float32))
    return out
