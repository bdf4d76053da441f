"""Forward drivers of the propagation engine (and the autograd seam for the backward row).

`propagate_forward` issues the HIP layer kernels for NGCF.py:120-147.  Round 1 ships the forward
path only: outputs carry no grad_fn (the backward pass, is the next row),
which is what `Experiment.eval` (experiment.py:66-119) and demo.py need.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import engine as _eng


def propagate_forward(owner, csrs: Sequence["_eng.LaplacianCSR"], user_w: torch.Tensor, item_w: torch.Tensor,
                      w1, b1, w2, b2, drop: Sequence[float], seeds: Sequence[int],
                      keep_carries: bool = False):
    """all_E [N, D] = [E0 | norm(E1) | ... | norm(En)] (NGCF.py:120-147).

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
    D = sum(widths)
    all_E = torch.empty((N, D), dtype=torch.float32, device=dev)
    _eng.copy_rows(user_w.detach(), all_E[:U, :d0])
    _eng.copy_rows(item_w.detach(), all_E[U:, :d0])
    prev = all_E[:, :d0]
    off = d0
    carries: List[Optional[torch.Tensor]] = []
    for k in range(n_layer):
        d_out = widths[k + 1]
        last = k == n_layer - 1
        carry = None
        if not last or keep_carries:
            if keep_carries:
                carry = torch.empty((N, d_out), dtype=torch.float32, device=dev)
            else:
                buf = owner._carry[k % 2]
                if buf is None or buf.device != dev or tuple(buf.shape) != (N, d_out):
                    buf = torch.empty((N, d_out), dtype=torch.float32, device=dev)
                    owner._carry[k % 2] = buf
                carry = buf
        _eng.layer_fused(csrs[k], prev, prev, w1[k].detach(), b1[k].detach(), w2[k].detach(), b2[k].detach(),
                         carry, all_E[:, off:off + d_out], owner._ws, drop[k], seeds[k])
        carries.append(carry)
        prev = carry
        off += d_out
    return (all_E, carries) if keep_carries else all_E


def propagate_with_grad(owner, csrs, user_w, item_w, w1, b1, w2, b2, drop, seeds) -> torch.Tensor:
    with torch.no_grad():
        return propagate_forward(owner, csrs, user_w, item_w, w1, b1, w2, b2, drop, seeds)


def gather_with_grad(table: torch.Tensor, idx: torch.Tensor, status: torch.Tensor) -> torch.Tensor:
    with torch.no_grad():
        return _eng.gather_rows(table, idx, status)
