"""`BPR` - the reference's loss module (`/root/reference/model/bprloss.py:9-22`) on the fused HIP kernel.

Same constructor and `forward(u, pos, neg)` -> 0-dim tensor.  Kept quirks: the `abs` around both
scores (bprloss.py:18), the divisor is the constructor's `batch_size`, not the actual batch
(bprloss.py:22), and a `[1, D]` positive row broadcasts against `[B, D]` users
(experiment.py:96-100) while its squared norm is counted once.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as _eng
from .autograd import BPRLoss


class BPR(nn.Module):
    def __init__(self, weight_decay, batch_size):
        super().__init__()
        self.weight_decay = weight_decay
        self.batch_size = batch_size
        self._ws = _eng.Workspace()

    def forward(self, u_idx, pos_idx, neg_idx):
        if torch.is_grad_enabled() and (u_idx.requires_grad or pos_idx.requires_grad or neg_idx.requires_grad):
            return BPRLoss.apply(u_idx, pos_idx, neg_idx, self.weight_decay, self.batch_size, self._ws)
        return _eng.bpr_loss(u_idx.detach(), pos_idx.detach(), neg_idx.detach(), self.weight_decay,
                             self.batch_size, self._ws)
