"""Loss of the training step on the HIP path.

``NLLLoss`` is a drop-in for ``torch.nn.NLLLoss(ignore_index=...)`` as the reference's tasks build it
(tasks/classification_task.py:125-127, tasks/open_ended_task.py:155-157): mean over the rows whose target is not
``ignore_index``, one launch forward and one backward (``ovqa_nll_loss``), fixed summation order.
``nll_loss_fwd_bwd`` is the fused form for a harness that takes ``(outputs, gradients)`` from its loss function
(train.TrainStep): loss and d loss / d log-probabilities out of ONE launch.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from . import ops


class NLLLoss(nn.Module):
    def __init__(self, ignore_index: int = -100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if logp.dim() > 2:  # (B, T, V) log-probabilities of a decoder, (B, T) targets: rows = positions
            logp, target = logp.reshape(-1, logp.shape[-1]), target.reshape(-1)
        return Fn.nll_loss(logp, target, self.ignore_index)


def nll_loss_fwd_bwd(logp: torch.Tensor, target: torch.Tensor, loss: torch.Tensor, ignore_index: int = -100,
                     accumulate: bool = False) -> torch.Tensor:
    """``loss`` (fp32 device scalar) (=|+=) NLLLoss(logp, target); returns d loss / d logp (fp32, dense)."""
    lp = logp.detach()
    lp = lp if lp.dtype == torch.float32 and lp.is_contiguous() else lp.float().contiguous()
    return ops.nll_loss(lp.reshape(-1, lp.shape[-1]), target.reshape(-1).contiguous(), ignore_index, loss=loss,
                        want_grad=True, accumulate=accumulate).view_as(logp)
