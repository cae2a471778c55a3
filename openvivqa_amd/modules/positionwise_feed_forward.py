"""Position-wise feed-forward block (drop-in for models/modules/positionwise_feed_forward.py).

LN(x + drop2(fc2(drop1(gelu(fc1 x))))) as two GEMM launches with fused
epilogues (bias+GELU+dropout, bias+dropout+residual) and one LayerNorm launch.
"""
from __future__ import annotations

from torch import nn

from .. import functional as Fn
from .. import runtime as rt


class PositionWiseFeedForward(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.fc1 = nn.Linear(config.D_MODEL, config.D_FF)
        self.fc2 = nn.Linear(config.D_FF, config.D_MODEL)
        self.dropout_1 = nn.Dropout(p=config.DROPOUT)
        self.dropout_2 = nn.Dropout(p=config.DROPOUT)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)
        self._site1, self._site2 = rt.new_dropout_site(), rt.new_dropout_site()

    def forward(self, input):
        arena = rt.ensure_arena(self)
        x = Fn.to_compute(input, arena.compute_dtype)
        st = dict(arena=arena, mod=self, params=list(self.parameters()),
                  drop1=rt.dropout_spec(self.dropout_1.p, self._site1, self.training, x.device),
                  drop2=rt.dropout_spec(self.dropout_2.p, self._site2, self.training, x.device))
        return Fn.ffn_block(x, st)
