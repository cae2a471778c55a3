"""Sinusoidal position table of the encoder prologue.

Reference: models/modules/pos_embeddings.py:39-72 recomputes cumsum/pow/sin/cos
on every forward; the values depend only on (N, D), so they are computed once
(fp32, on the host, same op order as the reference) and kept resident in HBM;
the LayerNorm kernel adds the table while it writes its output.
"""
from __future__ import annotations

import math

import torch
from torch import nn

_cache = {}


def sinusoid_table(n: int, d: int, temperature: float = 10000.0, normalize: bool = False, scale: float = 2 * math.pi,
                   device=None) -> torch.Tensor:
    key = (n, d, temperature, normalize, scale, str(device))
    t = _cache.get(key)
    if t is None:
        embed = torch.ones(1, n, dtype=torch.float32).cumsum(1)  # positions 1..n
        if normalize:
            embed = embed / (embed[:, -1:] + 1e-6) * scale
        dim_t = torch.arange(d, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / d)
        pos = embed[:, :, None] / dim_t
        t = torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=-1).flatten(-2)[0].contiguous()
        if device is not None:
            t = t.to(device)
        _cache[key] = t
    return t


class SinusoidPositionalEmbedding(nn.Module):
    """Same constructor/forward as the reference class; returns (B, N, D) fp32."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale

    def table(self, n: int, device) -> torch.Tensor:
        return sinusoid_table(n, self.num_pos_feats, float(self.temperature), self.normalize, self.scale, device)

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("padding-aware positions are not used on the hot path (encoders.py:113)")
        return self.table(x.shape[1], x.device).unsqueeze(0).expand(x.shape[0], -1, -1)
