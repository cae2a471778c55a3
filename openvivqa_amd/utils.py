"""Mask / position helpers with the reference's semantics (models/utils.py:32-73).

These are O(B*N) index manipulations executed with stock torch ops on whatever
device the inputs live on; they define what the attention kernels receive:
additive fp32 masks with value -1e5 (not -inf, not boolean).
"""
from __future__ import annotations

from typing import Optional

import torch

MASK_VALUE = -10e4  # == -100000.0, models/utils.py:56


def generate_padding_mask(sequences: Optional[torch.Tensor], padding_idx: int) -> Optional[torch.Tensor]:
    """(B,1,1,N) additive mask; a position is padding iff its feature sum equals
    padding_idx * D (token ids count as D == 1).  models/utils.py:44-57."""
    if sequences is None:
        return None
    s = sequences.unsqueeze(-1) if sequences.dim() == 2 else sequences
    is_pad = s.sum(dim=-1) == padding_idx * s.shape[-1]
    return (is_pad.long() * MASK_VALUE).unsqueeze(1).unsqueeze(1)


def generate_sequential_mask(seq_len: int, device=None) -> torch.Tensor:
    """(1,1,T,T) causal additive mask.  models/utils.py:59-66.  ``device`` (an addition): build it where it
    is used -- a host tensor + copy is not capturable into a hipGraph."""
    return (torch.triu(torch.ones(seq_len, seq_len, device=device), diagonal=1) * MASK_VALUE).unsqueeze(0).unsqueeze(0)


def generate_self_attention_masks(padding_masks: torch.Tensor, sequential_masks: torch.Tensor) -> torch.Tensor:
    """OR of the two masks -> (B,1,T,T).  models/utils.py:68-73."""
    return torch.logical_or(padding_masks != 0, sequential_masks != 0).long() * MASK_VALUE


def sinusoid_encoding_table(max_len: int, d_model: int, padding_idx: Optional[int] = None) -> torch.Tensor:
    """Decoder position table, rows 0..max_len-1.  models/utils.py:21-38."""
    pos = torch.arange(max_len, dtype=torch.float32).view(-1, 1)
    dim = torch.arange(d_model // 2, dtype=torch.float32).view(1, -1)
    ang = pos / 10000 ** (2 * dim / d_model)
    out = torch.zeros(max_len, d_model)
    out[:, ::2] = torch.sin(ang)
    out[:, 1::2] = torch.cos(ang)
    if padding_idx is not None:
        out[padding_idx] = 0
    return out


def box_relational_embedding(boxes: torch.Tensor, dim_g: int = 64, wave_len: float = 1000.0,
                             trignometric_embedding: bool = True) -> torch.Tensor:
    """Pairwise box geometry (B, N, N, dim_g) of the geometry-aware attention.  models/utils.py:102-162: log-scaled
    centre offsets (clamped at 1e-3) and log size ratios of boxes (x_min, y_min, x_max, y_max), optionally expanded
    to sin/cos features over dim_g/8 wavelengths (positions scaled by 100).  O(B*N*N) elementwise work."""
    B = boxes.size(0)
    x_min, y_min, x_max, y_max = torch.chunk(boxes, 4, dim=-1)
    cx, cy = (x_min + x_max) * 0.5, (y_min + y_max) * 0.5
    w, h = (x_max - x_min) + 1.0, (y_max - y_min) + 1.0
    dx = torch.log(torch.clamp(torch.abs((cx - cx.view(B, 1, -1)) / w), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy - cy.view(B, 1, -1)) / h), min=1e-3))
    dw = torch.log(w / w.view(B, 1, -1))
    dh = torch.log(h / h.view(B, 1, -1))
    pos = torch.stack((dx, dy, dw, dh), dim=-1)  # (B, N, N, 4)
    if not trignometric_embedding:
        return pos
    feat = torch.arange(dim_g / 8, device=boxes.device)
    dim_mat = 1.0 / torch.pow(torch.tensor(float(wave_len), device=boxes.device), feat / (dim_g / 8))
    mul = (100.0 * pos).unsqueeze(-1) * dim_mat.view(1, 1, 1, 1, -1)
    mul = mul.reshape(B, pos.size(1), pos.size(2), -1)
    return torch.cat((torch.sin(mul), torch.cos(mul)), dim=-1)
