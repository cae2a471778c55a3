"""Pointer-network scorers of the M4C family (HIP-backed).

OcrPtrNet              models/mmf_m4c.py:367-396   additive mask
DynamicPointerNetwork  models/m4c.py:19-33         boolean -inf fill on the key axis
                       models/iterative_m4c.py:18-32   ... on the query axis
Two projection GEMMs feed one batched score kernel (q.k^T * scale + mask / fill).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import functional as Fn
from .. import runtime as rt


class OcrPtrNet(nn.Module):
    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        if query_key_size is None:
            query_key_size = hidden_size
        self.hidden_size = hidden_size
        self.query_key_size = query_key_size
        self.query = nn.Linear(hidden_size, query_key_size)
        self.key = nn.Linear(hidden_size, query_key_size)

    def forward(self, query_inputs, key_inputs, attention_mask):
        arena = rt.ensure_arena(self)
        T = arena.compute_dtype
        squeeze = query_inputs.dim() == 2
        if squeeze:
            query_inputs = query_inputs.unsqueeze(1)
        q = Fn.linear(query_inputs.to(T), self.query, arena)
        k = Fn.linear(key_inputs.to(T), self.key, arena)
        B, nk = k.shape[0], k.shape[1]
        add_mask = attention_mask.reshape(B, nk).float().contiguous()
        s = Fn.pointer_score(q, k, 1.0 / math.sqrt(self.query_key_size), add_mask=add_mask)
        return s.squeeze(1) if squeeze else s


class DynamicPointerNetwork(nn.Module):
    """``axis='key'``: m4c.py variant (mask (B,1,1,Nk) bool); ``axis='query'``: iterative_m4c.py variant
    (mask (B,1,1,T) bool, rows of masked queries become -inf)."""

    def __init__(self, config, axis: str = "key"):
        super().__init__()
        self.query = nn.Linear(config.D_MODEL, config.D_MODEL)
        self.key = nn.Linear(config.D_MODEL, config.D_MODEL)
        self.d_model = config.D_MODEL
        assert axis in ("key", "query")
        self.axis = axis

    def forward(self, query_inputs, key_inputs, attention_mask):
        arena = rt.ensure_arena(self)
        T = arena.compute_dtype
        q = Fn.linear(query_inputs.to(T), self.query, arena)
        k = Fn.linear(key_inputs.to(T), self.key, arena)
        B = q.shape[0]
        fill = attention_mask.reshape(B, -1).to(torch.uint8).contiguous()
        kw = {"key_fill": fill} if self.axis == "key" else {"query_fill": fill}
        return Fn.pointer_score(q, k, 1.0 / math.sqrt(self.d_model), **kw)
