"""Embedding steps that feed the hot path ("next" row 2 of SURVEY 8f).

FeatureEmbedding   models/modules/vision_embeddings.py:10-25  (Linear+GELU+dropout, zero-row padding mask)
UsualEmbedding     models/modules/text_embeddings.py:56-80    (token lookup + masks, no pretrained vectors)
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from .. import runtime as rt
from ..builders.text_embedding_builder import META_TEXT_EMBEDDING
from ..builders.vision_embedding_builder import META_VISION_EMBEDDING
from ..utils import generate_padding_mask, generate_sequential_mask


@META_VISION_EMBEDDING.register()
class FeatureEmbedding(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.proj = nn.Linear(config.D_FEATURE, config.D_MODEL)
        self.gelu = nn.GELU()
        self.dropout = nn.Dropout(config.DROPOUT)
        self._site = rt.new_dropout_site()

    def forward(self, features):
        """One pass over the features for the zero-row padding mask (models/utils.py:44-58), one GEMM with the
        fused bias+GELU+dropout epilogue; differentiable (weights, bias and, if required, the features)."""
        arena = rt.ensure_arena(self)
        if features.dim() == 3:
            masks = ops.row_padding_mask(features.contiguous(), 0.0)
        else:  # other ranks: the generic torch helper
            masks = generate_padding_mask(features, padding_idx=0).to(features.device)
        x = features.to(arena.compute_dtype).contiguous()
        drop = rt.dropout_spec(self.dropout.p, self._site, self.training, x.device)
        return Fn.linear_gelu_dropout(x, self.proj, arena, drop), masks


@META_TEXT_EMBEDDING.register()
class UsualEmbedding(nn.Module):
    def __init__(self, config, vocab):
        super().__init__()
        self.padding_idx = vocab.padding_idx
        if config.WORD_EMBEDDING is not None:
            raise NotImplementedError("pretrained word vectors are a data-loading feature outside the hot path")
        self.components = nn.Embedding(len(vocab), config.D_MODEL, vocab.padding_idx)

    def forward(self, tokens):
        padding_masks = generate_padding_mask(tokens, padding_idx=self.padding_idx).to(tokens.device)
        sequential_masks = generate_sequential_mask(tokens.shape[-1], device=tokens.device)
        return self.components(tokens), (padding_masks, sequential_masks)


@META_TEXT_EMBEDDING.register()
class LSTMTextEmbedding(nn.Module):
    """Embedding -> Linear -> dropout -> LSTM (text_embeddings.py:222-246).  It sits in front of the hot path:
    the projection goes through the HIP GEMM, the recurrent part is torch's LSTM (MIOpen on ROCm) -- plumbing,
    not a kernel of this package."""

    def __init__(self, config, vocab):
        super().__init__()
        self.embedding = nn.Embedding(len(vocab), config.D_EMBEDDING, padding_idx=vocab.padding_idx)
        self.padding_idx = vocab.padding_idx
        if config.WORD_EMBEDDING is not None:
            raise NotImplementedError("pretrained word vectors are a data-loading feature outside the hot path")
        self.proj = nn.Linear(config.D_EMBEDDING, config.D_MODEL)
        self.dropout = nn.Dropout(config.DROPOUT)
        self.lstm = nn.LSTM(input_size=config.D_MODEL, hidden_size=config.D_MODEL, batch_first=True)

    def forward(self, tokens):
        padding_masks = generate_padding_mask(tokens, padding_idx=self.padding_idx).to(tokens.device)
        sequential_masks = generate_sequential_mask(tokens.shape[-1], device=tokens.device)
        arena = rt.ensure_arena(self)
        x = self.embedding(tokens).to(arena.compute_dtype)
        x = self.dropout(Fn.linear(x, self.proj, arena)).float()
        if not self.training and torch.is_grad_enabled() and x.is_cuda:
            # MIOpen's fused RNN refuses backward in eval mode; the native kernels do not
            with torch.backends.cudnn.flags(enabled=False):
                x, _ = self.lstm(x)
        else:
            x, _ = self.lstm(x)
        return x, (padding_masks, sequential_masks)
