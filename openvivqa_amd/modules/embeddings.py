"""Embedding steps that feed the hot path ("next" row 2 of SURVEY 8f).

FeatureEmbedding   models/modules/vision_embeddings.py:10-25  (Linear+GELU+dropout, zero-row padding mask)
UsualEmbedding     models/modules/text_embeddings.py:56-80    (token lookup + masks, no pretrained vectors)
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .. import runtime as rt
from .._lib import EPI_BIAS_GELU
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
        """Inference/feature-extraction form: GEMM with fused bias+GELU+dropout epilogue.
        (No autograd through this module yet: it sits before the hot path.)"""
        masks = generate_padding_mask(features, padding_idx=0).to(features.device)
        arena = rt.ensure_arena(self)
        x = features.to(arena.compute_dtype).contiguous()
        drop = rt.dropout_spec(self.dropout.p, self._site, self.training, x.device)
        with torch.no_grad():
            y = ops.linear_fwd(x, arena.compute(self.proj.weight), arena.master_of(self.proj.bias), EPI_BIAS_GELU,
                               drop=drop)
        return y, masks


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
        sequential_masks = generate_sequential_mask(tokens.shape[-1]).to(tokens.device)
        return self.components(tokens), (padding_masks, sequential_masks)
