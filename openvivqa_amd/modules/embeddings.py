"""Embedding steps that feed the hot path ("next" row 2 of SURVEY 8f).

FeatureEmbedding   models/modules/vision_embeddings.py:10-25  (Linear+GELU+dropout, zero-row padding mask)
UsualEmbedding     models/modules/text_embeddings.py:56-80    (token lookup + masks, no pretrained vectors)
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from .. import runtime as rt
from ..builders.text_embedding_builder import META_TEXT_EMBEDDING
from ..builders.vision_embedding_builder import META_VISION_EMBEDDING
from ..utils import generate_padding_mask, generate_sequential_mask


_seq_masks = {}


def _sequential_mask(n, device):
    """generate_sequential_mask (models/utils.py:60-66) is a constant of the sequence length: built once per device."""
    key = (int(n), str(device))
    m = _seq_masks.get(key)
    if m is None:
        m = generate_sequential_mask(n, device=device)
        _seq_masks[key] = m
    return m


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
        sequential_masks = _sequential_mask(tokens.shape[-1], tokens.device)
        if not tokens.is_cuda or tokens.dim() != 2:  # (shapes the gather kernel does not take: the stock lookup)
            padding_masks = generate_padding_mask(tokens, padding_idx=self.padding_idx).to(tokens.device)
            return self.components(tokens), (padding_masks, sequential_masks)
        arena = rt.ensure_arena(self)
        B, T = tokens.shape
        # fp32 rows out of the fp32 master table (what the stock lookup returns), padding mask from the same pass
        rows, padding_masks = Fn.embed_rows(tokens, self.components, arena, master=True, want_mask=True)
        D = self.components.weight.shape[1]
        rows = rows if rows.shape[1] == D else rows[:, :D].contiguous()
        return rows.view(B, T, D), (padding_masks, sequential_masks)


class LSTM(nn.Module):
    """Parameter holder of a one-layer LSTM under torch's names and initialisation (``weight_ih_l0`` [4H, I],
    ``weight_hh_l0`` [4H, H], ``bias_ih_l0`` / ``bias_hh_l0`` [4H], gate order i, f, g, o, all U(-1/sqrt(H), 1/sqrt(H)) drawn
    in that order), so ``lstm.*`` state_dict keys interchange with the reference's ``nn.LSTM``.  The recurrence itself is
    ``functional.lstm`` (ovqa_lstm_fwd / ovqa_lstm_bwd)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih_l0 = nn.Parameter(torch.empty(4 * hidden_size, input_size))
        self.weight_hh_l0 = nn.Parameter(torch.empty(4 * hidden_size, hidden_size))
        self.bias_ih_l0 = nn.Parameter(torch.empty(4 * hidden_size))
        self.bias_hh_l0 = nn.Parameter(torch.empty(4 * hidden_size))
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


@META_TEXT_EMBEDDING.register()
class LSTMTextEmbedding(nn.Module):
    """Embedding -> Linear -> dropout -> LSTM (text_embeddings.py:221-246), all on the HIP path: the token rows are
    laid out TIME-major (row t*B + b) so that every product of the recurrence's backward pass -- dx, dW_ih, dW_hh -- is a
    plain row-major GEMM over all time steps; the recurrence is one persistent launch each way (csrc/lstm.hip)."""

    def __init__(self, config, vocab):
        super().__init__()
        self.embedding = nn.Embedding(len(vocab), config.D_EMBEDDING, padding_idx=vocab.padding_idx)
        self.padding_idx = vocab.padding_idx
        if config.WORD_EMBEDDING is not None:
            raise NotImplementedError("pretrained word vectors are a data-loading feature outside the hot path")
        self.proj = nn.Linear(config.D_EMBEDDING, config.D_MODEL)
        self.dropout = nn.Dropout(config.DROPOUT)
        self.lstm = LSTM(config.D_MODEL, config.D_MODEL)
        self._site = rt.new_dropout_site()

    def forward(self, tokens):
        arena = rt.ensure_arena(self)
        B, T = tokens.shape
        sequential_masks = _sequential_mask(T, tokens.device)
        # token rows (time-major, zero-padded to the projection's 16-byte row width) and the padding mask in one pass
        x, padding_masks = Fn.embed_rows(tokens, self.embedding, arena, time_major=True, want_mask=True)
        x = Fn.linear(x, self.proj, arena)                                                  # [T*B, D]
        x = Fn.dropout(x, rt.dropout_spec(self.dropout.p, self._site, self.training, x.device))
        y = Fn.lstm(x, self.lstm, arena, B, T)                                              # fp32 [B, T, D]
        return y, (padding_masks, sequential_masks)
