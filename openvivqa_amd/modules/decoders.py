"""Transformer decoder (drop-in for models/modules/decoders.py:13-76).

Teacher-forced and stateful single-step decoding share the fused blocks; the
token + position embedding lookups and the final log-softmax are index / small
elementwise work left to stock torch ops, the vocabulary projection is the HIP
GEMM.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F

from .. import functional as Fn
from .. import runtime as rt
from ..builders.decoder_builder import META_DECODER
from ..builders.text_embedding_builder import build_text_embedding
from ..utils import (generate_padding_mask, generate_self_attention_masks, generate_sequential_mask,
                     sinusoid_encoding_table)
from .attentions import MultiHeadAttention
from .containers import Module, ModuleList
from .positionwise_feed_forward import PositionWiseFeedForward


class DecoderLayer(Module):
    """self-attention -> encoder attention -> FFN (built from ENC_ATTENTION, decoders.py:19)."""

    def __init__(self, config):
        super().__init__()
        self.self_attn = MultiHeadAttention(config.SELF_ATTENTION)
        self.enc_attn = MultiHeadAttention(config.ENC_ATTENTION)
        self.pwff = PositionWiseFeedForward(config.ENC_ATTENTION)

    def forward(self, queries, keys, values, self_attention_mask, enc_attention_mask, **kwargs):
        x = self.self_attn(queries, queries, queries, attention_mask=self_attention_mask, **kwargs)
        x = self.enc_attn(x, keys, values, attention_mask=enc_attention_mask, **kwargs)
        return self.pwff(x)


@META_DECODER.register()
class Decoder(Module):
    """N-layer decoder with causal + padding masking and running state for beam search."""

    def __init__(self, config, vocab):
        super().__init__()
        self.d_model = config.D_MODEL
        self.max_len = vocab.max_answer_length
        self.padding_idx = vocab.padding_idx
        self.N = config.LAYERS
        self.word_emb = build_text_embedding(config.TEXT_EMBEDDING, vocab)
        self.pos_emb = nn.Embedding.from_pretrained(
            sinusoid_encoding_table(max_len=self.max_len + 1, d_model=config.D_MODEL, padding_idx=0), freeze=True)
        self.layers = ModuleList([DecoderLayer(config.ATTENTION) for _ in range(config.LAYERS)])
        for layer in self.layers:  # in-place K / V caches of the decoding steps: sized once for the longest answer
            layer.self_attn.decode_capacity = self.max_len + 1
        self.fc = nn.Linear(config.D_MODEL, len(vocab), bias=False)
        self.register_state("running_mask_self_attention", torch.zeros((1, 1, 0)).bool())
        self.register_state("running_seq", torch.zeros((1,)).long())

    def forward(self, answer_tokens: torch.Tensor, encoder_features: torch.Tensor,
                encoder_attention_mask: torch.Tensor):
        b_s, seq_len = answer_tokens.shape
        dev = answer_tokens.device
        pad_mask = generate_padding_mask(answer_tokens, self.padding_idx).to(dev)
        self_mask = generate_self_attention_masks(pad_mask, generate_sequential_mask(seq_len, device=dev))
        if self._is_stateful:  # decoders.py:55-57
            self.running_mask_self_attention = torch.cat([self.running_mask_self_attention, self_mask], -1)
            self_mask = self.running_mask_self_attention
        seq = torch.arange(1, seq_len + 1, device=dev).view(1, -1).expand(b_s, -1)
        seq = seq.masked_fill(pad_mask.squeeze(1).squeeze(1) != 0, 0)
        if self._is_stateful:  # decoders.py:61-63
            self.running_seq.add_(1)
            seq = self.running_seq
        embedded, _ = self.word_emb(answer_tokens)
        out = embedded + self.pos_emb(seq)
        for layer in self.layers:
            out = layer(queries=out, keys=encoder_features, values=encoder_features,
                        self_attention_mask=self_mask, enc_attention_mask=encoder_attention_mask)
        arena = rt.ensure_arena(self.fc)
        logits = Fn.linear(out.to(arena.compute_dtype), self.fc, arena)
        return F.log_softmax(logits.float(), dim=-1)
