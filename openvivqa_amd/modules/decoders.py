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

    def forward(self, queries, keys, values, self_attention_mask, enc_attention_mask, encoder_group: int = 1, **kwargs):
        """``encoder_group`` (an addition, default 1): the encoder features / mask have one row per SAMPLE and serve
        ``encoder_group`` consecutive query rows (its beams); see MultiHeadAttention.forward."""
        x = self.self_attn(queries, queries, queries, attention_mask=self_attention_mask, **kwargs)
        x = self.enc_attn(x, keys, values, attention_mask=enc_attention_mask, encoder_group=encoder_group, **kwargs)
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
        self._mask_cache = self._mask_spare = self._enc_mask = None

    # ------------------------------------------------------------------ one-position decoding step
    def enable_statefulness(self, batch_size: int) -> None:
        super().enable_statefulness(batch_size)
        self._mask_cache = self._mask_spare = self._enc_mask = None

    def disable_statefulness(self) -> None:
        super().disable_statefulness()
        self._mask_cache = self._mask_spare = self._enc_mask = None

    def _seat_mask(self, R, dev):
        """The running self-attention mask as an in-place fp32 (R, max_len + 1) buffer; the state buffer is its live
        prefix viewed as (R, 1, 1, t) -- same name and shape as the reference's concatenated mask (decoders.py:55-57),
        so ``apply_to_states`` / ``reorder_states`` keep working (whatever they store is copied back in here)."""
        cur = self._buffers["running_mask_self_attention"]
        cache = getattr(self, "_mask_cache", None)
        if cur.numel() > 0 and cur.dim() != 4:
            raise RuntimeError("running_mask_self_attention must keep the reference's (rows, 1, 1, t) shape, got "
                               f"{tuple(cur.shape)}")
        n = cur.shape[-1] if cur.dim() == 4 else 0
        if (cache is not None and cache.shape[0] == R and n < cache.shape[1] and cur.dim() == 4 and cur.shape[0] == R
                and cur.dtype == torch.float32 and cur.data_ptr() == cache.data_ptr()):
            return cache, n
        cap = max(self.max_len + 1, 2 * (n + 1))
        new = torch.zeros(R, cap, dtype=torch.float32, device=dev)
        if n > 0:
            assert cur.shape[0] == R, "the state buffers' batch dimension must follow the tokens'"
            new[:, :n].copy_(cur.reshape(R, n))
        self._mask_cache, self._mask_spare = new, None
        return new, n

    def _cache_destination(self, name, rows):
        """reorder_states: the gathered mask rows go into the live prefix of the spare buffer (see MultiHeadAttention)."""
        cache = getattr(self, "_mask_cache", None)
        if cache is None or name != "running_mask_self_attention":
            return None
        s = self._buffers[name]
        if s.dim() != 4 or s.data_ptr() != cache.data_ptr():
            return None
        spare = getattr(self, "_mask_spare", None)
        if spare is None or spare.shape[0] != rows or spare.shape[1] != cache.shape[1]:
            spare = self._mask_spare = torch.zeros(rows, cache.shape[1], dtype=torch.float32, device=cache.device)
        return spare[:, None, None, :s.shape[-1]]

    def _caches_reordered(self):
        cur, spare = self._buffers.get("running_mask_self_attention"), getattr(self, "_mask_spare", None)
        if spare is not None and cur is not None and cur.dim() == 4 and cur.data_ptr() == spare.data_ptr():
            self._mask_cache, self._mask_spare = spare, self._mask_cache

    def _float_mask(self, mask, group=1):
        """The encoder attention mask in fp32 with one row per query row, converted / expanded to the ``group`` beams of
        each sample once per decode (the layers would each do it per step); keyed on the identity of the given tensor."""
        if mask is None or (mask.dtype == torch.float32 and (group == 1 or mask.shape[0] == 1)):
            return mask
        ident = (mask.data_ptr(), tuple(mask.shape), mask._version, mask.dtype, group)
        held = getattr(self, "_enc_mask", None)
        if held is None or held[0] != ident:
            m = mask.float()
            if group > 1 and m.shape[0] != 1:
                m = m.repeat_interleave(group, 0)
            held = self._enc_mask = (ident, m)
        return held[1]

    def _decode_step(self, tokens, encoder_features, encoder_attention_mask, return_logits, encoder_group=1):
        """Stateful step for one new position per row with the embedding sum, the position counter and the new mask column
        in ONE launch (ovqa_decode_embed) instead of ~12 index / elementwise launches."""
        from .. import ops
        from ..utils import MASK_VALUE
        arena = rt.ensure_arena(self.fc)
        T = arena.compute_dtype
        R, dev = tokens.shape[0], tokens.device
        cache, n = self._seat_mask(R, dev)
        seq = self._buffers["running_seq"]
        if not (seq.dtype == torch.int64 and seq.is_contiguous() and seq.numel() == R):
            seq = self._buffers["running_seq"] = seq.to(torch.int64).reshape(R, -1)[:, :1].contiguous()
        x32, x = ops.decode_embed(tokens.reshape(-1).contiguous(), self.word_emb.components.weight.detach(),
                                  self.pos_emb.weight.detach(), seq.view(-1), self.padding_idx, float(MASK_VALUE), cache,
                                  n, T)
        self_mask = self._buffers["running_mask_self_attention"] = cache[:, None, None, :n + 1]
        out = x32.view(R, 1, -1)
        if T == torch.bfloat16:
            out = Fn.attach_residual(x.view(R, 1, -1), out)
        enc_mask = self._float_mask(encoder_attention_mask, encoder_group)
        for layer in self.layers:
            out = layer(queries=out, keys=encoder_features, values=encoder_features,
                        self_attention_mask=self_mask, enc_attention_mask=enc_mask, encoder_group=encoder_group)
        logits = Fn.linear(out.to(T), self.fc, arena)
        return logits if return_logits else F.log_softmax(logits.float(), dim=-1)

    def forward(self, answer_tokens: torch.Tensor, encoder_features: torch.Tensor,
                encoder_attention_mask: torch.Tensor, return_logits: bool = False, encoder_group: int = 1):
        """``return_logits`` (an addition, default off): hand back the vocabulary logits instead of their log-softmax --
        the fused beam-search step takes the log-softmax inside its candidate kernel.  ``encoder_group`` (an addition,
        default 1 = the reference's contract): ``encoder_features`` / ``encoder_attention_mask`` carry one row per SAMPLE
        and each serves ``encoder_group`` consecutive token rows (the beams of that sample), instead of the per-beam
        copies beam_search.py:19-34,61 gathers -- the encoder K / V projections are then computed once per sample."""
        b_s, seq_len = answer_tokens.shape
        dev = answer_tokens.device
        if encoder_group > 1 and encoder_features.shape[0] * encoder_group != b_s:
            raise ValueError(f"encoder_group={encoder_group}: {encoder_features.shape[0]} encoder rows cannot serve "
                             f"{b_s} token rows")
        if (self._is_stateful and seq_len == 1 and answer_tokens.is_cuda and not torch.is_grad_enabled()
                and type(self.word_emb).__name__ == "UsualEmbedding" and self.d_model % 4 == 0
                and self.pos_emb.weight.dtype == torch.float32):
            return self._decode_step(answer_tokens, encoder_features, encoder_attention_mask, return_logits,
                                     encoder_group)
        if encoder_group > 1:  # the general path works on the reference's per-beam copies
            encoder_features = encoder_features.repeat_interleave(encoder_group, 0)
            if encoder_attention_mask is not None and encoder_attention_mask.shape[0] != 1:
                encoder_attention_mask = encoder_attention_mask.repeat_interleave(encoder_group, 0)
        pos = self.pos_emb.weight
        if (answer_tokens.is_cuda and not self._is_stateful and type(self.word_emb).__name__ == "UsualEmbedding"
                and self.d_model % 4 == 0 and pos.dtype == torch.float32 and seq_len + 1 <= pos.shape[0]):
            # decoders.py:50-60,66 in one launch: the padding + causal mask and the position rows added to the word rows
            embedded, _ = self.word_emb(answer_tokens)
            out, self_mask = Fn.decoder_inputs(embedded.float(), answer_tokens.contiguous(), pos, self.padding_idx)
        else:
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
        if not return_logits and out.is_cuda:
            # vocabulary projection + log_softmax on the library's kernels (decoders.py:75-76): the GEMM on the zero-padded
            # footprint of `fc`, ovqa_log_softmax_fwd / _bwd over the real words
            return Fn.classify_log_softmax(out.to(arena.compute_dtype), self.fc, arena)
        logits = Fn.linear(out.to(arena.compute_dtype), self.fc, arena)
        return logits if return_logits else F.log_softmax(logits.float(), dim=-1)
