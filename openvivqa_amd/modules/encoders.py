"""Encoder stacks (drop-in for models/modules/encoders.py): same registered names,
constructor config keys, forward kwargs and state_dict keys; every layer runs on
the fused HIP blocks.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fn
from .. import runtime as rt
from ..builders.encoder_builder import META_ENCODER
from .attentions import MultiHeadAttention, ScaledDotProductAttention
from .pos_embeddings import SinusoidPositionalEmbedding
from .positionwise_feed_forward import PositionWiseFeedForward


class EncoderLayer(nn.Module):
    """MHA block followed by the FFN block (encoders.py:9-19)."""

    def __init__(self, config):
        super().__init__()
        self.mhatt = MultiHeadAttention(config)
        self.pwff = PositionWiseFeedForward(config)

    def forward(self, queries, keys, values, attention_mask, **kwargs):
        return self.pwff(self.mhatt(queries=queries, keys=keys, values=values, attention_mask=attention_mask, **kwargs))


class GuidedEncoderLayer(nn.Module):
    """MCAN SGA unit: self-attention, guided attention over the other modality, FFN (encoders.py:74-99)."""

    def __init__(self, config):
        super().__init__()
        self.self_mhatt = MultiHeadAttention(config)
        self.guided_mhatt = MultiHeadAttention(config)
        self.pwff = PositionWiseFeedForward(config)

    def forward(self, queries, keys, values, self_attention_mask, guided_attention_mask, projected_kv=None, **kwargs):
        x = self.self_mhatt(queries=queries, keys=queries, values=queries, attention_mask=self_attention_mask, **kwargs)
        x = self.guided_mhatt(queries=x, keys=keys, values=values, attention_mask=guided_attention_mask,
                              projected_kv=projected_kv, **kwargs)
        return self.pwff(x)


class CrossModalityEncoderLayer(nn.Module):
    """LXMERT-style layer (encoders.py:21-72).

    In the reference the two cross-attention results are overwritten by the
    self-attention results before use (lines 39-66), so they never reach the
    output and their parameters never receive a gradient.  The parameters are
    kept (state_dict contract); the dead compute is skipped unless
    ``compute_dead_cross_attention`` is set.
    """

    compute_dead_cross_attention = False

    def __init__(self, config):
        super().__init__()
        self.vision_language_mhattn = MultiHeadAttention(config.VISION_LANGUAGE_ATTENTION)
        self.language_vision_mhattn = MultiHeadAttention(config.LANGUAGE_VISION_ATTENTION)
        self.vision_mhattn = MultiHeadAttention(config.VISION_SELF_ATTENTION)
        self.language_mhattn = MultiHeadAttention(config.LANGUAGE_SELF_ATTENTION)
        self.vision_pff = PositionWiseFeedForward(config.VISION_SELF_ATTENTION)
        self.language_pff = PositionWiseFeedForward(config.LANGUAGE_SELF_ATTENTION)

    def forward(self, vision_features, vision_padding_mask, language_features, language_padding_mask, **kwargs):
        if self.compute_dead_cross_attention:
            self.vision_language_mhattn(queries=vision_features, keys=language_features, values=language_features,
                                        attention_mask=language_padding_mask, **kwargs)
            self.language_vision_mhattn(queries=language_features, keys=vision_features, values=vision_features,
                                        attention_mask=vision_padding_mask)
        v = self.vision_mhattn(queries=vision_features, keys=vision_features, values=vision_features,
                               attention_mask=vision_padding_mask, **kwargs)
        l = self.language_mhattn(queries=language_features, keys=language_features, values=language_features,
                                 attention_mask=language_padding_mask)
        return self.vision_pff(v), self.language_pff(l)


class _Prologued(nn.Module):
    """LN(x) + sinusoid positions as one kernel (encoders.py:113,154,192-193,243-244)."""

    def _prologue(self, layer_norm: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
        arena = rt.ensure_arena(layer_norm)
        T = arena.compute_dtype
        if T == torch.float32 and x.dtype != torch.float32:
            x = x.float()
        pos = self.pos_embedding.table(x.shape[1], x.device)
        return Fn.prologue(x, layer_norm, pos, arena, T)


@META_ENCODER.register()
class Encoder(_Prologued):
    """Self-attention stack (MCAN SA over the question): encoders.py:101-117."""

    def __init__(self, config):
        super().__init__()
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)
        self.d_model = config.D_MODEL
        self.layers = nn.ModuleList([EncoderLayer(config.SELF_ATTENTION) for _ in range(config.LAYERS)])

    def forward(self, features: torch.Tensor, padding_mask: torch.Tensor):
        out = self._prologue(self.layer_norm, features)
        for i, layer in enumerate(self.layers):
            if i:
                out = rt.grad_milestone(out)
            out = layer(queries=out, keys=out, values=out, attention_mask=padding_mask)
        return Fn.finalize(out, features.dtype)


@META_ENCODER.register()
class GuidedAttentionEncoder(_Prologued):
    """MCAN guided-attention stack (encoders.py:137-164).  Both attention blocks of a
    layer are built from GUIDED_ATTENTION; the YAML's SELF_ATTENTION node is ignored,
    as in the reference (encoders.py:150)."""

    def __init__(self, config):
        super().__init__()
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)
        self.d_model = config.D_MODEL
        self.guided_attn_layers = nn.ModuleList(
            [GuidedEncoderLayer(config.GUIDED_ATTENTION) for _ in range(config.LAYERS)])

    def _kv_modules(self):
        return [layer.guided_mhatt.attention for layer in self.guided_attn_layers]

    def _ovqa_param_groups(self):
        """Arena adjacency: [K_0 V_0 K_1 V_1 ...] of the guided attentions, so that one GEMM projects the
        question features for all layers (functional.kv_project_all)."""
        att = self._kv_modules()
        if not all(type(a) is ScaledDotProductAttention for a in att):
            return []
        return [[w for a in att for w in (a.fc_k.weight, a.fc_v.weight)],
                [b for a in att for b in (a.fc_k.bias, a.fc_v.bias)]]

    def _hoisted_kv(self, lang):
        att = self._kv_modules()
        if len(att) < 2 or not all(type(a) is ScaledDotProductAttention for a in att):
            return None
        if any(l.guided_mhatt.can_be_stateful and l.guided_mhatt._is_stateful for l in self.guided_attn_layers):
            return None
        if lang.shape[-1] != att[0].fc_k.weight.shape[1] or lang.dim() != 3:
            return None
        return Fn.kv_project_all(lang, att, rt.ensure_arena(self))

    def forward(self, vision_features: torch.Tensor, vision_padding_mask: torch.Tensor,
                language_features: torch.Tensor, language_padding_mask: torch.Tensor):
        out = self._prologue(self.layer_norm, vision_features)
        lang = language_features.to(out.dtype)
        # every layer attends to the same question features: project K/V for all layers in one GEMM
        kv_all = self._hoisted_kv(lang)
        shared = {}
        if kv_all is not None:
            kv_all = rt.grad_milestone(kv_all, barrier=True)
        else:
            lang = rt.grad_milestone(lang, barrier=True)
        for i, layer in enumerate(self.guided_attn_layers):
            if i:
                out = rt.grad_milestone(out)
            out = layer(queries=out, keys=lang, values=lang, self_attention_mask=vision_padding_mask,
                        projected_kv=None if kv_all is None else (kv_all, i, shared),
                        guided_attention_mask=language_padding_mask)
        return Fn.finalize(out, vision_features.dtype)


@META_ENCODER.register()
class CoAttentionEncoder(_Prologued):
    """ViLBERT-style chained co-attention (encoders.py:166-224)."""

    def __init__(self, config):
        super().__init__()
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL)
        self.vision_layer_norm = nn.LayerNorm(config.D_MODEL)
        self.language_layer_norm = nn.LayerNorm(config.D_MODEL)
        self.d_model = config.D_MODEL

        def stack(cfg):
            return nn.ModuleList([EncoderLayer(cfg) for _ in range(config.LAYERS)])

        self.vision_language_attn_layers = stack(config.VISION_LANGUAGE_ATTENTION)
        self.language_vision_attn_layers = stack(config.LANGUAGE_VISION_ATTENTION)
        self.vision_self_attn_layers = stack(config.VISION_SELF_ATTENTION)
        self.language_self_attn_layers = stack(config.LANGUAGE_SELF_ATTENTION)

    def forward(self, vision_features: torch.Tensor, vision_padding_mask: torch.Tensor,
                language_features: torch.Tensor, language_padding_mask: torch.Tensor):
        vdt, ldt = vision_features.dtype, language_features.dtype
        v = self._prologue(self.vision_layer_norm, vision_features)
        l = self._prologue(self.language_layer_norm, language_features)
        for vl, lv, vs, ls in zip(self.vision_language_attn_layers, self.language_vision_attn_layers,
                                  self.vision_self_attn_layers, self.language_self_attn_layers):
            v = vl(queries=v, keys=l, values=l, attention_mask=language_padding_mask)
            l = lv(queries=l, keys=v, values=v, attention_mask=vision_padding_mask)
            v = vs(queries=v, keys=v, values=v, attention_mask=vision_padding_mask)
            l = ls(queries=l, keys=l, values=l, attention_mask=language_padding_mask)
        return Fn.finalize(v, vdt), Fn.finalize(l, ldt)


@META_ENCODER.register()
class CrossModalityEncoder(_Prologued):
    """LXMERT-style encoder (encoders.py:226-253)."""

    def __init__(self, config):
        super().__init__()
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL)
        self.vision_layer_norm = nn.LayerNorm(config.D_MODEL)
        self.language_layer_norm = nn.LayerNorm(config.D_MODEL)
        self.d_model = config.D_MODEL
        self.layers = nn.ModuleList([CrossModalityEncoderLayer(config) for _ in range(config.LAYERS)])

    def forward(self, vision_features: torch.Tensor, vision_padding_mask: torch.Tensor,
                language_features: torch.Tensor, language_padding_mask: torch.Tensor):
        vdt, ldt = vision_features.dtype, language_features.dtype
        v = self._prologue(self.vision_layer_norm, vision_features)
        l = self._prologue(self.language_layer_norm, language_features)
        for i, layer in enumerate(self.layers):
            if i:  # the two modality chains never mix (the cross-attention results are dead): independent cuts
                v, l = rt.grad_milestone(v), rt.grad_milestone(l)
            v, l = layer(vision_features=v, vision_padding_mask=vision_padding_mask, language_features=l,
                         language_padding_mask=language_padding_mask)
        return Fn.finalize(v, vdt), Fn.finalize(l, ldt)
