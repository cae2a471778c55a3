"""CrossModalityTransformer classification model (models/cross_modality_transformer.py:12-78, LXMERT-style) on the HIP
hot path: region + token embeddings -> CrossModalityEncoder -> softmax attention pooling of both modalities ->
LN(proj_v + proj_t) -> classifier.  Same constructor ``(config, vocab)``, attribute names and ``state_dict`` keys as the
reference (BASELINE configs[2] builds it from the unmodified ``configs/cross_modality_transformer.yaml``).

As upstream, ``forward`` returns the raw classifier logits (``cross_modality_transformer.py:78``; MCAN returns
log-probabilities) -- the reference feeds them to NLLLoss as they are.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fn
from .. import runtime as rt
from ..builders.encoder_builder import build_encoder
from ..builders.model_builder import META_ARCHITECTURE
from ..builders.text_embedding_builder import build_text_embedding
from ..builders.vision_embedding_builder import build_vision_embedding
from .mcan import MLP


@META_ARCHITECTURE.register()
class CrossModalityTransformer(nn.Module):
    def __init__(self, config, vocab):
        super().__init__()
        self.d_model = config.D_MODEL
        self.device = torch.device(config.DEVICE)
        self.region_embedding = build_vision_embedding(config.REGION_EMBEDDING)
        self.text_embedding = build_text_embedding(config.TEXT_EMBEDDING, vocab)
        self.encoder = build_encoder(config.ENCODER)
        self.vision_attr_reduce = MLP(config.VISION_ATTR_REDUCE)
        self.text_attr_reduce = MLP(config.TEXT_ATTR_REDUCE)
        self.vision_proj = nn.Linear(config.D_MODEL, config.D_MODEL)
        self.text_proj = nn.Linear(config.D_MODEL, config.D_MODEL)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)
        self.classify = nn.Linear(config.D_MODEL, vocab.total_answers)

    def init_weights(self):  # base_classification.py:12-15
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, input_features):
        arena = rt.ensure_arena(self)
        T = arena.compute_dtype
        vision, vision_mask = self.region_embedding(input_features.region_features)
        text, (text_mask, _) = self.text_embedding(input_features.question_tokens)
        vision, text = self.encoder(vision_features=vision, vision_padding_mask=vision_mask,
                                    language_features=text, language_padding_mask=text_mask)
        wv = self.vision_attr_reduce.pool(vision, arena)
        wt = self.text_attr_reduce.pool(text, arena)
        fused = Fn.linear_residual(wt, Fn.linear(wv, self.vision_proj, arena), self.text_proj, arena)
        out = Fn.prologue(fused, self.layer_norm, None, arena, T)
        return Fn.linear(out, self.classify, arena).float()
