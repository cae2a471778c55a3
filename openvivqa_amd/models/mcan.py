"""MCAN classification model (models/mcan.py:12-81) on the HIP hot path.

Same constructor ``(config, vocab)``, attribute names and ``state_dict`` keys as the reference, so
``build_model`` + the reference's ``configs/mcan.yaml`` resolve to it and checkpoints interchange.
The two encoder stacks (97 % of the FLOPs) are the fused HIP blocks; the embedding projection is the fused
GEMM+GELU(+dropout) epilogue; the attention-pooling head is the fc1 GEMM + ONE kernel per modality (relu, dropout, the
D -> 1 product, softmax over the positions, weighted sum: csrc/model_ends.hip), the two pooled projections are summed in the
second GEMM's residual epilogue, and the classifier + log_softmax run on the zero-padded footprint of the (ragged, 353-way)
classifier weights.
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


class MLP(nn.Module):
    """fc2(dropout(relu(fc1 x))): one attention-pooling logit per position (mcan.py:12-25)."""

    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.D_MODEL, config.D_MODEL)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(config.DROPOUT)
        self.fc2 = nn.Linear(config.D_MODEL, 1)
        self._site = rt.new_dropout_site()

    def forward(self, features: torch.Tensor):
        """The logits themselves (the reference's MLP.forward); the models below call ``pool`` instead."""
        arena = rt.ensure_arena(self)
        h = Fn.linear(features.to(arena.compute_dtype), self.fc1, arena)
        h = self.dropout(self.relu(h))
        # D -> 1: a matrix-vector product; fp32 torch op on the arena's master weights
        return torch.nn.functional.linear(h.float(), self.fc2.weight, self.fc2.bias)

    def pool(self, features: torch.Tensor, arena):
        """sum_n softmax_n(self(features))[n] * features[:, n]  (mcan.py:70-76), [B, D] in the compute dtype."""
        drop = rt.dropout_spec(self.dropout.p, self._site, self.training, features.device)
        return Fn.attention_pool(features, self, arena, drop)


@META_ARCHITECTURE.register()
class MCAN(nn.Module):
    def __init__(self, config, vocab):
        super().__init__()
        self.d_model = config.D_MODEL
        self.device = torch.device(config.DEVICE)
        self.text_embedding = build_text_embedding(config.TEXT_EMBEDDING, vocab)
        self.vision_embedding = build_vision_embedding(config.VISION_EMBEDDING)
        self.self_encoder = build_encoder(config.SELF_ENCODER)
        self.guided_encoder = build_encoder(config.GUIDED_ENCODER)
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
        vision, vision_mask = self.vision_embedding(input_features.region_features)
        text, (text_mask, _) = self.text_embedding(input_features.question_tokens)
        # (the LSTM embedding makes its output in the compute dtype too: the question stack takes that twin -- and hands its
        # own output over in the compute dtype -- instead of casting the fp32 tensor back and forth)
        text = self.self_encoder(features=Fn.compute_twin(text, T), padding_mask=text_mask)
        vision = self.guided_encoder(vision_features=vision, vision_padding_mask=vision_mask,
                                     language_features=text, language_padding_mask=text_mask)
        # attention pooling over each sequence (softmax over dim=1, padded positions included as in the
        # reference, mcan.py:70-76)
        wv = self.vision_attr_reduce.pool(vision, arena)
        wt = self.text_attr_reduce.pool(text, arena)
        fused = Fn.linear_residual(wt, Fn.linear(wv, self.vision_proj, arena), self.text_proj, arena)
        out = Fn.prologue(fused, self.layer_norm, None, arena, T)
        return Fn.classify_log_softmax(out, self.classify, arena)
