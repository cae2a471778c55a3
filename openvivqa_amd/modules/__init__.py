"""HIP-backed modules registered under the reference's names."""
from .containers import Module, ModuleDict, ModuleList
from .pos_embeddings import SinusoidPositionalEmbedding
from .attentions import (AdaptiveScaledDotProductAttention, AugmentedGeometryScaledDotProductAttention,
                         AugmentedMemoryScaledDotProductAttention, MultiHeadAttention, ScaledDotProductAttention)
from .positionwise_feed_forward import PositionWiseFeedForward
from .encoders import (CoAttentionEncoder, CrossModalityEncoder, CrossModalityEncoderLayer, Encoder, EncoderLayer,
                       GuidedAttentionEncoder, GuidedEncoderLayer)
from .embeddings import FeatureEmbedding, LSTMTextEmbedding, UsualEmbedding
from .decoders import Decoder, DecoderLayer
from .pointer import DynamicPointerNetwork, OcrPtrNet
from .mmt import MMT, BertEncoder, M4CDecodingHead, PrevPredEmbeddings

__all__ = [
    "Module", "ModuleDict", "ModuleList", "SinusoidPositionalEmbedding", "MultiHeadAttention",
    "ScaledDotProductAttention", "AugmentedMemoryScaledDotProductAttention", "AdaptiveScaledDotProductAttention",
    "AugmentedGeometryScaledDotProductAttention", "PositionWiseFeedForward", "CoAttentionEncoder", "CrossModalityEncoder",
    "CrossModalityEncoderLayer", "Encoder", "EncoderLayer", "GuidedAttentionEncoder", "GuidedEncoderLayer",
    "FeatureEmbedding", "LSTMTextEmbedding", "UsualEmbedding", "Decoder", "DecoderLayer", "DynamicPointerNetwork", "OcrPtrNet", "MMT", "BertEncoder", "PrevPredEmbeddings", "M4CDecodingHead",
]
