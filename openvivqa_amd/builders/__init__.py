"""Registry + factory surface kept from the reference.

Importing this package imports ``openvivqa_amd.modules`` so that the
decorators run, the way /root/reference/builders/__init__.py:1-9 star-imports
its module packages.
"""
from .registry import Registry
from .attention_builder import META_ATTENTION, build_attention
from .encoder_builder import META_ENCODER, build_encoder
from .decoder_builder import META_DECODER, build_decoder
from .text_embedding_builder import META_TEXT_EMBEDDING, build_text_embedding
from .vision_embedding_builder import META_VISION_EMBEDDING, build_vision_embedding
from .model_builder import META_ARCHITECTURE, build_model

__all__ = [
    "Registry", "META_ATTENTION", "build_attention", "META_ENCODER", "build_encoder",
    "META_DECODER", "build_decoder", "META_TEXT_EMBEDDING", "build_text_embedding",
    "META_VISION_EMBEDDING", "build_vision_embedding", "META_ARCHITECTURE", "build_model",
]
