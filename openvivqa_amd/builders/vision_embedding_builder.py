"""``build_vision_embedding`` factory (reference: builders/vision_embedding_builder.py)."""
from .registry import Registry

META_VISION_EMBEDDING = Registry("META_VISION_EMBEDDING")


def build_vision_embedding(config):
    return META_VISION_EMBEDDING.get(config.ARCHITECTURE)(config)
