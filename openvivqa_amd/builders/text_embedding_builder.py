"""``build_text_embedding`` factory (reference: builders/text_embedding_builder.py)."""
from .registry import Registry

META_TEXT_EMBEDDING = Registry("TEXT_EMBEDDING")


def build_text_embedding(config, vocab):
    return META_TEXT_EMBEDDING.get(config.ARCHITECTURE)(config, vocab)
