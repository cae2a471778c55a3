"""``build_attention`` factory (reference: builders/attention_builder.py:3-8)."""
from .registry import Registry

META_ATTENTION = Registry("META_ATTENTION")


def build_attention(config):
    return META_ATTENTION.get(config.ARCHITECTURE)(config)
