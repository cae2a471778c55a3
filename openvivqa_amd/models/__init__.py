"""Model classes registered under the reference's META_ARCHITECTURE names ("next" row 4 of SURVEY 8f)."""
from .mcan import MCAN, MLP
from .cross_modality_transformer import CrossModalityTransformer

__all__ = ["MCAN", "MLP", "CrossModalityTransformer"]
