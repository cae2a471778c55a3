"""Model classes registered under the reference's META_ARCHITECTURE names ("next" row 4 of SURVEY 8f)."""
from .mcan import MCAN, MLP

__all__ = ["MCAN", "MLP"]
