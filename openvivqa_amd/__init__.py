"""openvivqa_amd -- MI355X-native cross-modal attention hot path of OpenViVQA.

Import side effect (like /root/reference/builders/__init__.py): the HIP-backed
classes register themselves under the reference's names, so
``build_attention / build_encoder / build_decoder`` resolve the reference's
YAML ``ARCHITECTURE`` strings to them.
"""
from . import builders  # noqa: F401
from .builders import (META_ATTENTION, META_DECODER, META_ENCODER, build_attention, build_decoder,  # noqa: F401
                       build_encoder)
from .config import ConfigNode, attention_config, get_config  # noqa: F401
from .runtime import (get_compute_dtype, manual_seed, prepare, set_compute_dtype)  # noqa: F401
from . import modules  # noqa: F401  (registration)
from . import models  # noqa: F401  (registration)
from .builders import META_ARCHITECTURE, build_model  # noqa: F401

__version__ = "0.1.0"
