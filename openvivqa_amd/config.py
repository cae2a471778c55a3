"""yacs-free attribute-access config loader.

The reference reads its YAML files through ``yacs.config.CfgNode``
(/root/reference/configs/utils.py:1-5); every constructor on the hot path only
uses attribute access (``config.D_MODEL``, ``config.SELF_ATTENTION.HEAD`` ...).
``ConfigNode`` gives the same access semantics over ``yaml.safe_load`` so the
reference's YAML files load verbatim without yacs.
"""
from __future__ import annotations

import copy
from typing import Any, Mapping

import yaml


class ConfigNode(dict):
    """dict with recursive attribute access (``node.A.B``), like yacs CfgNode."""

    def __init__(self, init: Mapping[str, Any] | None = None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, Mapping) and not isinstance(v, ConfigNode):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, name: str):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value):
        self[name] = self._wrap(value)

    def __deepcopy__(self, memo):
        return ConfigNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self) -> "ConfigNode":
        return copy.deepcopy(self)

    def get(self, key, default=None):  # keep dict.get semantics explicit
        return self[key] if key in self else default


def get_config(yaml_file: str) -> ConfigNode:
    """Same entry point name as /root/reference/configs/utils.py:4."""
    with open(yaml_file, "r") as f:
        return ConfigNode(yaml.safe_load(f))


def attention_config(d_model=512, head=8, d_key=64, d_value=64, d_ff=2048, dropout=0.1,
                     use_aoa=False, can_be_stateful=False,
                     architecture="ScaledDotProductAttention") -> ConfigNode:
    """Programmatic equivalent of one attention sub-node of the reference YAMLs
    (e.g. /root/reference/configs/mcan.yaml:67-77)."""
    return ConfigNode(dict(ARCHITECTURE=architecture, HEAD=head, D_MODEL=d_model, D_KEY=d_key,
                           D_VALUE=d_value, D_FF=d_ff, D_FEATURE=2048, USE_AOA=use_aoa,
                           CAN_BE_STATEFUL=can_be_stateful, DROPOUT=dropout))
