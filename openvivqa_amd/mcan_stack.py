"""The MCAN encoder stack as the reference's model composes it (models/mcan.py:37-38,57-68):
``self_encoder`` (question SA stack) runs to completion, then ``guided_encoder`` attends the
image regions to the FINAL question features.  Built through the registry factories, so the
YAML ``ARCHITECTURE`` strings select the classes exactly as in the reference."""
from __future__ import annotations

import torch
from torch import nn

from .builders import build_encoder


class MCANEncoderStack(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        self.self_encoder = build_encoder(model_config.SELF_ENCODER)
        self.guided_encoder = build_encoder(model_config.GUIDED_ENCODER)
        self.d_model = model_config.D_MODEL

    def forward(self, vision_features, vision_padding_mask, text_features, text_padding_mask):
        text = self.self_encoder(features=text_features, padding_mask=text_padding_mask)
        vision = self.guided_encoder(vision_features=vision_features, vision_padding_mask=vision_padding_mask,
                                     language_features=text,
                                     language_padding_mask=text_padding_mask)
        return vision, text


def synthetic_batch(batch, regions, tokens, d_model, min_regions, min_tokens, seed, device, dtype):
    """SURVEY 8d stack-level inputs: N(0,1) features, per-sample valid lengths, padded rows zeroed,
    additive -1e5 masks of shape (B,1,1,N)."""
    gen = torch.Generator().manual_seed(seed)
    v = torch.randn(batch, regions, d_model, generator=gen)
    t = torch.randn(batch, tokens, d_model, generator=gen)
    nv = torch.randint(min_regions, regions + 1, (batch,), generator=gen)
    nt = torch.randint(min_tokens, tokens + 1, (batch,), generator=gen)
    vmask = (torch.arange(regions)[None, :] >= nv[:, None])
    tmask = (torch.arange(tokens)[None, :] >= nt[:, None])
    v[vmask] = 0
    t[tmask] = 0
    vm = (vmask.float() * -10e4)[:, None, None, :]
    tm = (tmask.float() * -10e4)[:, None, None, :]
    return (v.to(device=device, dtype=dtype), vm.to(device), t.to(device=device, dtype=dtype), tm.to(device))
