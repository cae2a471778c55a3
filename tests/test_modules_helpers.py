"""Shared builders for module-level tests (CPU and GPU)."""
import torch


def mcan_pair(ns, layers, seed, d=512, heads=8, dff=2048):
    """(Encoder, GuidedAttentionEncoder) of namespace ``ns`` (the product's modules or the oracle)."""
    from openvivqa_amd.config import ConfigNode, attention_config
    torch.manual_seed(seed)
    sa = attention_config(d_model=d, head=heads, d_key=d // heads, d_value=d // heads, d_ff=dff)
    enc = getattr(ns, "Encoder", None) or ns.OracleEncoder
    gen = getattr(ns, "GuidedAttentionEncoder", None) or ns.OracleGuidedAttentionEncoder
    te = enc(ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=d, LAYERS=layers, SELF_ATTENTION=sa)))
    ve = gen(ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=d, LAYERS=layers, SELF_ATTENTION=sa,
                             GUIDED_ATTENTION=sa)))
    return te, ve
