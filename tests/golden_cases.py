"""Golden-case table shared by the oracle tests (CPU) and the HIP parity tests (GPU).

Each case knows how to build its module from a *namespace* of classes (the
oracle's or the product's -- same constructor signatures, same state_dict
keys), how to call it, and which inputs carry gradients.  The fixtures were
produced by ``tests/golden/make_golden.py`` from the real reference.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from openvivqa_amd.config import ConfigNode

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class FakeVocab:
    max_answer_length = 6
    padding_idx = 0
    bos_idx = 1
    eos_idx = 2

    def __len__(self):
        return 11


class GenVocab:
    """The vocab of G16 (beam search through the reference's BaseTransformer): what BaseTransformer, Decoder and
    UsualEmbedding read (base_transformer.py:13-16,34; decoders.py:35-44)."""

    def __init__(self, meta):
        v = meta["vocab"]
        self.n, self.max_answer_length = v["len"], v["max_answer_length"]
        self.padding_idx, self.bos_idx, self.eos_idx = v["padding_idx"], v["bos_idx"], v["eos_idx"]

    def __len__(self):
        return self.n


def teacher_forced_inputs(tokens, bos_idx, eos_idx, padding_idx):
    """For sequences a search produced, (b, T): the decoder input [<bos>, w_0 .. w_{T-2}]; ``live`` = positions whose
    word was chosen before the sequence had produced <eos> (only there is the recorded score the word's log-probability,
    beam_search.py:52 zeroes it afterwards); ``clean`` = positions before any pad word was fed back (a pad input keeps
    its position embedding in the stateful pass and loses it teacher-forced, decoders.py:59-63: not comparable after)."""
    b, T = tokens.shape
    inp = torch.cat([torch.full((b, 1), bos_idx, dtype=tokens.dtype), tokens[:, :-1]], 1)
    is_eos = (tokens == eos_idx).long()
    live = (torch.cumsum(is_eos, 1) - is_eos) == 0
    clean = torch.cumsum((inp == padding_idx).long(), 1) == 0
    return inp, live, clean


class ModelVocab:
    padding_idx = 0

    def __init__(self, n_tokens, total_answers):
        self.n_tokens, self.total_answers = n_tokens, total_answers

    def __len__(self):
        return self.n_tokens


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    groups = {"in": {}, "w": {}, "out": {}, "lw": {}, "gin": {}, "gw": {}}
    for k in z.files:
        if k == "meta":
            continue
        g, rest = k.split("/", 1)
        groups[g][rest] = torch.from_numpy(z[k])
    meta = json.loads(str(z["meta"]))
    return SimpleNamespace(name=name, meta=meta, **{("inputs" if g == "in" else g): v for g, v in groups.items()})


def oracle_namespace():
    import oracle as O
    return SimpleNamespace(
        SDPA=O.OracleSDPA, MemorySDPA=O.OracleMemorySDPA, AdaptiveSDPA=O.OracleAdaptiveSDPA,
        GeometrySDPA=O.OracleGeometrySDPA, MHA=O.OracleMHA, PWFF=O.OraclePWFF, EncoderLayer=O.OracleEncoderLayer,
        GuidedEncoderLayer=O.OracleGuidedEncoderLayer, CrossModalityEncoderLayer=O.OracleCrossModalityEncoderLayer,
        Encoder=O.OracleEncoder, GuidedAttentionEncoder=O.OracleGuidedAttentionEncoder,
        CoAttentionEncoder=O.OracleCoAttentionEncoder, CrossModalityEncoder=O.OracleCrossModalityEncoder,
        DecoderLayer=O.OracleDecoderLayer, Decoder=O.OracleDecoder, OcrPtrNet=O.OracleOcrPtrNet,
        DynamicPointerNetwork=O.OracleDynamicPointerNetwork, MCAN=O.OracleMCAN,
        LSTMTextEmbedding=O.OracleLSTMTextEmbedding,
        BertEncoder=O.OracleBertEncoder, MMT=O.OracleMMT)


def hip_namespace():
    import openvivqa_amd.modules as M
    return SimpleNamespace(
        SDPA=M.ScaledDotProductAttention, MemorySDPA=M.AugmentedMemoryScaledDotProductAttention,
        AdaptiveSDPA=M.AdaptiveScaledDotProductAttention, GeometrySDPA=M.AugmentedGeometryScaledDotProductAttention,
        MHA=M.MultiHeadAttention, PWFF=M.PositionWiseFeedForward,
        EncoderLayer=M.EncoderLayer, GuidedEncoderLayer=M.GuidedEncoderLayer,
        CrossModalityEncoderLayer=M.CrossModalityEncoderLayer, Encoder=M.Encoder,
        GuidedAttentionEncoder=M.GuidedAttentionEncoder, CoAttentionEncoder=M.CoAttentionEncoder,
        CrossModalityEncoder=M.CrossModalityEncoder, DecoderLayer=M.DecoderLayer, Decoder=M.Decoder,
        OcrPtrNet=M.OcrPtrNet, DynamicPointerNetwork=M.DynamicPointerNetwork, MCAN=_hip_mcan(),
        LSTMTextEmbedding=M.LSTMTextEmbedding,
        BertEncoder=M.BertEncoder, MMT=M.MMT)


def _hip_mcan():
    from openvivqa_amd.models import MCAN
    return MCAN


def _cfg(case):
    return ConfigNode(case.meta["cfg"])


def _sdpa(ns, c):
    return ns.SDPA(_cfg(c))


def _call_sdpa(m, i):
    out, att = m(i["queries"], i["keys"], i["values"], attention_mask=i["mask"])
    return {"out": out, "att": att}


def _call_sdpa_causal(m, i):
    out, att = m(i["queries"], i["queries"], i["queries"], attention_mask=i["mask"])
    out_nm, _ = m(i["queries"], i["queries"], i["queries"])
    return {"out": out, "att": att, "out_nomask": out_nm}


def _call_vl_single(m, i):
    return {"out": m(vision_features=i["vision"], vision_padding_mask=i["vmask"],
                     language_features=i["language"], language_padding_mask=i["lmask"])}


def _call_vl_pair(m, i):
    v, l = m(vision_features=i["vision"], vision_padding_mask=i["vmask"],
             language_features=i["language"], language_padding_mask=i["lmask"])
    return {"vision": v, "language": l}


def _call_mmt(m, i):
    out = m(i["txt"], i["tmask"], i["obj"], i["omask"], i["ocr"], i["cmask"], i["ans"], i["prev_inds"])
    dec = m.prev_pred_embeddings(i["ans"], i["ocr"], i["prev_inds"])
    return {"seq": out["mmt_seq_output"], "txt_out": out["mmt_txt_output"], "ocr_out": out["mmt_ocr_output"],
            "dec_out": out["mmt_dec_output"], "dec_emb": dec}


# name -> (build(ns, case), call(module, inputs), grad_inputs)
CASES = {
    "G1_sdpa_5x7": (_sdpa, _call_sdpa, ["queries", "keys", "values"]),
    "G1_sdpa_7x7": (_sdpa, _call_sdpa, ["queries", "keys", "values"]),
    "G1_sdpa_causal": (_sdpa, _call_sdpa_causal, ["queries"]),
    "G14_memory_sdpa": (lambda ns, c: ns.MemorySDPA(_cfg(c)), _call_sdpa, ["queries", "keys", "values"]),
    "G15_adaptive_sdpa": (lambda ns, c: ns.AdaptiveSDPA(_cfg(c)),
                          lambda m, i: (lambda r: {"out": r[0], "att": torch.cat(r[1], dim=2)})(
                              m(i["queries"], i["keys"], i["values"], i["signals"], attention_mask=i["mask"])),
                          ["queries", "keys", "values", "signals"]),
    "G2_mha_aoa0": (lambda ns, c: ns.MHA(_cfg(c)),
                    lambda m, i: {"out": m(i["queries"], i["keys"], i["values"], i["mask"])},
                    ["queries", "keys", "values"]),
    "G2_mha_aoa1": (lambda ns, c: ns.MHA(_cfg(c)),
                    lambda m, i: {"out": m(i["queries"], i["keys"], i["values"], i["mask"])},
                    ["queries", "keys", "values"]),
    "G3_pwff": (lambda ns, c: ns.PWFF(_cfg(c)), lambda m, i: {"out": m(i["x"])}, ["x"]),
    "G4_encoder_layer": (lambda ns, c: ns.EncoderLayer(_cfg(c)),
                         lambda m, i: {"out": m(queries=i["x"], keys=i["x"], values=i["x"], attention_mask=i["mask"])},
                         ["x"]),
    "G4_guided_layer": (lambda ns, c: ns.GuidedEncoderLayer(_cfg(c)),
                        lambda m, i: {"out": m(queries=i["vision"], keys=i["language"], values=i["language"],
                                               self_attention_mask=i["vmask"], guided_attention_mask=i["lmask"])},
                        ["vision", "language"]),
    "G4_crossmodality_layer": (lambda ns, c: ns.CrossModalityEncoderLayer(_cfg(c)), _call_vl_pair,
                               ["vision", "language"]),
    "G5_encoder": (lambda ns, c: ns.Encoder(_cfg(c)),
                   lambda m, i: {"out": m(features=i["features"], padding_mask=i["mask"])}, ["features"]),
    "G5_guided_encoder": (lambda ns, c: ns.GuidedAttentionEncoder(_cfg(c)), _call_vl_single, ["vision", "language"]),
    "G5_coattention_encoder": (lambda ns, c: ns.CoAttentionEncoder(_cfg(c)), _call_vl_pair, ["vision", "language"]),
    "G5_crossmodality_encoder": (lambda ns, c: ns.CrossModalityEncoder(_cfg(c)), _call_vl_pair,
                                 ["vision", "language"]),
    "G7_decoder_layer": (lambda ns, c: ns.DecoderLayer(_cfg(c)),
                         lambda m, i: {"out": m(queries=i["x"], keys=i["enc"], values=i["enc"],
                                                self_attention_mask=i["self_mask"],
                                                enc_attention_mask=i["enc_mask"])}, ["x", "enc"]),
    "G7_decoder": (lambda ns, c: ns.Decoder(_cfg(c), FakeVocab()),
                   lambda m, i: {"logp": m(answer_tokens=i["tokens"], encoder_features=i["enc"],
                                           encoder_attention_mask=i["enc_mask"])}, ["enc"]),
    "G8_ocrptr": (lambda ns, c: ns.OcrPtrNet(c.meta["hidden"]),
                  lambda m, i: {"s3": m(i["q3"], i["k"], i["mask"]), "s2": m(i["q2"], i["k"], i["mask"])},
                  ["q3", "q2", "k"]),
    "G12_mcan_model": (lambda ns, c: ns.MCAN(_cfg(c), ModelVocab(c.meta["vocab_len"], c.meta["total_answers"])),
                       lambda m, i: {"logp": m(SimpleNamespace(region_features=i["regions"],
                                                               question_tokens=i["tokens"]))}, ["regions"]),
    "G17_lstm_text_embedding": (lambda ns, c: ns.LSTMTextEmbedding(_cfg(c), ModelVocab(c.meta["vocab_len"],
                                                                                        c.meta["total_answers"])),
                                lambda m, i: (lambda r: {"features": r[0], "pad_mask": r[1][0], "seq_mask": r[1][1]})(
                                    m(i["tokens"])), []),
    "G13_bert_encoder": (lambda ns, c: ns.BertEncoder(SimpleNamespace(**c.meta["cfg"])),
                         lambda m, i: {"out": m(i["x"], i["mask"], head_mask=[None] * 2)[0]}, ["x"]),
    "G13_mmt": (lambda ns, c: ns.MMT(SimpleNamespace(**c.meta["cfg"])), _call_mmt, ["txt", "obj", "ocr", "ans"]),
    "G8_dynptr_query_axis": (lambda ns, c: ns.DynamicPointerNetwork(ConfigNode(dict(D_MODEL=c.meta["d_model"])),
                                                                    axis="query"),
                             lambda m, i: {"scores": m(i["q"], i["k"], i["qmask"])}, ["q", "k"]),
    "G8_dynptr_key_axis": (lambda ns, c: ns.DynamicPointerNetwork(ConfigNode(dict(D_MODEL=c.meta["d_model"])),
                                                                  axis="key"),
                           lambda m, i: {"scores": m(i["q"], i["k"], i["kmask"])}, ["q", "k"]),
}


def run_case(ns, name, device="cpu", dtype=torch.float32, prepare=None):
    """Build the module from ``ns``, load golden weights, run fwd+bwd.

    Returns (case, outs, gin, gw, module)."""
    case = load_case(name)
    build, call, grad_inputs = CASES[name]
    module = build(ns, case)
    missing, unexpected = module.load_state_dict(case.w, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "running_" not in k and "pos_emb" not in k], missing
    module = module.to(device)
    if prepare is not None:
        module = prepare(module)
    module.eval()
    ins = {}
    for k, v in case.inputs.items():
        v = v.clone().to(device)
        if v.dtype.is_floating_point and "mask" not in k:
            v = v.to(dtype)
        if k in grad_inputs:
            v.requires_grad_(True)
        ins[k] = v
    outs = call(module, ins)
    loss = 0
    for k, o in outs.items():
        if k in case.lw:
            lw = case.lw[k].to(device)
            of = o.float()
            loss = loss + (torch.where(torch.isfinite(of), of, torch.zeros_like(of)) * lw).sum()
    loss.backward()
    gin = {k: ins[k].grad for k in grad_inputs}
    gw = {k: p.grad for k, p in module.named_parameters()}
    return case, outs, gin, gw, module
