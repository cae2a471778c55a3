#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

* needs /root/reference (read-only); it is never copied -- only tensors leave.
* import recipe = SURVEY.md section 8c: bare package shells for ``builders``,
  ``models``, ``models.modules`` and ``data_utils`` so the reference's fan-out
  ``__init__`` files do not run, plus an in-memory ``termcolor`` stub.
* every case is seeded, eval mode (dropout off) unless stated, fp32.
* outputs: tests/golden/G*.npz (+ manifest.json).  Small (<1 MB in total).

npz key scheme:  in/<name> inputs, w/<state_dict key> weights, out/<name>
outputs, lw/<name> loss weights (loss = sum_i (out_i * lw_i).sum()),
gin/<name> input grads, gw/<param> param grads.  ``meta`` holds a JSON string.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

for _name in ["builders", "models", "models.modules", "data_utils"]:
    _m = types.ModuleType(_name)
    _m.__path__ = [os.path.join(REF, _name.replace(".", "/"))]
    sys.modules[_name] = _m
_tc = types.ModuleType("termcolor")
_tc.colored = lambda s, *a, **k: s
sys.modules["termcolor"] = _tc

import models.modules.attentions as R_att  # noqa: E402
import models.modules.positionwise_feed_forward as R_ff  # noqa: E402
import models.modules.encoders as R_enc  # noqa: E402
import models.modules.pos_embeddings as R_pos  # noqa: E402
import models.utils as R_utils  # noqa: E402
import models.modules.text_embeddings as R_txt  # noqa: E402,F401  (registers UsualEmbedding)
import models.modules.decoders as R_dec  # noqa: E402
import models.mmf_m4c as R_m4c  # noqa: E402
import models.iterative_m4c as R_im4c  # noqa: E402
import models.base_transformer as R_bt  # noqa: E402  (BaseTransformer: step / beam_search, base_transformer.py:9-54)
import models.modules.beam_search as R_bs  # noqa: E402

from openvivqa_amd.config import ConfigNode, attention_config  # noqa: E402

torch.set_num_threads(4)
D, H, DK, DFF = 32, 4, 8, 64


def att_cfg(**kw):
    base = dict(d_model=D, head=H, d_key=DK, d_value=DK, d_ff=DFF, dropout=0.1)
    base.update(kw)
    return attention_config(**base)


def np_(t):
    return t.detach().cpu().numpy().copy()


class Case:
    def __init__(self, name):
        self.name = name
        self.arrays = {}
        self.meta = {}

    def add(self, prefix, key, t):
        self.arrays[f"{prefix}/{key}"] = np_(t) if torch.is_tensor(t) else np.asarray(t)

    def weights(self, module):
        for k, v in module.state_dict().items():
            self.add("w", k, v)

    def save(self):
        self.arrays["meta"] = np.array(json.dumps(self.meta))
        path = os.path.join(HERE, self.name + ".npz")
        np.savez_compressed(path, **self.arrays)
        return path


def run_with_grads(case, module, inputs, call, grad_inputs):
    """inputs: dict name->tensor; call(module, inputs)->dict of outputs."""
    module.eval()
    case.weights(module)
    ins = {}
    for k, v in inputs.items():
        v = v.clone()
        if k in grad_inputs:
            v.requires_grad_(True)
        ins[k] = v
        case.add("in", k, v)
    outs = call(module, ins)
    g = torch.Generator().manual_seed(4242)
    loss = 0
    for k, o in outs.items():
        case.add("out", k, o)
        if o.dtype.is_floating_point and o.requires_grad:
            lw = torch.randn(o.shape, generator=g)
            case.add("lw", k, lw)
            fin = torch.isfinite(o)
            loss = loss + (torch.where(fin, o, torch.zeros_like(o)) * lw).sum()
    loss.backward()
    for k in grad_inputs:
        case.add("gin", k, ins[k].grad)
    none = []
    for k, p in module.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            case.add("gw", k, p.grad)
    case.meta["grad_none"] = none
    case.meta["loss"] = float(loss.detach())
    return outs


def feats(b, n, d, gen, pad_rows=None):
    x = torch.randn(b, n, d, generator=gen)
    if pad_rows:
        for bi, rows in pad_rows.items():
            x[bi, rows] = 0
    return x


manifest = {}


def finish(case):
    p = case.save()
    manifest[case.name] = {"bytes": os.path.getsize(p), "keys": len(case.arrays)}
    print(f"{case.name}: {os.path.getsize(p)} B, {len(case.arrays)} arrays")


# ---------------------------------------------------------------- G1 SDPA
def g1():
    for tag, (nq, nk) in {"5x7": (5, 7), "7x7": (7, 7)}.items():
        torch.manual_seed(101)
        m = R_att.ScaledDotProductAttention(att_cfg())
        with torch.no_grad():  # biases are zero-initialised; make them matter
            for lin in (m.fc_q, m.fc_k, m.fc_v, m.fc_o):
                lin.bias.normal_(0, 0.1)
        gen = torch.Generator().manual_seed(7)
        q = feats(3, nq, D, gen)
        kv = feats(3, nk, D, gen, pad_rows={1: [nk - 2, nk - 1], 2: list(range(nk))})
        mask = R_utils.generate_padding_mask(kv, 0)
        c = Case(f"G1_sdpa_{tag}")
        c.meta.update(cfg=dict(att_cfg()), nq=nq, nk=nk)

        def call(mod, ins):
            out, att = mod(ins["queries"], ins["keys"], ins["values"], attention_mask=ins["mask"])
            return {"out": out, "att": att}
        run_with_grads(c, m, {"queries": q, "keys": kv, "values": kv.clone(), "mask": mask},
                       call, ["queries", "keys", "values"])
        finish(c)
    # no-mask + (B,1,nq,nk) mask variant
    torch.manual_seed(102)
    m = R_att.ScaledDotProductAttention(att_cfg())
    gen = torch.Generator().manual_seed(8)
    q = feats(2, 6, D, gen)
    pm = R_utils.generate_padding_mask(torch.tensor([[3, 4, 5, 6, 0, 0], [3, 4, 5, 6, 7, 8]]), 0)
    sam = R_utils.generate_self_attention_masks(pm, R_utils.generate_sequential_mask(6))
    c = Case("G1_sdpa_causal")
    c.meta.update(cfg=dict(att_cfg()))

    def call2(mod, ins):
        out, att = mod(ins["queries"], ins["queries"], ins["queries"], attention_mask=ins["mask"])
        out_nm, _ = mod(ins["queries"], ins["queries"], ins["queries"])
        return {"out": out, "att": att, "out_nomask": out_nm}
    run_with_grads(c, m, {"queries": q, "mask": sam}, call2, ["queries"])
    finish(c)


# ---------------------------------------------------------------- G2 MHA
def g2():
    for aoa in (False, True):
        torch.manual_seed(201)
        cfg = att_cfg(use_aoa=aoa)
        m = R_att.MultiHeadAttention(cfg)
        with torch.no_grad():
            m.layer_norm.weight.uniform_(0.5, 1.5)
            m.layer_norm.bias.normal_(0, 0.1)
        gen = torch.Generator().manual_seed(9)
        q = feats(3, 5, D, gen)
        kv = feats(3, 7, D, gen, pad_rows={1: [5, 6], 2: list(range(7))})
        mask = R_utils.generate_padding_mask(kv, 0)
        c = Case(f"G2_mha_aoa{int(aoa)}")
        c.meta.update(cfg=dict(cfg))
        run_with_grads(c, m, {"queries": q, "keys": kv, "values": kv.clone(), "mask": mask},
                       lambda mod, ins: {"out": mod(ins["queries"], ins["keys"], ins["values"], ins["mask"])},
                       ["queries", "keys", "values"])
        finish(c)
    # stateful: 3 single steps vs one shot with causal mask
    torch.manual_seed(202)
    cfg = att_cfg(can_be_stateful=True)
    m = R_att.MultiHeadAttention(cfg).eval()
    gen = torch.Generator().manual_seed(10)
    x = feats(2, 3, D, gen)
    c = Case("G2_mha_stateful")
    c.meta.update(cfg=dict(cfg))
    c.weights(m)
    c.add("in", "x", x)
    with torch.no_grad():
        full = m(x, x, x, R_utils.generate_sequential_mask(3))
        c.add("out", "oneshot", full)
        with m.statefulness(2):
            steps = []
            for t in range(3):
                xt = x[:, t:t + 1]
                steps.append(m(xt, xt, xt, torch.zeros(1, 1, 1, t + 1)))
            c.add("out", "running_keys_final", m.running_keys)
        c.add("out", "steps", torch.cat(steps, 1))
        c.meta["state_keys_after_disable"] = [list(m.running_keys.shape), list(m.running_values.shape)]
    finish(c)


# ---------------------------------------------------------------- G3 PWFF
def g3():
    torch.manual_seed(301)
    m = R_ff.PositionWiseFeedForward(att_cfg())
    with torch.no_grad():
        m.layer_norm.weight.uniform_(0.5, 1.5)
        m.layer_norm.bias.normal_(0, 0.1)
    gen = torch.Generator().manual_seed(11)
    x = feats(2, 7, D, gen) * 2.0
    c = Case("G3_pwff")
    c.meta.update(cfg=dict(att_cfg()))
    run_with_grads(c, m, {"x": x}, lambda mod, ins: {"out": mod(ins["x"])}, ["x"])
    finish(c)


def cm_cfg(layers=2):
    return ConfigNode(dict(ARCHITECTURE="CrossModalityEncoder", D_MODEL=D, LAYERS=layers,
                           VISION_LANGUAGE_ATTENTION=att_cfg(), LANGUAGE_VISION_ATTENTION=att_cfg(),
                           VISION_SELF_ATTENTION=att_cfg(), LANGUAGE_SELF_ATTENTION=att_cfg()))


def vl_inputs(seed, b=3, nv=9, nl=5):
    gen = torch.Generator().manual_seed(seed)
    v = feats(b, nv, D, gen, pad_rows={1: [7, 8]})
    l = feats(b, nl, D, gen, pad_rows={2: [3, 4]})
    return v, l, R_utils.generate_padding_mask(v, 0), R_utils.generate_padding_mask(l, 0)


# ---------------------------------------------------------------- G4 layers
def g4():
    v, l, vm, lm = vl_inputs(12)
    torch.manual_seed(401)
    m = R_enc.EncoderLayer(att_cfg())
    c = Case("G4_encoder_layer")
    c.meta.update(cfg=dict(att_cfg()))
    run_with_grads(c, m, {"x": v, "mask": vm},
                   lambda mod, ins: {"out": mod(queries=ins["x"], keys=ins["x"], values=ins["x"],
                                                attention_mask=ins["mask"])}, ["x"])
    finish(c)

    torch.manual_seed(402)
    m = R_enc.GuidedEncoderLayer(att_cfg())
    c = Case("G4_guided_layer")
    c.meta.update(cfg=dict(att_cfg()))
    run_with_grads(c, m, {"vision": v, "language": l, "vmask": vm, "lmask": lm},
                   lambda mod, ins: {"out": mod(queries=ins["vision"], keys=ins["language"], values=ins["language"],
                                                self_attention_mask=ins["vmask"], guided_attention_mask=ins["lmask"])},
                   ["vision", "language"])
    finish(c)

    torch.manual_seed(403)
    m = R_enc.CrossModalityEncoderLayer(cm_cfg())
    c = Case("G4_crossmodality_layer")
    c.meta.update(cfg=json.loads(json.dumps(cm_cfg())))

    def call(mod, ins):
        vo, lo = mod(vision_features=ins["vision"], vision_padding_mask=ins["vmask"],
                     language_features=ins["language"], language_padding_mask=ins["lmask"])
        return {"vision": vo, "language": lo}
    run_with_grads(c, m, {"vision": v, "language": l, "vmask": vm, "lmask": lm}, call, ["vision", "language"])
    finish(c)


# ---------------------------------------------------------------- G5 encoders
def g5():
    v, l, vm, lm = vl_inputs(13)
    sa = att_cfg()
    enc_cfg = ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=D, LAYERS=2, SELF_ATTENTION=sa))
    torch.manual_seed(501)
    m = R_enc.Encoder(enc_cfg)
    c = Case("G5_encoder")
    c.meta.update(cfg=json.loads(json.dumps(enc_cfg)))
    run_with_grads(c, m, {"features": l, "mask": lm},
                   lambda mod, ins: {"out": mod(features=ins["features"], padding_mask=ins["mask"])}, ["features"])
    finish(c)

    g_cfg = ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=D, LAYERS=2,
                            SELF_ATTENTION=sa, GUIDED_ATTENTION=att_cfg()))
    torch.manual_seed(502)
    m = R_enc.GuidedAttentionEncoder(g_cfg)
    c = Case("G5_guided_encoder")
    c.meta.update(cfg=json.loads(json.dumps(g_cfg)))

    def callg(mod, ins):
        return {"out": mod(vision_features=ins["vision"], vision_padding_mask=ins["vmask"],
                           language_features=ins["language"], language_padding_mask=ins["lmask"])}
    run_with_grads(c, m, {"vision": v, "language": l, "vmask": vm, "lmask": lm}, callg, ["vision", "language"])
    finish(c)

    def call2(mod, ins):
        vo, lo = mod(vision_features=ins["vision"], vision_padding_mask=ins["vmask"],
                     language_features=ins["language"], language_padding_mask=ins["lmask"])
        return {"vision": vo, "language": lo}

    co_cfg = cm_cfg()
    co_cfg.ARCHITECTURE = "CoAttentionEncoder"
    torch.manual_seed(503)
    m = R_enc.CoAttentionEncoder(co_cfg)
    c = Case("G5_coattention_encoder")
    c.meta.update(cfg=json.loads(json.dumps(co_cfg)))
    run_with_grads(c, m, {"vision": v, "language": l, "vmask": vm, "lmask": lm}, call2, ["vision", "language"])
    finish(c)

    torch.manual_seed(504)
    m = R_enc.CrossModalityEncoder(cm_cfg())
    c = Case("G5_crossmodality_encoder")
    c.meta.update(cfg=json.loads(json.dumps(cm_cfg())))
    run_with_grads(c, m, {"vision": v, "language": l, "vmask": vm, "lmask": lm}, call2, ["vision", "language"])
    finish(c)


# ---------------------------------------------------------------- G6 pos / masks
def g6():
    c = Case("G6_pos_masks")
    for n, d in [(3, 8), (100, 512)]:
        pe = R_pos.SinusoidPositionalEmbedding(d)(torch.zeros(2, n, d))
        c.add("out", f"sinusoid_{n}_{d}", pe[0] if n == 3 else pe[0, ::33, ::37])
        assert torch.equal(pe[0], pe[1])
    c.add("out", "table_6_8_pad0", R_utils.sinusoid_encoding_table(6, 8, padding_idx=0))
    c.add("out", "table_6_8_nopad", R_utils.sinusoid_encoding_table(6, 8))
    toks = torch.tensor([[5, 6, 7, 0, 0], [1, 2, 3, 4, 5], [0, 0, 0, 0, 0]])
    c.add("in", "tokens", toks)
    pm = R_utils.generate_padding_mask(toks, 0)
    c.add("out", "padmask_tokens", pm)
    gen = torch.Generator().manual_seed(14)
    f = feats(2, 4, 6, gen, pad_rows={0: [3]})
    c.add("in", "feats", f)
    c.add("out", "padmask_feats", R_utils.generate_padding_mask(f, 0))
    sm = R_utils.generate_sequential_mask(5)
    c.add("out", "seqmask_5", sm)
    c.add("out", "selfmask", R_utils.generate_self_attention_masks(pm, sm))
    c.meta["dtypes"] = {"padmask": str(pm.dtype), "seqmask": str(sm.dtype)}
    finish(c)


class FakeVocab:
    """The 5 attributes Decoder/UsualEmbedding read (decoders.py:35-44,
    text_embeddings.py:61-64)."""
    max_answer_length = 6
    padding_idx = 0
    bos_idx = 1
    eos_idx = 2

    def __len__(self):
        return 11


def dec_cfg(layers=2):
    return ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=D, LAYERS=layers,
        ATTENTION=dict(SELF_ATTENTION=att_cfg(can_be_stateful=True), ENC_ATTENTION=att_cfg()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=D, D_EMBEDDING=16,
                            WORD_EMBEDDING=None, WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))


# ---------------------------------------------------------------- G7 decoder
def g7():
    gen = torch.Generator().manual_seed(15)
    enc = feats(2, 9, D, gen, pad_rows={1: [7, 8]})
    emask = R_utils.generate_padding_mask(enc, 0)
    toks = torch.tensor([[1, 5, 6, 7, 2, 0], [1, 8, 9, 2, 0, 0]])

    torch.manual_seed(701)
    m = R_dec.DecoderLayer(dec_cfg().ATTENTION)
    x = feats(2, 6, D, gen)
    pm = R_utils.generate_padding_mask(toks, 0)
    sam = R_utils.generate_self_attention_masks(pm, R_utils.generate_sequential_mask(6))
    c = Case("G7_decoder_layer")
    c.meta.update(cfg=json.loads(json.dumps(dec_cfg().ATTENTION)))
    run_with_grads(c, m, {"x": x, "enc": enc, "self_mask": sam, "enc_mask": emask},
                   lambda mod, ins: {"out": mod(queries=ins["x"], keys=ins["enc"], values=ins["enc"],
                                                self_attention_mask=ins["self_mask"],
                                                enc_attention_mask=ins["enc_mask"])}, ["x", "enc"])
    finish(c)

    torch.manual_seed(702)
    m = R_dec.Decoder(dec_cfg(), FakeVocab())
    c = Case("G7_decoder")
    c.meta.update(cfg=json.loads(json.dumps(dec_cfg())), vocab=dict(len=11, max_answer_length=6, padding_idx=0))
    run_with_grads(c, m, {"tokens": toks, "enc": enc, "enc_mask": emask},
                   lambda mod, ins: {"logp": mod(answer_tokens=ins["tokens"], encoder_features=ins["enc"],
                                                 encoder_attention_mask=ins["enc_mask"])}, ["enc"])
    # stateful greedy-style stepping with the teacher tokens (no padding tokens fed)
    m.eval()
    with torch.no_grad():
        with m.statefulness(2):
            steps = []
            for t in range(4):
                steps.append(m(toks[:, t:t + 1], enc, emask))
            c.add("out", "running_seq_final", m.running_seq)
        c.add("out", "step_logp", torch.cat(steps, 1))
    finish(c)


# ---------------------------------------------------------------- G8 pointers
def g8():
    gen = torch.Generator().manual_seed(16)
    torch.manual_seed(801)
    m = R_m4c.OcrPtrNet(24)
    q3 = torch.randn(2, 4, 24, generator=gen)
    k = feats(2, 5, 24, gen, pad_rows={1: [3, 4]})
    mask = R_utils.generate_padding_mask(k, 0)
    c = Case("G8_ocrptr")
    c.meta.update(hidden=24)
    run_with_grads(c, m, {"q3": q3, "q2": q3[:, 0].clone(), "k": k, "mask": mask},
                   lambda mod, ins: {"s3": mod(ins["q3"], ins["k"], ins["mask"]),
                                     "s2": mod(ins["q2"], ins["k"], ins["mask"])}, ["q3", "q2", "k"])
    finish(c)

    torch.manual_seed(802)
    cfg = ConfigNode(dict(D_MODEL=24))
    m = R_im4c.DynamicPointerNetwork(cfg)
    qmask = torch.tensor([[False, False, False, True], [False, False, True, True]])[:, None, None, :]
    c = Case("G8_dynptr_query_axis")
    c.meta.update(d_model=24, note="models/iterative_m4c.py:18-32 (query-axis fill); the key-axis twin "
                  "models/m4c.py:19-33 cannot be imported here (pytorch_transformers missing) and is "
                  "pinned only through the shared bilinear term.")
    run_with_grads(c, m, {"q": q3, "k": k, "qmask": qmask},
                   lambda mod, ins: {"scores": mod(ins["q"], ins["k"], ins["qmask"])}, ["q", "k"])
    finish(c)


def _reference_class(relpath, name, namespace):
    """One class of a reference file whose MODULE cannot be imported here (models/m4c.py needs pytorch_transformers):
    the class statement alone is parsed out of the file with ``ast`` and executed, in memory, in ``namespace`` (the
    modules its body uses).  Nothing is written anywhere; what is executed is the reference's own class body."""
    import ast
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    ns = dict(namespace)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def g8k():
    """a17, key-axis variant: the REAL ``DynamicPointerNetwork`` of models/m4c.py:19-33 (boolean mask over the OCR
    tokens -> -inf columns), same inputs as the query-axis case."""
    import math
    cls = _reference_class("models/m4c.py", "DynamicPointerNetwork", dict(torch=torch, nn=torch.nn, np=np, math=math))
    gen = torch.Generator().manual_seed(16)
    q3 = torch.randn(2, 4, 24, generator=gen)
    k = feats(2, 5, 24, gen, pad_rows={1: [3, 4]})
    torch.manual_seed(803)
    m = cls(ConfigNode(dict(D_MODEL=24)))
    kmask = torch.tensor([[False, False, False, False, True], [False, False, False, True, True]])[:, None, None, :]
    c = Case("G8_dynptr_key_axis")
    c.meta.update(d_model=24, note="models/m4c.py:19-33 (key-axis fill): the class body executed from the reference "
                  "file by ast extraction (its module needs pytorch_transformers, absent here)")
    run_with_grads(c, m, {"q": q3, "k": k, "kmask": kmask},
                   lambda mod, ins: {"scores": mod(ins["q"], ins["k"], ins["kmask"])}, ["q", "k"])
    finish(c)


# ---------------------------------------------------------------- G9 full size checksum
def g9():
    torch.manual_seed(901)
    sa = attention_config()
    enc_cfg = ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa))
    g_cfg = ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=6,
                            SELF_ATTENTION=sa, GUIDED_ATTENTION=attention_config()))
    text_enc = R_enc.Encoder(enc_cfg).eval()
    vis_enc = R_enc.GuidedAttentionEncoder(g_cfg).eval()
    gen = torch.Generator().manual_seed(902)
    v = torch.randn(4, 100, 512, generator=gen)
    l = torch.randn(4, 20, 512, generator=gen)
    v[1, 90:] = 0
    l[2, 12:] = 0
    v.requires_grad_(True)
    l.requires_grad_(True)
    vm, lm = R_utils.generate_padding_mask(v, 0), R_utils.generate_padding_mask(l, 0)
    lo = text_enc(features=l, padding_mask=lm)
    vo = vis_enc(vision_features=v, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    # NB: mean(LN-output^2) is analytically constant (its gradient is pure rounding noise), so the
    # checksum loss is a random linear functional of the outputs instead.
    wv = torch.randn(vo.shape, generator=gen)
    wl = torch.randn(lo.shape, generator=gen)
    loss = (vo * wv).mean() + (lo * wl).mean()
    loss.backward()
    c = Case("G9_mcan_fullsize_checksum")
    c.meta.update(seed_weights=901, seed_inputs=902, B=4, NV=100, NT=20, D=512, L=6,
                  recipe="torch.manual_seed(901); Encoder(cfg); GuidedAttentionEncoder(cfg); "
                         "gen=Generator(902); v=randn(4,100,512); l=randn(4,20,512); v[1,90:]=0; l[2,12:]=0; "
                         "wv=randn(vo.shape); wl=randn(lo.shape) (same gen); loss = (vo*wv).mean()+(lo*wl).mean()",
                  n_params=sum(p.numel() for p in text_enc.parameters()) + sum(p.numel() for p in vis_enc.parameters()))
    c.add("out", "loss", loss)
    c.add("out", "vision_first8", vo[0, 0, :8])
    c.add("out", "language_first8", lo[0, 0, :8])
    c.add("out", "vision_stats", torch.stack([vo.mean(), vo.abs().max(), vo.std()]))
    c.add("out", "language_stats", torch.stack([lo.mean(), lo.abs().max(), lo.std()]))
    c.add("out", "vision_sample", vo[:, ::17, ::61])
    c.add("out", "language_sample", lo[:, ::3, ::61])
    c.add("out", "gin_vision_sample", v.grad[:, ::17, ::61])
    c.add("out", "gin_language_sample", l.grad[:, ::3, ::61])
    c.add("out", "input_checksum", torch.stack([v.detach().sum(), l.detach().sum()]))
    names, norms = [], []
    for pre, mod in (("self_encoder.", text_enc), ("guided_encoder.", vis_enc)):
        for k, p in mod.named_parameters():
            names.append(pre + k)
            norms.append(p.grad.norm())
    c.meta["grad_norm_names"] = names
    c.add("out", "grad_norms", torch.stack(norms))
    c.add("out", "weight_checksum", torch.stack([sum(p.detach().sum() for p in text_enc.parameters()),
                                                  sum(p.detach().sum() for p in vis_enc.parameters())]))
    finish(c)


# ---------------------------------------------------------------- G10 manifests
def g10():
    sa = attention_config()
    out = {}
    enc_cfg = ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=3, SELF_ATTENTION=sa))
    out["Encoder"] = {k: list(v.shape) for k, v in R_enc.Encoder(enc_cfg).state_dict().items()}
    g_cfg = ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=3,
                            SELF_ATTENTION=sa, GUIDED_ATTENTION=sa))
    out["GuidedAttentionEncoder"] = {k: list(v.shape) for k, v in R_enc.GuidedAttentionEncoder(g_cfg).state_dict().items()}
    cm = ConfigNode(dict(ARCHITECTURE="CrossModalityEncoder", D_MODEL=512, LAYERS=3,
                         VISION_LANGUAGE_ATTENTION=sa, LANGUAGE_VISION_ATTENTION=sa,
                         VISION_SELF_ATTENTION=sa, LANGUAGE_SELF_ATTENTION=sa))
    out["CrossModalityEncoder"] = {k: list(v.shape) for k, v in R_enc.CrossModalityEncoder(cm).state_dict().items()}
    out["CoAttentionEncoder"] = {k: list(v.shape) for k, v in R_enc.CoAttentionEncoder(cm).state_dict().items()}
    dcfg = dec_cfg(3)
    out["Decoder_D32_V11"] = {k: list(v.shape) for k, v in R_dec.Decoder(dcfg, FakeVocab()).state_dict().items()}
    mha = R_att.MultiHeadAttention(attention_config(use_aoa=True, can_be_stateful=True))
    out["MultiHeadAttention_aoa_stateful"] = {k: list(v.shape) for k, v in mha.state_dict().items()}
    out["OcrPtrNet_768"] = {k: list(v.shape) for k, v in R_m4c.OcrPtrNet(768).state_dict().items()}
    with open(os.path.join(HERE, "G10_state_dict_manifest.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("G10 manifest:", {k: len(v) for k, v in out.items()})


# ---------------------------------------------------------------- G11 train step
def g11():
    from torch.optim import Adam
    from torch.optim.lr_scheduler import LambdaLR
    torch.manual_seed(1101)
    enc_cfg = ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=D, LAYERS=1, SELF_ATTENTION=att_cfg(dropout=0.0)))
    m = R_enc.Encoder(enc_cfg)
    head = torch.nn.Linear(D, 5)
    m.train()
    params = list(m.parameters()) + list(head.parameters())
    warmup, d_model = 4, D
    optim = Adam(params, lr=1.0, betas=(0.9, 0.98))
    sched = LambdaLR(optim, lambda s: (d_model ** -.5) * min((s + 1) ** -.5, (s + 1) * warmup ** -1.5))
    loss_fn = torch.nn.NLLLoss(ignore_index=0)
    gen = torch.Generator().manual_seed(17)
    x = feats(4, 5, D, gen, pad_rows={3: [4]})
    y = torch.tensor([1, 2, 0, 4])
    c = Case("G11_train_two_steps")
    c.meta.update(cfg=json.loads(json.dumps(enc_cfg)), warmup=warmup, lr=1.0, betas=[0.9, 0.98])
    c.weights(m)
    c.add("w", "head.weight", head.weight)
    c.add("w", "head.bias", head.bias)
    c.add("in", "x", x)
    c.add("in", "y", y)
    losses, lrs = [], []
    for step in range(2):
        mask = R_utils.generate_padding_mask(x, 0)
        out = torch.log_softmax(head(m(features=x, padding_mask=mask).mean(1)), -1)
        optim.zero_grad()
        loss = loss_fn(out, y)
        loss.backward()
        lrs.append(optim.param_groups[0]["lr"])
        optim.step()
        losses.append(loss.item())
        sched.step()
    c.add("out", "losses", torch.tensor(losses))
    c.add("out", "lrs", torch.tensor(lrs, dtype=torch.float64))
    for k, v in m.state_dict().items():
        c.add("out", "w2/" + k, v)
    c.add("out", "w2/head.weight", head.weight)
    c.add("out", "w2/head.bias", head.bias)
    finish(c)


# ---------------------------------------------------------------- G12 full MCAN model ("next" rows 2 + 4)
def mcan_cfg():
    att = dict(att_cfg())
    mlp = dict(D_MODEL=D, DROPOUT=0.1)
    return ConfigNode(dict(
        ARCHITECTURE="MCAN", DEVICE="cpu", D_MODEL=D,
        VISION_EMBEDDING=dict(ARCHITECTURE="FeatureEmbedding", D_FEATURE=20, D_MODEL=D, DROPOUT=0.1),
        TEXT_EMBEDDING=dict(ARCHITECTURE="LSTMTextEmbedding", D_MODEL=D, D_EMBEDDING=12, DROPOUT=0.1,
                            WORD_EMBEDDING=None, WORD_EMBEDDING_CACHE=None),
        SELF_ENCODER=dict(ARCHITECTURE="Encoder", D_MODEL=D, LAYERS=2, SELF_ATTENTION=att),
        GUIDED_ENCODER=dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=D, LAYERS=2, SELF_ATTENTION=att,
                            GUIDED_ATTENTION=att),
        VISION_ATTR_REDUCE=mlp, TEXT_ATTR_REDUCE=mlp))


class ModelVocab:
    padding_idx = 0
    total_answers = 7

    def __len__(self):
        return 11


def g12():
    """models/mcan.py MCAN end to end.  vision_embeddings.py imports two names that transformers 5 no longer
    exports (ViTFeatureExtractor; unused by FeatureEmbedding): they are aliased in memory for the import."""
    import transformers
    if not hasattr(transformers, "ViTFeatureExtractor"):
        transformers.ViTFeatureExtractor = getattr(transformers, "ViTImageProcessor", object)
    import models.modules.vision_embeddings  # noqa: F401  (registers FeatureEmbedding)
    import models.mcan as R_mcan
    torch.manual_seed(1201)
    cfg = mcan_cfg()
    m = R_mcan.MCAN(cfg, ModelVocab())
    with torch.no_grad():
        m.layer_norm.weight.uniform_(0.5, 1.5)
        m.layer_norm.bias.normal_(0, 0.1)
    gen = torch.Generator().manual_seed(33)
    regions = feats(3, 9, 20, gen, pad_rows={1: [7, 8], 2: [4, 5, 6, 7, 8]})
    tokens = torch.tensor([[3, 4, 5, 6, 7, 0], [8, 9, 10, 3, 0, 0], [4, 4, 5, 0, 0, 0]])
    c = Case("G12_mcan_model")
    c.meta.update(cfg=json.loads(json.dumps(cfg, default=dict)), total_answers=7, vocab_len=11)

    def call(mod, ins):
        return {"logp": mod(types.SimpleNamespace(region_features=ins["regions"], question_tokens=ins["tokens"]))}
    run_with_grads(c, m, {"regions": regions, "tokens": tokens}, call, ["regions"])
    finish(c)


# ---------------------------------------------------------------- G13 M4C multimodal transformer ("next" row 3)
def bert_cfg():
    from transformers import BertConfig
    return BertConfig(hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96)


def g13():
    """(a) the third-party layer body: installed transformers BertEncoder, called as mmf_m4c.py:349-351 does;
    (b) the reference's PrevPredEmbeddings; (c) the reference's MMT.forward (mask construction, concatenation,
    slicing) run as an unbound method on a stand-in ``self`` -- MMT.__init__ itself does not run under
    transformers 5 (init_weights API drift)."""
    import transformers
    from transformers.models.bert.modeling_bert import BertEncoder
    cfg = bert_cfg()
    meta_cfg = dict(hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                    layer_norm_eps=cfg.layer_norm_eps, hidden_dropout_prob=cfg.hidden_dropout_prob,
                    attention_probs_dropout_prob=cfg.attention_probs_dropout_prob)
    # (a)
    torch.manual_seed(1301)
    enc = BertEncoder(cfg)
    with torch.no_grad():
        for n_, p_ in enc.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0, 0.1)
            elif "LayerNorm.weight" in n_:
                p_.uniform_(0.5, 1.5)
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(3, 9, 48, generator=gen)
    pad = torch.zeros(3, 1, 1, 9)
    pad[1, ..., 5:7] = -10e4
    ext = pad.repeat(1, 1, 9, 1)
    ext[:, :, -3:, -3:] = R_utils.generate_sequential_mask(3)
    c = Case("G13_bert_encoder")
    c.meta.update(cfg=meta_cfg, transformers=transformers.__version__)
    run_with_grads(c, enc, {"x": x, "mask": ext},
                   lambda mod, ins: {"out": mod(ins["x"], ins["mask"], head_mask=[None] * 2)[0]}, ["x"])
    finish(c)
    # (b) + (c)
    torch.manual_seed(1302)
    prev = R_m4c.PrevPredEmbeddings(cfg)
    enc2 = BertEncoder(cfg)
    with torch.no_grad():
        for ln in (prev.ans_layer_norm, prev.ocr_layer_norm, prev.emb_layer_norm):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.1)
    holder = torch.nn.Module()
    holder.prev_pred_embeddings, holder.encoder = prev, enc2
    holder.config = cfg
    txt, obj, ocr = feats(2, 4, 48, gen), feats(2, 5, 48, gen), feats(2, 3, 48, gen)
    tmask, omask, cmask = torch.zeros(2, 1, 1, 4), torch.zeros(2, 1, 1, 5), torch.zeros(2, 1, 1, 3)
    tmask[0, ..., 3:] = -10e4
    omask[1, ..., 2:] = -10e4
    cmask[1, ..., 2:] = -10e4
    ans = torch.randn(6, 48, generator=gen)
    prev_inds = torch.tensor([[1, 7, 0, 3], [5, 6, 8, 2]])  # < 6: fixed vocabulary, >= 6: OCR token
    c = Case("G13_mmt")
    c.meta.update(cfg=meta_cfg, transformers=transformers.__version__)

    def call(mod, ins):
        out = R_m4c.MMT.forward(mod, ins["txt"], ins["tmask"], ins["obj"], ins["omask"], ins["ocr"], ins["cmask"],
                                ins["ans"], ins["prev_inds"])
        dec = mod.prev_pred_embeddings(ins["ans"], ins["ocr"], ins["prev_inds"])
        return {"seq": out["mmt_seq_output"], "txt_out": out["mmt_txt_output"], "ocr_out": out["mmt_ocr_output"],
                "dec_out": out["mmt_dec_output"], "dec_emb": dec}
    run_with_grads(c, holder, dict(txt=txt, tmask=tmask, obj=obj, omask=omask, ocr=ocr, cmask=cmask, ans=ans,
                                   prev_inds=prev_inds), call, ["txt", "obj", "ocr", "ans"])
    finish(c)


# ---------------------------------------------------------------- G14 memory-augmented attention (row 4)
def g14():
    torch.manual_seed(1401)
    cfg = att_cfg()
    cfg["MEMORY"] = 3
    m = R_att.AugmentedMemoryScaledDotProductAttention(cfg)
    with torch.no_grad():
        for lin in (m.fc_q, m.fc_k, m.fc_v, m.fc_o):
            lin.bias.normal_(0, 0.1)
    gen = torch.Generator().manual_seed(17)
    q = feats(3, 5, D, gen)
    kv = feats(3, 7, D, gen, pad_rows={1: [5, 6], 2: list(range(7))})
    mask = R_utils.generate_padding_mask(kv, 0)
    c = Case("G14_memory_sdpa")
    c.meta.update(cfg=dict(cfg))

    def call(mod, ins):
        out, att = mod(ins["queries"], ins["keys"], ins["values"], attention_mask=ins["mask"])
        return {"out": out, "att": att}
    run_with_grads(c, m, {"queries": q, "keys": kv, "values": kv.clone(), "mask": mask}, call,
                   ["queries", "keys", "values"])
    finish(c)


# ---------------------------------------------------------------- G15 adaptive / geometry attention (row 4)
def g15():
    """Adaptive attention: the real class, forward + gradients.  Geometry attention: upstream's forward raises
    NameError on every call (attentions.py:134-137), so what the reference CAN give is pinned: its own
    box_relational_embedding outputs (both embedding forms) and its constructor's parameter shapes."""
    torch.manual_seed(1501)
    cfg = att_cfg()
    m = R_att.AdaptiveScaledDotProductAttention(cfg)
    with torch.no_grad():
        for lin in (m.fc_q, m.fc_k, m.fc_v, m.fc_o, m.fc_s):
            lin.bias.normal_(0, 0.1)
    gen = torch.Generator().manual_seed(19)
    q = feats(3, 5, D, gen)
    kv = feats(3, 7, D, gen, pad_rows={1: [5, 6]})
    sig = feats(3, 5, D, gen)
    mask = R_utils.generate_padding_mask(kv, 0)
    c = Case("G15_adaptive_sdpa")
    c.meta.update(cfg=dict(cfg))

    def call(mod, ins):
        out, att = mod(ins["queries"], ins["keys"], ins["values"], ins["signals"], attention_mask=ins["mask"])
        return {"out": out, "att": torch.cat(att, dim=2)}
    run_with_grads(c, m, {"queries": q, "keys": kv, "values": kv.clone(), "signals": sig, "mask": mask}, call,
                   ["queries", "keys", "values", "signals"])
    finish(c)

    c = Case("G15_box_geometry")
    boxes = torch.rand(2, 6, 4, generator=gen)
    boxes[..., 2:] = boxes[..., :2] + 0.05 + boxes[..., 2:] * 0.5  # x_max > x_min, y_max > y_min
    c.add("in", "boxes", boxes)
    c.add("out", "trig", R_utils.box_relational_embedding(boxes, dim_g=DK, trignometric_embedding=True))
    c.add("out", "plain", R_utils.box_relational_embedding(boxes, dim_g=4, trignometric_embedding=False))
    gcfg = att_cfg()
    gcfg["TRIGNOMETRIC_EMBEDDING"] = True
    gm = R_att.AugmentedGeometryScaledDotProductAttention(gcfg)
    c.meta.update(cfg=dict(gcfg), dim_g=DK, state_dict_shapes={k: list(v.shape) for k, v in gm.state_dict().items()})
    finish(c)


# ---------------------------------------------------------------- G16 beam search through the reference's own classes
class GenVocab:
    """What BaseTransformer / Decoder / UsualEmbedding read from a vocab (base_transformer.py:13-16,34; decoders.py:35-44)."""
    max_answer_length = 8
    padding_idx = 0
    bos_idx = 1
    eos_idx = 2

    def __len__(self):
        return 13


def g16():
    """The reference's UNMODIFIED ``BaseTransformer.beam_search`` (base_transformer.py:46-54: statefulness ->
    encoder_forward -> BeamSearch.apply) over its own ``Decoder`` and ``BeamSearch`` (beam_search.py:4-118), beam 1 and
    3, all beams returned.  The vocabulary projection is scaled up so that candidate scores are well separated (a search
    over near-ties pins nothing), and the seed is the first one for which (i) some sample's best sequence reaches <eos>
    early (after 1-4 words), (ii) some sample's does not reach it at all, (iii) no two candidates that decide a selection are closer than
    5e-3, (iv) at least 48 words over all beams were chosen while live and before any pad word was fed back (those
    are comparable with a teacher-forced pass), (v) a live sequence emits the pad word at least once -- recorded in ``meta``."""
    class Gen(R_bt.BaseTransformer):
        def __init__(self, cfg, vocab):
            super().__init__(cfg, vocab)
            self.device = torch.device("cpu")
            self.decoder = R_dec.Decoder(cfg, vocab)

        def encoder_forward(self, inp):
            return inp["enc"], inp["enc_mask"]

    class GapSearch(R_bs.BeamSearch):  # same search; records how close the deciding candidates were
        gaps = []

        def select(self, candidate_logprob):
            v, _ = torch.sort(candidate_logprob.view(self.b_s, -1), -1, descending=True)
            GapSearch.gaps.append(float((v[:, :self.beam_size] - v[:, 1:self.beam_size + 1]).min()))
            return super().select(candidate_logprob)

    vocab, b_s = GenVocab(), 5
    T = vocab.max_answer_length
    for seed in range(400):
        gen = torch.Generator().manual_seed(1600 + seed)
        enc = feats(b_s, 9, D, gen, pad_rows={1: [7, 8], 3: [5, 6, 7, 8]})
        emask = R_utils.generate_padding_mask(enc, 0)
        torch.manual_seed(1600 + seed)
        m = Gen(dec_cfg(), vocab)
        m.eval()
        with torch.no_grad():
            m.decoder.fc.weight.mul_(12.0)
        res, ok = {}, True
        GapSearch.gaps = []
        saved = R_bt.BeamSearch
        R_bt.BeamSearch = GapSearch
        try:
            with torch.no_grad():
                for beam in (1, 3):
                    res[beam] = m.beam_search({"enc": enc, "enc_mask": emask}, batch_size=b_s, beam_size=beam,
                                              out_size=beam)
        finally:
            R_bt.BeamSearch = saved
        top = res[3][0][:, 0]
        eos_pos = [(row == vocab.eos_idx).nonzero() for row in top]
        early = sum(1 for e in eos_pos if len(e) and 1 <= int(e[0]) <= 4)
        never = sum(1 for e in eos_pos if len(e) == 0)
        g1_eos = int((res[1][0].reshape(b_s, T) == vocab.eos_idx).any(-1).sum())
        allb = res[3][0].reshape(b_s * 3, T)
        is_eos = (allb == vocab.eos_idx).long()
        live = (torch.cumsum(is_eos, 1) - is_eos) == 0            # no <eos> BEFORE this position
        fed = torch.cat([torch.ones(b_s * 3, 1, dtype=torch.long), allb[:, :-1]], 1)
        clean = torch.cumsum((fed == vocab.padding_idx).long(), 1) == 0   # no pad word fed back so far
        n_scored = int((live & clean).sum())
        n_live_pad = int((live & (allb == vocab.padding_idx)).sum())
        if (early >= 1 and never >= 1 and 1 <= g1_eos < b_s and min(GapSearch.gaps) > 5e-3 and n_scored >= 48
                and n_live_pad >= 1):
            break
    else:
        raise RuntimeError("no seed satisfied the G16 conditions")
    c = Case("G16_beam_search")
    c.meta.update(cfg=json.loads(json.dumps(dec_cfg())), seed=1600 + seed, min_gap=min(GapSearch.gaps),
                  vocab=dict(len=len(vocab), max_answer_length=T, padding_idx=0, bos_idx=1, eos_idx=2),
                  early_eos=early, never_eos=never, fc_scale=12.0, n_scored=n_scored,
                  n_live_pad=n_live_pad)
    for k, v in m.decoder.state_dict().items():
        c.add("w", k, v)
    c.add("in", "enc", enc)
    c.add("in", "enc_mask", emask)
    for beam in (1, 3):
        toks, lp = res[beam]
        c.add("out", f"beam{beam}_tokens", toks.reshape(b_s, beam, T))
        c.add("out", f"beam{beam}_logp", lp.reshape(b_s, beam, T))
    # the model the search walked, for the record: states registered by BaseTransformer + the decoder's
    c.meta["n_states"] = len(list(m.states()))
    finish(c)


# ---------------------------------------------------------------- G17 LSTM text embedding ("next" row 2)
def g17():
    """models/modules/text_embeddings.py:221-246 LSTMTextEmbedding (the reference class, unmodified): embedding -> proj ->
    dropout (eval: off) -> torch LSTM over ALL positions, padded ones included; forward + every parameter gradient
    (the embedding's padding row gets none); a second sequence length exercises a different number of steps."""
    cfg = ConfigNode(dict(ARCHITECTURE="LSTMTextEmbedding", D_MODEL=D, D_EMBEDDING=12, DROPOUT=0.1,
                          WORD_EMBEDDING=None, WORD_EMBEDDING_CACHE=None))
    torch.manual_seed(1701)
    m = R_txt.LSTMTextEmbedding(cfg, ModelVocab())
    tokens = torch.tensor([[3, 4, 5, 6, 7, 8, 9, 0], [8, 9, 10, 3, 0, 0, 0, 0], [4, 4, 5, 0, 0, 0, 0, 0],
                           [0, 0, 0, 0, 0, 0, 0, 0], [10, 9, 8, 7, 6, 5, 4, 3]])
    c = Case("G17_lstm_text_embedding")
    c.meta.update(cfg=json.loads(json.dumps(cfg, default=dict)), vocab_len=11, total_answers=7)

    def call(mod, ins):
        feats_, (pad, seq) = mod(ins["tokens"])
        return {"features": feats_, "pad_mask": pad, "seq_mask": seq}
    run_with_grads(c, m, {"tokens": tokens}, call, [])
    finish(c)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", help="e.g. g12 (default: all)")
    todo = ap.parse_args().cases
    table = dict(g1=g1, g2=g2, g3=g3, g4=g4, g5=g5, g6=g6, g7=g7, g8=g8, g8k=g8k, g9=g9, g10=g10, g11=g11, g12=g12, g13=g13, g14=g14, g15=g15, g16=g16, g17=g17)
    mpath = os.path.join(HERE, "manifest.json")
    if todo and os.path.exists(mpath):
        manifest.update(json.load(open(mpath))["cases"])
    for name in (todo or list(table)):
        table[name]()
    with open(mpath, "w") as f:
        json.dump(dict(torch=torch.__version__, cases=manifest), f, indent=1, sort_keys=True)
    print("total bytes:", sum(v["bytes"] for v in manifest.values()))
