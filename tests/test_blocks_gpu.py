"""Block-level parity (forward, input grads, weight grads) of each fused HIP block against the
oracle at BASELINE layer sizes (D=512, H=8, dff=2048, 100 regions x 20 tokens, padded samples).
fp32 mode: rel-L2 <= 2e-4 everywhere; bf16 mode: rel-L2 <= 2e-2 (stated per assert)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32, BF16 = torch.float32, torch.bfloat16


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape
    return ((a - b).norm() / max(b.norm().item(), 1e-30)).item()


@pytest.fixture(params=[F32, BF16], ids=["fp32", "bf16"])
def mode(request):
    import openvivqa_amd as A
    A.set_compute_dtype(request.param)
    yield request.param
    A.set_compute_dtype(BF16)


def _inputs(B=4, nv=100, nl=20, D=512, seed=0):
    import oracle as O
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, nv, D, generator=g)
    l = torch.randn(B, nl, D, generator=g)
    v[1, 90:] = 0
    l[0, 12:] = 0
    if B > 3:
        v[2, 64:] = 0
        l[3, 8:] = 0
    return v, l, O.padding_mask(v, 0), O.padding_mask(l, 0)


EMU_GRAD_BAR = 1.5e-2  # bf16 mode, every gradient tensor against the bf16-emulating oracle (relative L2)


def _compare(name, mode, oracle_mod, hip_mod, call, inputs, grad_names):
    """fp32 mode: outputs and every gradient against the fp32 oracle (rel-L2 2e-5 / 2e-4).  bf16 mode: outputs against
    the fp32 oracle (1e-2) and every gradient -- inputs and each parameter -- against the oracle in bf16-emulation mode
    (oracle.emulate_bf16), EMU_GRAD_BAR; the gradients' distance from the fp32 oracle is recorded, not asserted (it is
    the price of bf16 storage, which the emulation shares)."""
    import oracle as O
    from conftest import parity_record as rec
    hip_mod.load_state_dict(oracle_mod.state_dict())
    oracle_mod.eval()
    hip_mod = hip_mod.to(DEV).eval()
    ins_h = {k: (v.clone().to(DEV).requires_grad_(True) if k in grad_names else v.to(DEV)) for k, v in inputs.items()}
    out_h = call(hip_mod, ins_h)
    if not isinstance(out_h, tuple):
        out_h = (out_h,)
    gen = torch.Generator().manual_seed(99)
    ws = [torch.randn(o.shape, generator=gen) for o in out_h]
    sum((o.float() * w.to(DEV)).sum() for o, w in zip(out_h, ws)).backward()

    def oracle_pass(emulate):
        oracle_mod.zero_grad(set_to_none=True)
        ins_o = {k: (v.clone().requires_grad_(True) if k in grad_names else v) for k, v in inputs.items()}
        with O.emulate_bf16(emulate):
            out_o = call(oracle_mod, ins_o)
            if not isinstance(out_o, tuple):
                out_o = (out_o,)
            sum((o * w).sum() for o, w in zip(out_o, ws)).backward()
        return out_o, {k: ins_o[k].grad for k in grad_names}, \
            {k: (None if p.grad is None else p.grad.clone()) for k, p in oracle_mod.named_parameters()}
    out_o, gin_o, gw_o = oracle_pass(False)
    tag = f"block[{name},{'fp32' if mode == F32 else 'bf16'}]"
    tol_f = 2e-5 if mode == F32 else 1e-2
    report = [(f"out{i}", rec(tag, f"out{i} vs fp32 oracle", rel_l2(a, b), tol_f), tol_f)
              for i, (a, b) in enumerate(zip(out_h, out_o))]
    if mode == F32:
        gin_r, gw_r, tol_g, what = gin_o, gw_o, 2e-4, "fp32 oracle"
    else:
        for k in grad_names:
            rec(tag, f"d{k} vs fp32 oracle (recorded)", rel_l2(ins_h[k].grad, gin_o[k]), 0.0)
        _, gin_r, gw_r = oracle_pass(True)
        tol_g, what = EMU_GRAD_BAR, "emulation"
    for k in grad_names:
        report.append((f"d{k}", rec(tag, f"d{k} vs {what}", rel_l2(ins_h[k].grad, gin_r[k]), tol_g), tol_g))
    for k, p in hip_mod.named_parameters():
        if gw_r[k] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0
            continue
        if k.endswith("fc_k.bias") or k.endswith("self.key.bias"):
            continue
        report.append((f"dW[{k}]", rec(tag, f"dW[{k}] vs {what}", rel_l2(p.grad, gw_r[k]), tol_g), tol_g))
    bad = [(n, e, t) for n, e, t in report if not e < t]
    assert not bad, f"{name}: " + ", ".join(f"{n}={e:.2e}(>{t:.0e})" for n, e, t in bad)


def _cfg(**kw):
    from openvivqa_amd.config import attention_config
    return attention_config(**kw)


def test_block_prologue(mode):
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    cfg = ConfigNode(dict(D_MODEL=512, LAYERS=0, SELF_ATTENTION=_cfg()))
    torch.manual_seed(1)
    o, h = O.OracleEncoder(cfg), M.Encoder(cfg)
    with torch.no_grad():
        o.layer_norm.weight.uniform_(0.5, 1.5)
        o.layer_norm.bias.normal_(0, 0.1)
    v, l, vm, lm = _inputs()
    _compare("prologue", mode, o, h, lambda m, i: m(i["x"], i["mask"]), {"x": v, "mask": vm}, ["x"])


def test_feature_embedding(mode):
    """"next" row 2: FeatureEmbedding (Linear 1024->512 + GELU + dropout, zero-row padding mask) forward, mask,
    input and weight gradients vs the oracle at the MCAN region-feature size."""
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    cfg = ConfigNode(dict(D_FEATURE=1024, D_MODEL=512, DROPOUT=0.1))
    torch.manual_seed(2)
    o, h = O.OracleFeatureEmbedding(cfg), M.FeatureEmbedding(cfg)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 100, 1024, generator=g)
    x[1, 90:] = 0
    x[3, 37:] = 0
    _compare("feature_embedding", mode, o, h, lambda m, i: m(i["x"])[0], {"x": x}, ["x"])
    mo, mh = o(x)[1], h.to(DEV)(x.to(DEV))[1]
    assert mh.shape == mo.shape == (4, 1, 1, 100) and torch.equal(mh.cpu(), mo.float())


def test_feature_embedding_train_dropout_consistent():
    """Train mode (fp32): the backward regenerates the forward's dropout mask -- finite differences agree."""
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    A.set_compute_dtype(F32)
    try:
        torch.manual_seed(3)
        m = M.FeatureEmbedding(ConfigNode(dict(D_FEATURE=64, D_MODEL=32, DROPOUT=0.3))).to(DEV).train()
        x = torch.randn(2, 5, 64, device=DEV, requires_grad=True)
        w = torch.randn(2, 5, 32, device=DEV)

        def f(inp):
            A.manual_seed(77)
            return (m(inp)[0] * w).sum()
        f(x).backward()
        g = x.grad.clone()
        assert float((m(x)[0] == 0).float().mean()) > 0.15  # dropout is really on
        for idx in [(0, 0, 0), (1, 3, 7), (0, 4, 63)]:
            xp, xm = x.detach().clone(), x.detach().clone()
            xp[idx] += 1e-2
            xm[idx] -= 1e-2
            with torch.no_grad():
                fd = (f(xp) - f(xm)).item() / 2e-2
            assert abs(fd - g[idx].item()) < 2e-2 * max(1.0, abs(fd)), (idx, fd, g[idx].item())
    finally:
        A.set_compute_dtype(BF16)


@pytest.mark.parametrize("kind", ["self", "cross", "general"])
def test_block_mha(mode, kind):
    import oracle as O
    import openvivqa_amd.modules as M
    torch.manual_seed(2)
    o, h = O.OracleMHA(_cfg()), M.MultiHeadAttention(_cfg())
    with torch.no_grad():
        for lin in (o.attention.fc_q, o.attention.fc_k, o.attention.fc_v, o.attention.fc_o):
            lin.bias.normal_(0, 0.1)
        o.layer_norm.weight.uniform_(0.5, 1.5)
        o.layer_norm.bias.normal_(0, 0.1)
    v, l, vm, lm = _inputs()
    if kind == "self":
        _compare("mha-self", mode, o, h, lambda m, i: m(i["x"], i["x"], i["x"], i["mask"]), {"x": v, "mask": vm}, ["x"])
    elif kind == "cross":
        _compare("mha-cross", mode, o, h, lambda m, i: m(i["x"], i["kv"], i["kv"], i["mask"]),
                 {"x": v, "kv": l, "mask": lm}, ["x", "kv"])
    else:
        l2 = torch.randn(l.shape, generator=torch.Generator().manual_seed(5))
        _compare("mha-general", mode, o, h, lambda m, i: m(i["x"], i["k"], i["v"], i["mask"]),
                 {"x": v, "k": l, "v": l2, "mask": lm}, ["x", "k", "v"])


def test_block_ffn(mode):
    import oracle as O
    import openvivqa_amd.modules as M
    torch.manual_seed(3)
    o, h = O.OraclePWFF(_cfg()), M.PositionWiseFeedForward(_cfg())
    with torch.no_grad():
        o.layer_norm.weight.uniform_(0.5, 1.5)
        o.layer_norm.bias.normal_(0, 0.1)
    v, l, vm, lm = _inputs()
    _compare("ffn", mode, o, h, lambda m, i: m(i["x"]), {"x": v}, ["x"])


def test_block_guided_layer(mode):
    import oracle as O
    import openvivqa_amd.modules as M
    torch.manual_seed(4)
    o, h = O.OracleGuidedEncoderLayer(_cfg()), M.GuidedEncoderLayer(_cfg())
    v, l, vm, lm = _inputs()
    _compare("guided-layer", mode, o, h,
             lambda m, i: m(i["v"], i["l"], i["l"], i["vm"], i["lm"]), {"v": v, "l": l, "vm": vm, "lm": lm}, ["v", "l"])


def test_sdpa_direct_with_differentiable_att(mode):
    """ScaledDotProductAttention used directly returns (out, att) and BOTH are differentiable."""
    import oracle as O
    import openvivqa_amd.modules as M
    torch.manual_seed(6)
    o, h = O.OracleSDPA(_cfg()), M.ScaledDotProductAttention(_cfg())
    v, l, vm, lm = _inputs(B=2)
    _compare("sdpa", mode, o, h, lambda m, i: m(i["x"], i["kv"], i["kv"], i["mask"]),
             {"x": v, "kv": l, "mask": lm}, ["x", "kv"])


def test_decoder_config5_shapes(mode):
    """BASELINE config 5 shapes: Decoder L=3, T=20 answer tokens, 237 encoder positions (197 ViT patches +
    40 question tokens), d=512 -- teacher-forced forward + backward vs the oracle, then stateful greedy
    stepping (projected K/V cache) vs the teacher-forced log-probs of the same prefix."""
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = 20, 0, 1, 2

        def __len__(self):
            return 1000
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=512, LAYERS=3,
        ATTENTION=dict(SELF_ATTENTION=_cfg(can_be_stateful=True), ENC_ATTENTION=_cfg()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=512, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(8)
    o, h = O.OracleDecoder(cfg, Vocab()), M.Decoder(cfg, Vocab())
    h.load_state_dict(o.state_dict(), strict=False)
    o.eval()
    h = h.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    toks = torch.randint(3, 1000, (4, 20), generator=g)
    toks[:, 0] = 1
    toks[1, 15:] = 0
    toks[3, 9:] = 0
    enc = torch.randn(4, 237, 512, generator=g)
    enc[2, 200:] = 0
    emask = O.padding_mask(enc, 0)
    w = torch.randn(4, 20, 1000, generator=g)
    eo = enc.clone().requires_grad_(True)
    lo = o(toks, eo, emask)
    (lo * w).sum().backward()
    eh = enc.clone().to(DEV).requires_grad_(True)
    lh = h(toks.to(DEV), eh, emask.to(DEV))
    (lh * w.to(DEV)).sum().backward()
    from conftest import parity_record as rec
    tag = f"decoder-config5[{'fp32' if mode == F32 else 'bf16'}]"
    tol_f = 1e-4 if mode == F32 else 1e-2
    assert rec(tag, "log-probs vs fp32 oracle", rel_l2(lh, lo), tol_f) < tol_f, rel_l2(lh, lo)
    if mode == F32:
        tol_g, what = 5e-4, "fp32 oracle"
    else:  # bf16: every gradient against the bf16-emulating oracle
        rec(tag, "d enc vs fp32 oracle (recorded)", rel_l2(eh.grad, eo.grad), 0.0)
        o.zero_grad(set_to_none=True)
        eo = enc.clone().requires_grad_(True)
        with O.emulate_bf16():
            (o(toks, eo, emask) * w).sum().backward()
        tol_g, what = EMU_GRAD_BAR, "emulation"
    assert rec(tag, f"d enc vs {what}", rel_l2(eh.grad, eo.grad), tol_g) < tol_g, rel_l2(eh.grad, eo.grad)
    go = dict(o.named_parameters())
    for k, p in h.named_parameters():
        if k.endswith("fc_k.bias") or go[k].grad is None or not p.requires_grad:
            continue
        assert rec(tag, f"dW[{k}] vs {what}", rel_l2(p.grad, go[k].grad), tol_g) < tol_g, (k, rel_l2(p.grad, go[k].grad))
    if mode == BF16:
        o.zero_grad(set_to_none=True)
        with torch.no_grad():
            lo = o(toks, enc, emask)
    # stateful decoding: feed the first 9 (non-padding) tokens one by one
    with torch.no_grad():
        with h.statefulness(4):
            steps = [h(toks[:, t:t + 1].to(DEV), enc.to(DEV), emask.to(DEV)) for t in range(9)]
            assert tuple(h.layers[0].self_attn.running_keys.shape) == (4, 9, 512)
            # beam-search style reorder of every state buffer
            perm = torch.tensor([2, 0, 3, 1], device=DEV)
            h.apply_to_states(lambda s: s.index_select(0, perm) if s.shape[0] == 4 else s)
            assert tuple(h.layers[0].self_attn.running_keys.shape) == (4, 9, 512)
    assert rel_l2(torch.cat(steps, 1), lo[:, :9]) < (1e-4 if mode == F32 else 1e-2)


def test_mmt_config4_size(mode):
    """BASELINE configs[3] shape (mmf_m4c.yaml:92-95): hidden 768, 8 heads of 96, 4 layers, intermediate 3072,
    S = 20 txt + 100 obj + 50 ocr + 12 dec = 182 under the prefix-LM mask; B = 4.  Evaluation-mode forward (what the
    12-step decode runs) and gradients vs the oracle (= installed HF BertEncoder arithmetic, golden G13)."""
    from types import SimpleNamespace
    import oracle as O
    import openvivqa_amd.modules as M
    cfg = SimpleNamespace(hidden_size=768, num_hidden_layers=4, num_attention_heads=8, intermediate_size=3072,
                          layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(4)
    o, h = O.OracleMMT(cfg), M.MMT(cfg)
    with torch.no_grad():
        for n, p in o.named_parameters():  # BERT-style init (std 0.02) instead of torch's default
            if p.dim() == 2:
                p.normal_(0, 0.02)
    g = torch.Generator().manual_seed(8)
    B = 4
    txt, obj, ocr = torch.randn(B, 20, 768, generator=g), torch.randn(B, 100, 768, generator=g), torch.randn(B, 50, 768, generator=g)
    tmask, omask, cmask = torch.zeros(B, 1, 1, 20), torch.zeros(B, 1, 1, 100), torch.zeros(B, 1, 1, 50)
    tmask[0, ..., 14:] = -10e4
    omask[1, ..., 60:] = -10e4
    cmask[2, ..., 30:] = -10e4
    ans = torch.randn(200, 768, generator=g)
    prev = torch.randint(0, 250, (B, 12), generator=g)
    ins = dict(txt=txt, tmask=tmask, obj=obj, omask=omask, ocr=ocr, cmask=cmask, ans=ans, prev=prev)

    def call(m, i):
        return m(i["txt"], i["tmask"], i["obj"], i["omask"], i["ocr"], i["cmask"], i["ans"], i["prev"])["mmt_seq_output"]
    _compare("mmt", mode, o, h, call, ins, ["txt", "obj", "ocr"])


def test_m4c_graphed_greedy_decode_equals_eager():
    """modules/mmt.GraphedGreedyDecode (every pass of the multimodal transformer replayed from one hipGraph, the early-exit
    check on the host) against M4CDecodingHead.greedy_decode, bf16, configs[3] shapes, on TWO different batches through
    the same captured graph (the second call refreshes the graph's static prefix rows, mask row and input LayerNorms): same
    number of passes, same tokens, bit-identical scores."""
    from types import SimpleNamespace
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.modules.mmt import GraphedGreedyDecode
    A.set_compute_dtype(torch.bfloat16)
    cfg = SimpleNamespace(hidden_size=768, num_hidden_layers=4, num_attention_heads=8, intermediate_size=3072,
                          layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    num_choices, max_iter, bos, eos = 300, 12, 1, 2
    torch.manual_seed(21)
    mmt, head = M.MMT(cfg).to(DEV).eval(), M.M4CDecodingHead(768, num_choices).to(DEV).eval()
    with torch.no_grad():
        head.classifier.bias[eos] = 0.3
    graphed = GraphedGreedyDecode(head, mmt, max_iter, bos, eos)
    g = torch.Generator().manual_seed(5)
    B = 8
    for trial in range(2):
        txt, obj, ocr = (torch.randn(B, n, 768, generator=g).to(DEV) for n in (20, 100, 50))
        tm, om, cm = (torch.zeros(B, 1, 1, n, device=DEV) for n in (20, 100, 50))
        tm[trial, ..., 12:] = -10e4
        cm[2 + trial, ..., 33:] = -10e4
        s0, p0, n0 = head.greedy_decode(mmt, txt, tm, obj, om, ocr, cm, max_iter, bos, eos)
        s1, p1, n1 = graphed(txt, tm, obj, om, ocr, cm)
        assert n0 == n1 and torch.equal(p0, p1), (trial, n0, n1)
        assert torch.equal(s0, s1), trial
    assert graphed.graph is not None


def test_m4c_greedy_decode_config4(mode):
    """BASELINE configs[3] (mmf_m4c.yaml: 50 OCR + 100 region tokens + 20 question tokens, hidden 768, 4 layers x 8
    heads, 12 decoding iterations, classifier || OcrPtrNet(768) scores): the product's greedy decoding loop
    (modules/mmt.py M4CDecodingHead, mmf_m4c.py:221-256) against the oracle running the same loop.  fp32 mode: the
    two loops make the same number of passes, emit the same tokens and the same final scores.  bf16 mode: near-tied
    arg-maxes may flip, so every pass of the oracle's loop is replayed with ITS prev_inds and the scores are compared;
    the product's own loop must terminate within max_iter passes."""
    from types import SimpleNamespace
    import oracle as O
    import openvivqa_amd.modules as M
    cfg = SimpleNamespace(hidden_size=768, num_hidden_layers=4, num_attention_heads=8, intermediate_size=3072,
                          layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    num_choices, max_iter, bos, eos = 300, 12, 1, 2
    torch.manual_seed(14)
    mmt_o, head_o = O.OracleMMT(cfg).eval(), O.OracleM4CDecodingHead(768, num_choices).eval()
    with torch.no_grad():
        for m in (mmt_o, head_o):
            for n, p in m.named_parameters():
                if p.dim() == 2:
                    p.normal_(0, 0.02)
        head_o.classifier.weight.normal_(0, 0.05)
        head_o.classifier.bias[eos] = 0.5  # eos reachable: the early exit of the loop is exercised for some seeds
    mmt_h, head_h = M.MMT(cfg), M.M4CDecodingHead(768, num_choices)
    mmt_h.load_state_dict(mmt_o.state_dict())
    head_h.load_state_dict(head_o.state_dict())
    mmt_h, head_h = mmt_h.to(DEV).eval(), head_h.to(DEV).eval()
    g = torch.Generator().manual_seed(9)
    B = 4
    txt, obj, ocr = (torch.randn(B, n, 768, generator=g) for n in (20, 100, 50))
    tmask, omask, cmask = torch.zeros(B, 1, 1, 20), torch.zeros(B, 1, 1, 100), torch.zeros(B, 1, 1, 50)
    tmask[0, ..., 14:] = -10e4
    omask[1, ..., 60:] = -10e4
    cmask[2, ..., 30:] = -10e4
    s_o, prev_o, n_o, trace = head_o.greedy_decode(mmt_o, txt, tmask, obj, omask, ocr, cmask, max_iter, bos, eos)
    dev = lambda t: t.to(DEV)
    s_h, prev_h, n_h = head_h.greedy_decode(mmt_h, dev(txt), dev(tmask), dev(obj), dev(omask), dev(ocr), dev(cmask),
                                            max_iter, bos, eos)
    assert s_h.shape == s_o.shape == (B, max_iter, num_choices + 50) and 1 <= n_h <= max_iter
    if mode == F32:
        assert n_h == n_o and torch.equal(prev_h.cpu(), prev_o)
        assert rel_l2(s_h, s_o) < 1e-4
    with torch.no_grad():  # every pass of the oracle's loop, replayed on the HIP path with the same prev_inds
        for prev in trace:
            r_o = head_o.scores(mmt_o(txt, tmask, obj, omask, ocr, cmask, head_o.classifier.weight, prev), cmask)
            r_h = head_h.scores(mmt_h(dev(txt), dev(tmask), dev(obj), dev(omask), dev(ocr), dev(cmask),
                                      head_h.classifier.weight, dev(prev)), dev(cmask))
            fin = torch.isfinite(r_o)
            assert torch.equal(torch.isfinite(r_h).cpu(), fin)
            a, b = torch.where(fin, r_h.cpu(), torch.zeros_like(r_o)), torch.where(fin, r_o, torch.zeros_like(r_o))
            e = ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()
            assert e < (1e-3 if mode == F32 else 1e-2), e


def test_fused_beam_reorder_equals_reference_gather():
    """Module.reorder_states (one grouped row-gather launch over every state buffer) == the reference's
    BeamSearch._expand_state applied through apply_to_states (beam_search.py:19-34), on a stateful decoder after a
    few decoding steps: K/V caches of every layer (bf16), the running self-attention mask (bool) and the running
    position counter (int64), first with cur_beam 1 -> beam 3 (t = 0), then beam 3 -> 3."""
    import copy
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = 20, 0, 1, 2

        def __len__(self):
            return 50
    A.set_compute_dtype(BF16)
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=512, LAYERS=3,
        ATTENTION=dict(SELF_ATTENTION=_cfg(can_be_stateful=True), ENC_ATTENTION=_cfg()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=512, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(2)
    dec = M.Decoder(cfg, Vocab()).to(DEV).eval()
    b_s, beam = 4, 3
    enc = torch.randn(b_s, 30, 512, device=DEV)
    emask = torch.zeros(b_s, 1, 1, 30, device=DEV)
    g = torch.Generator().manual_seed(1)

    def reference_expand(selected_beam, cur):
        def fn(s):
            shape = [int(sh) for sh in s.shape]
            bm = selected_beam
            for _ in shape[1:]:
                bm = bm.unsqueeze(-1)
            s = torch.gather(s.view(*([b_s, cur] + shape[1:])), 1, bm.expand(*([b_s, beam] + shape[1:])))
            return s.view(*([-1] + shape[1:]))
        return fn
    with torch.no_grad(), dec.statefulness(b_s):
        dec(torch.randint(4, 50, (b_s, 1), device=DEV), enc, emask)
        for cur in (1, beam):
            sel = torch.randint(0, cur, (b_s, beam), generator=g).to(DEV)
            twin = copy.deepcopy(dec)
            twin.apply_to_states(reference_expand(sel, cur))
            dec.reorder_states(sel, b_s, cur, beam)
            for a, b in zip(dec.states(), twin.states()):
                assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
            e2 = enc.repeat_interleave(beam, 0) if cur == 1 else e2
            m2 = emask.repeat_interleave(beam, 0) if cur == 1 else m2
            dec(torch.randint(4, 50, (b_s * beam, 1), device=DEV), e2, m2)  # the caches keep decoding afterwards


@pytest.mark.parametrize("beam", [1, 3])
def test_beam_search_decode_matches_oracle(beam, mode):
    """Row f1 end to end: batched beam search (openvivqa_amd.beam = beam_search.py's control flow) over the stateful
    Decoder -- in-place projected K / V caches, single-query attention kernel, encoder K / V projected once per sample
    and shared by its beams, one fused gather per reorder -- against (a) the same search driven through the reference's
    own ``apply_to_states(_expand_state)`` protocol and (b) the ORACLE decoder (fp32, CPU) under that protocol.
    fp32 mode: identical tokens, log-probabilities 1e-3.  bf16 mode: the two HIP searches agree exactly with each other
    (same kernels, same cache contents); against the oracle near-ties may flip a beam, so the scores of the oracle's
    winning sequences are compared instead (teacher-forced through the HIP decoder)."""
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.beam import BeamSearch
    from openvivqa_amd.config import ConfigNode

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = 8, 0, 1, 2

        def __len__(self):
            return 60
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=512, LAYERS=2,
        ATTENTION=dict(SELF_ATTENTION=_cfg(can_be_stateful=True), ENC_ATTENTION=_cfg()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=512, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(12)
    o, h = O.OracleDecoder(cfg, Vocab()), M.Decoder(cfg, Vocab())
    with torch.no_grad():
        o.fc.weight.mul_(6.0)  # spread the vocabulary distribution: fewer near-ties between candidates
    h.load_state_dict(o.state_dict(), strict=False)
    o.eval()
    h = h.to(DEV).eval()
    b_s, T = 5, 8
    g = torch.Generator().manual_seed(4)
    enc = torch.randn(b_s, 37, 512, generator=g)
    enc[1, 30:] = 0
    emask = O.padding_mask(enc, 0)

    def search(dec, dev, reorder, kernels=False):
        e, m = enc.to(dev), emask.to(dev)
        st = {}

        def step(t, prev, **kw):
            if t == 0:
                st["e"], st["m"] = e, m
                prev = torch.full((b_s, 1), 1, dtype=torch.long, device=dev)
            elif t == 1 and beam > 1:
                st["e"], st["m"] = e.repeat_interleave(beam, 0), m.repeat_interleave(beam, 0)
            return dec(prev, st["e"], st["m"], **kw)
        with torch.no_grad(), dec.statefulness(b_s):
            return BeamSearch(dec, step, b_s, T, 2, beam, dev, reorder=reorder,
                              logits_step=(lambda t, prev: step(t, prev, return_logits=True)) if kernels else None).apply(1)
    out_f, lp_f = search(h, DEV, "fused")
    out_r, lp_r = search(h, DEV, "reference")
    assert torch.equal(out_f, out_r) and torch.equal(lp_f, lp_r)
    # selection + bookkeeping as two kernels per step (ovqa_beam_candidates / ovqa_beam_commit): the same search
    out_k, lp_k = search(h, DEV, "fused", kernels=True)
    assert torch.equal(out_k, out_f), (out_k, out_f)
    assert (lp_k - lp_f).abs().max().item() < 1e-4
    trace = []
    out_o, lp_o = O.oracle_generate(o, enc, emask, 1, 2, beam, max_len=T, trace=trace)
    if mode == F32:
        assert torch.equal(out_f.cpu(), out_o), (out_f, out_o)
        assert rel_l2(lp_f, lp_o) < 1e-3
    # every mode, EVERY sample and beam: the HIP decoder driven statefully through the ORACLE's choices (its words fed
    # back, its source beams handed to reorder_states) gives the oracle's log-probabilities at every step -- caches,
    # reorder and decoder numerics without the near-tie flips a free-running bf16 search may take
    _replay_search_choices(h, enc, emask, trace, beam, 1, 1e-3 if mode == F32 else 1e-2, f"beam-replay[{beam}]")
    # the caches the search left behind are gone: a teacher-forced pass still matches
    with torch.no_grad():
        toks = torch.cat([torch.ones(b_s, 1, dtype=torch.long), out_o[:, :-1]], 1)
        lo = o(toks, enc, emask)
        lh = h(toks.to(DEV), enc.to(DEV), emask.to(DEV))
    assert rel_l2(lh, lo) < (1e-4 if mode == F32 else 1e-2)


def _replay_search_choices(dec, enc, emask, trace, beam, bos, bar, tag):
    """Drive the stateful HIP decoder through a recorded search (oracle.oracle_beam_search's ``trace``): at step t feed
    the recorded words, compare the log-probabilities of EVERY row and word with the recorded ones (the suite's
    normalised max error max|a - b| / max(1, max|b|) <= bar; the worst probability difference is recorded next to it),
    then reorder the caches by the recorded source beams (Module.reorder_states, the fused gather)."""
    from conftest import parity_record as rec
    b_s = enc.shape[0]
    e, m = enc.to(DEV), emask.to(DEV)
    worst, worst_p = 0.0, 0.0
    with torch.no_grad(), dec.statefulness(b_s):
        prev = torch.full((b_s, 1), bos, dtype=torch.long, device=DEV)
        for rec_t in trace:
            t, cur = rec_t["t"], rec_t["cur"]
            lp = dec(prev, e, m, encoder_group=(1 if t == 0 else beam)).float().cpu().view(b_s, cur, -1)
            ref = rec_t["step_logp"]
            worst = max(worst, float((lp - ref).abs().max() / max(1.0, float(ref.abs().max()))))
            worst_p = max(worst_p, float((lp.exp() - ref.exp()).abs().max()))
            dec.reorder_states(rec_t["from_beam"].to(DEV), b_s, cur, beam)
            prev = rec_t["word"].reshape(-1, 1).to(DEV)
    rec(tag, "log-probabilities of every step / row / word, normalised max error", worst, bar)
    rec(tag, "max |p - p_oracle| (recorded, not a bar)", worst_p, float("nan"))
    assert worst <= bar, (tag, worst)


@pytest.mark.parametrize("beam", [1, 3])
def test_beam_search_reproduces_reference_search_golden(beam, mode):
    """G16 = the reference's unmodified BaseTransformer.beam_search over its own Decoder and BeamSearch
    (base_transformer.py:46-54, beam_search.py:36-118; tasks/open_ended_task.py:135), every beam returned.  The HIP
    decoder under this package's search -- eager, fused selection kernels, and the whole decode replayed from one
    hipGraph -- fp32 mode: identical words for every beam, word scores 1e-3.  Both modes, EVERY sample and beam (no
    majority clause): the teacher-forced HIP distribution at every live, comparable position of G16's sequences is
    within the bar (1e-3 / 1e-2, normalised max) of the oracle's, whose score of G16's words is G16's; and the decoder
    replayed statefully through the oracle's search choices matches the oracle step by step."""
    import oracle as O
    import openvivqa_amd.modules as M
    from conftest import parity_record as rec
    from golden_cases import GenVocab, load_case, teacher_forced_inputs
    from openvivqa_amd.beam import GraphedBeamSearch
    from openvivqa_amd.config import ConfigNode
    case = load_case("G16_beam_search")
    vocab, cfg = GenVocab(case.meta), ConfigNode(case.meta["cfg"])
    dec = M.Decoder(cfg, vocab)
    dec.load_state_dict(case.w)
    dec = dec.to(DEV).eval()
    enc, mask = case.inputs["enc"], case.inputs["enc_mask"]
    b_s, T = enc.shape[0], vocab.max_answer_length
    ref_t, ref_lp = case.out[f"beam{beam}_tokens"], case.out[f"beam{beam}_logp"]
    tag = f"G16[{'fp32' if mode == F32 else 'bf16'},beam{beam}]"
    if mode == F32:
        for fused in (False, True):
            search = GraphedBeamSearch(dec, b_s, T, vocab.bos_idx, vocab.eos_idx, beam, out_size=beam, fused=fused)
            for use_graph in (False, True):
                toks, lp = search(enc.to(DEV), mask.to(DEV), use_graph=use_graph)
                assert torch.equal(toks.cpu().reshape(ref_t.shape), ref_t), (fused, use_graph)
                err = float((lp.cpu().reshape(ref_lp.shape) - ref_lp).abs().max())
                rec(tag, f"word scores fused={fused} graph={use_graph}", err, 1e-3)
                assert err < 1e-3
    o = O.OracleDecoder(cfg, vocab)
    o.load_state_dict(case.w)
    o.eval()
    seqs, seq_lp = ref_t.reshape(b_s * beam, T), ref_lp.reshape(b_s * beam, T)
    inp, live, clean = teacher_forced_inputs(seqs, vocab.bos_idx, vocab.eos_idx, vocab.padding_idx)
    enc_b, mask_b = enc.repeat_interleave(beam, 0), mask.repeat_interleave(beam, 0)
    with torch.no_grad():
        full_h = dec(inp.to(DEV), enc_b.to(DEV), mask_b.to(DEV)).float().cpu()      # (b_s * beam, T, |V|)
        full_o = o(inp, enc_b, mask_b)
    sel = live & clean
    assert int(sel.sum()) >= 20
    # the oracle's teacher-forced score of G16's words IS G16's score (also checked on the CPU) ...
    assert float((full_o.gather(-1, seqs.unsqueeze(-1)).squeeze(-1) - seq_lp)[sel].abs().max()) < 1e-5
    # ... and the HIP decoder's whole distribution at those positions is within the bar for EVERY sample and beam
    # (the suite's normalised max error; the vocabulary projection of G16 is scaled x12, log-probabilities reach -30)
    bar = 1e-3 if mode == F32 else 1e-2
    per_seq = []
    for r in range(b_s * beam):
        if sel[r].any():
            a, b = full_h[r][sel[r]], full_o[r][sel[r]]
            per_seq.append(float((a - b).abs().max() / max(1.0, float(b.abs().max()))))
    word_err = float((full_h.gather(-1, seqs.unsqueeze(-1)).squeeze(-1) - seq_lp)[sel].abs().max())
    rec(tag, "teacher-forced distribution at G16's positions, worst sequence (normalised max)", max(per_seq), bar)
    rec(tag, "teacher-forced score of G16's words, worst |difference| (recorded, not a bar)", word_err, float("nan"))
    assert len(per_seq) == b_s * beam and max(per_seq) <= bar, per_seq  # EVERY sample and beam
    if mode == BF16:  # what the KERNELS add on top of bf16 storage: against the oracle in bf16-emulation mode
        with torch.no_grad(), O.emulate_bf16():
            full_e = o(inp, enc_b, mask_b)
        emu_err = float((full_h - full_e)[sel].abs().max() / max(1.0, float(full_e[sel].abs().max())))
        rec(tag, "teacher-forced distribution vs the bf16-emulating oracle (normalised max)", emu_err, bar)
        assert emu_err <= bar
    trace = []
    toks_o, _ = O.oracle_generate(o, enc, mask, vocab.bos_idx, vocab.eos_idx, beam, out_size=beam, trace=trace)
    assert torch.equal(toks_o.reshape(ref_t.shape), ref_t)  # (the oracle's search is G16's: tests/test_oracle_golden.py)
    _replay_search_choices(dec, enc, mask, trace, beam, vocab.bos_idx, bar, tag + "-replay")


@pytest.mark.parametrize("beam", [1, 3])
def test_graphed_beam_search_equals_eager(beam):
    """GraphedBeamSearch: the whole decode (T decoder steps + selections + state reorders) replayed from ONE hipGraph
    gives bit-identical tokens and scores to the eager search, also on a second batch (static inputs refilled) and
    after the eager path has run in between (the module's decode caches are per decode)."""
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.beam import GraphedBeamSearch
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.utils import generate_padding_mask
    A.set_compute_dtype(BF16)

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = 10, 0, 1, 2

        def __len__(self):
            return 200
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=512, LAYERS=3,
        ATTENTION=dict(SELF_ATTENTION=_cfg(can_be_stateful=True), ENC_ATTENTION=_cfg()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=512, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(21)
    dec = M.Decoder(cfg, Vocab()).to(DEV).eval()
    with torch.no_grad():
        dec.fc.weight.mul_(6.0)
    b_s = 6
    search = GraphedBeamSearch(dec, b_s, 10, 1, 2, beam)
    g = torch.Generator().manual_seed(2)
    for trial in range(3):
        enc = torch.randn(b_s, 50, 512, generator=g)
        enc[trial % b_s, 40:] = 0
        enc = enc.to(DEV)
        mask = generate_padding_mask(enc, 0)
        out_e, lp_e = search(enc, mask, use_graph=False)
        out_g, lp_g = search(enc, mask)
        assert search.graph is not None
        assert torch.equal(out_g, out_e) and torch.equal(lp_g, lp_e), trial
    assert out_g.shape == (b_s, 10)


def test_ragged_model_dimension_is_refused_with_an_explanation_and_plain_linear_takes_the_padded_footprint():
    """a parameter whose reduction length is not a multiple of 8 (d_model = 300) is zero-padded in the
    arena, so `arena.compute(p)` is a non-contiguous corner view.  `functional.linear` takes the padded footprint and
    computes the layer; the fused feed-forward block needs d_model % 8 == 0 and says so (a RuntimeError that names the
    cause, not a bare assertion)."""
    import torch.nn as nn
    import openvivqa_amd.functional as Fn
    import openvivqa_amd.runtime as rt
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.modules.positionwise_feed_forward import PositionWiseFeedForward
    torch.manual_seed(3)
    lin = nn.Linear(300, 64).to(DEV)
    arena = rt.ensure_arena(lin)
    x = torch.randn(40, 300, device=DEV)
    with torch.no_grad():
        y = Fn.linear(x.to(arena.compute_dtype), lin, arena)
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    assert y.shape == (40, 64) and rel_l2(y, ref) < 1e-2
    ff = PositionWiseFeedForward(ConfigNode(dict(D_MODEL=300, D_FF=512, DROPOUT=0.0))).to(DEV)
    with pytest.raises(RuntimeError, match="d_model % 8 == 0"):
        ff(torch.randn(2, 5, 300, device=DEV))
