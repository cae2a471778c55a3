"""Module-level parity of the HIP path: golden vectors of the REAL reference, and the
oracle on seeded inputs at BASELINE sizes.  Everything goes through the C ABI.

Bars (north star): fp32 mode <= 1e-3, bf16 mode <= 1e-2, both as normalised max
error max|a-b| / max(1, max|b|), at every depth (the residual stream is kept in fp32 between
blocks: tests/bf16_noise_floor.py); bf16 gradients <= 3e-2 relative L2.
On top of that the bf16 path is compared with the oracle in bf16-EMULATION mode (oracle.emulate_bf16:
rounds where the HIP path stores bf16) -- the bug detector: at shallow depth only accumulation order
and fast exp / erf are left between the two (<= 2e-3 at L=1); at L=6 early 1-ulp flips have been
amplified through 30 blocks and the bar is 6e-3 (test_stack_forward_and_gradients_vs_bf16_emulating_oracle).
"""
import json
import os

import pytest
import torch

from golden_cases import CASES, GOLDEN_DIR, FakeVocab, hip_namespace, load_case, oracle_namespace, run_case

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32, BF16 = torch.float32, torch.bfloat16


def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    ia, ib = torch.isinf(a), torch.isinf(b)
    assert torch.equal(ia, ib)
    a, b = torch.where(ia, torch.zeros_like(a), a), torch.where(ib, torch.zeros_like(b), b)
    if a.numel() == 0:
        return 0.0
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / max(b.norm().item(), 1e-12)).item()


@pytest.fixture(params=[F32, BF16], ids=["fp32", "bf16"])
def mode(request):
    import openvivqa_amd as A
    A.set_compute_dtype(request.param)
    yield request.param
    A.set_compute_dtype(BF16)


# Bars of the bf16 mode, gradients: every gradient -- inputs and every parameter, small ones
# included -- is held against the oracle in bf16-EMULATION mode (oracle.emulate_bf16: rounds values and gradients where
# the HIP path stores bf16, so that accumulation order and the fast exp / erf are all that is left between the two),
# relative L2 per tensor.  A parameter whose gradient is analytically zero (fc_k.bias: softmax shift invariance) is
# pure rounding noise on both sides and is skipped.
EMU_GRAD_BAR = 1.5e-2
# bf16 gradients against the REFERENCE's fp32 gradients (golden cases; input gradients per tensor, parameter gradients all
# together), relative L2: 2e-2 since round 5 (3e-2 before; worst measured 9.7e-3, profiles/r05_parity_report.tsv)
FP32_REF_GRAD_BAR = 2e-2
# 30 chained blocks (MCAN L=6): a 1-ulp bf16 flip early in the stack is amplified by every later rounding, so the
# gap to the emulation grows with depth (scripts/parity_depth.py); every gradient tensor of the L=6 stacks is held to this
G9_EMU_BAR = 2.5e-2
ZERO_GRAD = ("fc_k.bias", "self.key.bias", "attr_reduce.fc2.bias")


def _emu_case(name):
    """Inputs' and parameters' gradients of golden case ``name`` from the oracle in bf16-emulation mode (CPU)."""
    import oracle as O
    with O.emulate_bf16():
        _, outs, gin, gw, _ = run_case(oracle_namespace(), name)
    return outs, gin, gw


@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_modules_match_reference_golden(name, mode):
    from conftest import grad_close, parity_record as rec
    case, outs, gin, gw, _ = run_case(hip_namespace(), name, device=DEV)
    tag = f"golden[{name},{'fp32' if mode == F32 else 'bf16'}]"
    fwd_tol = 1e-3 if mode == F32 else 1e-2
    for k, ref in case.out.items():
        if k in outs and outs[k] is not None:
            assert rec(tag, f"out/{k}", nerr(outs[k], ref), fwd_tol) < fwd_tol, f"{name} out/{k}: {nerr(outs[k], ref):.3e}"
    for k in case.meta["grad_none"]:
        assert gw[k] is None or float(gw[k].abs().max()) == 0.0, f"{name}: {k} must not receive a gradient"
    if mode == F32:  # fp32 mode: the reference's own gradients, normalised max error 1e-3
        for k, ref in case.gin.items():
            assert rec(tag, f"gin/{k}", nerr(gin[k], ref), 1e-3) < 1e-3, f"{name} gin/{k}"
        for k, ref in case.gw.items():
            assert gw[k] is not None, f"{name}: missing grad for {k}"
            if not k.endswith(ZERO_GRAD):
                assert rec(tag, f"gw/{k}", nerr(gw[k], ref), 1e-3) < 1e-3, f"{name} gw/{k}"
        return
    # bf16 mode: (1) against the reference's fp32 gradients, relative L2 of ALL parameter gradients taken together
    # and of each input gradient, 3e-2; (2) every tensor against the bf16-emulating oracle, EMU_GRAD_BAR
    _, egin, egw = _emu_case(name)
    for k, ref in case.gin.items():
        assert rec(tag, f"gin/{k} vs fp32 reference", rel_l2(gin[k], ref), FP32_REF_GRAD_BAR) < FP32_REF_GRAD_BAR, f"{name} gin/{k}"
        rec(tag, f"gin/{k} vs emulation", rel_l2(gin[k], egin[k]), EMU_GRAD_BAR)
        assert grad_close(rel_l2(gin[k], egin[k]), rel_l2(egin[k], ref), EMU_GRAD_BAR, tag, f"gin/{k}"), \
            f"{name} gin/{k} vs emulation: {rel_l2(gin[k], egin[k]):.3e}"
    keys = [k for k in case.gw if not k.endswith(ZERO_GRAD)]
    for k in case.gw:
        assert gw[k] is not None, f"{name}: missing grad for {k}"
    if keys:
        allh = torch.cat([gw[k].detach().double().cpu().flatten() for k in keys])
        allr = torch.cat([case.gw[k].double().flatten() for k in keys])
        assert rec(tag, "gw/* together vs fp32 reference", rel_l2(allh, allr), FP32_REF_GRAD_BAR) < FP32_REF_GRAD_BAR, f"{name} gw"
    for k in keys:
        e, fmt = rel_l2(gw[k], egw[k]), rel_l2(egw[k], case.gw[k])
        rec(tag, f"gw/{k} vs emulation", e, EMU_GRAD_BAR)
        assert grad_close(e, fmt, EMU_GRAD_BAR, tag, f"gw/{k}"), f"{name} gw/{k} vs emulation: {e:.3e} (format itself: {fmt:.3e})"


def test_state_dict_manifest_matches_reference():
    import openvivqa_amd as A
    from openvivqa_amd.config import ConfigNode, attention_config
    import openvivqa_amd.modules as M
    with open(os.path.join(GOLDEN_DIR, "G10_state_dict_manifest.json")) as f:
        man = json.load(f)
    sa = attention_config()
    cm = ConfigNode(dict(D_MODEL=512, LAYERS=3, VISION_LANGUAGE_ATTENTION=sa, LANGUAGE_VISION_ATTENTION=sa,
                         VISION_SELF_ATTENTION=sa, LANGUAGE_SELF_ATTENTION=sa))
    built = {
        "Encoder": A.build_encoder(ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=3, SELF_ATTENTION=sa))),
        "GuidedAttentionEncoder": A.build_encoder(ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512,
                                                                   LAYERS=3, SELF_ATTENTION=sa, GUIDED_ATTENTION=sa))),
        "CrossModalityEncoder": M.CrossModalityEncoder(cm),
        "CoAttentionEncoder": M.CoAttentionEncoder(cm),
        "OcrPtrNet_768": M.OcrPtrNet(768),
        "MultiHeadAttention_aoa_stateful": M.MultiHeadAttention(attention_config(use_aoa=True, can_be_stateful=True)),
    }
    for name, mod in built.items():
        got = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert got == man[name], name


def test_mha_stateful_matches_reference(mode):
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.utils import generate_sequential_mask
    case = load_case("G2_mha_stateful")
    m = M.MultiHeadAttention(ConfigNode(case.meta["cfg"]))
    m.load_state_dict(case.w, strict=False)
    m = m.to(DEV).eval()
    x = case.inputs["x"].to(DEV)
    tol = 1e-3 if mode == F32 else 1e-2
    with torch.no_grad():
        full = m(x, x, x, generate_sequential_mask(3).to(DEV))
        with m.statefulness(2):
            steps = [m(x[:, t:t + 1], x[:, t:t + 1], x[:, t:t + 1], torch.zeros(1, 1, 1, t + 1, device=DEV))
                     for t in range(3)]
            assert tuple(m.running_keys.shape) == (2, 3, 32)
        assert tuple(m.running_keys.shape) == (0, 32)
    assert nerr(full, case.out["oneshot"]) < tol
    assert nerr(torch.cat(steps, 1), case.out["steps"]) < tol


def test_decoder_stateful_steps(mode):
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    case = load_case("G7_decoder")
    m = M.Decoder(ConfigNode(case.meta["cfg"]), FakeVocab())
    m.load_state_dict(case.w, strict=False)
    m = m.to(DEV).eval()
    toks, enc, emask = (case.inputs[k].to(DEV) for k in ("tokens", "enc", "enc_mask"))
    with torch.no_grad():
        with m.statefulness(2):
            steps = [m(toks[:, t:t + 1], enc, emask) for t in range(4)]
            assert torch.equal(m.running_seq.cpu(), case.out["running_seq_final"])
    assert nerr(torch.cat(steps, 1), case.out["step_logp"]) < (1e-3 if mode == F32 else 1e-2)


def _mcan_pair(ns, layers, seed):
    from openvivqa_amd.config import ConfigNode, attention_config
    torch.manual_seed(seed)
    sa = attention_config()
    te = ns.Encoder(ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=layers, SELF_ATTENTION=sa)))
    ve = ns.GuidedAttentionEncoder(ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=layers,
                                                   SELF_ATTENTION=sa, GUIDED_ATTENTION=attention_config())))
    return te, ve


def test_fullsize_mcan_against_reference_checksum(mode):
    """G9: MCAN encoders L=6, B=4, 100x20, D=512 rebuilt from seeds; the fixture holds the REAL
    reference's loss, output samples, input-grad samples and per-parameter grad norms."""
    import openvivqa_amd.utils as U
    c = load_case("G9_mcan_fullsize_checksum")
    te, ve = _mcan_pair(hip_namespace(), 6, c.meta["seed_weights"])
    te, ve = te.to(DEV).eval(), ve.to(DEV).eval()
    gen = torch.Generator().manual_seed(c.meta["seed_inputs"])
    v = torch.randn(4, 100, 512, generator=gen)
    l = torch.randn(4, 20, 512, generator=gen)
    v[1, 90:] = 0
    l[2, 12:] = 0
    v, l = v.to(DEV).requires_grad_(True), l.to(DEV).requires_grad_(True)
    vm, lm = U.generate_padding_mask(v, 0), U.generate_padding_mask(l, 0)
    lo = te(features=l, padding_mask=lm)
    vo = ve(vision_features=v, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    wv = torch.randn(vo.shape, generator=gen).to(DEV)
    wl = torch.randn(lo.shape, generator=gen).to(DEV)
    loss = (vo.float() * wv).mean() + (lo.float() * wl).mean()
    loss.backward()
    tol = 1e-3 if mode == F32 else 1e-2
    assert abs(loss.item() - c.out["loss"].item()) < tol
    assert nerr(vo[:, ::17, ::61], c.out["vision_sample"]) < tol
    assert nerr(lo[:, ::3, ::61], c.out["language_sample"]) < tol
    if mode == BF16:
        assert rel_l2(vo[:, ::17, ::61], c.out["vision_sample"]) < 1e-2
        assert rel_l2(lo[:, ::3, ::61], c.out["language_sample"]) < 1e-2
    gtol = 1e-3 if mode == F32 else FP32_REF_GRAD_BAR
    assert rel_l2(v.grad[:, ::17, ::61], c.out["gin_vision_sample"]) < gtol
    assert rel_l2(l.grad[:, ::3, ::61], c.out["gin_language_sample"]) < gtol
    names = c.meta["grad_norm_names"]
    got = {}
    for pre, mod in (("self_encoder.", te), ("guided_encoder.", ve)):
        for k, p in mod.named_parameters():
            got[pre + k] = p.grad.norm().item()
    refs = c.out["grad_norms"].tolist()
    import math
    from conftest import parity_record as rec
    num = math.sqrt(sum((got[n] - r) ** 2 for n, r in zip(names, refs) if not n.endswith("fc_k.bias")))
    den = math.sqrt(sum(r ** 2 for n, r in zip(names, refs) if not n.endswith("fc_k.bias")))
    tag = f"G9[{'fp32' if mode == F32 else 'bf16'}]"
    assert rec(tag, "per-parameter grad-norm vector vs reference", num / den, gtol) < gtol
    if mode == F32:  # every parameter's gradient norm against the real reference's
        for n, ref in zip(names, refs):
            if not n.endswith("fc_k.bias"):
                assert abs(got[n] - ref) <= 2 * gtol * max(ref, 1e-6) + 1e-7, (n, got[n], ref)
        return
    # bf16: EVERY parameter gradient, the small ones included (fc_q / fc_k of a freshly initialised stack are ~100x
    # smaller than the rest: near-uniform softmax, dS = P(dP - delta) is a cancellation), as full tensors against the
    # oracle in bf16-emulation mode rebuilt from the fixture's seeds (the fixture itself holds only norms and samples)
    import oracle as O
    from conftest import grad_close
    te_o, ve_o = _mcan_pair(oracle_namespace(), 6, c.meta["seed_weights"])
    te_o.eval(), ve_o.eval()

    def oracle_grads(emulate):
        te_o.zero_grad(set_to_none=True), ve_o.zero_grad(set_to_none=True)
        v_r, l_r = v.detach().cpu().clone().requires_grad_(), l.detach().cpu().clone().requires_grad_()
        with O.emulate_bf16(emulate):
            lo_r = te_o(l_r, O.padding_mask(l_r.detach(), 0))
            vo_r = ve_o(v_r, O.padding_mask(v_r.detach(), 0), lo_r, O.padding_mask(l_r.detach(), 0))
            ((vo_r * wv.cpu()).mean() + (lo_r * wl.cpu()).mean()).backward()
        g = {"d vision": v_r.grad, "d language": l_r.grad}
        for pre, m in (("self_encoder.", te_o), ("guided_encoder.", ve_o)):
            g.update({"gw/" + pre + k: p.grad.clone() for k, p in m.named_parameters()})
        return g
    g32, gem = oracle_grads(False), oracle_grads(True)
    ghip = {"d vision": v.grad, "d language": l.grad}
    for pre, m in (("self_encoder.", te), ("guided_encoder.", ve)):
        ghip.update({"gw/" + pre + k: p.grad for k, p in m.named_parameters()})
    bad = []
    for k in ghip:
        if k.endswith("fc_k.bias"):
            continue
        e, fmt = rel_l2(ghip[k], gem[k]), rel_l2(gem[k], g32[k])
        rec(tag, f"{k} vs emulation", e, G9_EMU_BAR)
        rec(tag, f"{k}: emulation vs fp32 oracle (no kernel involved)", fmt, 0.0)
        if not grad_close(e, fmt, G9_EMU_BAR, tag, k):
            bad.append((k, e, fmt))
    assert not bad, bad


@pytest.mark.parametrize("B", [64])
def test_baseline_size_vs_oracle_bf16(B):
    """BASELINE config: B=64, 100 regions x 20 tokens, D=512, L=6 (padding included), bf16 HIP
    path vs the fp32 CPU oracle on identical seeded weights/inputs -- forward only (the oracle
    needs ~2 s for this), plus size-independent properties."""
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    import oracle as O
    A.set_compute_dtype(BF16)
    te_o, ve_o = _mcan_pair(oracle_namespace(), 6, 77)
    te, ve = _mcan_pair(hip_namespace(), 6, 78)
    te.load_state_dict(te_o.state_dict())
    ve.load_state_dict(ve_o.state_dict())
    te, ve = te.to(DEV).eval(), ve.to(DEV).eval()
    te_o.eval(), ve_o.eval()
    gen = torch.Generator().manual_seed(5)
    v = torch.randn(B, 100, 512, generator=gen)
    l = torch.randn(B, 20, 512, generator=gen)
    nv = torch.randint(80, 101, (B,), generator=gen)
    nt = torch.randint(8, 21, (B,), generator=gen)
    for i in range(B):
        v[i, nv[i]:] = 0
        l[i, nt[i]:] = 0
    with torch.no_grad():
        lo_ref = te_o(l, O.padding_mask(l, 0))
        vo_ref = ve_o(v, O.padding_mask(v, 0), lo_ref, O.padding_mask(l, 0))
        vd, ld = v.to(DEV), l.to(DEV)
        vm, lm = U.generate_padding_mask(vd, 0), U.generate_padding_mask(ld, 0)
        lo = te(features=ld, padding_mask=lm)
        vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
        # north star: 1e-2 in bf16, at L=6 too
        assert nerr(lo, lo_ref) < 1e-2 and nerr(vo, vo_ref) < 1e-2, (nerr(lo, lo_ref), nerr(vo, vo_ref))
        assert rel_l2(lo, lo_ref) < 1e-2 and rel_l2(vo, vo_ref) < 1e-2, (rel_l2(lo, lo_ref), rel_l2(vo, vo_ref))
        # bug detector: against the oracle rounding where the HIP path stores bf16
        with O.emulate_bf16():
            lo_emu = te_o(l, O.padding_mask(l, 0))
            vo_emu = ve_o(v, O.padding_mask(v, 0), lo_emu, O.padding_mask(l, 0))
        assert nerr(lo, lo_emu) < 6e-3 and nerr(vo, vo_emu) < 6e-3, (nerr(lo, lo_emu), nerr(vo, vo_emu))
        # property: samples are independent (data-parallel shardability): a half batch gives the same rows
        lo_h = te(features=ld[:32], padding_mask=lm[:32])
        vo_h = ve(vision_features=vd[:32], vision_padding_mask=vm[:32], language_features=lo_h,
                  language_padding_mask=lm[:32])
        assert torch.equal(vo_h, vo[:32]) and torch.equal(lo_h, lo[:32])  # (bit-equal: no tile tier changes the summation order)
        assert nerr(vo_h, vo[:32]) < 2e-2 and nerr(lo_h, lo[:32]) < 2e-2


def _sharpen(modules, factor, value=0.3):
    """A NON-DEGENERATE operating point for the attention gradients: every fc_q x ``factor`` (attention scores x factor)
    and every fc_v x ``value``.

    Why the fresh-weight point is degenerate (measured, round 4): with i.i.d. inputs and Xavier weights every attention
    sub-layer adds nearly the SAME vector (the mean of V under a near-uniform softmax) to all tokens, the common component
    grows from layer to layer, and in the last guided layer the tokens are almost identical.  dQ_i = sum_j dS_ij K_j with
    sum_j dS_ij = 0 then cancels everything but the tokens' small individual parts: |d fc_q| / |d fc_v| falls 0.22 ->
    0.062 -> 0.016 -> 0.004 -> 0.001 -> 0.0003 over the six image self-attentions, and bf16 STORAGE alone (the emulating
    oracle against the fp32 one, no kernel involved) is 8e-2 off on that tensor -- the only tensors that ever needed
    grad_close's second arm.  Damping the value path (x0.3) keeps the tokens apart (ratio 0.08 in the last layer, the
    format's own error <= 1.4e-2 on EVERY tensor: scripts/operating_point.py) and sharper scores (x2) move the softmax
    away from uniform: there the fc_q / fc_k gradients of the last layers are ordinary numbers."""
    with torch.no_grad():
        for m in modules:
            for k, p in m.named_parameters():
                if ".fc_q." in k:
                    p.mul_(factor)
                if ".fc_v." in k:
                    p.mul_(value)


@pytest.mark.parametrize("layers,B,sharp", [(1, 16, 1.0), (6, 16, 1.0), (6, 16, 2.0), (6, 64, 2.0)],
                         ids=["L1", "L6", "L6-sharp", "L6-B64-sharp"])
def test_stack_forward_and_gradients_vs_bf16_emulating_oracle(layers, B, sharp):
    """The bug detector: MCAN stacks, 100x20 -- the HIP bf16 path against the oracle in bf16-EMULATION mode
    (oracle.emulate_bf16: values and gradients rounded where the HIP path stores bf16, fp32 everywhere else).
    At L=1 (5 chained blocks) nothing but accumulation order and fast exp/erf separates the two: forward <= 2e-3.
    At L=6 a 1-ulp bf16 flip early in the stack has been amplified through 30 blocks, so the gap approaches the
    rounding noise itself: forward <= 6e-3 (measured 3.3e-3 text / 4.6e-3 vision; against the fp32 oracle the same
    outputs are at 6e-3 / 7e-3 and held to 1e-2 by test_baseline_size_vs_oracle_bf16).
    ``sharp`` (round 4): the same stacks at a NON-DEGENERATE operating point (attention scores x2,
    value path x0.3: ``_sharpen``), the second one at the BASELINE batch (B = 64, configs[1]): there EVERY gradient tensor -- fc_q / fc_k
    of the last layers included -- is held to the bar against the emulation with NO escape clause."""
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    import oracle as O
    A.set_compute_dtype(BF16)
    te_o, ve_o = _mcan_pair(oracle_namespace(), layers, 41)
    te, ve = _mcan_pair(hip_namespace(), layers, 42)
    if sharp != 1.0:
        _sharpen((te_o, ve_o), sharp)
    te.load_state_dict(te_o.state_dict())
    ve.load_state_dict(ve_o.state_dict())
    te, ve = te.to(DEV).eval(), ve.to(DEV).eval()
    te_o.eval(), ve_o.eval()
    gen = torch.Generator().manual_seed(8)
    v, l = torch.randn(B, 100, 512, generator=gen), torch.randn(B, 20, 512, generator=gen)
    for i in range(B):
        v[i, 84 + i % 16:] = 0
        l[i, 8 + i % 12:] = 0
    wv, wl = torch.randn(v.shape, generator=gen), torch.randn(l.shape, generator=gen)
    v_r, l_r = v.clone().requires_grad_(), l.clone().requires_grad_()
    lo_r = te_o(l_r, O.padding_mask(l, 0))  # fp32 pass first: the storage format's own error per gradient tensor
    vo_r = ve_o(v_r, O.padding_mask(v, 0), lo_r, O.padding_mask(l, 0))
    ((vo_r * wv).mean() + (lo_r * wl).mean()).backward()
    g32 = {(pre + k): p.grad.clone() for pre, m in (("self_encoder.", te_o), ("guided_encoder.", ve_o))
           for k, p in m.named_parameters()}
    lo32, vo32 = lo_r.detach(), vo_r.detach()
    te_o.zero_grad(set_to_none=True), ve_o.zero_grad(set_to_none=True)
    v_r, l_r = v.clone().requires_grad_(), l.clone().requires_grad_()
    with O.emulate_bf16():
        lo_r = te_o(l_r, O.padding_mask(l, 0))
        vo_r = ve_o(v_r, O.padding_mask(v, 0), lo_r, O.padding_mask(l, 0))
        ((vo_r * wv).mean() + (lo_r * wl).mean()).backward()
    vd, ld = v.to(DEV).requires_grad_(), l.to(DEV).requires_grad_()
    vm, lm = U.generate_padding_mask(vd.detach(), 0), U.generate_padding_mask(ld.detach(), 0)
    lo = te(features=ld, padding_mask=lm)
    vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    ((vo.float() * wv.to(DEV)).mean() + (lo.float() * wl.to(DEV)).mean()).backward()
    from conftest import ESCAPES, grad_close, parity_record as rec
    tag = f"stack-emulation[L={layers}]" if sharp == 1.0 else f"stack-emulation[L={layers},B={B},scores x{sharp:g}]"
    ftol, gtol, wtol = (2e-3, 8e-3, 1.2e-2) if layers == 1 else (6e-3, 2e-2, 2.5e-2)
    if sharp != 1.0:  # the operating point itself: how far from uniform the last layer's attention is (recorded)
        rec(tag, "vision out vs fp32 oracle (north-star bar)", nerr(vo, vo32), 1e-2)
        rec(tag, "language out vs fp32 oracle (north-star bar)", nerr(lo, lo32), 1e-2)
        assert nerr(vo, vo32) < 1e-2 and nerr(lo, lo32) < 1e-2
    rec(tag, "language out vs emulation", nerr(lo, lo_r), ftol)
    rec(tag, "vision out vs emulation", nerr(vo, vo_r), ftol)
    assert nerr(lo, lo_r) < ftol and nerr(vo, vo_r) < ftol, (nerr(lo, lo_r), nerr(vo, vo_r))
    assert rec(tag, "d vision vs emulation", rel_l2(vd.grad, v_r.grad), gtol) < gtol
    assert rec(tag, "d language vs emulation", rel_l2(ld.grad, l_r.grad), gtol) < gtol
    bad, n_before = [], len(ESCAPES)
    for (pre, hip_m, ref_m) in (("self_encoder.", te, te_o), ("guided_encoder.", ve, ve_o)):
        gref = dict(ref_m.named_parameters())
        for k, p in hip_m.named_parameters():
            if k.endswith("fc_k.bias"):  # analytically zero
                continue
            e, fmt = rel_l2(p.grad, gref[k].grad), rel_l2(gref[k].grad, g32[pre + k])
            rec(tag, f"gw/{pre}{k} vs emulation", e, wtol)
            if ".fc_q." in k or ".fc_k." in k:  # how non-degenerate: size of this gradient next to the layer's fc_v one
                rec(tag, f"gw/{pre}{k}: norm / norm of the same block's fc_v gradient (recorded)",
                    float(p.grad.norm() / dict(hip_m.named_parameters())[k.replace("fc_q", "fc_v").replace("fc_k", "fc_v")].grad.norm()),
                    float("nan"))
            if sharp != 1.0:  # the operating point is non-degenerate iff the FORMAT resolves every tensor
                rec(tag, f"gw/{pre}{k}: emulation vs fp32 oracle (no kernel involved)", fmt, wtol)
                assert fmt < wtol, (pre + k, fmt)
            if not grad_close(e, fmt, wtol, tag, pre + k, allow_escape=(sharp == 1.0)):
                bad.append((pre + k, e, fmt))
    rec(tag, "gradient tensors that passed through the escape clause", float(len(ESCAPES) - n_before), 0.0)
    assert not bad, bad
    if sharp != 1.0:
        assert len(ESCAPES) == n_before


@pytest.mark.parametrize("arch,layers", [("CrossModalityEncoder", 6), ("CoAttentionEncoder", 4), ("CoAttentionEncoder", 6)])
def test_config3_size_pair_encoders_vs_oracle_bf16(arch, layers):
    """BASELINE configs[2] (cross_modality_transformer.yaml shape: d=512, L=6, 100 regions x 20 tokens; B=16 here,
    samples are independent) and its ViLBERT-style sibling: bf16 HIP path vs the fp32 oracle (outputs, 1e-2) and vs the
    bf16-emulating oracle (outputs and every gradient).
    The co-attention stack chains FOUR EncoderLayers per layer and modality (8 blocks): L=4 is 32 chained blocks, the
    depth of MCAN L=6 (30).  At L=6 (48 blocks; no shipped config uses it) the output is a DOCUMENTED MISS of the 1e-2
    bar against the fp32 oracle: bf16 storage of weights and activations alone -- the emulating oracle, no kernel
    involved -- is that far from fp32 there (asserted below), and the HIP path stays within 8e-3 of the emulation."""
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    import oracle as O
    from conftest import parity_record as rec
    from openvivqa_amd.config import ConfigNode, attention_config
    A.set_compute_dtype(BF16)
    sa = attention_config()
    cfg = ConfigNode(dict(D_MODEL=512, LAYERS=layers, VISION_LANGUAGE_ATTENTION=sa, LANGUAGE_VISION_ATTENTION=sa,
                          VISION_SELF_ATTENTION=sa, LANGUAGE_SELF_ATTENTION=sa))
    torch.manual_seed(31)
    ref = getattr(oracle_namespace(), arch)(cfg).eval()
    hip = getattr(hip_namespace(), arch)(cfg)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV).eval()
    gen = torch.Generator().manual_seed(6)
    B = 16
    v = torch.randn(B, 100, 512, generator=gen)
    l = torch.randn(B, 20, 512, generator=gen)
    for i in range(B):
        v[i, 80 + i:] = 0
        l[i, 8 + i % 12:] = 0
    wv, wl = torch.randn(v.shape, generator=gen), torch.randn(l.shape, generator=gen)
    vo_r, lo_r = v.clone().requires_grad_(), l.clone().requires_grad_()
    a, b = ref(vo_r, O.padding_mask(v, 0), lo_r, O.padding_mask(l, 0))
    ((a * wv).mean() + (b * wl).mean()).backward()
    a, b = a.detach(), b.detach()
    g32 = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    g32.update({"d vision": vo_r.grad, "d language": lo_r.grad})
    ref.zero_grad(set_to_none=True)
    vo_r, lo_r = v.clone().requires_grad_(), l.clone().requires_grad_()
    with O.emulate_bf16():
        ae, be = ref(vo_r, O.padding_mask(v, 0), lo_r, O.padding_mask(l, 0))
        ((ae * wv).mean() + (be * wl).mean()).backward()
    vd, ld = v.to(DEV).requires_grad_(), l.to(DEV).requires_grad_()
    ah, bh = hip(vision_features=vd, vision_padding_mask=U.generate_padding_mask(vd.detach(), 0), language_features=ld,
                 language_padding_mask=U.generate_padding_mask(ld.detach(), 0))
    ((ah.float() * wv.to(DEV)).mean() + (bh.float() * wl.to(DEV)).mean()).backward()
    tag = f"pair[{arch},L={layers}]"
    e_f = max(rec(tag, "vision out vs fp32 oracle", nerr(ah, a), 1e-2), rec(tag, "language out vs fp32 oracle", nerr(bh, b), 1e-2))
    e_e = max(rec(tag, "vision out vs emulation", nerr(ah, ae), 8e-3), rec(tag, "language out vs emulation", nerr(bh, be), 8e-3))
    emu_gap = max(rec(tag, "emulation vs fp32 oracle (no kernel involved), vision", nerr(ae, a), 1e-2),
                  rec(tag, "emulation vs fp32 oracle (no kernel involved), language", nerr(be, b), 1e-2))
    assert e_e < 8e-3, (e_e, e_f)
    # (round 3 carried a 1.5e-2 escape for CoAttention L=6, 48 chained blocks: it measures 9.0e-3 since the delta fix)
    assert e_f < 1e-2 and rel_l2(ah, a) < 1e-2 and rel_l2(bh, b) < 1e-2, (e_f, emu_gap, rel_l2(ah, a), rel_l2(bh, b))
    from conftest import grad_close
    gbar = G9_EMU_BAR  # 24-48 chained blocks
    bad = []
    for what, gh, ge in (("d vision", vd.grad, vo_r.grad), ("d language", ld.grad, lo_r.grad)):
        e, fmt = rec(tag, f"{what} vs emulation", rel_l2(gh, ge), gbar), rel_l2(ge, g32[what])
        if not grad_close(e, fmt, gbar, tag, what):
            bad.append((what, e, fmt))
    gref = dict(ref.named_parameters())
    # CoAttentionEncoder at L = 6 chains 48 blocks (the deepest shipped stack: 30) and no shipped config builds it: its
    # late fc_q / fc_k gradients are cancellations the bf16 FORMAT cannot resolve (emulation vs fp32 oracle up to 0.10, no
    # kernel involved).  Round 5: such tensors no longer pass through grad_close's escape clause here (it needed a ceiling
    # of 0.15 for this case alone); a tensor is asserted when the format resolves it (its own error <= half the bar: the
    # HIP path sits at ~sqrt(2) x the format's error, two realisations of the same rounding noise), the others are counted
    # and recorded.  Every other case keeps the clause, under the lower ceiling.
    no_escape = arch == "CoAttentionEncoder" and layers == 6
    unresolved = 0
    for k, p in hip.named_parameters():
        if gref[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if k.endswith("fc_k.bias"):
            continue
        e, fmt = rec(tag, f"gw/{k} vs emulation", rel_l2(p.grad, gref[k].grad), gbar), rel_l2(gref[k].grad, g32[k])
        if no_escape and fmt > 0.5 * gbar:
            unresolved += 1
            rec(tag, f"[not asserted: the format's own error is {fmt:.3e}] gw/{k}", e, gbar)
            continue
        if not grad_close(e, fmt, gbar, tag, "gw/" + k, allow_escape=not no_escape):
            bad.append((k, e, fmt))
    assert not bad, bad
    assert unresolved <= 60, unresolved  # (of 576 tensors; measured 50 in round 5: recorded, NOT asserted -- DESIGN.md section 2)


def test_crossmodality_dead_branch_and_unused_grads(mode):
    """cross-attention parameters exist in the state_dict, get no gradient, and skipping
    their dead compute leaves outputs identical to computing it."""
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    case = load_case("G4_crossmodality_layer")
    outs = []
    for dead in (False, True):
        m = M.CrossModalityEncoderLayer(ConfigNode(case.meta["cfg"]))
        m.compute_dead_cross_attention = dead
        m.load_state_dict(case.w)
        m = m.to(DEV).eval()
        i = {k: v.to(DEV) for k, v in case.inputs.items()}
        with torch.no_grad():
            outs.append(m(i["vision"], i["vmask"], i["language"], i["lmask"]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_train_mode_dropout_runs_and_is_consistent():
    """Train mode: dropout masks are regenerated in backward from (seed, site, step).  Check the
    gradient of one FFN block against finite differences of its own forward (fp32 mode)."""
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import attention_config
    from openvivqa_amd import runtime as rt
    A.set_compute_dtype(F32)
    try:
        torch.manual_seed(3)
        m = M.PositionWiseFeedForward(attention_config(d_model=32, d_ff=64, dropout=0.3)).to(DEV).train()
        x = torch.randn(2, 5, 32, device=DEV, requires_grad=True)
        w = torch.randn(2, 5, 32, device=DEV)

        def f(inp):
            A.manual_seed(1234)  # same masks on every evaluation
            return (m(inp) * w).sum()
        y = f(x)
        y.backward()
        g = x.grad.clone()
        eps = 1e-2
        for idx in [(0, 0, 0), (1, 3, 7), (0, 4, 31)]:
            xp, xm = x.detach().clone(), x.detach().clone()
            xp[idx] += eps
            xm[idx] -= eps
            with torch.no_grad():
                fd = (f(xp) - f(xm)).item() / (2 * eps)
            assert abs(fd - g[idx].item()) < 2e-2 * max(1.0, abs(fd)), (idx, fd, g[idx].item())
        m.eval()
        with torch.no_grad():
            a, b = m(x), m(x)
        assert torch.equal(a, b)
    finally:
        A.set_compute_dtype(BF16)


@pytest.mark.parametrize("axis", ["key", "query"])
def test_dynamic_pointer_network_module_vs_oracle(axis, mode):
    """a17 at the M4C size (D_MODEL 768, 12 decode positions x 50 OCR tokens): the key-axis variant of
    models/m4c.py:19-33 (boolean mask over the OCR tokens -> -inf columns) and the query-axis variant of
    models/iterative_m4c.py:18-32 (-inf rows), module level, forward and gradients, against the oracle."""
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    cfg = ConfigNode(dict(D_MODEL=768))
    torch.manual_seed(17)
    ref = O.OracleDynamicPointerNetwork(cfg, axis)
    hip = M.DynamicPointerNetwork(cfg, axis)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    g = torch.Generator().manual_seed(3)
    B, T, No = 4, 12, 50
    q, k = torch.randn(B, T, 768, generator=g), torch.randn(B, No, 768, generator=g)
    n = No if axis == "key" else T
    mask = torch.zeros(B, 1, 1, n, dtype=torch.bool)
    mask[1, ..., n - 7:] = True
    mask[3, ..., n // 2:] = True
    qr, kr = q.clone().requires_grad_(), k.clone().requires_grad_()
    s_r = ref(qr, kr, mask)
    qh, kh = q.to(DEV).requires_grad_(), k.to(DEV).requires_grad_()
    s_h = hip(qh, kh, mask.to(DEV))
    assert s_h.shape == s_r.shape == (B, T, No)
    assert torch.equal(torch.isinf(s_h).cpu(), torch.isinf(s_r))
    assert nerr(s_h, s_r) < (1e-3 if mode == F32 else 1e-2), nerr(s_h, s_r)
    w = torch.randn(s_r.shape, generator=g)
    fin = torch.isfinite(s_r)
    (torch.where(fin, s_r, torch.zeros_like(s_r)) * w).sum().backward()
    (torch.where(fin.to(DEV), s_h, torch.zeros_like(s_h)) * w.to(DEV)).sum().backward()
    gt = 1e-3 if mode == F32 else 3e-2
    assert rel_l2(qh.grad, qr.grad) < gt and rel_l2(kh.grad, kr.grad) < gt
    for name, p in hip.named_parameters():
        assert rel_l2(p.grad, dict(ref.named_parameters())[name].grad) < gt, name


@pytest.mark.parametrize("trig", [True, False])
def test_geometry_attention_vs_oracle(trig, mode):
    """AugmentedGeometryScaledDotProductAttention built working (upstream's forward raises NameError): the
    log-geometry-biased softmax on the HIP attention kernel (bias = per-head additive mask) against the oracle's
    restatement of the intended computation, forward, input gradients and the fc_g / projection weight gradients."""
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode, attention_config
    cfg = attention_config(d_model=512, head=8, d_key=64, d_value=64)
    cfg["TRIGNOMETRIC_EMBEDDING"] = trig
    torch.manual_seed(21)
    ref = O.OracleGeometrySDPA(cfg)
    with torch.no_grad():
        for g_ in ref.fc_gs:
            g_.bias.fill_(0.3)  # keep a good share of the relu alive
    hip = M.AugmentedGeometryScaledDotProductAttention(cfg)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    g = torch.Generator().manual_seed(5)
    B, N = 3, 36
    x = torch.randn(B, N, 512, generator=g)
    boxes = torch.rand(B, N, 4, generator=g)
    boxes[..., 2:] = boxes[..., :2] + 0.05 + boxes[..., 2:] * 0.5
    mask = torch.zeros(B, 1, 1, N)
    mask[1, ..., 30:] = -10e4
    xr = x.clone().requires_grad_()
    out_r, att_r = ref(xr, xr, xr, boxes, attention_mask=mask)
    xh = x.to(DEV).requires_grad_()
    out_h, att_h = hip(xh, xh, xh, boxes.to(DEV), attention_mask=mask.to(DEV))
    out_h2, _ = hip(xh, xh, xh, mask.to(DEV), boxes=boxes.to(DEV))  # the order MultiHeadAttention uses
    assert torch.equal(out_h2, out_h)
    tol = 1e-3 if mode == F32 else 1e-2
    assert nerr(out_h, out_r) < tol and nerr(att_h, att_r) < tol, (nerr(out_h, out_r), nerr(att_h, att_r))
    w = torch.randn(out_r.shape, generator=g)
    (out_r * w).sum().backward()
    (out_h.float() * w.to(DEV)).sum().backward()
    gt = 1e-3 if mode == F32 else 3e-2
    assert rel_l2(xh.grad, xr.grad) < gt, rel_l2(xh.grad, xr.grad)
    gref = dict(ref.named_parameters())
    for k, p in hip.named_parameters():
        if k.endswith("fc_k.bias"):
            continue
        assert p.grad is not None, k
        if k.startswith("fc_gs"):  # 8 heads x d_g weights: compare them together
            continue
        assert rel_l2(p.grad, gref[k].grad) < gt, (k, rel_l2(p.grad, gref[k].grad))
    gh = torch.cat([p.grad.flatten() for k, p in hip.named_parameters() if k.startswith("fc_gs")])
    gr = torch.cat([p.grad.flatten() for k, p in ref.named_parameters() if k.startswith("fc_gs")])
    assert rel_l2(gh, gr) < gt, rel_l2(gh, gr)


def test_mask_and_position_helpers_bit_exact_on_gpu():
    """a12 / a13 on the device: the mask helpers, the decoder position table, the encoder
    sinusoid table and the row-padding-mask kernel against G6 -- outputs of the reference's models/utils.py:32-73 and
    pos_embeddings.py:58-72 -- bit for bit, with CUDA inputs."""
    from openvivqa_amd import ops
    from openvivqa_amd.modules.pos_embeddings import SinusoidPositionalEmbedding
    from openvivqa_amd.utils import (generate_padding_mask, generate_self_attention_masks, generate_sequential_mask,
                                     sinusoid_encoding_table)
    c = load_case("G6_pos_masks")
    pe = SinusoidPositionalEmbedding(8)(torch.zeros(2, 3, 8, device=DEV))
    assert pe.is_cuda and pe.shape == (2, 3, 8) and torch.equal(pe[0].cpu(), c.out["sinusoid_3_8"])
    big = SinusoidPositionalEmbedding(512)(torch.zeros(1, 100, 512, device=DEV))[0]
    assert torch.equal(big[::33, ::37].cpu(), c.out["sinusoid_100_512"])
    assert torch.equal(sinusoid_encoding_table(6, 8, 0), c.out["table_6_8_pad0"])
    assert torch.equal(sinusoid_encoding_table(6, 8), c.out["table_6_8_nopad"])
    toks, feats = c.inputs["tokens"].to(DEV), c.inputs["feats"].to(DEV)
    pm = generate_padding_mask(toks, 0)
    assert pm.is_cuda and pm.dtype == c.out["padmask_tokens"].dtype and torch.equal(pm.cpu(), c.out["padmask_tokens"])
    pf = generate_padding_mask(feats, 0)
    assert torch.equal(pf.cpu(), c.out["padmask_feats"])
    sm = generate_sequential_mask(5, device=DEV)
    assert sm.is_cuda and torch.equal(sm.cpu(), c.out["seqmask_5"])
    both = generate_self_attention_masks(pm, sm)
    assert both.dtype == c.out["selfmask"].dtype and torch.equal(both.cpu(), c.out["selfmask"])
    # the fused kernel form of generate_padding_mask for feature rows (FeatureEmbedding): same values, fp32 and bf16
    for dt in (F32, BF16):
        km = ops.row_padding_mask(feats.to(dt).contiguous(), 0.0)
        assert km.shape == (2, 1, 1, 4) and torch.equal(km.cpu(), c.out["padmask_feats"].float())
    # the encoder prologue's fused table add reads the same table: LN(x) + pos == LN(x) + G6's rows
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 8, generator=g).to(DEV)
    ones, zeros = torch.ones(8, device=DEV), torch.zeros(8, device=DEV)
    y_pos, _, _ = ops.layernorm_fwd(x, ones, zeros, 1e-5, out_dtype=F32, pos=pe[0].contiguous())
    y, _, _ = ops.layernorm_fwd(x, ones, zeros, 1e-5, out_dtype=F32)
    assert torch.equal(y_pos.cpu(), (y + c.out["sinusoid_3_8"].to(DEV)).cpu())


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_train_step_reference_trajectory_on_gpu(graph):
    """Row T on the device: G11 -- the reference's own two Adam steps (losses and post-step
    weights of Encoder L=2 + head under NLLLoss, Adam(0.9, 0.98), Noam LambdaLR) -- replayed through TrainStep in fp32
    mode with the real kernels, eager launches and captured hipGraph."""
    import openvivqa_amd as A
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.train import TrainStep, noam_lr_scale
    from openvivqa_amd.utils import generate_padding_mask
    A.set_compute_dtype(F32)
    try:
        case = load_case("G11_train_two_steps")
        enc = M.Encoder(ConfigNode(case.meta["cfg"]))
        head = torch.nn.Linear(32, 5)

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.enc, self.head = enc, head

            def forward(self, x):
                return torch.log_softmax(self.head(self.enc(x, generate_padding_mask(x, 0)).mean(1)), -1)
        net = Net()
        enc.load_state_dict({k: v for k, v in case.w.items() if not k.startswith("head.")})
        head.load_state_dict({"weight": case.w["head.weight"], "bias": case.w["head.bias"]})
        net = net.to(DEV).train()
        y = case.inputs["y"].to(DEV)
        nll = torch.nn.NLLLoss(ignore_index=0)
        ts = TrainStep(net, lambda x: nll(net(x), y), lr=case.meta["lr"], betas=tuple(case.meta["betas"]),
                       lr_lambda=lambda s: noam_lr_scale(s, 32, case.meta["warmup"]), use_graph=graph,
                       compute_dtype=F32)
        x = case.inputs["x"].to(DEV)
        losses = [float(ts.step(x).item()) for _ in range(2)]
        one_graph = os.environ.get("OVQA_WHOLE_STEP_GRAPH", "1") != "0"  # (the A/B switch keeps the per-phase graphs)
        assert graph == ts.captured and (not graph or (ts.whole is not None) == one_graph)  # Adam + Noam schedule in the graph
        assert nerr(torch.tensor(losses), case.out["losses"]) < 1e-4, (losses, case.out["losses"])
        for k, v in enc.state_dict().items():
            if k.endswith("fc_k.bias"):
                continue  # analytically zero gradient: Adam turns its rounding noise into +-lr steps
            assert nerr(v, case.out["w2/" + k]) < 1e-3, ("post-step " + k, nerr(v, case.out["w2/" + k]))
        assert nerr(head.weight, case.out["w2/head.weight"]) < 1e-3
    finally:
        A.set_compute_dtype(BF16)


def test_npy_ingestion_feeds_feature_embedding(tmp_path, mode):
    """``.npy`` dict-of-arrays files -> FeatureCollator (pinned staging, side-stream copy, fixed padded length) ->
    FeatureEmbedding: the same features and zero-row padding mask as the oracle's embedding on the reference-collated
    batch (utils/instance.py:31-54,155-170; vision_embeddings.py:10-25)."""
    import numpy as np
    import oracle as O
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.ingest import FeatureCollator, load_features
    rng = np.random.default_rng(3)
    lens = [100, 37, 81, 64]
    for i, n in enumerate(lens):
        np.save(tmp_path / f"{i}.npy", {"region_features": rng.standard_normal((n, 2048)).astype(np.float32)},
                allow_pickle=True)
    samples = [load_features(str(tmp_path / f"{i}.npy")) for i in range(len(lens))]
    col = FeatureCollator(["region_features"], DEV, pad_to={"region_features": 100})
    cfg = ConfigNode(dict(ARCHITECTURE="FeatureEmbedding", D_FEATURE=2048, D_MODEL=512, DROPOUT=0.1))
    torch.manual_seed(1)
    o = O.OracleFeatureEmbedding(cfg).eval()
    h = M.FeatureEmbedding(cfg)
    h.load_state_dict(o.state_dict())
    h = h.to(DEV).eval()
    ref = torch.zeros(len(lens), 100, 2048)
    for i, s_ in enumerate(samples):
        ref[i, :lens[i]] = torch.tensor(s_["region_features"])
    for _ in range(3):  # the buffers are reused from the third batch on
        batch = col.collate(samples)
        col.wait()
        assert torch.equal(batch["region_features"].cpu(), ref)
    with torch.no_grad():
        fh, mh = h(batch["region_features"])
        fo, mo = o(ref)
    assert torch.equal(mh.cpu() != 0, mo != 0)
    assert nerr(fh, fo) < (1e-3 if mode == F32 else 1e-2)


# ---- size-independent properties at the BASELINE configuration (configs[1]: B = 64, L = 6, 100 regions x 20 tokens): no
# ---- oracle involved, so the full size costs nothing on the CPU
def _baseline_stack(dropout_eval=True):
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    A.set_compute_dtype(BF16)
    te, ve = _mcan_pair(hip_namespace(), 6, 123)
    te, ve = te.to(DEV).eval(), ve.to(DEV).eval()
    gen = torch.Generator().manual_seed(17)
    B = 64
    v, l = torch.randn(B, 100, 512, generator=gen), torch.randn(B, 20, 512, generator=gen)
    nv, nt = torch.randint(80, 101, (B,), generator=gen), torch.randint(8, 21, (B,), generator=gen)
    for i in range(B):
        v[i, nv[i]:] = 0
        l[i, nt[i]:] = 0
    v, l = v.to(DEV), l.to(DEV)
    return te, ve, v, l, U.generate_padding_mask(v, 0), U.generate_padding_mask(l, 0), nv, nt


def test_baseline_size_backward_is_linear_in_the_upstream_gradient():
    """Backward is linear: an upstream gradient scaled by a power of two scales EVERY gradient -- inputs and all 256
    parameter tensors of the L = 6 stacks at B = 64 -- by exactly that factor, bit for bit (scaling by 2^k commutes with
    every rounding in the bf16 / fp32 pipeline; the kernels are deterministic).  A gradient kernel that dropped, doubled or
    mis-accumulated a contribution anywhere in the 30 chained blocks fails this without any oracle."""
    te, ve, v, l, vm, lm, _, _ = _baseline_stack()
    gen = torch.Generator().manual_seed(3)
    gv, gl = torch.randn(v.shape, generator=gen).to(DEV, BF16), torch.randn(l.shape, generator=gen).to(DEV, BF16)
    grads = []
    for scale in (1.0, 4.0):
        for p in list(te.parameters()) + list(ve.parameters()):
            p.grad = None
        vd, ld = v.clone().requires_grad_(), l.clone().requires_grad_()
        lo = te(features=ld, padding_mask=lm)
        vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
        torch.autograd.backward([vo, lo], [gv * scale, gl * scale])
        g = {"d vision": vd.grad.clone(), "d language": ld.grad.clone()}
        for pre, m in (("te.", te), ("ve.", ve)):
            g.update({pre + k: p.grad.detach().clone() for k, p in m.named_parameters()})
        grads.append(g)
    assert len(grads[0]) == 2 + 98 + 158  # the two inputs, 98 tensors of the question stack, 158 of the guided stack
    for k, g1 in grads[0].items():
        assert torch.equal(grads[1][k], g1 * 4.0), k
        assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0 or k.endswith("fc_k.bias"), k


def test_baseline_size_padding_and_sample_order_properties():
    """(1) Samples are independent (what data parallelism shards on): a permutation of the batch permutes the outputs,
    bit for bit.  (2) The outputs of VALID positions do not depend on how many padded positions follow them: with fewer
    padded regions per sample (96 instead of 100 columns, same masks) every sample that still fits reproduces its valid
    rows -- the additive -1e5 mask removes padded keys exactly (exp underflows to 0), padded query rows feed only padded
    keys of later layers."""
    te, ve, v, l, vm, lm, nv, nt = _baseline_stack()
    with torch.no_grad():
        lo = te(features=l, padding_mask=lm)
        vo = ve(vision_features=v, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(DEV)
        lo_p = te(features=l[perm], padding_mask=lm[perm])
        vo_p = ve(vision_features=v[perm], vision_padding_mask=vm[perm], language_features=lo_p,
                  language_padding_mask=lm[perm])
        assert torch.equal(lo_p, lo[perm]) and torch.equal(vo_p, vo[perm])
        # fewer padded positions: 100 -> 96 regions keeps every sample with <= 96 valid regions intact
        keep = (nv <= 96).to(DEV)
        assert int(keep.sum()) >= 20
        vo_t = ve(vision_features=v[:, :96].contiguous(), vision_padding_mask=vm[..., :96].contiguous(),
                  language_features=lo, language_padding_mask=lm)
        for i in torch.nonzero(keep).flatten().tolist():
            n = int(nv[i])
            assert nerr(vo_t[i, :n], vo[i, :n]) < 4e-3, i  # (another tile count: the same math in another summation order)


# ---- BASELINE configs[2] at ITS batch: CrossModalityEncoder d = 512, L = 6, B = 64 per GPU ------------
def _config2_encoder(namespace, seed=31):
    from openvivqa_amd.config import ConfigNode, attention_config
    sa = attention_config()
    cfg = ConfigNode(dict(D_MODEL=512, LAYERS=6, VISION_LANGUAGE_ATTENTION=sa, LANGUAGE_VISION_ATTENTION=sa,
                          VISION_SELF_ATTENTION=sa, LANGUAGE_SELF_ATTENTION=sa))
    torch.manual_seed(seed)
    return namespace.CrossModalityEncoder(cfg)


def _config2_batch(B=64):
    gen = torch.Generator().manual_seed(6)
    v, l = torch.randn(B, 100, 512, generator=gen), torch.randn(B, 20, 512, generator=gen)
    nv, nt = torch.randint(80, 101, (B,), generator=gen), torch.randint(8, 21, (B,), generator=gen)
    for i in range(B):
        v[i, nv[i]:] = 0
        l[i, nt[i]:] = 0
    return v, l


def test_config2_forward_at_batch_64_vs_oracle():
    """configs[2]'s own size (B = 64; the gradient test above runs at B = 16, samples being independent): forward of the
    bf16 HIP path against the fp32 oracle (north-star 1e-2, normalised max and relative L2) and the bf16-emulating one."""
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    import oracle as O
    from conftest import parity_record as rec
    A.set_compute_dtype(BF16)
    ref = _config2_encoder(oracle_namespace()).eval()
    hip = _config2_encoder(hip_namespace())
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV).eval()
    v, l = _config2_batch()
    with torch.no_grad():
        a, b = ref(v, O.padding_mask(v, 0), l, O.padding_mask(l, 0))
        with O.emulate_bf16():
            ae, be = ref(v, O.padding_mask(v, 0), l, O.padding_mask(l, 0))
        vd, ld = v.to(DEV), l.to(DEV)
        ah, bh = hip(vision_features=vd, vision_padding_mask=U.generate_padding_mask(vd, 0), language_features=ld,
                     language_padding_mask=U.generate_padding_mask(ld, 0))
    tag = "config2[B=64]"
    for what, h, r, e in (("vision", ah, a, ae), ("language", bh, b, be)):
        assert rec(tag, f"{what} out vs fp32 oracle", nerr(h, r), 1e-2) < 1e-2 and rel_l2(h, r) < 1e-2, what
        assert rec(tag, f"{what} out vs emulation", nerr(h, e), 8e-3) < 8e-3, what


def test_config2_linearity_and_sample_order_properties():
    """The size-independent properties of the MCAN stacks, on configs[2]'s CrossModalityEncoder at B = 64, L = 6: (1) an
    upstream gradient scaled by 4 scales both input gradients and every live parameter gradient by exactly 4, bit for bit,
    and the dead cross-attention parameters (encoders.py:39-66) get none; (2) a permutation of the batch permutes both
    outputs bit for bit."""
    import openvivqa_amd as A
    import openvivqa_amd.utils as U
    A.set_compute_dtype(BF16)
    hip = _config2_encoder(hip_namespace()).to(DEV).eval()
    v, l = _config2_batch()
    v, l = v.to(DEV), l.to(DEV)
    vm, lm = U.generate_padding_mask(v, 0), U.generate_padding_mask(l, 0)
    gen = torch.Generator().manual_seed(3)
    gv, gl = torch.randn(v.shape, generator=gen).to(DEV, BF16), torch.randn(l.shape, generator=gen).to(DEV, BF16)
    grads = []
    for scale in (1.0, 4.0):
        for p in hip.parameters():
            p.grad = None
        vd, ld = v.clone().requires_grad_(), l.clone().requires_grad_()
        vo, lo = hip(vision_features=vd, vision_padding_mask=vm, language_features=ld, language_padding_mask=lm)
        torch.autograd.backward([vo, lo], [gv.to(vo.dtype) * scale, gl.to(lo.dtype) * scale])
        g = {"d vision": vd.grad.clone(), "d language": ld.grad.clone()}
        g.update({k: (None if p.grad is None else p.grad.detach().clone()) for k, p in hip.named_parameters()})
        grads.append(g)
    dead = [k for k, g in grads[0].items() if g is None or float(g.abs().max()) == 0.0]
    assert len([k for k in dead if "vision_language_mhattn" in k or "language_vision_mhattn" in k]) == 120  # 20 per layer
    for k, g1 in grads[0].items():
        if g1 is None:
            assert grads[1][k] is None
            continue
        assert torch.equal(grads[1][k], g1 * 4.0), k
    with torch.no_grad():
        vo, lo = hip(vision_features=v, vision_padding_mask=vm, language_features=l, language_padding_mask=lm)
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(DEV)
        vo_p, lo_p = hip(vision_features=v[perm], vision_padding_mask=vm[perm], language_features=l[perm],
                         language_padding_mask=lm[perm])
    assert torch.equal(vo_p, vo[perm]) and torch.equal(lo_p, lo[perm])


# ---- BASELINE configs[4], the training half, at ITS size: Decoder d = 512, L = 3, B = 64, T = 20, 237 encoder positions, 4000 words ----
def _baseline_decoder(seed=41):
    import openvivqa_amd.modules as M
    from openvivqa_amd.config import ConfigNode, attention_config
    from openvivqa_amd.utils import generate_padding_mask
    V, T, NE, B = 4000, 20, 237, 64

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = T, 0, 1, 2

        def __len__(self):
            return V
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=512, LAYERS=3,
        ATTENTION=dict(SELF_ATTENTION=attention_config(can_be_stateful=True), ENC_ATTENTION=attention_config()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=512, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(seed)
    m = M.Decoder(cfg, Vocab()).to(DEV).eval()
    g = torch.Generator().manual_seed(seed + 1)
    ans = torch.randint(3, V, (B, T), generator=g)
    ans[:, 0] = 1
    na = torch.randint(6, T + 1, (B,), generator=g)
    ans[torch.arange(T)[None, :] >= na[:, None]] = 0  # <bos> a_1 .. a_n <pad> ...
    enc = torch.randn(B, NE, 512, generator=g)
    ne = torch.randint(200, NE + 1, (B,), generator=g)
    enc[torch.arange(NE)[None, :] >= ne[:, None]] = 0
    enc = enc.to(DEV, BF16)
    return m, ans.to(DEV), enc, generate_padding_mask(enc, 0), na


def test_decoder_train_size_backward_is_linear_and_samples_are_independent():
    """The teacher-forced Decoder at the size bench.py's `decoder_train` runs (no oracle, so the full size costs nothing on the
    CPU): (1) an upstream gradient of the log-probabilities scaled by 4 scales the gradient of the encoder features and of
    every trainable parameter -- the word embeddings' deterministic scatter, the row-per-workgroup log-softmax backward, the
    4000-word classifier on its zero-padded footprint included -- by exactly 4, bit for bit; (2) a permutation of the batch
    permutes the log-probabilities bit for bit; (3) rows of padded answer positions carry no gradient into the table beyond
    the padding row's zeros."""
    m, ans, enc, emask, na = _baseline_decoder()
    gen = torch.Generator().manual_seed(5)
    up = torch.randn(64, 20, 4000, generator=gen).to(DEV) * 1e-3
    grads = []
    for scale in (1.0, 4.0):
        for p in m.parameters():
            p.grad = None
        e = enc.clone().requires_grad_()
        logp = m(ans, e, emask)
        assert logp.shape == (64, 20, 4000) and logp.dtype == F32
        logp.backward(up * scale)
        g = {"d encoder": e.grad.clone()}
        g.update({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None})
        grads.append(g)
    assert len(grads[0]) >= 1 + 3 * 26  # the encoder features and every tensor of the three layers, the table, the classifier
    for k, g1 in grads[0].items():
        assert torch.equal(grads[1][k], g1 * 4.0), k
        assert bool(torch.isfinite(g1).all()), k
    table = grads[0]["word_emb.components.weight"]
    assert float(table[0].abs().max()) == 0.0  # padding_idx
    with torch.no_grad():
        lp = m(ans, enc, emask)
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(DEV)
        lp_p = m(ans[perm], enc[perm], emask[perm])
        assert torch.equal(lp_p, lp[perm])
        assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-3  # rows of a log-softmax
