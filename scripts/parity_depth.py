#!/usr/bin/env python3
"""HIP bf16 path vs (a) the fp32 oracle and (b) the oracle in bf16-emulation mode, by stack depth L (B=16, 100x20):
forward normalised-max error, and at each depth the input-gradient relative L2 against both."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import openvivqa_amd as A  # noqa: E402
import openvivqa_amd.utils as U  # noqa: E402
import oracle as O  # noqa: E402
from golden_cases import hip_namespace, oracle_namespace  # noqa: E402
from test_modules_gpu import _mcan_pair, nerr, rel_l2  # noqa: E402

A.set_compute_dtype(torch.bfloat16)
B = 16
for L in (1, 2, 3, 6):
    te_o, ve_o = _mcan_pair(oracle_namespace(), L, 41)
    te, ve = _mcan_pair(hip_namespace(), L, 42)
    te.load_state_dict(te_o.state_dict())
    ve.load_state_dict(ve_o.state_dict())
    te, ve = te.to("cuda").eval(), ve.to("cuda").eval()
    te_o.eval(), ve_o.eval()
    gen = torch.Generator().manual_seed(8)
    v, l = torch.randn(B, 100, 512, generator=gen), torch.randn(B, 20, 512, generator=gen)
    for i in range(B):
        v[i, 84 + i:] = 0
        l[i, 8 + i % 12:] = 0
    wv, wl = torch.randn(v.shape, generator=gen), torch.randn(l.shape, generator=gen)
    res = {}
    for name, emu in (("fp32", False), ("emu", True)):
        v_r, l_r = v.clone().requires_grad_(), l.clone().requires_grad_()
        for m in (te_o, ve_o):
            m.zero_grad()
        with O.emulate_bf16(emu):
            lo_r = te_o(l_r, O.padding_mask(l, 0))
            vo_r = ve_o(v_r, O.padding_mask(v, 0), lo_r, O.padding_mask(l, 0))
            ((vo_r * wv).mean() + (lo_r * wl).mean()).backward()
        gw = {("t." + k): p.grad.clone() for k, p in te_o.named_parameters()}
        gw.update({("v." + k): p.grad.clone() for k, p in ve_o.named_parameters()})
        res[name] = (lo_r.detach(), vo_r.detach(), v_r.grad, l_r.grad, gw)
    vd, ld = v.to("cuda").requires_grad_(), l.to("cuda").requires_grad_()
    vm, lm = U.generate_padding_mask(vd.detach(), 0), U.generate_padding_mask(ld.detach(), 0)
    lo = te(features=ld, padding_mask=lm)
    vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    ((vo.float() * wv.to("cuda")).mean() + (lo.float() * wl.to("cuda")).mean()).backward()
    gh = {("t." + k): p.grad for k, p in te.named_parameters()}
    gh.update({("v." + k): p.grad for k, p in ve.named_parameters()})
    for name in ("fp32", "emu"):
        lo_r, vo_r, gv, gl, gw = res[name]
        gmax = max(float(g.norm()) for g in gw.values())
        worst = max(((rel_l2(gh[k], g), k) for k, g in gw.items()
                     if not k.endswith("fc_k.bias") and float(g.norm()) >= 0.05 * gmax))
        print(f"L={L} vs {name:4s}: text nerr {nerr(lo, lo_r):.2e} vision nerr {nerr(vo, vo_r):.2e} | "
              f"dV relL2 {rel_l2(vd.grad, gv):.2e} dL relL2 {rel_l2(ld.grad, gl):.2e} | worst dW {worst[0]:.2e} {worst[1]}",
              flush=True)
