#!/usr/bin/env python3
"""bf16 HIP path vs the fp32 CPU oracle at the BASELINE size (L=6, B=64, 100x20, D=512): normalised max error and
relative L2 of both stack outputs for a few weight/input seeds (what tests/test_modules_gpu.py asserts at 1e-2)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for seed in (77, 78, 79):
    te_o, ve_o = _mcan_pair(oracle_namespace(), 6, seed)
    te, ve = _mcan_pair(hip_namespace(), 6, seed + 100)
    te.load_state_dict(te_o.state_dict())
    ve.load_state_dict(ve_o.state_dict())
    te, ve = te.to("cuda").eval(), ve.to("cuda").eval()
    te_o.eval(), ve_o.eval()
    gen = torch.Generator().manual_seed(seed)
    v, l = torch.randn(B, 100, 512, generator=gen), torch.randn(B, 20, 512, generator=gen)
    nv, nt = torch.randint(80, 101, (B,), generator=gen), torch.randint(8, 21, (B,), generator=gen)
    for i in range(B):
        v[i, nv[i]:] = 0
        l[i, nt[i]:] = 0
    with torch.no_grad():
        lo_ref = te_o(l, O.padding_mask(l, 0))
        vo_ref = ve_o(v, O.padding_mask(v, 0), lo_ref, O.padding_mask(l, 0))
        vd, ld = v.to("cuda"), l.to("cuda")
        vm, lm = U.generate_padding_mask(vd, 0), U.generate_padding_mask(ld, 0)
        lo = te(features=ld, padding_mask=lm)
        vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    print(f"seed {seed}: text nerr {nerr(lo, lo_ref):.2e} relL2 {rel_l2(lo, lo_ref):.2e} | "
          f"vision nerr {nerr(vo, vo_ref):.2e} relL2 {rel_l2(vo, vo_ref):.2e}", flush=True)
