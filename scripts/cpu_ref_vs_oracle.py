#!/usr/bin/env python3
"""BASELINE.md section 4.1 (build container only: needs /root/reference): the REAL reference modules and the oracle
restatement timed side by side on the headline workload -- Encoder + GuidedAttentionEncoder, L=6, B=64, 100 regions
x 20 tokens, d=512, fp32, train mode (dropout on), forward + loss + backward + Adam(0.9, 0.98) -- with the same
thread count, 2 warm-up + N timed steps each, median.  Shows that bench.py's cpu_baseline (the oracle, which is all
that travels to the GPU box) is a fair stand-in for the reference's own CPU path.  Import recipe: SURVEY.md 8c.

    PYTHONDONTWRITEBYTECODE=1 python scripts/cpu_ref_vs_oracle.py [threads] [timed steps]
"""
import json
import os
import statistics
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
import models.modules.encoders as R_enc  # noqa: E402

import oracle as O  # noqa: E402
from openvivqa_amd.config import ConfigNode, attention_config  # noqa: E402
from openvivqa_amd.mcan_stack import synthetic_batch  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.set_num_threads(threads)
sa = attention_config()
cfg_t = ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa))
cfg_v = ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa,
                        GUIDED_ATTENTION=sa))
v, vm, t, tm = synthetic_batch(64, 100, 20, 512, 80, 8, 1234, "cpu", torch.float32)
gt = torch.Generator().manual_seed(1)
tv, tt = torch.randn(v.shape, generator=gt), torch.randn(t.shape, generator=gt)


def timed(te, ve):
    te.train(), ve.train()
    opt = torch.optim.Adam(list(te.parameters()) + list(ve.parameters()), lr=1e-4, betas=(0.9, 0.98))

    def one():
        lo = te(t, tm)
        vo = ve(v, vm, lo, tm)
        loss = (vo - tv).pow(2).mean() + (lo - tt).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()
    for _ in range(2):
        one()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    return ts


out = {"threads": threads, "timed_steps": steps}
for name, (te, ve) in {"reference": (R_enc.Encoder(cfg_t), R_enc.GuidedAttentionEncoder(cfg_v)),
                       "oracle": (O.OracleEncoder(cfg_t), O.OracleGuidedAttentionEncoder(cfg_v))}.items():
    torch.manual_seed(0)
    ts = timed(te, ve)
    med = statistics.median(ts)
    out[name] = {"s_per_step_median": round(med, 3), "min": round(min(ts), 3), "max": round(max(ts), 3),
                 "samples_per_s": round(64 / med, 2)}
    print(name, out[name], flush=True)
out["oracle_over_reference"] = round(out["oracle"]["samples_per_s"] / out["reference"]["samples_per_s"], 3)
print(json.dumps(out))
