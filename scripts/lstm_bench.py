#!/usr/bin/env python3
"""Time the persistent LSTM launches alone (hipGraph replay of fwd + bwd, HIP events), (one hand-off form is left: the data is the flag).
    python scripts/lstm_bench.py [B T]"""
import os
import sys
import json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvivqa_amd import ops  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 20)
H, dev = 512, "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(T * B, H, generator=g).to(dev, torch.bfloat16)
w_ih = (torch.rand(4 * H, H, generator=g) * 2 - 1).mul(H ** -0.5).to(dev, torch.bfloat16)
w_hh = (torch.rand(4 * H, H, generator=g) * 2 - 1).mul(H ** -0.5).to(dev, torch.bfloat16)
b0, b1 = torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)
dy = torch.randn(B, T, H, generator=g).to(dev)
wt = w_hh.t().contiguous()
out = {}
for form in ("sentinel",):  # (rounds 4-5 also timed the counter / fence hand-off forms here: profiles/README.md)
    res = {}
    for which in ("fwd", "bwd"):
        y, hseq, saved, _ = ops.lstm_fwd(x, w_ih, w_hh, b0, b1, B, T)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                if which == "fwd":
                    ops.lstm_fwd(x, w_ih, w_hh, b0, b1, B, T)
                else:
                    ops.lstm_bwd(dy, w_hh, wt, saved, B, T, H)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[which + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
    res["us_per_step"] = round((res["fwd_us"] + res["bwd_us"]) / (2 * T), 2)
    out[form] = res
print(json.dumps({"B": B, "T": T, "H": H, "includes": "the memset node(s) in front of each launch", "forms": out}))
