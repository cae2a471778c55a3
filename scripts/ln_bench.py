#!/usr/bin/env python3
"""Time the LayerNorm launches alone, as the MCAN step uses them (fp32 pre-LN sum in, bf16 operand out; backward with the
dropout twin), from a replayed hipGraph.

    python scripts/ln_bench.py [M ...]      (default: 6400 1280)

Two cache states per size: `cold` = a graph of NBUF launches over NBUF different buffer sets (> 256 MB together: neither
the L2s nor the infinity cache hold a launch's operands when its turn comes), `warm` = the same launch NBUF times on one
buffer set.  The in-step figures of profiles/*_step_kernel_stats.csv lie between the two.  (Round 5 swept kernel forms through
environment switches with this script: profiles/r05_ln_bench*.jsonl; the forms that lost are gone from the library.)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvivqa_amd import ops  # noqa: E402

D, dev = 512, "cuda"
sizes = [int(a) for a in sys.argv[1:]] or [6400, 1280]
REPLAYS = 20


def timed(fn, n):
    gr = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPLAYS):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / (REPLAYS * n), 2)


out = {"D": D, "env": {k: v for k, v in os.environ.items() if k.startswith("OVQA_LN")}}
for M in sizes:
    nbuf = max(8, int(400e6 // (M * D * 10)))
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(M, D, generator=g).to(dev) for _ in range(nbuf)]
    dys = [torch.randn(M, D, generator=g).to(dev, torch.bfloat16) for _ in range(nbuf)]
    gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    dgamma, dbeta = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    drop = ops.DropSpec(0.1, 1, 7)
    _, mean, rstd = ops.layernorm_fwd(xs[0], gamma, beta, out_dtype=torch.bfloat16)
    q = ops.WgradQueue()
    res = {}

    def fwd(bufs):
        for x in bufs:
            ops.layernorm_fwd(x, gamma, beta, out_dtype=torch.bfloat16)

    def bwd(pairs):
        for dy, x in pairs:
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, drop=drop, dx_dtype=torch.bfloat16, defer=q)
        q.reduces.clear()

    res["fwd_cold_us"] = timed(lambda: fwd(xs), nbuf)
    res["fwd_warm_us"] = timed(lambda: fwd([xs[0]] * nbuf), nbuf)
    res["bwd_cold_us"] = timed(lambda: bwd(list(zip(dys, xs))), nbuf)
    res["bwd_warm_us"] = timed(lambda: bwd([(dys[0], xs[0])] * nbuf), nbuf)
    res["nbuf"] = nbuf
    out[str(M)] = res
    del xs, dys
print(json.dumps(out))
