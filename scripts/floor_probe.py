#!/usr/bin/env python3
"""Where does the ~4.7 us a tiny kernel shows inside the captured training step come from?  scripts/boundary_bench.py
measures 1.6 us per dependent launch for a chain of ONE trivial kernel; this probe replays chains of the library's
REAL kernels from a torch CUDAGraph (no profiler) and reports microseconds per launch:

  A  N x ln_fwd(M=1280) on one buffer            (same code, warm data)
  B  N x ln_fwd(M=1280) cycling over 16 buffers   (same code, colder data)
  C  N x increment_step                           (1-wave kernel)
  D  alternating increment_step / ln_fwd / tiny GEMM / attention (different code every launch)
  E  the text EncoderLayer forward sequence (7 launches) x 6 layers, no-grad  -> us per layer (58 in the step trace)
  F  the guided layer forward sequence x 6                                   -> us per layer (155 in the step trace)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openvivqa_amd as A  # noqa: E402
from openvivqa_amd import ops  # noqa: E402
from openvivqa_amd.mcan_stack import MCANEncoderStack, synthetic_batch  # noqa: E402


def graph_time(fn, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        g.replay()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us per replay


def main():
    dev = torch.device("cuda", 0)
    out = {}
    D, N = 512, 64
    g = torch.ones(D, device=dev)
    b = torch.zeros(D, device=dev)
    xs = [torch.randn(1280, D, device=dev).bfloat16() for _ in range(16)]
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    out["A ln_fwd same buffer"] = graph_time(lambda: [ops.layernorm_fwd(xs[0], g, b) for _ in range(N)]) / N
    out["B ln_fwd 16 buffers"] = graph_time(lambda: [ops.layernorm_fwd(xs[i % 16], g, b) for i in range(N)]) / N
    out["C increment_step"] = graph_time(lambda: [ops.increment_step(step) for _ in range(N)]) / N
    w = (torch.randn(512, 512, device=dev) * 0.04).bfloat16()
    qkv = torch.randn(64, 20, 1536, device=dev).bfloat16()
    mask = torch.zeros(64, 1, 1, 20, device=dev)

    def mixed():
        for i in range(N // 4):
            ops.increment_step(step)
            ops.layernorm_fwd(xs[i % 16], g, b)
            ops.linear_fwd(xs[i % 16], w, b)
            ops.attention_fwd(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], mask, 8)
    out["D mixed 4 kernels"] = graph_time(mixed) / N
    out["D' ln_fwd only (same count)"] = out["B ln_fwd 16 buffers"]
    single = {
        "linear 1280x512x512": lambda: ops.linear_fwd(xs[0], w, b),
        "attention 20x20": lambda: ops.attention_fwd(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], mask, 8),
    }
    for k, f in single.items():
        out["chain of " + k] = graph_time(lambda: [f() for _ in range(N)]) / N

    cfg = A.get_config(os.path.join(ROOT, "configs", "mcan_bench.yaml"))
    A.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = MCANEncoderStack(cfg.MODEL).to(dev).eval()
    v, vm, t, tm = synthetic_batch(64, 100, 20, 512, 80, 8, 3, dev, torch.bfloat16)
    with torch.no_grad():
        model(v, vm, t, tm)
        out["E text stack fwd (6 layers), us per layer"] = graph_time(lambda: model.self_encoder(t, tm)) / 6
        lo = model.self_encoder(t, tm)
        out["F guided stack fwd (6 layers), us per layer"] = graph_time(lambda: model.guided_encoder(v, vm, lo, tm)) / 6
    for k, val in out.items():
        print(f"{k:48s} {val:8.2f} us")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "floor_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
