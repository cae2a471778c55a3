#!/usr/bin/env python3
"""Feasibility probe: does running the step as TWO independent half-batch chains on two streams (one fork, one join,
inside one captured graph) beat one full-batch chain?  Forward of both stacks, no-grad; B=64 as one chain vs 2 x B=32
concurrently vs 2 x B=32 back to back on one stream."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openvivqa_amd as A  # noqa: E402
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
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda", 0)
    cfg = A.get_config(os.path.join(ROOT, "configs", "mcan_bench.yaml"))
    A.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = MCANEncoderStack(cfg.MODEL).to(dev).eval()
    v, vm, t, tm = synthetic_batch(64, 100, 20, 512, 80, 8, 3, dev, torch.bfloat16)
    halves = [tuple(x[i * 32:(i + 1) * 32].contiguous() for x in (v, vm, t, tm)) for i in range(2)]
    quarters = [tuple(x[i * 16:(i + 1) * 16].contiguous() for x in (v, vm, t, tm)) for i in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]

    def full():
        model(v, vm, t, tm)

    def serial(parts):
        def f():
            for h in parts:
                model(*h)
        return f

    def concurrent(parts):
        def f():
            main_s = torch.cuda.current_stream()
            for s, h in zip(streams, parts):
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    model(*h)
            for s, _ in zip(streams, parts):
                main_s.wait_stream(s)
        return f
    with torch.no_grad():
        model(v, vm, t, tm)
        for name, fn in (("one chain, B=64", full), ("two chains back to back, 2 x B=32", serial(halves)),
                         ("two chains on two streams, 2 x B=32", concurrent(halves)),
                         ("four chains on four streams, 4 x B=16", concurrent(quarters))):
            print(f"{name:45s} {graph_time(fn):9.1f} us per forward of 64 samples", flush=True)


if __name__ == "__main__":
    main()
