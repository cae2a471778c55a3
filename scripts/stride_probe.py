#!/usr/bin/env python3
"""Does a power-of-two row pitch of the activation operand (K = 512 bf16 = 1 KiB rows, K = 2048 = 4 KiB) cost L2
channel conflicts in the direct-to-LDS staging?  linear_fwd on x with padded row pitches (views of a wider buffer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvivqa_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 10)


for (M, N, K) in [(6400, 512, 512), (6400, 2048, 512), (6400, 512, 2048), (1280, 512, 2048), (1280, 512, 512)]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = {}
    for pad in (0, 8, 64, 128, 192):
        buf = torch.randn(M, K + pad, device="cuda").bfloat16()
        x = buf[:, :K]
        row[pad] = round(timeit(lambda: ops.linear_fwd(x, w, b, out=y)), 2)
    print(f"M={M} N={N} K={K}: us by x row-pitch padding (elements) {row}", flush=True)
