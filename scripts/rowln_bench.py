#!/usr/bin/env python3
"""ovqa_linear_fwd_res32_ln: the row-complete kernel (csrc/gemm_rowln.h) against its two launches (OVQA_ROWLN=0), per shape,
hipGraph replay of 10 calls, HIP events.  The in-step comparison is scripts/gpu_ab_libs.sh / bench.py under OVQA_ROWLN."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvivqa_amd import ops


def timeit(fn, reps=30):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * 10)


dev = "cuda"
for (M, K) in [(6400, 512), (6400, 2048), (1280, 512), (1280, 2048), (640, 512), (64, 512), (64, 2048), (12800, 512)]:
    N = 512
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev); g = torch.ones(N, device=dev); be = torch.zeros(N, device=dev)
    prev = torch.randn(M, N, device=dev)
    _, mean, rstd = ops.layernorm_fwd(prev, g, be, 1e-5, out_dtype=torch.bfloat16)
    res = ops.LnRef(prev, mean, rstd, g, be, 1e-5)
    row = {"M": M, "N": N, "K": K}
    for name, env in (("one_kernel", "1"), ("two_launches", "0")):
        os.environ["OVQA_ROWLN"] = env
        t = timeit(lambda: ops.linear_fwd_res32_ln(x, w, b, res, g, be, 1e-5))
        row[name + "_us"] = round(t * 1e6, 2)
    print(json.dumps(row), flush=True)
