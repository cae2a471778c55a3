#!/usr/bin/env python3
"""hipBLASLt (torch.nn.functional.linear) at the shapes of scripts/gemm256_dev.hip, same box, same call: yardstick only."""
import torch, json
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, N, K in [(6400, 2048, 512), (6400, 1536, 512), (6400, 512, 2048), (6400, 512, 512), (6400, 3072, 768), (8192, 4096, 4096)]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 0.2 - 0.1).bfloat16()
    us = t(lambda: torch.nn.functional.linear(x, w))
    print(json.dumps({"M": M, "N": N, "K": K, "hipblaslt_us": round(us, 2), "TF": round(2 * M * N * K / us * 1e-6, 1)}), flush=True)
