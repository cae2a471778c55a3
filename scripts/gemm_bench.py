#!/usr/bin/env python3
"""Per-shape timing of the GEMM entry points (hipGraph replay, HIP events)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvivqa_amd import ops

dev = "cuda"
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

shapes = [(6400, 2048, 512), (6400, 512, 2048), (6400, 1536, 512), (6400, 512, 512), (1280, 2048, 512),
          (1280, 512, 2048), (1280, 1536, 512), (1280, 6144, 512), (1280, 512, 512), (8192, 4096, 4096)]
which = sys.argv[1:] or ["fwd", "bwd_data", "bwd_weight"]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); u = torch.empty_like(y)
    dy = torch.randn(M, N, device=dev).bfloat16(); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    fl = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K}
    if "fwd" in which:
        for name, fn in (("bias", lambda: ops.linear_fwd(x, w, b, out=y)),
                         ("gelu", lambda: ops.linear_fwd(x, w, b, ops.EPI_BIAS_GELU, out=y, preact_out=u)),
                         ("gelu_no_preact_store", lambda: ops.linear_fwd(x, w, b, ops.EPI_BIAS_GELU, out=y)),
                         ("resid", lambda: ops.linear_fwd(x, w, b, ops.EPI_BIAS_RESIDUAL, residual=r, out=y))):
            t = timeit(fn); row[name] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
        t = timeit(lambda: torch.nn.functional.linear(x, w)); row["torch"] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
    if "bwd_data" in which and M < 8000:
        wt = w.t().contiguous()
        t = timeit(lambda: ops.linear_bwd_data_wt(dy, wt, out=dx)); row["dX_wt"] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
        t = timeit(lambda: torch.matmul(dy, w)); row["torch_dX"] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
    if "bwd_weight" in which and M < 8000:
        t = timeit(lambda: ops.linear_bwd_weight(dy, x, dw, db)); row["dW"] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
        t = timeit(lambda: torch.matmul(dy.t(), x)); row["torch_dW"] = f"{t*1e6:.1f}us {fl/t/1e12:.0f}TF"
    print(json.dumps(row), flush=True)
