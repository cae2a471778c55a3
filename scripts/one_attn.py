import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvivqa_amd import ops
B, H, nq, nk = 64, 8, int(sys.argv[1]), int(sys.argv[2]); mode = sys.argv[3]
dev = "cuda"
qkv = torch.randn(B, nq, 3 * 512, device=dev).bfloat16()
kv = torch.randn(B, nk, 2 * 512, device=dev).bfloat16()
q = qkv[..., :512]; k = kv[..., :512]; v = kv[..., 512:]
mask = torch.zeros(B, 1, 1, nk, device=dev); mask[:, :, :, nk - 5:] = -1e5
o, lse, _ = ops.attention_fwd(q, k, v, mask, H)
do = torch.randn_like(o)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
if mode == "time":
    def timeit(fn):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 200
    dq = torch.empty_like(q.contiguous()); dk = torch.empty_like(k.contiguous()); dv = torch.empty_like(v.contiguous())
    print(f"nq={nq} nk={nk} fwd {timeit(lambda: ops.attention_fwd(q, k, v, mask, H)):.1f}us  bwd {timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, mask, H, dq=dq, dk=dk, dv=dv)):.1f}us")
else:
    for _ in range(reps):
        if mode == "fwd": ops.attention_fwd(q, k, v, mask, H)
        else: ops.attention_bwd(do, q, k, v, o, lse, mask, H)
    torch.cuda.synchronize()
