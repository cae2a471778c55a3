import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvivqa_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dev = "cuda"
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, N, device=dev).bfloat16(); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
for _ in range(20):
    if mode == "fwd": ops.linear_fwd(x, w, b, out=y)
    else: ops.linear_bwd_data(dy, w, out=dx)
torch.cuda.synchronize()
