#!/usr/bin/env python3
"""Phase stamps of one wave of the persistent LSTM forward (OVQA_LSTM_PROBE=1): per step, microseconds between
step start -> operands ready + MFMAs issued -> gates + LDS tile -> barrier -> stores issued -> input half of t + 1."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OVQA_LSTM_PROBE"] = "1"
from openvivqa_amd import ops  # noqa: E402
B, T, H, dev = 64, 20, 512, "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(T * B, H, generator=g).to(dev, torch.bfloat16)
w = [(torch.rand(4 * H, H, generator=g) * 2 - 1).mul(H ** -0.5).to(dev, torch.bfloat16) for _ in range(2)]
b = [torch.zeros(4 * H, device=dev) for _ in range(2)]
for form in ("sentinel",):
    for rep in range(3):
        y, hseq, saved, scratch = ops.lstm_fwd(x, w[0], w[1], b[0], b[1], B, T)
    torch.cuda.synchronize()
    st = scratch.view(torch.int32)[512:512 + 8 * T].view(T, 8).cpu().long() & 0xFFFFFFFF
    print(form, "columns: wait+mfma | gates+tile | barrier | stores | xpart   (us, 100 MHz stamps); last: whole step")
    for t in range(1, T):
        r = st[t]
        d = [(int(r[i + 1]) - int(r[i])) / 100.0 for i in range(5)]
        print(f"  t={t:2d} " + " ".join(f"{v:6.2f}" for v in d) + f"   {(int(r[5]) - int(r[0])) / 100.0:6.2f}")
