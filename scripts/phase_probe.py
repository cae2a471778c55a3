"""Development probe: where the time goes INSIDE the small attention backward kernels.

Builds the library with -DOVQA_PHASE_PROBE (wall-clock timestamps of thread 0 of two workgroups at marked points),
runs the MCAN question self-attention (20 x 20) and guided attention (100 x 20) backward shapes, prints the phase
deltas in microseconds (100 MHz counter: 10 ns resolution).  Run on the GPU box:  python scripts/phase_probe.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OVQA_EXTRA_HIPCC_FLAGS"] = "-DOVQA_PHASE_PROBE"
from openvivqa_amd import build as B  # noqa: E402

B.build(force=True, verbose=False)
import torch  # noqa: E402

from openvivqa_amd import _lib, ops  # noqa: E402

lib = _lib.load()
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["start", "loads issued", "staged", "barrier", "dQ done", "dK/dV mfma", "reduced", "end"]


def run(nq, nk, label, flush=False):
    Bn, H, D = 64, 8, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(Bn, nq, 3 * D, device="cuda", generator=g).bfloat16()
    kv = torch.randn(Bn, nk, 2 * D, device="cuda", generator=g).bfloat16()
    q = qkv[..., :D]
    k, v = (qkv[..., D:2 * D], qkv[..., 2 * D:]) if nq == nk else (kv[..., :D], kv[..., D:])
    mask = torch.zeros(Bn, 1, 1, nk, device="cuda")
    mask[:, :, :, nk - 3:] = -1e5
    o, lse, _ = ops.attention_fwd(q, k, v, mask, H)
    d_o = torch.randn_like(o)
    junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for it in range(4):
        if flush:
            junk.fill_(1.0)  # push the operands out of L2 / MALL
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_bwd(d_o, q, k, v, o, lse, mask, H)
        e1.record()
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.ovqa_debug_probe(out) == 0
    for wg in range(2):
        ts = [out[wg * 16 + i] for i in range(len(names))]
        t0 = ts[0]
        print(f"{label} {'cold' if flush else 'warm'} wg{'0' if wg == 0 else 'mid'}: " +
              "  ".join(f"{n} {((t - t0) / 100.0):.2f}" for n, t in zip(names, ts) if t >= t0) +
              f"   [events {e0.elapsed_time(e1) * 1e3:.1f} us]")


for flush in (False, True):
    run(20, 20, "text 20x20", flush)
    run(100, 20, "guided 100x20", flush)
def run_fused(n, label, flush=False):
    Bn, H, D = 64, 8, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(Bn, n, D, device="cuda", generator=g).bfloat16()
    w = (torch.randn(3 * D, D, device="cuda", generator=g) * D ** -0.5).bfloat16()
    bias = torch.zeros(3 * D, device="cuda")
    mask = torch.zeros(Bn, 1, 1, n, device="cuda")
    mask[:, :, :, n - 3:] = -1e5
    junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for it in range(4):
        if flush:
            junk.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_qkv_fwd(x, w, bias, mask, H)
        e1.record()
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.ovqa_debug_probe(out) == 0
    nm = ["start", "K loop", "barrier", "epilogue", "barrier", "attention"]
    for wg in range(2):
        ts = [out[wg * 16 + i] for i in range(len(nm))]
        print(f"{label} {'cold' if flush else 'warm'} wg{'0' if wg == 0 else 'mid'}: " +
              "  ".join(f"{a} {((t - ts[0]) / 100.0):.2f}" for a, t in zip(nm, ts)) + f"   [events {e0.elapsed_time(e1) * 1e3:.1f} us]")


def run_gemm(M, N, K, label, flush=False, kind="dx"):
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(K, N, device="cuda", generator=g) * N ** -0.5).bfloat16()
    add = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for it in range(4):
        if flush:
            junk.fill_(1.0)
        torch.cuda.synchronize()
        ops.linear_bwd_data_wt(dy, wt, addend=add)
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.ovqa_debug_probe_gemm(out) == 0
    nm = ["start", "prologue issued", "first K tile done", "K loop done", "epilogue done"]
    for wg in range(2):
        ts = [out[wg * 16 + i] for i in range(len(nm))]
        print(f"{label} {'cold' if flush else 'warm'} wg{'0' if wg == 0 else 'mid'}: " +
              "  ".join(f"{a} {((t - ts[0]) / 100.0):.2f}" for a, t in zip(nm, ts)))


for flush in (False, True):
    run_gemm(6400, 512, 512, "dX 6400x512<-512", flush)
    run_gemm(6400, 1536, 512, "dX 6400x512<-1536", flush)
    run_gemm(6400, 2048, 512, "dX 6400x512<-2048", flush)
    run_gemm(6400, 512, 2048, "dX 6400x2048<-512", flush)
    run_gemm(1280, 512, 512, "dX 1280x512<-512", flush)
for flush in (False, True):
    run_fused(100, "fused fwd 100", flush)
    run_fused(20, "fused fwd 20", flush)
names = ["start", "images", "delta", "barrier", "dQ waves end", "dK/dV waves end"]
for flush in (False, True):
    run(100, 100, "image 100x100", flush)


def run_do(nq, nk, label, names, flush=False):
    """ovqa_attention_bwd_do: the fc_o dX product inside the attention backward (20 x 20: two heads per 4-wave workgroup;
    100 x 100: the role-split kernel)."""
    Bn, H, D = 64, 8, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(Bn, nq, 3 * D, device="cuda", generator=g).bfloat16()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    mask = torch.zeros(Bn, 1, 1, nk, device="cuda")
    mask[:, :, :, nk - 3:] = -1e5
    lo = []
    o, lse, _ = ops.attention_fwd(q, k, v, mask, H, lo_out=lo)
    dy = torch.randn(Bn, nq, D, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(D, D, device="cuda", generator=g) * D ** -0.5).bfloat16()
    if not ops.attention_bwd_do_ok(dy, wt, q, k, mask, H):
        print(label, "not covered by the fused form")
        return
    junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for it in range(4):
        if flush:
            junk.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_bwd_do(dy, wt, q, k, v, o, lse, mask, H, o_lo=lo[0])
        e1.record()
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.ovqa_debug_probe(out) == 0
    for wg in range(2):
        ts = [out[wg * 16 + i] for i, _ in names]
        print(f"{label} {'cold' if flush else 'warm'} wg{'0' if wg == 0 else 'mid'}: " +
              "  ".join(f"{n} {((t - ts[0]) / 100.0):.2f}" for (_, n), t in zip(names, ts)) +
              f"   [events {e0.elapsed_time(e1) * 1e3:.1f} us]")


for flush in (False, True):
    run_do(20, 20, "fused dO backward 20x20",
           [(0, "start"), (1, "projection loop"), (2, "loads issued"), (3, "images + delta"), (4, "dQ waves"), (7, "end")], flush)
    run_do(100, 100, "fused dO backward 100x100",
           [(0, "start"), (6, "projection loop"), (1, "loads issued"), (2, "images + delta"), (3, "barrier"),
            (4, "dQ waves end"), (5, "dK/dV waves end")], flush)
