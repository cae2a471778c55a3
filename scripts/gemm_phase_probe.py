"""Development probe: phases inside the direct-to-LDS GEMM tile (library built with -DOVQA_PHASE_PROBE).
(Rounds 3-5 compared tile forms through environment switches here; the forms that lost are gone from the library.)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OVQA_EXTRA_HIPCC_FLAGS"] = "-DOVQA_PHASE_PROBE"
from openvivqa_amd import build as B  # noqa: E402

if os.environ.get("OVQA_PROBE_BUILD", "1") == "1":
    B.build(force=True, verbose=False)
import torch  # noqa: E402

from openvivqa_amd import _lib, ops  # noqa: E402

lib = _lib.load()


def run_gemm(M, N, K, label, flush=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(K, N, device="cuda", generator=g) * N ** -0.5).bfloat16()
    add = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for it in range(4):
        if flush:
            junk.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_bwd_data_wt(dy, wt, addend=add)
        e1.record()
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.ovqa_debug_probe_gemm(out) == 0
    nm = ["start", "prologue issued", "first K tile done", "K loop done", "epilogue done", "exchanged"]
    for wg in range(2):
        ts = [out[wg * 16 + i] for i in range(len(nm))]
        print(f"{label} {'cold' if flush else 'warm'} wg{'0' if wg == 0 else 'mid'}: " +
              "  ".join(f"{a} {((t - ts[0]) / 100.0):.2f}" for a, t in zip(nm, ts) if t >= ts[0]) +
              f"   [events {e0.elapsed_time(e1) * 1e3:.1f} us]")


for flush in (False, True):
    run_gemm(6400, 512, 512, "dX 6400x512<-512", flush)
    run_gemm(6400, 2048, 512, "dX 6400x512<-2048", flush)
    run_gemm(6400, 512, 2048, "dX 6400x2048<-512", flush)
