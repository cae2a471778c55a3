"""Development probe: start / K-loop-done / end stamps and the CU of EVERY workgroup of one GEMM launch (library built
with -DOVQA_PHASE_PROBE), to see where a launch's time goes beyond one workgroup's own timeline."""
import ctypes as C
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OVQA_EXTRA_HIPCC_FLAGS"] = "-DOVQA_PHASE_PROBE"
from openvivqa_amd import build as B  # noqa: E402

if os.environ.get("OVQA_PROBE_BUILD", "1") == "1":
    B.build(force=True, verbose=False)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from openvivqa_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.ovqa_debug_probe_gemm_wgs.argtypes = [C.c_void_p, C.c_int]


def pct(a, qs=(0, 10, 50, 90, 100)):
    return " ".join(f"{np.percentile(a, q):.2f}" for q in qs)


def timeline(label, fn, n_wg_hint=4096):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4 * 4096))()
    assert lib.ovqa_debug_probe_gemm_wgs(buf, 4096) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).copy()
    t_last = a[:, 0].max()
    live = a[:, 0] > t_last - 100 * 200          # stamps of the last launch only (within 200 us)
    a = a[live]
    n = len(a)
    t0 = a[:, 0].min()
    st, kd, en = [(a[:, i].astype(np.int64) - int(t0)) / 100.0 for i in range(3)]
    hw = a[:, 3]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) * 1000 + \
        ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) * 100 + \
        ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64) * 50 + ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    per_cu = collections.Counter(cu.tolist())
    hist = collections.Counter(per_cu.values())
    print(f"== {label}: {n} workgroups on {len(per_cu)} CUs; workgroups per CU histogram {dict(sorted(hist.items()))}")
    print(f"   span (first start -> last end) {en.max():.2f} us")
    print(f"   start   p0/10/50/90/100: {pct(st)}")
    print(f"   end     p0/10/50/90/100: {pct(en)}")
    print(f"   own time (end - start)  : {pct(en - st)}")
    print(f"   K loop (start -> K done): {pct(kd - st)}   epilogue: {pct(en - kd)}")
    # second-round workgroups: started after some workgroup had already ended
    late = st > en.min()
    print(f"   workgroups that started after the first one ended: {int(late.sum())}; their start p50 {np.median(st[late]) if late.any() else 0:.2f}")
    if os.environ.get("OVQA_WG_PLACEMENT") == "1":  # where did the k-th workgroup of XCD x go?  (blockIdx = 8 k + x)
        idx_live = np.nonzero(live)[0]
        xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
        se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
        cu_in = (((hw >> np.uint64(12)) & np.uint64(1)) * 16 + ((hw >> np.uint64(8)) & np.uint64(15))).astype(np.int64)
        for x in (0, 5):
            sel = [i for i, b in enumerate(idx_live) if b % 8 == x]
            print(f"   blockIdx % 8 == {x}: XCC ids {sorted(set(xcc[sel].tolist()))}; (SE, CU) of k = 0.. :",
                  " ".join(f"{se[i]}.{cu_in[i]}" for i in sel[:72]))
    by_cnt = {c: [] for c in hist}
    for k, c in per_cu.items():
        by_cnt[c].append(en[cu == k].max())
    for c in sorted(by_cnt):
        print(f"   CUs holding {c}: last end p50 {np.median(by_cnt[c]):.2f} max {np.max(by_cnt[c]):.2f}")


def dx(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(0)
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(K, N, device="cuda", generator=g) * N ** -0.5).bfloat16()
    add = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    return lambda: ops.linear_bwd_data_wt(dy, wt, addend=add)


def fwd(M, N, K, epi=0):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda")
    return lambda: ops.linear_fwd(x, w, b, epilogue=epi)


timeline("dX 6400x512 <- 512", dx(6400, 512, 512))
timeline("dX 6400x512 <- 2048", dx(6400, 2048, 512))
timeline("dX 6400x2048 <- 512", dx(6400, 512, 2048))
timeline("fwd 6400x512 <- 512", fwd(6400, 512, 512))
timeline("fwd 6400x2048 <- 512 gelu", fwd(6400, 2048, 512, 1))
timeline("dX 1280x512 <- 512", dx(1280, 512, 512))
