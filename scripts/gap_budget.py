#!/usr/bin/env python3
"""Per-kernel-family gap budget of the headline step (profiles/README.md).

    python scripts/gap_budget.py [profiles/r06_step_kernel_stats.csv profiles/r06_traffic.json]

For every family of the replayed step: launches and microseconds per step (rocprofv3 kernel trace, replay window),
ALGORITHMIC flops and bytes (every operand once, bf16 unless the path stores fp32), measured HBM bytes (separate FETCH_SIZE
/ WRITE_SIZE passes, FETCH x 2 on gfx950), the floor max(flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) and the gap.
Shapes: BASELINE configs[1] (B = 64, 100 regions x 20 tokens, d = 512, dff = 2048, L = 6)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (gemm_launch_list: the launch list of one step)

stats = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_step_kernel_stats.csv")
traffic = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_traffic.json")
B, NV, NT, D, DFF, L, H = 64, 100, 20, 512, 2048, 6, 8
rows = [r for r in csv.reader(open(stats)) if r and not r[0].startswith("#") and r[0] != "kernel"]
tr = json.load(open(traffic))


def meas(pred):
    tot = n = 0
    for k, v in tr.items():
        if pred(k):
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return tot / n if n else None


fused_adam = any("grouped_wgrad_adam" in r[0] for r in rows)
fam = bench.gemm_launch_list(B, NV, NT, D, DFF, L)
mv, mt = B * NV, B * NT
# the fc_o dX products that run inside the attention backward kernels (guided, question self-attention and -- round 5 -- the
# image self-attention) are not GEMM launches
fused_dx = [(mv, D, D)] * (2 * L) + [(mt, D, D)] * L
dx = list(fam["dx"])
for sh in fused_dx:
    for i, s in enumerate(dx):
        if s[:3] == sh and s[3] == "":
            dx.pop(i)
            break


def gemm_bytes(sh, kind):
    M, N, K = sh[:3]
    if kind == "bias":
        return 2 * (M * K + N * K + M * N)
    if kind == "gelu":
        return 2 * (M * K + N * K + 2 * M * N)               # y and the pre-activation
    if kind == "residual":
        return 2 * (M * K + N * K) + 4 * M * N * 2           # fp32 residual stream in and out
    flag = sh[3]
    return 2 * (M * N + N * K + M * K) + (2 * M * K if flag in ("a", "g") else 0)  # + addend / pre-activation


families = [
    ("dX GEMMs (MEpiBwdData)", lambda k: "MEpiBwdData" in k and "gemm_bf16" in k, sum(2.0 * s[0] * s[1] * s[2] for s in dx),
     sum(gemm_bytes(s, "dx") for s in dx)),
    ("forward GEMMs, bias (QKV / KV projections)", lambda k: "MEpiBias," in k or "MEpiBias>" in k or k.endswith("MEpiBias)"),
     None, None),
    ("forward GEMMs, bias + GELU (fc1)", lambda k: "MEpiBiasGelu" in k, sum(2.0 * s[0] * s[1] * s[2] for s in fam["gelu"]),
     sum(gemm_bytes(s, "gelu") for s in fam["gelu"])),
    ("forward GEMMs, fp32 residual epilogue (fc_o, fc2)", lambda k: "MEpiBiasRes32" in k,
     sum(2.0 * s[0] * s[1] * s[2] for s in fam["residual"]), sum(gemm_bytes(s, "residual") for s in fam["residual"])),
    # (round 5, N = 1: Adam of the weight matrices runs in this launch's epilogue -- 28 B per weight instead of the 4 B of a
    # stored gradient)
    ("grouped dW + Adam of the weight matrices (one launch)" if fused_adam else "grouped dW (one launch)",
     lambda k: "grouped_wgrad" in k, 0.348e12,
     2 * sum(s[0] * (s[1] + s[2]) for f in ("bias", "gelu", "residual") for s in fam[f]) + (28 if fused_adam else 4) * 44.1e6),
    ("attention forward, projections inside", lambda k: "attn_qkv_fwd" in k or "attn_q_fwd" in k, None, None),
    ("attention backward", lambda k: "attn_bwd" in k, None, None),
    ("LayerNorm forward", lambda k: "ln_fwd" in k, 0.0, None),
    ("LayerNorm backward + parameter reduce", lambda k: "ln_bwd" in k, 0.0, None),
    ("Adam of the 1-D parameters" if fused_adam else "Adam (tiled: master, moments, shadow, transposed shadow)",
     lambda k: "adam" in k, 0.0, None if fused_adam else 44.1e6 * 30),
    ("loss, step counter, stray elementwise", lambda k: True, 0.0, None),
]
# the one hoisted K | V projection + the six fused-QKV kernels carry the 'bias' family's flops inside other rows
bias_flops = sum(2.0 * s[0] * s[1] * s[2] for s in fam["bias"] if s[0] == mt and s[1] == 2 * D * L)
bias_bytes = sum(gemm_bytes(s, "bias") for s in fam["bias"] if s[0] == mt and s[1] == 2 * D * L)
qkv_flops = sum(2.0 * s[0] * s[1] * s[2] for s in fam["bias"] if not (s[0] == mt and s[1] == 2 * D * L))
att_fwd_core = 4.0 * B * H * 64 * (NV * NV + NV * NT + NT * NT) * L
att_bwd_core = 10.0 * B * H * 64 * (NV * NV + NV * NT + NT * NT) * L + sum(2.0 * s[0] * s[1] * s[2] for s in fused_dx)
att_fwd_bytes = L * 2 * ((mv * D + 3 * D * D + mv * 3 * D + 2 * mv * D) + (mv * D + D * D + mv * D + 2 * mt * D + 2 * mv * D)
                         + (mt * D + 3 * D * D + mt * 3 * D + 2 * mt * D))
att_bwd_bytes = L * 2 * ((4 * mv * D + 2 * mv * D + mv * D + 2 * mv * D) + (mv * D + D * D + 3 * mv * D + 2 * mt * D + mv * D + 2 * mt * D)
                         + (mt * D + D * D + 3 * mt * D + 2 * mt * D + 3 * mt * D))
fixed = {"forward GEMMs, bias (QKV / KV projections)": (bias_flops, bias_bytes),
         "attention forward, projections inside": (qkv_flops - bias_flops * 0 + att_fwd_core, att_fwd_bytes),
         "attention backward": (att_bwd_core, att_bwd_bytes)}
used = set()
out = []
for name, pred, flops, nbytes in families:
    sel = [r for r in rows if r[0] not in used and pred(r[0])]
    used.update(r[0] for r in sel)
    if not sel:
        continue
    launches = sum(float(r[1]) for r in sel)
    us = sum(float(r[2]) for r in sel)
    if name in fixed:
        flops, nbytes = fixed[name]
    m = meas(lambda k: any(k.startswith(r[0][:60]) or r[0].startswith(k[:60]) for r in sel))
    mbytes = m * launches if m else None
    if nbytes is None:
        nbytes = mbytes  # streaming kernels: what they moved is what they had to move (to first order)
    floor = max((flops or 0.0) / 2.5e15, (nbytes or 0.0) / 8e12) * 1e6
    out.append((name, launches, us, flops, nbytes, mbytes, floor, us - floor))
print("| family | launches | us / step | algorithmic GFLOP | algorithmic MB | measured HBM MB | floor us = max(MFMA, HBM) | gap us |")
print("|---|---|---|---|---|---|---|---|")
for name, launches, us, flops, nbytes, mbytes, floor, gap in out:
    f = lambda v, s=1.0: "-" if v is None else f"{v / s:.0f}"
    print(f"| {name} | {launches:.0f} | {us:.0f} | {f(flops, 1e9)} | {f(nbytes, 1e6)} | {f(mbytes, 1e6)} | {floor:.0f} | {gap:.0f} |")
tot = sum(o[2] for o in out)
print(f"| **all** | {sum(o[1] for o in out):.0f} | **{tot:.0f}** | {sum((o[3] or 0) for o in out) / 1e9:.0f} | | | {sum(o[6] for o in out):.0f} | {sum(o[7] for o in out):.0f} |")
