#!/usr/bin/env python3
"""Headline benchmark: VQA samples/s, forward+backward+optimizer step of the MCAN encoder stack
(d=512, L=6, B=64 per GPU, 100 regions x 20 tokens, bf16) -- BASELINE.json `metric`, configs[1].

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU (RCCL = torch.distributed "nccl"), weak scaling (64 samples per GPU).
A "step" is: forward -> loss -> backward -> gradient all-reduce -> Adam, on synthetic inputs
already resident in HBM.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP
events, live) and `cpu_baseline` (the oracle restatement timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_SAMPLE_FWD_BWD = 16.31e9  # MCAN encoders L=6, matmul FLOPs, fwd+bwd = 3x fwd
PEAK_BF16 = 2.5e15                   # dense MFMA peak, MI355X_MICROARCH.md chip table
PEAK_HBM = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "mcan_bench.yaml"))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="dtype of the gradient all-reduce.  fp32 (default) is the reference's arithmetic (DDP sums fp32 "
                         "gradients); bf16 halves the bytes per xGMI link and moves 0.08 %% of the weights by more than "
                         "one bf16 step of their own value after two Adam steps (tests/test_train_gpu.py::"
                         "test_data_parallel_exchange_bf16_vs_fp32_vs_single_process, DESIGN.md section 6)")
    ap.add_argument("--overlap-mb", type=float, default=48.0,
                    help="N>1: release a gradient segment to the all-reduce stream every this many MB (fp32) of "
                         "finished gradients during backward; 0 = one exchange after backward")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="diagnostic at N=1: run the N>1 code path (phased backward, comm stream, RCCL) on a "
                         "single-rank group")
    ap.add_argument("--beam", type=int, default=1, help="--workload decode: beam size (1 = greedy)")
    ap.add_argument("--workload", default="stack", choices=["stack", "model", "cross_modality", "decoder_train", "m4c_decode", "decode"],
                    help="stack = BASELINE's metric (the two encoder stacks, default); cross_modality = SECONDARY line for "
                         "BASELINE configs[2]: the whole CrossModalityTransformer model (configs/cross_modality_bench.yaml "
                         "= the reference's MODEL node at L=6: FeatureEmbedding 2048->512, UsualEmbedding, 6 "
                         "CrossModalityEncoder layers, pooling head, classifier, NLLLoss), 64 samples/GPU, data parallel "
                         "over --gpus ranks; model = SECONDARY diagnostic: "
                         "the whole MCAN model (FeatureEmbedding + LSTM text embedding + stacks + pooling head + "
                         "classifier + NLLLoss) on synthetic region features / token ids (SURVEY 8d); m4c_decode = "
                         "SECONDARY diagnostic for BASELINE configs[3]: M4C's multimodal transformer (hidden 768, 4 layers x "
                         "8 heads, 20 question + 100 region + 50 OCR + 12 decoding positions) run through the 12-pass "
                         "greedy decoding loop with the classifier || OcrPtrNet head, evaluation mode; decode = SECONDARY "
                         "diagnostic for BASELINE configs[4] (configs/vit_mbert_generation.yaml): autoregressive "
                         "answer decoding, Decoder L=3 over 237 encoder positions (ViT 197 + 40 question tokens), T=20, "
                         "--beam 1|3, tokens/s and the fraction of HBM peak against the bytes a step must stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed steps of the CPU leg at the best thread count")
    ap.add_argument("--cpu-threads", default="8,16,32,64,128",
                    help="thread counts the CPU leg sweeps (one B=16 step each) before timing at the best")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-shard", action="store_true",
                    help="N > 1: the replicated optimiser (all-reduce + the same Adam on every rank) instead of the sharded one "
                         "(reduce-scatter, Adam on the owned chunks, all-gather of the bf16 shadow)")
    ap.add_argument("--rehearse-shard", type=int, default=0,
                    help="with --rehearse-comm on ONE rank: run the sharded optimiser's machinery with the slice sizes of N ranks "
                         "(Adam on 1/N of every exchanged bucket; timing only, the other chunks are not updated)")
    ap.add_argument("--no-fuse-adam", action="store_true",
                    help="keep Adam a launch of its own at N = 1 too (default at N = 1: the weight matrices are updated inside "
                         "the last grouped weight-gradient launch, TrainStep(fuse_adam=True); with N > 1 the gradient exchange "
                         "stands between gradient and update and Adam is always separate)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (model, cross_modality, decoder_train, decode beam 1 / 3, m4c_decode) that the default "
                         "single-GPU run of the headline workload appends under `secondary`")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher self-test (no GPU): start the ranks exactly as a real run would, bring up a gloo "
                         "process group, count the ranks with an all-reduce and print the JSON skeleton with `n_gpus` = "
                         "the group's world size and `value` null")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of --steps steps: the first is the contract's measurement (`value`), the rest "
                         "give median / min / max in extra keys")
    return ap.parse_args()


def gemm_launch_list(B, NV, NT, D, DFF, L, in_step=False):
    """(M, N, K[, flag]) of every MFMA GEMM launch of one step, grouped by kernel family.
    Forward families: y[M,N] = x[M,K] w[N,K]^T (+epilogue).  dX family: dx[M,K] = dy[M,N] w[N,K] (computed
    from the transposed weight copy wt[K,N], as the training step does) with
    flag 'g' (fused dropout*GELU' epilogue, FFN seam), 'a' (residual-branch addend) or ''."""
    mv, mt = B * NV, B * NT
    bias, gelu, resid, dx = [], [], [], []
    for _ in range(L):  # text EncoderLayer
        bias.append((mt, 3 * D, D))
        resid.append((mt, D, D))
        gelu.append((mt, DFF, D))
        resid.append((mt, D, DFF))
        dx += [(mt, D, DFF, "g"), (mt, DFF, D, "a"), (mt, D, D, ""), (mt, 3 * D, D, "a")]
    for _ in range(L):  # GuidedEncoderLayer
        bias.append((mv, 3 * D, D))
        resid.append((mv, D, D))
        bias.append((mv, D, D))
        resid.append((mv, D, D))
        gelu.append((mv, DFF, D))
        resid.append((mv, D, DFF))
        dx += [(mv, D, DFF, "g"), (mv, DFF, D, "a"), (mv, D, D, ""), (mv, D, D, "a"), (mv, D, D, ""),
               (mv, 3 * D, D, "a")]
    # the guided layers' K/V projections of the question features are hoisted into one GEMM (fwd and dX)
    bias.append((mt, 2 * D * L, D))
    dx.append((mt, 2 * D * L, D, ""))
    if in_step:
        # what the step LAUNCHES: the fc_o dX products run inside the attention backward kernels (guided and question
        # self-attention since round 3, image self-attention since round 5), the self-attention QKV projections inside the
        # attention forward kernels
        for sh in [(mv, D, D, "")] * (2 * L) + [(mt, D, D, "")] * L:
            dx.remove(sh)
        bias = [sh for sh in bias if sh[1] != 3 * D]
        bias = [sh for sh in bias if not (sh[0] == mv and sh[1] == D)]  # (the guided query projection: inside attn_q_fwd)
    return {"bias": bias, "gelu": gelu, "residual": resid, "dx": dx}


KERNEL_OF_FAMILY = {
    "bias": "gemm_bf16_glds_kernel<false, false, MEpiBias,",
    "gelu": "gemm_bf16_glds_kernel<false, false, MEpiBiasGelu,",
    "residual": "gemm_bf16_glds_kernel<false, false, MEpiBiasResidual,",
    "dx": "gemm_bf16_glds_kernel<false, false, MEpiBwdData,",  # reads the transposed weight copy (row-major tile)
}


TRAFFIC_FILES = ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")
PMC_FILES = ("r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")


def pmc_traffic(kernel_prefix):
    """Per-launch HBM bytes of a kernel family from the newest COMMITTED PMC pass (profiles/rNN_traffic.json, produced by
    scripts/gpu_profile_step.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950, KB -> bytes).  Returns
    (bytes, file name): the figure is read from the file, not measured by this run -- PMC passes need rocprofv3."""
    path = next((p for p in (os.path.join(ROOT, "profiles", f) for f in TRAFFIC_FILES) if os.path.exists(p)), None)
    if path is None:
        return None, None
    key = kernel_prefix.replace("(anonymous namespace)::", "")
    tot_bytes, tot_n = 0.0, 0
    table = json.load(open(path))
    meta = table.pop("_meta", None)  # (round 6 on: commit, box and time of the collection)
    for name, v in table.items():  # a family = all tile-size instantiations of the kernel
        if key in name.replace("(anonymous namespace)::", ""):
            tot_bytes += v["hbm_bytes_per_launch"] * v["launches"]
            tot_n += v["launches"]
    src = "profiles/" + os.path.basename(path)
    if meta:
        src += f" (collected at commit {meta.get('commit')} on box {meta.get('box')}, {meta.get('collected_utc')})"
    return (round(tot_bytes / tot_n) if tot_n else None), src


def roofline_probe(device, B, NV, NT, D, DFF, L, reps=20):
    """Time every MFMA GEMM kernel family over exactly the launch list of one step (hipGraph replay of the
    list, HIP events on the replay stream) and report the one that takes the most time per step."""
    from openvivqa_amd import ops
    fam = gemm_launch_list(B, NV, NT, D, DFF, L, in_step=True)
    results = {}
    for name, shapes in fam.items():
        bufs = {}
        for sh in set(shapes):
            M, N, K = sh[:3]
            bufs[sh] = dict(x=torch.randn(M, K, device=device).bfloat16(),
                            w=(torch.randn(N, K, device=device) * K ** -0.5).bfloat16(),
                            wt=(torch.randn(K, N, device=device) * K ** -0.5).bfloat16(),
                            b=torch.randn(N, device=device), r=torch.randn(M, N, device=device).bfloat16(),
                            y=torch.empty(M, N, device=device, dtype=torch.bfloat16),
                            u=torch.empty(M, N, device=device, dtype=torch.bfloat16),
                            dx=torch.empty(M, K, device=device, dtype=torch.bfloat16),
                            pre=torch.randn(M, K, device=device).bfloat16())

        def run_all():
            for sh in shapes:
                t = bufs[sh]
                if name == "dx":
                    ops.linear_bwd_data_wt(t["r"], t["wt"], preact=t["pre"] if sh[3] == "g" else None, out=t["dx"],
                                           addend=t["x"] if sh[3] == "a" else None)
                else:
                    epi = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "residual": ops.EPI_BIAS_RESIDUAL}[name]
                    ops.linear_fwd(t["x"], t["w"], t["b"], epi, residual=t["r"] if name == "residual" else None,
                                   out=t["y"], preact_out=t["u"] if name == "gelu" else None)
        # replay the launch list from a hipGraph so that host launch latency is not what is timed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_all()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from openvivqa_amd import runtime as rt
        with torch.cuda.graph(g, capture_error_mode=rt.capture_error_mode()):
            run_all()
        g.replay()
        torch.cuda.synchronize()
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
        total_s = e0.elapsed_time(e1) * 1e-3 / reps
        flops = sum(2.0 * sh[0] * sh[1] * sh[2] for sh in shapes)
        results[name] = dict(launches=len(shapes), time_s=total_s, flops=flops,
                             avg_launch_us=total_s / len(shapes) * 1e6)
    dom = max(results, key=lambda k: results[k]["time_s"])
    r = results[dom]
    achieved = r["flops"] / r["time_s"] / 1e12
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
        "frac": round(achieved * 1e12 / PEAK_BF16, 4), "traffic": pmc_traffic(KERNEL_OF_FAMILY[dom])[0],
        "kernel": KERNEL_OF_FAMILY[dom] + " {2|3|4}, 8, {128|64|32}>", "launches_per_step": r["launches"],
        "avg_launch_us": round(r["avg_launch_us"], 2),
        "algorithmic_flops_per_launch": round(r["flops"] / r["launches"]),
        "families": {k: {"avg_launch_us": round(v["avg_launch_us"], 2),
                         "tflops": round(v["flops"] / v["time_s"] / 1e12, 1)} for k, v in results.items()},
    }


def instep_probe(ts, dom_hint=None):
    """The GEMM kernel families INSIDE a real step: one eager forward+loss+backward of the captured workload with the
    library's launch timing armed (ovqa_launch_timing_begin/_end): every launch of the bf16 GEMM kernels carries its
    own start / stop HIP events (hipExtLaunchKernel), which take the dispatch packet's begin / end timestamps on the
    stream the kernel runs on -- the kernel's execution time as rocprofv3 --kernel-trace reports it, with no
    event-record packets in between to calibrate away.  A long gate kernel goes first, so that all ~240 launches of
    the step are queued before the GPU starts the first one (back to back, as in the graph replay).  Operands are
    whatever the previous kernel of the step left behind (cold L2), unlike roofline_probe's replay of a family on
    reused buffers.  profiles/ holds the rocprofv3 --kernel-trace --stats summary of the same step: its per-family
    average must agree with the figure returned here."""
    import ctypes as C
    from openvivqa_amd import _lib, ops
    lib = _lib.load()
    recs = []

    def tagged(fam_of, fn):
        def wrapped(*a, **kw):
            n0 = lib.ovqa_launch_timing_count()
            out = fn(*a, **kw)
            n1 = lib.ovqa_launch_timing_count()
            if n1 == n0 + 1:  # exactly one timed GEMM launch (a VALU fallback records none)
                recs.append((n0,) + fam_of(*a, **kw))
            return out
        return wrapped

    def rows(x):
        return x.numel() // x.shape[-1]

    def fam_fwd(x, w, bias=None, epilogue=ops.EPI_BIAS, **kw):
        name = {ops.EPI_BIAS: "bias", ops.EPI_BIAS_GELU: "gelu", ops.EPI_BIAS_RESIDUAL: "residual"}[epilogue]
        return name, 2.0 * rows(x) * w.shape[0] * w.shape[1]

    def fam_res32(x, w, *a, **kw):
        return "residual", 2.0 * rows(x) * w.shape[0] * w.shape[1]

    def fam_dx(dy, wt, **kw):
        return "dx", 2.0 * rows(dy) * wt.shape[0] * wt.shape[1]

    # the attention kernels of the step (the library times their launches the same way): per launch the ALGORITHMIC
    # bytes -- every operand once, bf16 unless said -- and flops, so that each kernel can be put on its own roofline
    def shp(q, k):
        B, nq, F = q.shape
        return B, nq, k.shape[1], F

    def fam_qkv(x, w, bias, mask, H, **kw):  # x, W (3F x D), qkv out, o + o_lo out, lse (fp32)
        B, n, Dm = x.shape
        F = w.shape[0] // 3
        nbytes = 2 * (B * n * Dm + w.numel() + B * n * 3 * F + 2 * B * n * F) + 4 * B * H * n
        return f"attn_qkv_fwd {n}x{n}", 2.0 * B * n * Dm * 3 * F + 4.0 * B * n * n * F, nbytes

    def fam_q(x, w, bias, k, v, mask, H, **kw):  # x, W_q, q out, k, v in, o + o_lo out, lse
        (B, nq, _), F, nk = x.shape, w.shape[0], k.shape[1]
        nbytes = 2 * (x.numel() + w.numel() + B * nq * F + 2 * B * nk * F + 2 * B * nq * F) + 4 * B * H * nq
        return f"attn_q_fwd {nq}x{nk}", 2.0 * B * nq * x.shape[-1] * F + 4.0 * B * nq * nk * F, nbytes

    def fam_bwd(d_o, q, k, v, o, lse, mask, H, **kw):  # dO, q, o, o_lo in; k, v in; dq, dk, dv out
        B, nq, nk, F = shp(q, k)
        nbytes = 2 * (4 * B * nq * F + 2 * B * nk * F + B * nq * F + 2 * B * nk * F) + 4 * B * H * nq
        return f"attn_bwd {nq}x{nk}", 10.0 * B * nq * nk * F, nbytes

    def fam_bwd_do(dy, wt, q, k, v, o, lse, mask, H, **kw):  # dY, W_o^T in instead of dO
        B, nq, nk, F = shp(q, k)
        nbytes = 2 * (dy.numel() + wt.numel() + 3 * B * nq * F + 2 * B * nk * F + B * nq * F + 2 * B * nk * F) + 4 * B * H * nq
        return f"attn_bwd_do {nq}x{nk}", 2.0 * B * nq * dy.shape[-1] * F + 10.0 * B * nq * nk * F, nbytes

    def fam_att(q, k, v, mask, H, **kw):
        B, nq, nk, F = shp(q, k)
        return f"attn_fwd {nq}x{nk}", 4.0 * B * nq * nk * F, 2 * (2 * B * nq * F + 2 * B * nk * F) + 4 * B * H * nq
    names = ("linear_fwd", "linear_fwd_res32", "linear_bwd_data_wt", "attention_qkv_fwd", "attention_q_fwd",
             "attention_bwd", "attention_bwd_do", "attention_fwd")
    saved = tuple(getattr(ops, n) for n in names)
    for n, fam in zip(names, (fam_fwd, fam_res32, fam_dx, fam_qkv, fam_q, fam_bwd, fam_bwd_do, fam_att)):
        setattr(ops, n, tagged(fam, getattr(ops, n)))
    cap = 4096
    us = (C.c_float * cap)()
    torch.cuda.synchronize()
    _lib.check(lib.ovqa_launch_timing_begin(cap), "launch_timing_begin")
    try:
        torch.cuda._sleep(200_000_000)  # gate (~0.1 s): the host queues the whole step behind it
        ts._fwd_bwd()
    finally:
        for nm, fn in zip(names, saved):
            setattr(ops, nm, fn)
        n = lib.ovqa_launch_timing_end(us, cap)
    if n < 0:
        _lib.check(n, "launch_timing_end")
    torch.cuda.synchronize()
    fams = {}
    for rec in recs:
        idx, fam, flops = rec[:3]
        f = fams.setdefault(fam, {"launches": 0, "time_s": 0.0, "flops": 0.0, "bytes": 0.0})
        f["launches"] += 1
        f["time_s"] += us[idx] * 1e-6
        f["flops"] += flops
        f["bytes"] += rec[3] if len(rec) > 3 else 0.0
    return fams


# family of the in-step probe -> the kernel instantiation the library picks for it (rocprofv3's name, as a prefix)
ATTN_KERNEL_OF = {"attn_qkv_fwd 100x100": "attn_qkv_fwd_mfma_kernel<128, 1,", "attn_qkv_fwd 20x20": "attn_qkv_fwd_mfma_kernel<32, 2,",
                  "attn_qkv_fwd": "attn_qkv_fwd_mfma_kernel", "attn_q_fwd": "attn_q_fwd_mfma_kernel",
                  "attn_bwd 100x100": "attn_bwd_roles_mfma_kernel", "attn_bwd_do 100x100": "attn_bwd_roles_mfma_kernel",
                  "attn_bwd_do 100x20": "attn_bwd_do_smallk_mfma_kernel",
                  "attn_bwd_do 20x20": "attn_bwd_do_smallk1_mfma_kernel", "attn_bwd": "attn_bwd_smallk_mfma_kernel",
                  "attn_fwd": "attn_fwd_mfma_kernel"}


def pmc_mfma_busy(kernel):
    """`mfma_busy_frac` of a kernel from the newest committed PMC summary (profiles/rNN_pmc_summary.json), or None."""
    for f in PMC_FILES:
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            vals = [v.get("mfma_busy_frac") for k, v in json.load(open(path)).items() if kernel in k and v.get("launches_sampled")]
            vals = [v for v in vals if v is not None]
            return (round(max(vals), 4), "profiles/" + f) if vals else (None, "profiles/" + f)
    return None, None


def attention_rooflines(fams):
    """The attention kernels that are IN the timed step (the old figure probed a kernel the step no
    longer launches), from the same in-step launch timing as the GEMM families: launches per step, average duration, the
    algorithmic bytes and flops of a launch, and both rooflines -- HBM (the bound SURVEY 8d names for the attention core)
    and MFMA (the fused-projection forms are GEMM-shaped) -- plus the matrix-pipe busy fraction out of the committed PMC
    pass of the same step."""
    out = {}
    for fam, f in sorted(fams.items()):
        if not fam.startswith("attn_"):
            continue
        kern = next((v for k, v in ATTN_KERNEL_OF.items() if fam.startswith(k)), None)
        t = f["time_s"] / f["launches"]
        busy, src = pmc_mfma_busy(kern) if kern else (None, None)
        out[fam] = {"kernel": kern, "launches_per_step": f["launches"], "avg_launch_us": round(t * 1e6, 2),
                    "algorithmic_bytes_per_launch": round(f["bytes"] / f["launches"]),
                    "achieved_GBps": round(f["bytes"] / f["time_s"] / 1e9, 1),
                    "frac_of_hbm_peak": round(f["bytes"] / f["time_s"] / 8e12, 4),
                    "algorithmic_flops_per_launch": round(f["flops"] / f["launches"]),
                    "tflops": round(f["flops"] / f["time_s"] / 1e12, 1),
                    "frac_of_mfma_peak": round(f["flops"] / f["time_s"] / PEAK_BF16, 4),
                    "mfma_busy_frac": busy, "mfma_busy_source": src}
    return {"peak_hbm_GBps": 8000.0, "peak_mfma_TFLOPs": PEAK_BF16 / 1e12, "method": "in-step launch timing, as `roofline`",
            "kernels": out}


def cpu_model():
    model, phys = "unknown", None
    try:
        cores = set()
        phys_id = core_id = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys_id = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core_id = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys_id is not None and core_id is not None:
                    cores.add((phys_id, core_id))
                phys_id = core_id = None
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys, os.cpu_count()


def cpu_baseline(cfg, steps, thread_list):
    """The oracle (plain-PyTorch CPU restatement, fp32, train mode) on the same workload, on the host cores of this
    box: a thread-count sweep on a quarter batch (one timed step each) picks the fastest setting, then `steps` full
    B=64 steps are timed there (median).  BASELINE.md section 4.1 has the oracle next to the real reference at equal
    threads (the reference itself cannot travel to the GPU box)."""
    import statistics
    import oracle as O
    from openvivqa_amd.mcan_stack import synthetic_batch
    b = cfg.BENCH
    torch.manual_seed(0)
    te = O.build_oracle_encoder(cfg.MODEL.SELF_ENCODER).train()
    ve = O.build_oracle_encoder(cfg.MODEL.GUIDED_ENCODER).train()
    params = list(te.parameters()) + list(ve.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98))
    v, vm, t, tm = synthetic_batch(b.BATCH_PER_GPU, b.REGIONS, b.TOKENS, cfg.MODEL.D_MODEL, b.MIN_REGIONS,
                                   b.MIN_TOKENS, b.SEED, "cpu", torch.float32)
    gt = torch.Generator().manual_seed(1)
    tv, tt = torch.randn(v.shape, generator=gt), torch.randn(t.shape, generator=gt)

    def one(n):
        lo = te(t[:n], tm[:n])
        vo = ve(v[:n], vm[:n], lo, tm[:n])
        loss = (vo - tv[:n]).pow(2).mean() + (lo - tt[:n]).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()

    model, phys, logical = cpu_model()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (logical or 1)
    cands = sorted({min(int(x), avail) for x in thread_list.split(",") if x.strip()})
    sweep, q = {}, max(1, b.BATCH_PER_GPU // 4)
    for n in cands:
        torch.set_num_threads(n)
        one(q)
        t0 = time.perf_counter()
        one(q)
        sweep[n] = round(q / (time.perf_counter() - t0), 2)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    one(b.BATCH_PER_GPU)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        one(b.BATCH_PER_GPU)
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    return {"value": round(b.BATCH_PER_GPU / dt, 2), "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"median of {steps} timed steps (+1 warm-up) of the full B={b.BATCH_PER_GPU} L=6 fwd+bwd+Adam step, "
                      f"fp32, train mode, {dt:.2f} s/step (min {min(ts):.2f}, max {max(ts):.2f}) at {best} threads, the "
                      f"fastest of a sweep on B={q}",
            "thread_sweep_samples_per_s": sweep, "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
            "cpus_available": avail}


def m4c_decode_bench(args, device, world, rank, dist, B, seed):
    """SECONDARY diagnostic (not the headline metric): BASELINE configs[3], `configs/mmf_m4c.yaml:92-95` -- samples/s of
    M4C's evaluation path, max_iter = 12 passes of the multimodal transformer per batch (mmf_m4c.py:236-256; the early
    exit is disabled by an unreachable eos index so that every batch costs the same), random weights, synthetic
    embeddings of the reference's shapes.  One "step" = one full greedy decode of a batch."""
    from types import SimpleNamespace
    from openvivqa_amd.modules.mmt import MMT, M4CDecodingHead
    cfg = SimpleNamespace(hidden_size=768, num_hidden_layers=4, num_attention_heads=8, intermediate_size=3072,
                          layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(seed)
    mmt, head = MMT(cfg).to(device).eval(), M4CDecodingHead(768, 5000).to(device).eval()
    g = torch.Generator().manual_seed(seed + rank)
    txt, obj, ocr = (torch.randn(B, n, 768, generator=g).to(device) for n in (20, 100, 50))
    z = lambda n: torch.zeros(B, 1, 1, n, device=device)
    from openvivqa_amd.modules.mmt import GraphedGreedyDecode
    masks = (z(20), z(100), z(50))
    graphed = GraphedGreedyDecode(head, mmt, 12, 1, -1)
    use_graph = os.environ.get("OVQA_M4C_GRAPH", "1") != "0"
    run = lambda: graphed(txt, masks[0], obj, masks[1], ocr, masks[2], use_graph=use_graph)
    for _ in range(max(1, args.warmup)):
        run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, _, passes = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    res = None
    if rank == 0:
        res = ({
            "metric": "SECONDARY: M4C greedy decode samples/sec (12 MMT passes per sample), hidden 768, S=182",
            "value": round(world * B * args.steps / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "secondary, BASELINE configs[3]: MMT 4 layers x 8 heads of 96, intermediate 3072, "
                       f"20 txt + 100 obj + 50 ocr + 12 dec positions, classifier(5000) || OcrPtrNet(768), B={B}/GPU, "
                       f"{passes} passes per decode, " + ("every pass replayed from one hipGraph, the early-exit check "
                       "on the host as upstream" if use_graph else "eager launches"), "global_batch": world * B,
                       "parallelism": f"dp{world}"},
            "ms_per_mmt_pass": round(dt / args.steps / passes * 1e3, 3)})
        # algorithmic work of one MMT pass (forward only): per position and layer the four 768 x 768 projections and the
        # two 768 x 3072 FFN products (2 flops per multiply-add) + scores and weighted sum over S = 182 positions; the
        # classifier / pointer heads on the 12 decoding positions are < 1 % and left out
        S, Hd, Ff, Lm = 20 + 100 + 50 + 12, 768, 3072, 4
        gf_pass = B * S * Lm * (2 * (4 * Hd * Hd + 2 * Hd * Ff) + 4 * S * Hd) / 1e9
        res["algorithmic_gflop_per_mmt_pass"] = round(gf_pass, 1)
        res["step_frac_of_bf16_peak"] = round(gf_pass * passes * args.steps * world / dt / 1e3 / (world * 2500.0), 4)
    return res


def decode_bench(args, device, world, rank, dist, B, seed):
    """SECONDARY diagnostic (not the headline metric): BASELINE configs[4], `configs/vit_mbert_generation.yaml:68-98` --
    the stateful `Decoder` (L=3, d=512, H=8, dff=2048) decoding T=20 tokens per sample over 237 encoder positions with
    batched beam search (openvivqa_amd.beam = beam_search.py's control flow), evaluation mode, random weights, synthetic
    encoder features (the ViT / mBERT encoders in front are out of scope).  One "step" = one full decode of a batch.
    `roofline_decode`: bytes ONE decoding step must stream from HBM -- decoder weights once, the self-attention K/V
    prefix of every live beam, the projected encoder K/V of every sample (shared by its beams) -- against the measured
    time per step."""
    from types import SimpleNamespace
    import openvivqa_amd as A
    from openvivqa_amd.beam import BeamSearch
    from openvivqa_amd.config import ConfigNode, attention_config
    T, NE, V, L, D = 20, 237, 4000, 3, 512

    class Vocab:
        max_answer_length, padding_idx, bos_idx, eos_idx = T, 0, 1, 2

        def __len__(self):
            return V
    cfg = ConfigNode(dict(
        ARCHITECTURE="Decoder", D_MODEL=D, LAYERS=L,
        ATTENTION=dict(SELF_ATTENTION=attention_config(can_be_stateful=True), ENC_ATTENTION=attention_config()),
        TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=D, D_EMBEDDING=300, WORD_EMBEDDING=None,
                            WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
    torch.manual_seed(seed)
    dec = A.build_decoder(cfg, Vocab()).to(device).eval()
    g = torch.Generator().manual_seed(seed + rank)
    enc = torch.randn(B, NE, D, generator=g)
    ne = torch.randint(200, NE + 1, (B,), generator=g)
    enc[torch.arange(NE)[None, :] >= ne[:, None]] = 0
    enc = enc.to(device)
    from openvivqa_amd.utils import generate_padding_mask
    emask = generate_padding_mask(enc, 0)
    beam = args.beam
    from openvivqa_amd.beam import GraphedBeamSearch
    search = GraphedBeamSearch(dec, B, T, 1, -1, beam)  # eos unreachable: every batch costs the same

    def run():
        return search(enc, emask, use_graph=not args.no_graph)
    for _ in range(max(1, args.warmup)):
        run()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    res = None
    if rank == 0:
        per_step = dt / args.steps / T
        es = 2 if args.dtype == "bf16" else 4
        n_w = sum(p.numel() for n, p in dec.named_parameters() if "word_emb" not in n and "pos_emb" not in n)
        # per decoding step: every weight once; self K/V prefix (mean length T/2) per live beam; encoder K/V per sample
        bytes_step = n_w * es + L * (B * beam * (T / 2) * 2 * D * es + B * NE * 2 * D * es)
        res = ({
            "metric": "SECONDARY: autoregressive decode tokens/sec (best beam), Decoder L=3 d=512, 237 encoder positions, T=20",
            "value": round(world * B * T * args.steps / dt, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"secondary, BASELINE configs[4]: Decoder L={L} d={D} H=8 dff=2048, |V|={V}, B={B}/GPU, "
                                   f"{NE} encoder positions, T={T} decoding steps, beam {beam}, "
                                   + ("eager launches" if args.no_graph else "whole decode replayed from one hipGraph"),
                       "global_batch": world * B, "parallelism": f"dp{world}", "beam": beam},
            "us_per_decoding_step": round(per_step * 1e6, 1),
            "roofline_decode": {"bound": "hbm", "bytes_per_decoding_step": int(bytes_step), "peak": 8000.0, "unit": "GB/s",
                                "achieved": round(bytes_step / per_step / 1e9, 1),
                                "frac": round(bytes_step / per_step / PEAK_HBM, 4)}})
    return res


def ensure_library(local_rank):
    """The C-ABI library is a build artefact.  Decide about it BEFORE anything touches the GPU or the process group
    (hipcc must never run in a process that initialised HIP, and never under a profiler): local rank 0 compiles a
    missing / stale library, the other ranks of the node wait for the file; OVQA_NO_BUILD=1 (set by the profiling
    scripts, which build first) turns a missing library into an error."""
    from openvivqa_amd import build as _build
    if not _build.needs_build():
        return
    if os.environ.get("OVQA_NO_BUILD", "0") == "1":
        raise SystemExit(f"bench.py: {_build.LIB} is missing or older than its sources; "
                         "run `python -m openvivqa_amd.build` first (OVQA_NO_BUILD=1 forbids building here)")
    if local_rank == 0:
        _build.build(verbose=False)
        return
    deadline = time.time() + 900
    while _build.needs_build():
        if time.time() > deadline:
            raise SystemExit("bench.py: timed out waiting for local rank 0 to build libovqa_hip.so")
        time.sleep(1.0)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only the launcher.
    It builds the library, checks that the node has N devices (`torch.cuda.device_count()` does not initialise HIP) and
    starts N FRESH rank processes through torch.distributed.run -- it never touches the GPU itself and never execs --
    then relays their output (rank 0's single JSON line on stdout) and exits with the children's code."""
    import subprocess
    ensure_library(0)
    rehearse = os.environ.get("OVQA_REHEARSE_BACKEND", "nccl") != "nccl"
    ndev = torch.cuda.device_count()
    if not args.dry_launch and not rehearse and ndev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {ndev} GPU(s); refusing to run fewer ranks "
                         "than asked for (one process per GPU; OVQA_REHEARSE_BACKEND=gloo shares one GPU for rehearsals)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OVQA_NO_BUILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def dry_launch(args, world, rank):
    """--dry-launch: what a rank does before any GPU work, on gloo -- proves that `--gpus N` became N ranks."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        counted, pg_world = int(t.item()), dist.get_world_size()
    else:
        counted = pg_world = 1
    if rank == 0:
        print(json.dumps({"metric": "VQA samples/sec fwd+bwd, MCAN d=512 L=6, B=64, 100 regions x 20 tokens",
                          "value": None, "unit": "samples/s", "n_gpus": pg_world, "steps": args.steps,
                          "warmup": args.warmup, "dry_launch": True, "world_size": pg_world, "ranks_counted": counted,
                          "device_count": torch.cuda.device_count(),
                          "config": {"parallelism": f"dp{pg_world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different "
                         "number of ranks than the command line asks for")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    ensure_library(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback for the product path)")
    # rehearsal on a one-GPU box: OVQA_REHEARSE_BACKEND=gloo runs all ranks on GPU 0 with gloo transporting the
    # CUDA tensors (RCCL refuses two ranks on one device); the product path (backend nccl = RCCL) is the default
    backend = os.environ.get("OVQA_REHEARSE_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # RCCL prints a version banner on stdout when a communicator is created: keep stdout for the ONE JSON line
    # by pointing fd 1 at stderr while the process group (and its first collective) comes up
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=device)
            else:
                dist.init_process_group(backend=backend)
        elif args.rehearse_comm:
            import tempfile
            import torch.distributed as dist
            dist.init_process_group(backend="nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(), "rdv"),
                                    rank=0, world_size=1, device_id=device)
        else:
            dist = None
        if dist is not None:
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    import openvivqa_amd as A

    def teardown():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload in ("m4c_decode", "decode"):
        b = A.get_config(args.config).BENCH
        A.set_compute_dtype(dtype)
        A.manual_seed(b.SEED + rank)
        torch.manual_seed(b.SEED)
        fn = m4c_decode_bench if args.workload == "m4c_decode" else decode_bench
        res = fn(args, device, world, rank, dist, int(b.BATCH_PER_GPU), int(b.SEED))
        if rank == 0:
            print(json.dumps(res), flush=True)
        return teardown()

    out, ts, cfg = train_bench(args, args.workload, device, world, rank, dist, dtype)
    if rank == 0:
        b = cfg.BENCH
        if args.workload == "stack":
            if not args.no_roofline and args.dtype == "bf16":
                add_rooflines(out, ts, cfg, device)
            if world == 1 and not args.no_secondary and not args.rehearse_comm:
                out["secondary"] = secondary_lines(args, device, dtype)
                out["ms_per_step_unfused"] = out["secondary"].get("stack_unfused_adam", {}).get("ms_per_step")
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps, args.cpu_threads)
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    teardown()


# algorithmic matmul FLOPs per sample, forward + backward (= 3 x forward), of the whole models (the encoder figures are
# SURVEY 8d's; the rest is 2 m n k of every nn.Linear in front of / behind the stacks at the bench shapes)
MODEL_GFLOP_PER_SAMPLE = {
    # stacks 16.31 + FeatureEmbedding 1024->512 (100 rows) 0.315 + word projection 300->512 (20 rows) 0.018 + LSTM
    # 2 x [2048 x 512] (20 rows) 0.252 + pooling MLPs fc1 (120 rows) 0.189 + projections / classifier 0.004
    "model": 17.09,
    # CrossModalityEncoder L=6 live work 13.97 (dead cross-attention skipped) + FeatureEmbedding 2048->512
    # 0.629 + pooling MLPs 0.189 + projections / classifier 0.004
    "cross_modality": 14.79,
    # Decoder L=3, T=20 answer positions, 237 encoder positions, |V|=4000: BASELINE.md section 3 (1.218 forward) -- the
    # K / V projections of the 237 encoder rows, self- and encoder-attention, FFN and the 512 -> 4000 vocabulary product
    "decoder_train": 3.65,
}


def train_bench(args, workload, device, world, rank, dist, dtype, steps=None, warmup=None, repeats=None):
    """Build the workload (stack = BASELINE's metric; model / cross_modality = whole models through build_model), capture
    its step, time `steps` steps as the contract says.  Returns (JSON dict on rank 0 else None, TrainStep, config)."""
    import openvivqa_amd as A
    from openvivqa_amd import ops
    from openvivqa_amd.mcan_stack import MCANEncoderStack, synthetic_batch
    from openvivqa_amd.train import TrainStep, noam_lr_scale
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    repeats = args.repeats if repeats is None else repeats
    config = args.config
    if workload == "cross_modality":
        config = os.path.join(ROOT, "configs", "cross_modality_bench.yaml")
    cfg = A.get_config(config)
    b = cfg.BENCH
    A.set_compute_dtype(dtype)
    A.manual_seed(b.SEED + rank)
    torch.manual_seed(b.SEED)  # identical initial weights on every rank
    whole_model = workload in ("model", "cross_modality", "decoder_train")
    D = cfg.MODEL.D_MODEL
    loss_buf = torch.zeros(1, device=device)
    if not whole_model:
        model = MCANEncoderStack(cfg.MODEL).to(device).train()
        v, vm, t, tm = synthetic_batch(b.BATCH_PER_GPU, b.REGIONS, b.TOKENS, D, b.MIN_REGIONS, b.MIN_TOKENS,
                                       b.SEED + rank, device, dtype)
        # MSE against fixed random targets: mean(out^2) alone is constant for LayerNorm outputs
        # (degenerate gradient), see DESIGN.md section 6.
        gt = torch.Generator().manual_seed(b.SEED + 7919 * (rank + 1))
        tgt_v = torch.randn(v.shape, generator=gt).to(device=device, dtype=dtype)
        tgt_t = torch.randn(t.shape, generator=gt).to(device=device, dtype=dtype)

        def forward_loss(v_, vm_, t_, tm_):
            vo, lo = model(v_, vm_, t_, tm_)
            dvo = ops.sq_loss_fwd_bwd(vo.detach(), loss_buf, accumulate=False, target=tgt_v)
            dlo = ops.sq_loss_fwd_bwd(lo.detach(), loss_buf, accumulate=True, target=tgt_t)
            return (vo, lo), (dvo, dlo)
        batch = (v, vm, t, tm)
    elif workload == "decoder_train":
        # BASELINE configs[4], the TRAINING half (tasks/open_ended_task.py:150-169): the Decoder of
        # configs/vit_mbert_generation.yaml:68-98 teacher-forced on the right-shifted answer, NLLLoss(ignore_index = pad)
        # over every answer position, Adam + Noam.  The 237 encoder positions (197 ViT patches + 40 question tokens) are
        # synthetic features: the ViT / mBERT encoders in front are out of scope.
        from openvivqa_amd.config import ConfigNode, attention_config
        from openvivqa_amd.losses import nll_loss_fwd_bwd
        from openvivqa_amd.utils import generate_padding_mask
        T_, NE_, V_, L_ = 20, 237, 4000, 3

        class DVocab:
            max_answer_length, padding_idx, bos_idx, eos_idx = T_, 0, 1, 2

            def __len__(self):
                return V_
        dcfg = ConfigNode(dict(
            ARCHITECTURE="Decoder", D_MODEL=D, LAYERS=L_,
            ATTENTION=dict(SELF_ATTENTION=attention_config(can_be_stateful=True), ENC_ATTENTION=attention_config()),
            TEXT_EMBEDDING=dict(ARCHITECTURE="UsualEmbedding", D_MODEL=D, D_EMBEDDING=300, WORD_EMBEDDING=None,
                                WORD_EMBEDDING_CACHE=None, DROPOUT=0.1)))
        model = A.build_decoder(dcfg, DVocab()).to(device).train()
        g = torch.Generator().manual_seed(b.SEED + rank)
        enc = torch.randn(b.BATCH_PER_GPU, NE_, D, generator=g)
        ne = torch.randint(200, NE_ + 1, (b.BATCH_PER_GPU,), generator=g)
        enc[torch.arange(NE_)[None, :] >= ne[:, None]] = 0
        ans = torch.randint(3, V_, (b.BATCH_PER_GPU, T_ + 1), generator=g)
        ans[:, 0] = 1
        na = torch.randint(6, T_ + 1, (b.BATCH_PER_GPU,), generator=g)
        ans[torch.arange(T_ + 1)[None, :] > na[:, None]] = 0  # <bos> a_1 .. a_n <pad> ...
        enc = enc.to(device=device, dtype=dtype)
        emask = generate_padding_mask(enc, 0)
        tgt = ans[:, 1:].contiguous().to(device)  # shifted_right_answer_tokens (open_ended_task.py:157)

        def forward_loss(tokens_, enc_, emask_):
            logp = model(tokens_, enc_, emask_)
            return [logp], [nll_loss_fwd_bwd(logp, tgt, loss_buf, ignore_index=0)]
        batch = (ans[:, :-1].contiguous().to(device), enc, emask)
    else:
        from types import SimpleNamespace
        from openvivqa_amd.builders import build_model
        from openvivqa_amd.losses import NLLLoss, nll_loss_fwd_bwd

        class Vocab:
            padding_idx, total_answers = 0, int(b.ANSWERS)

            def __len__(self):
                return int(b.VOCAB)
        model = build_model(cfg.MODEL, Vocab()).train()
        g = torch.Generator().manual_seed(b.SEED + rank)
        feats = torch.randn(b.BATCH_PER_GPU, b.REGIONS, int(b.D_FEATURE), generator=g)
        nreg = torch.randint(b.MIN_REGIONS, b.REGIONS + 1, (b.BATCH_PER_GPU,), generator=g)
        feats[torch.arange(b.REGIONS)[None, :] >= nreg[:, None]] = 0
        toks = torch.randint(4, int(b.VOCAB), (b.BATCH_PER_GPU, b.TOKENS), generator=g)
        ntok = torch.randint(b.MIN_TOKENS, b.TOKENS + 1, (b.BATCH_PER_GPU,), generator=g)
        toks[torch.arange(b.TOKENS)[None, :] >= ntok[:, None]] = 0
        ans = torch.randint(0, int(b.ANSWERS), (b.BATCH_PER_GPU,), generator=g).to(device)
        nll = NLLLoss()  # (drop-in for the reference's nn.NLLLoss: classification_task.py:125-127)

        def forward_loss(feats_, toks_):
            out_ = model(SimpleNamespace(region_features=feats_, question_tokens=toks_))
            if workload == "model":  # MCAN returns log-probabilities: loss and its gradient from ONE launch
                return [out_], [nll_loss_fwd_bwd(out_, ans, loss_buf)]
            return nll(out_, ans)  # (CrossModalityTransformer feeds raw logits to NLLLoss, as upstream)
        batch = (feats.to(device=device, dtype=dtype), toks.to(device))
    comm = torch.bfloat16 if args.comm_dtype == "bf16" else torch.float32
    ts = TrainStep(model, forward_loss, lr=float(b.LEARNING_RATE), betas=(0.9, 0.98),
                   lr_lambda=lambda s_: noam_lr_scale(s_, D, int(b.WARMUP)), use_graph=not args.no_graph,
                   comm_dtype=comm, compute_dtype=dtype, overlap_mb=args.overlap_mb,
                   force_comm=args.rehearse_comm, fuse_adam=not getattr(args, "no_fuse_adam", False) and os.environ.get("OVQA_FUSE_ADAM", "1") != "0",
                   shard_optimizer=False if getattr(args, "no_shard", False) else None,
                   rehearse_shard=getattr(args, "rehearse_shard", 0) if args.rehearse_comm else 0)
    if workload == "cross_modality":
        loss_buf = ts.loss  # (the scalar-loss protocol: TrainStep copies the loss into its own buffer)
    else:
        ts.loss = loss_buf

    ts.prepare(*batch)  # graph capture happens here, never inside the timed region (even with --warmup 0)
    # the synthetic batch is resident in HBM: hand the step the graph's own input buffers (what a data loader
    # would fill in place) instead of paying four device-to-device copies per step
    batch = tuple(ts.static_inputs)
    for _ in range(warmup):
        ts.step(*batch)

    def window():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(*batch)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        w = time.perf_counter() - t0
        if dist is not None:
            tw = torch.tensor([w], device=device, dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            w = tw.item()
        return w
    dt = window()
    # further windows of the same K steps (not part of `value`): spread of the measurement
    windows = [dt] + [window() for _ in range(max(0, repeats - 1))]
    final_loss = float(loss_buf.item())
    ts.check_device_status()  # (a persistent-kernel hand-off that gave up during the timed steps is an error, not a number)
    comm_stats = ts.timed_comm_step(*batch) if (dist is not None) else {}
    if rank != 0:
        return None, ts, cfg
    import statistics
    ms = dt / steps * 1e3
    value = world * b.BATCH_PER_GPU * steps / dt
    flops = FLOPS_PER_SAMPLE_FWD_BWD if not whole_model else MODEL_GFLOP_PER_SAMPLE[workload] * 1e9
    out = {
        "metric": "VQA samples/sec fwd+bwd, MCAN d=512 L=6, B=64, 100 regions x 20 tokens",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "world_size": dist.get_world_size() if (dist is not None and world > 1) else 1,
        "device_count": torch.cuda.device_count(),
        "config": {"workload": "configs[1]: MCAN encoder stack (Encoder + GuidedAttentionEncoder) d=512 L=6 H=8 "
                   "dff=2048, 64 samples/GPU x (100 regions + 20 tokens), padded lengths, dropout 0.1, "
                   "fwd+loss+bwd+grad all-reduce+Adam(0.9,0.98)+Noam LR",
                   "global_batch": world * b.BATCH_PER_GPU, "parallelism": f"dp{world}",
                   "hipgraph": not args.no_graph, "comm_dtype": args.comm_dtype if (world > 1 or args.rehearse_comm) else None,
                   "grad_segments": ts.n_exchanges, "backward_phases": len(ts.segments), "rehearse_comm": bool(args.rehearse_comm),
                   "adam": ("inside the last weight-gradient launch for the weight matrices (TrainStep.fuse_adam: world size 1), "
                            "one launch for the 1-D parameters" if getattr(ts, "_fused", None) is not None and ts._fused.began
                            else (f"sharded over {ts.reducer.shard_world} ranks: reduce-scatter of the weight matrices' gradients, "
                                  "Adam on the owned chunks, all-gather of the bf16 shadow, transposed shadow rebuilt locally"
                                  + (" (single-rank REHEARSAL of the slice sizes: the other chunks are not updated)"
                                     if ts.reducer.shard_world != ts.reducer.world else "")
                                  if getattr(ts, "shard", False)
                                  else "separate launch(es) behind backward / the gradient exchange"))},
        "final_loss": round(final_loss, 6),
        "repeats": len(windows),
        "ms_per_step_median": round(statistics.median(windows) / steps * 1e3, 3),
        "ms_per_step_min": round(min(windows) / steps * 1e3, 3),
        "ms_per_step_max": round(max(windows) / steps * 1e3, 3),
        "gradient_exchange": comm_stats or None,
        "step_tflops": round(value * flops / 1e12, 1),
        "step_frac_of_bf16_peak": round(value * flops / world / PEAK_BF16, 4),
    }
    if workload == "model":
        out["metric"] = "SECONDARY: VQA samples/sec fwd+bwd, whole MCAN model (embeddings + stacks + head), L=6, B=64"
        out["config"]["workload"] = ("secondary (SURVEY 8d model-level): MCAN via build_model from configs/mcan_bench.yaml (the "
                                     "reference YAML's MODEL node, L=6): FeatureEmbedding 1024->512, LSTMTextEmbedding |V|=4000 "
                                     "(embedding rows, 300->512 projection, persistent-kernel LSTM), encoder stacks, "
                                     "attention-pooling head, 353-way classifier + log_softmax, NLLLoss, Adam + Noam: every "
                                     "launch of the step is the library's")
    elif workload == "cross_modality":
        out["metric"] = ("SECONDARY (BASELINE configs[2]): VQA samples/sec fwd+bwd, CrossModalityTransformer "
                         "d=512 L=6, 64 samples/GPU")
        out["config"]["workload"] = ("secondary, BASELINE configs[2]: CrossModalityTransformer via build_model from "
                                     "configs/cross_modality_bench.yaml (the reference YAML's MODEL node, L=6): "
                                     "FeatureEmbedding 2048->512, UsualEmbedding |V|=4000, 6 CrossModalityEncoder "
                                     "layers (4 attention + 2 feed-forward blocks each), pooling head, 353-way "
                                     "classifier, NLLLoss on the logits as upstream, Adam + Noam; 100 regions x 20 "
                                     "tokens, data parallel")
    elif workload == "decoder_train":
        out["metric"] = ("SECONDARY (BASELINE configs[4], training half): samples/sec fwd+bwd, teacher-forced Decoder L=3 "
                         "d=512, T=20, 237 encoder positions, |V|=4000, 64 samples/GPU")
        out["config"]["workload"] = ("secondary, BASELINE configs[4] training: Decoder (configs/vit_mbert_generation.yaml:68-98: "
                                     "L=3, d=512, H=8, dff=2048, UsualEmbedding 300 -> 512, |V|=4000) teacher-forced over "
                                     "T=20 answer positions and 237 synthetic encoder positions, causal + padding masks, "
                                     "log_softmax + NLLLoss(ignore_index=pad) over every position, dropout 0.1, Adam + Noam "
                                     "(tasks/open_ended_task.py:150-169)")
        out["value_tokens_per_s"] = round(value * 20, 1)
    if whole_model:
        out["algorithmic_gflop_per_sample_fwd_bwd"] = MODEL_GFLOP_PER_SAMPLE[workload]
    return out, ts, cfg


def add_rooflines(out, ts, cfg, device):
    b = cfg.BENCH
    D = cfg.MODEL.D_MODEL
    sa = cfg.MODEL.SELF_ENCODER.SELF_ATTENTION
    warm = roofline_probe(device, b.BATCH_PER_GPU, b.REGIONS, b.TOKENS, D, sa.D_FF, cfg.MODEL.SELF_ENCODER.LAYERS)
    fams = instep_probe(ts)
    gemm_fams = {k: v for k, v in fams.items() if not k.startswith("attn_")}
    dom = max(gemm_fams, key=lambda k: gemm_fams[k]["time_s"])
    f = fams[dom]
    achieved = f["flops"] / f["time_s"] / 1e12
    traffic, tfile = pmc_traffic(KERNEL_OF_FAMILY[dom])
    traffic_src = (f"{tfile}: FETCH_SIZE / WRITE_SIZE PMC passes of this same step under rocprofv3, committed with "
                   "the round -- read from the file, not collected by this run") if tfile else None
    out["roofline"] = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
        "frac": round(achieved * 1e12 / PEAK_BF16, 4), "traffic": traffic, "traffic_source": traffic_src,
        "kernel": KERNEL_OF_FAMILY[dom] + " ...>", "launches_per_step": f["launches"],
        "avg_launch_us": round(f["time_s"] / f["launches"] * 1e6, 2),
        "algorithmic_flops_per_launch": round(f["flops"] / f["launches"]),
        "method": "in-step: every launch of the family inside one eager step of the real workload carries its own "
                  "start/stop HIP events (hipExtLaunchKernel via ovqa_launch_timing_begin/_end: the dispatch "
                  "packet's begin/end timestamps on the launch stream; gate kernel first; cold operands); "
                  "profiles/ holds the rocprofv3 --kernel-trace averages of the same step (replay window)",
        "families_in_step": {k: {"launches": v["launches"], "avg_launch_us": round(v["time_s"] / v["launches"] * 1e6, 2),
                                 "tflops": round(v["flops"] / v["time_s"] / 1e12, 1)} for k, v in gemm_fams.items()},
        "families_warm_replay": warm["families"],
    }
    # secondary: the attention kernels of the step, each on its rooflines
    out["roofline_attention"] = attention_rooflines(fams)


def secondary_lines(args, device, dtype):
    """The other BASELINE configurations on the same clock as the headline: after the headline window,
    in this process, each in try / except -- a failure becomes an "error" string, never a non-zero exit of the headline.
    Short windows (20 steps / 5 decodes) so that the whole default run stays within a couple of minutes."""
    import copy
    res = {}

    def compact(line, extra=()):
        keep = ("value", "unit", "ms_per_step", "steps", "warmup") + tuple(extra)
        d = {k: line[k] for k in keep if k in line}
        d["config"] = line["config"]["workload"]
        return d

    def guarded(name, fn):
        try:
            torch.cuda.synchronize()
            res[name] = fn()
        except Exception as exc:  # noqa: BLE001
            res[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finally:
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass

    # the headline workload with Adam as a launch of its own (what every rank of an N > 1 run executes: the fused form needs
    # world size 1), so that a scaling curve can be read against the same code path
    a_unf = copy.copy(args)
    a_unf.no_fuse_adam = True
    guarded("stack_unfused_adam", lambda: compact(train_bench(a_unf, "stack", device, 1, 0, None, dtype, steps=30, warmup=5, repeats=1)[0]))
    for wl in ("model", "cross_modality", "decoder_train"):
        guarded(wl, lambda wl=wl: compact(train_bench(args, wl, device, 1, 0, None, dtype, steps=20, warmup=3, repeats=1)[0],
                                          ("step_frac_of_bf16_peak", "algorithmic_gflop_per_sample_fwd_bwd")))
    a2 = copy.copy(args)
    a2.steps, a2.warmup = 5, 2
    import openvivqa_amd as A
    cfg = A.get_config(args.config)
    B, seed = int(cfg.BENCH.BATCH_PER_GPU), int(cfg.BENCH.SEED)
    for beam in (1, 3):
        a3 = copy.copy(a2)
        a3.beam = beam

        def run_decode(a3=a3):
            line = decode_bench(a3, device, 1, 0, None, B, seed)
            d = compact(line, ("us_per_decoding_step",))
            d["frac_of_hbm_peak"] = line["roofline_decode"]["frac"]
            return d
        guarded(f"decode_beam{beam}", run_decode)
    guarded("m4c_decode", lambda: compact(m4c_decode_bench(a2, device, 1, 0, None, B, seed),
                                                 ("ms_per_mmt_pass", "algorithmic_gflop_per_mmt_pass", "step_frac_of_bf16_peak")))
    return res


if __name__ == "__main__":
    main()
