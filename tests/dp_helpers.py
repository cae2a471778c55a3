"""Helpers for the data-parallel CPU tests: a small MCAN stack driven by the product's TrainStep, with the
kernel wrappers replaced by tests/mock_ops.py (same patching as test_plumbing_cpu.py's fixture, done by hand
because the workers are spawned processes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ATT = dict(ARCHITECTURE="ScaledDotProductAttention", HEAD=4, D_MODEL=32, D_KEY=8, D_VALUE=8, D_FF=64,
           USE_AOA=False, CAN_BE_STATEFUL=False, DROPOUT=0.0)
CFG = dict(ARCHITECTURE="MCAN", D_MODEL=32,
           SELF_ENCODER=dict(ARCHITECTURE="Encoder", D_MODEL=32, LAYERS=3, SELF_ATTENTION=ATT),
           GUIDED_ENCODER=dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=32, LAYERS=4, SELF_ATTENTION=ATT,
                               GUIDED_ATTENTION=ATT))


def patch_cpu_ops():
    import mock_ops
    import openvivqa_amd as A
    import openvivqa_amd.functional as Fn
    import openvivqa_amd.runtime as rt
    import openvivqa_amd.train as tr
    import openvivqa_amd.modules.embeddings as emb
    for mod in (Fn, rt, tr, emb):
        mod.ops = mock_ops

    def build_arena_cpu(module, device=None, compute_dtype=None):
        params = list(module.parameters())
        return rt.ParamArena(rt.collect_groups(module), params[0].device, compute_dtype or rt.get_compute_dtype())
    rt.build_arena = build_arena_cpu
    rt.step_tensor = lambda device: torch.zeros(1, dtype=torch.int32)
    A.set_compute_dtype(torch.float32)


CM_CFG = dict(D_MODEL=32, LAYERS=3, VISION_LANGUAGE_ATTENTION=ATT, LANGUAGE_VISION_ATTENTION=ATT,
              VISION_SELF_ATTENTION=ATT, LANGUAGE_SELF_ATTENTION=ATT)


def make_step(overlap_mb, comm_dtype=torch.float32, seed=7, kind="mcan", shard=None):
    import mock_ops
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.mcan_stack import MCANEncoderStack
    from openvivqa_amd.train import TrainStep
    torch.manual_seed(seed)
    if kind == "mcan":
        model = MCANEncoderStack(ConfigNode(CFG))
    else:  # BASELINE configs[2]: LXMERT-style pair encoder (dead cross-attention parameters included)
        import openvivqa_amd.modules as M
        model = M.CrossModalityEncoder(ConfigNode(CM_CFG))
    model.train()
    loss_buf = torch.zeros(1)

    def forward_loss(v, vm, t, tm, tgt_v, tgt_t):
        vo, lo = model(v, vm, t, tm)
        dvo = mock_ops.sq_loss_fwd_bwd(vo.detach(), loss_buf, accumulate=False, target=tgt_v)
        dlo = mock_ops.sq_loss_fwd_bwd(lo.detach(), loss_buf, accumulate=True, target=tgt_t)
        return (vo, lo), (dvo, dlo)
    ts = TrainStep(model, forward_loss, lr=1e-2, betas=(0.9, 0.98), use_graph=False, comm_dtype=comm_dtype,
                   compute_dtype=torch.float32, overlap_mb=overlap_mb, bucket_mb=0.01, shard_optimizer=shard)
    return model, ts


def batch(rank, B=3, NV=6, NT=4, D=32):
    from openvivqa_amd.mcan_stack import synthetic_batch
    v, vm, t, tm = synthetic_batch(B, NV, NT, D, NV - 2, NT - 2, 100 + rank, "cpu", torch.float32)
    g = torch.Generator().manual_seed(500 + rank)
    return v, vm, t, tm, torch.randn(B, NV, D, generator=g), torch.randn(B, NT, D, generator=g)


def dp_worker(rank, world, rdv, overlap_mb, comm_bf16, q, kind="mcan", shard=None):
    import torch.distributed as dist
    patch_cpu_ops()
    dist.init_process_group("gloo", init_method="file://" + rdv, rank=rank, world_size=world)
    try:
        model, ts = make_step(overlap_mb, torch.bfloat16 if comm_bf16 else torch.float32, kind=kind, shard=shard)
        for _ in range(2):
            ts.step(*batch(rank))
        assert ts.shard == (bool(shard) if shard is not None else world > 1), (ts.shard, shard, world)
        owned = ts.reducer.owned([(0, ts.arena.numel)]) if ts.shard else None
        sd = ts.state_dict()  # (a collective under the sharded optimiser: masters and moments come home from their owners)
        q.put((rank, ts.arena.master.clone().numpy(), [list(map(list, s)) for s in ts.segments],
               sd["optim"]["exp_avg"].numpy(), owned))
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- real kernels, several ranks on ONE GPU over gloo
def gpu_stack_step(B, comm_dtype, seed=11, layers=1, lr=1e-4):
    """A small MCAN stack (d = 512, one layer each, dropout off) under TrainStep with the REAL kernels on cuda:0."""
    import openvivqa_amd as A
    from openvivqa_amd import ops
    from openvivqa_amd.config import ConfigNode
    from openvivqa_amd.mcan_stack import MCANEncoderStack
    from openvivqa_amd.train import TrainStep
    dev = torch.device("cuda", 0)
    A.set_compute_dtype(torch.bfloat16)
    A.manual_seed(seed)
    torch.manual_seed(seed)
    att = dict(ARCHITECTURE="ScaledDotProductAttention", HEAD=8, D_MODEL=512, D_KEY=64, D_VALUE=64, D_FF=2048,
               USE_AOA=False, CAN_BE_STATEFUL=False, DROPOUT=0.0)
    cfg = ConfigNode(dict(ARCHITECTURE="MCAN", D_MODEL=512,
                          SELF_ENCODER=dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=layers, SELF_ATTENTION=att),
                          GUIDED_ENCODER=dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=layers,
                                              SELF_ATTENTION=att, GUIDED_ATTENTION=att)))
    model = MCANEncoderStack(cfg).to(dev).train()
    loss = torch.zeros(1, device=dev)
    tgt = {}

    def forward_loss(v_, vm_, t_, tm_):
        vo, lo = model(v_, vm_, t_, tm_)
        dvo = ops.sq_loss_fwd_bwd(vo.detach(), loss, accumulate=False, target=tgt["v"])
        dlo = ops.sq_loss_fwd_bwd(lo.detach(), loss, accumulate=True, target=tgt["t"])
        return (vo, lo), (dvo, dlo)
    ts = TrainStep(model, forward_loss, lr=lr, betas=(0.9, 0.98), compute_dtype=torch.bfloat16, use_graph=False,
                   comm_dtype=comm_dtype, overlap_mb=0.0)
    ts.loss = loss
    return model, ts, tgt


def gpu_rank_batch(rank, B):
    from openvivqa_amd.mcan_stack import synthetic_batch
    dev = torch.device("cuda", 0)
    v, vm, t, tm = synthetic_batch(B, 100, 20, 512, 80, 8, 100 + rank, dev, torch.bfloat16)
    g = torch.Generator().manual_seed(500 + rank)
    tv = torch.randn(B, 100, 512, generator=g).to(dev, torch.bfloat16)
    tt = torch.randn(B, 20, 512, generator=g).to(dev, torch.bfloat16)
    return (v, vm, t, tm), (tv, tt)


DP_GPU_BATCH = 8  # samples per rank: 160 question rows -- every product on the tiled GEMM kernels in the ranks AND in the
# single process that replays their shards (<= 128 rows take the one-wave form: same values, another summation order)


def dp_gpu_worker(rank, world, rdv, comm_bf16, steps, q):
    """One data-parallel rank with the real kernels on cuda:0, gradients exchanged over gloo (RCCL refuses several
    ranks on one device; the exchange arithmetic -- cast, sum, scale -- is the product's either way)."""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + rdv, rank=rank, world_size=world)
    try:
        model, ts, tgt = gpu_stack_step(DP_GPU_BATCH, torch.bfloat16 if comm_bf16 else torch.float32)
        batch, (tv, tt) = gpu_rank_batch(rank, DP_GPU_BATCH)
        tgt["v"], tgt["t"] = tv, tt
        for _ in range(steps):
            ts.step(*batch)
        ts.gather_state()  # (sharded optimiser: the fp32 masters of a chunk live on its owner -- a collective, every rank)
        torch.cuda.synchronize()
        if rank == 0:
            q.put(ts.arena.master.detach().cpu().numpy())
    finally:
        dist.destroy_process_group()
