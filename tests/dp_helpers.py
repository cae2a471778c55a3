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


def make_step(overlap_mb, comm_dtype=torch.float32, seed=7, kind="mcan"):
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
                   compute_dtype=torch.float32, overlap_mb=overlap_mb, bucket_mb=0.01)
    return model, ts


def batch(rank, B=3, NV=6, NT=4, D=32):
    from openvivqa_amd.mcan_stack import synthetic_batch
    v, vm, t, tm = synthetic_batch(B, NV, NT, D, NV - 2, NT - 2, 100 + rank, "cpu", torch.float32)
    g = torch.Generator().manual_seed(500 + rank)
    return v, vm, t, tm, torch.randn(B, NV, D, generator=g), torch.randn(B, NT, D, generator=g)


def dp_worker(rank, world, rdv, overlap_mb, comm_bf16, q, kind="mcan"):
    import torch.distributed as dist
    patch_cpu_ops()
    dist.init_process_group("gloo", init_method="file://" + rdv, rank=rank, world_size=world)
    try:
        model, ts = make_step(overlap_mb, torch.bfloat16 if comm_bf16 else torch.float32, kind=kind)
        for _ in range(2):
            ts.step(*batch(rank))
        q.put((rank, ts.arena.master.clone().numpy(), [list(map(list, s)) for s in ts.segments]))
    finally:
        dist.destroy_process_group()
