"""GPU tests of the training harness (row T / SURVEY 8e): hipGraph-captured fwd+loss+bwd, fused Adam, and the
data-parallel code path rehearsed on ONE GPU through a single-rank RCCL group (phased backward captured as
several hipGraphs over one memory pool, gradient segments reduced on the communication stream)."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(layers, dropout):
    from openvivqa_amd.config import ConfigNode
    att = dict(ARCHITECTURE="ScaledDotProductAttention", HEAD=8, D_MODEL=512, D_KEY=64, D_VALUE=64, D_FF=2048,
               USE_AOA=False, CAN_BE_STATEFUL=False, DROPOUT=dropout)
    return ConfigNode(dict(ARCHITECTURE="MCAN", D_MODEL=512,
                           SELF_ENCODER=dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=layers, SELF_ATTENTION=att),
                           GUIDED_ENCODER=dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=layers,
                                               SELF_ATTENTION=att, GUIDED_ATTENTION=att)))


def _make(layers, **kw):
    import openvivqa_amd as A
    from openvivqa_amd import ops
    from openvivqa_amd.mcan_stack import MCANEncoderStack, synthetic_batch
    from openvivqa_amd.train import TrainStep
    dev = torch.device("cuda", 0)
    A.set_compute_dtype(torch.bfloat16)
    A.manual_seed(11)
    torch.manual_seed(11)
    model = MCANEncoderStack(_cfg(layers, 0.0)).to(dev).train()
    v, vm, t, tm = synthetic_batch(16, 100, 20, 512, 80, 8, 5, dev, torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    tv = torch.randn(v.shape, generator=g).to(dev, torch.bfloat16)
    tt = torch.randn(t.shape, generator=g).to(dev, torch.bfloat16)
    loss = torch.zeros(1, device=dev)

    def forward_loss(v_, vm_, t_, tm_):
        vo, lo = model(v_, vm_, t_, tm_)
        dvo = ops.sq_loss_fwd_bwd(vo.detach(), loss, accumulate=False, target=tv)
        dlo = ops.sq_loss_fwd_bwd(lo.detach(), loss, accumulate=True, target=tt)
        return (vo, lo), (dvo, dlo)
    ts = TrainStep(model, forward_loss, lr=1e-4, betas=(0.9, 0.98), compute_dtype=torch.bfloat16, **kw)
    ts.loss = loss
    return model, ts, (v, vm, t, tm)


@pytest.fixture
def single_rank_group():
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="ovqa_"), "rdv"),
                            rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("comm", [torch.float32, torch.bfloat16])
def test_phased_backward_with_comm_stream_matches_single_graph(single_rank_group, use_graph, comm):
    """The N>1 path (several backward phases, each its own hipGraph; segments cast / all-reduced / cast back on
    the communication stream while later phases run) yields the gradients and the weights of the plain
    single-graph step."""
    _, ref, batch = _make(3)
    for _ in range(3):
        ref.step(*batch)
    torch.cuda.synchronize()
    _, ts, batch = _make(3, force_comm=True, overlap_mb=16.0, comm_dtype=comm, use_graph=use_graph)
    for _ in range(3):
        ts.step(*batch)
    torch.cuda.synchronize()
    assert ref.graphs is not None and len(ref.graphs) == 1 and len(ref.segments) == 1
    assert len(ts.segments) >= 4, ts.segments
    assert (ts.graphs is not None and len(ts.graphs) == len(ts.segments)) if use_graph else ts.graphs is None
    flat = sorted(r for s in ts.segments for r in s)
    assert flat[0][0] == 0 and flat[-1][1] == ts.arena.numel and all(a[1] == b[0] for a, b in zip(flat, flat[1:]))
    assert abs(float(ts.loss) - float(ref.loss)) <= 2e-3 * abs(float(ref.loss))
    # gradients of the last step: identical kernels; only the order in which the question-feature gradient is
    # summed differs (bf16 rounding), plus bf16 transport when comm is bf16
    assert _rel(ts.arena.grad, ref.arena.grad) <= (2e-2 if comm == torch.bfloat16 else 1e-2)
    # weights moved by 3 Adam steps of lr 1e-4
    moved = (ref.arena.master - ts.arena.master).abs().max().item()
    assert moved <= 6.5e-4


def test_graph_replay_equals_eager_steps():
    _, a, batch = _make(2, use_graph=True)
    _, b, _ = _make(2, use_graph=False)
    a.step(*batch)
    b.step(*batch)
    torch.cuda.synchronize()
    # same kernels on the same weights: only the order of the fp32 atomics (bias / LayerNorm gradients) differs
    assert _rel(a.arena.grad, b.arena.grad) <= 1e-5
    for _ in range(2):  # afterwards Adam amplifies that noise on parameters with ~zero gradient (fc_k.bias)
        a.step(*batch)
        b.step(*batch)
    torch.cuda.synchronize()
    assert _rel(a.arena.grad, b.arena.grad) <= 5e-3
    assert abs(float(a.loss) - float(b.loss)) <= 1e-4 * abs(float(b.loss))
