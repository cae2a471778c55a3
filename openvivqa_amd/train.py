"""Data-parallel training step for the hot path (row T of SURVEY 8a).

Mirrors the per-batch op order of the reference's train loops
(tasks/classification_task.py:120-139, tasks/open_ended_task.py:150-169,
optimizer from tasks/base_task.py:46-48,73-76):

    forward -> (zero_grad) -> loss -> backward -> [all-reduce] -> Adam(0.9, 0.98) -> LR schedule

MI355X-first choices:
  * one process per GPU; gradients live in ONE flat fp32 arena buffer that the
    backward kernels write directly, so the data-parallel exchange is a single
    large RCCL all-reduce over xGMI (optionally in bf16: half the bytes per link)
    instead of hundreds of per-tensor collectives;
  * forward + loss + backward are captured once into a hipGraph and replayed
    (the step is ~400 short kernels: eager launch overhead would dominate);
  * Adam runs as one fused kernel over the arena and rewrites the bf16 shadow
    weights in the same pass; LR schedule and bias correction read device scalars,
    so the captured graph never goes stale;
  * parameters without a gradient (the dead cross-attention of
    CrossModalityEncoderLayer) simply keep zeros in the arena: the
    all-reduce is layout-identical on every rank by construction.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from . import runtime as rt


def noam_lr_scale(step: int, d_model: int, warmup: int) -> float:
    """tasks/base_task.py:73-76 (step is 0-based, as LambdaLR passes it)."""
    s = step + 1
    return (d_model ** -0.5) * min(s ** -0.5, s * warmup ** -1.5)


class FlatAdam:
    """torch.optim.Adam semantics (betas 0.9/0.98, eps 1e-8) on a ParamArena, one kernel launch."""

    def __init__(self, arena: rt.ParamArena, lr: float = 1.0, betas=(0.9, 0.98), eps: float = 1e-8,
                 weight_decay: float = 0.0, lr_lambda: Optional[Callable[[int], float]] = None):
        self.arena, self.lr, self.betas, self.eps, self.weight_decay = arena, lr, betas, eps, weight_decay
        dev = arena.device
        self.exp_avg = torch.zeros_like(arena.master)
        self.exp_avg_sq = torch.zeros_like(arena.master)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)  # number of optimiser steps taken
        self.lr_scale = torch.ones(1, dtype=torch.float32, device=dev)
        self.lr_lambda = lr_lambda
        self.host_step = 0
        if lr_lambda is not None:
            self.lr_scale.fill_(lr_lambda(0))

    def step(self, grad: Optional[torch.Tensor] = None, grad_scale: float = 1.0) -> None:
        a = self.arena
        ops.increment_step(self.step_t)
        ops.adam_step(a.master, a.grad if grad is None else grad, self.exp_avg, self.exp_avg_sq, a.shadow, self.lr,
                      self.step_t, lr_scale=self.lr_scale, betas=self.betas, eps=self.eps,
                      weight_decay=self.weight_decay, grad_scale=grad_scale)
        self.host_step += 1
        if self.lr_lambda is not None:  # scheduler.step(): value used by the NEXT optimiser step
            self.lr_scale.fill_(self.lr_lambda(self.host_step))


class GradAllReducer:
    """Sum-all-reduce of the flat gradient buffer over the default process group.

    ``comm_dtype=torch.bfloat16`` halves the bytes on every xGMI link (the arena is cast by one
    streaming kernel, reduced, and Adam consumes the fp32 view after a cast back); buckets keep
    individual collectives at ``bucket_mb`` so RCCL can pipeline them.  Works with the ``gloo``
    backend on CPU tensors too (used by the world_size-2 CPU tests)."""

    def __init__(self, numel: int, device, comm_dtype: torch.dtype = torch.float32, bucket_mb: float = 64.0,
                 group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        elt = 2 if comm_dtype == torch.bfloat16 else 4
        self.bucket = max(1, int(bucket_mb * (1 << 20) / elt))
        self.staging = torch.empty(numel, dtype=comm_dtype, device=device) if comm_dtype != torch.float32 else None

    def bounds(self, numel: int):
        return [(s, min(s + self.bucket, numel)) for s in range(0, numel, self.bucket)]

    def __call__(self, grad: torch.Tensor) -> torch.Tensor:
        """Returns the buffer holding the SUM over ranks (callers scale by 1/world)."""
        if self.world == 1:
            return grad
        buf = grad
        if self.staging is not None:
            if grad.is_cuda:
                ops.cast(grad, self.staging)
            else:
                self.staging.copy_(grad)
            buf = self.staging
        handles = [dist.all_reduce(buf[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for s, e in self.bounds(buf.numel())]
        for h in handles:
            h.wait()
        if self.staging is not None:
            if grad.is_cuda:
                ops.cast(self.staging, grad)
            else:
                grad.copy_(self.staging)
        return grad


class TrainStep:
    """forward -> loss -> backward -> all-reduce -> Adam, with the first three captured in a hipGraph.

    ``forward_loss(*static_inputs)`` must run the model and return ``(outputs, grads)`` where
    ``grads[i]`` is d loss / d outputs[i] (so that the loss kernel can emit its own gradient), or a
    scalar loss tensor (then autograd differentiates it).  It must be capture-safe: no host syncs.
    """

    def __init__(self, model: nn.Module, forward_loss: Callable, lr: float = 1.0, betas=(0.9, 0.98),
                 lr_lambda: Optional[Callable[[int], float]] = None, use_graph: bool = True,
                 comm_dtype: torch.dtype = torch.float32, bucket_mb: float = 64.0, device=None,
                 compute_dtype: Optional[torch.dtype] = None):
        self.model = model
        self.arena = rt.prepare(model, device=device, compute_dtype=compute_dtype)
        self.arena.overwrite_grads = True
        self.arena.attach_grads()
        self.optim = FlatAdam(self.arena, lr=lr, betas=betas, lr_lambda=lr_lambda)
        self.reducer = GradAllReducer(self.arena.numel, self.arena.device, comm_dtype, bucket_mb)
        self.forward_loss = forward_loss
        self.use_graph = use_graph
        self.graph = None
        self.static_inputs = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.arena.device)
        self.drop_step = rt.step_tensor(self.arena.device)
        self._foreign = [(0, self.arena.numel)]  # until the first backward tells which grads the kernels own

    # -- one fwd+bwd on the static inputs (this is what gets captured) ------
    def _fwd_bwd(self):
        a = self.arena
        if a.small_lo < a.numel:  # bias / LayerNorm gradients: atomically reduced, so zero them (one memset)
            a.grad[a.small_lo:].zero_()
        for s, e in self._foreign:  # grads that autograd accumulates into / that nobody writes
            if s < a.small_lo:
                a.grad[s:min(e, a.small_lo)].zero_()
        res = self.forward_loss(*self.static_inputs)
        if isinstance(res, tuple):
            outs, grads = res
            torch.autograd.backward(list(outs), list(grads))
        else:
            res.backward()
            self.loss.copy_(res.detach().float().reshape(1))

    def _discover_foreign(self):
        self._fwd_bwd()
        self._foreign = self.arena.foreign_ranges()

    def _capture(self, inputs: Sequence[torch.Tensor]):
        self.static_inputs = [t.clone() for t in inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._discover_foreign()
            for _ in range(2):  # warm-up: allocator pools, lazy arenas, workspace
                self._fwd_bwd()
            from . import functional as _fn
            _fn.wgrad_queue().reserve(32)  # table buffers for the grouped dW launches of the capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._fwd_bwd()

    def step(self, *inputs: torch.Tensor) -> torch.Tensor:
        if self.static_inputs is None:
            self._capture(inputs)
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._fwd_bwd()
        g = self.reducer(self.arena.grad)
        self.optim.step(g, grad_scale=1.0 / self.reducer.world)
        ops.increment_step(self.drop_step)
        return self.loss
