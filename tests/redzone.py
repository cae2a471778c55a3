"""Red zones around device buffers (SURVEY section 5 asked for a bounds-checking debug build; GPU AddressSanitizer is not
available on this pool, so the kernel tests bring their own guard bands).

Every tensor ``openvivqa_amd.ops`` allocates for a C-ABI call (outputs, saved blocks, workspaces, scratch) and every input
the kernel tests build with ``rnd()`` lives in the middle of a larger byte buffer with a 4-KiB band of 0xFF bytes directly
in front of it and directly behind its last byte.  After the test the bands must be intact: an out-of-bounds STORE of any
kernel shows as a changed band byte; an out-of-bounds LOAD reads 0xFF.. = NaN in bf16 and fp32, which the test's own parity
check then sees in the result.  Test infrastructure: nothing in the product imports this."""
from __future__ import annotations

import math

import torch

BAND = 4096
POISON = 0xFF


class RedZone:
    def __init__(self):
        self.live = []  # (raw uint8 buffer, payload bytes, description)

    def alloc(self, shape, dtype, device, what="alloc"):
        shape = tuple(int(s) for s in shape)
        n = math.prod(shape) * torch.empty(0, dtype=dtype).element_size()
        raw = torch.empty(2 * BAND + n, dtype=torch.uint8, device=device)
        raw[:BAND] = POISON
        raw[BAND + n:] = POISON
        self.live.append((raw, n, f"{what} {tuple(shape)} {dtype}"))
        return raw[BAND:BAND + n].view(dtype).view(shape)

    def guard(self, t: torch.Tensor, what="input") -> torch.Tensor:
        """A copy of ``t`` (contiguous CUDA tensors only; anything else is returned as it is) inside guard bands."""
        if not (torch.is_tensor(t) and t.is_cuda and t.is_contiguous() and t.numel() > 0):
            return t
        g = self.alloc(t.shape, t.dtype, t.device, what)
        g.copy_(t)
        return g.requires_grad_(t.requires_grad)

    def check(self):
        bad = []
        for raw, n, what in self.live:
            front, back = raw[:BAND], raw[BAND + n:]
            if not bool((front == POISON).all()):
                i = int((front != POISON).nonzero()[-1])
                bad.append(f"{what}: write {BAND - i} bytes IN FRONT of the buffer")
            if not bool((back == POISON).all()):
                i = int((back != POISON).nonzero()[0])
                bad.append(f"{what}: write {i} bytes BEHIND the end of the buffer")
        self.live.clear()
        assert not bad, "red zone damaged: " + "; ".join(bad)


class _TorchProxy:
    """Stands in for the ``torch`` module inside ``openvivqa_amd.ops``: the allocating calls return guarded tensors."""

    def __init__(self, rz: RedZone):
        self._rz = rz

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def _shape(size):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            return tuple(size[0])
        return tuple(size)

    def _cuda(self, device):
        return device is not None and torch.device(device).type == "cuda"

    def empty(self, *size, dtype=None, device=None, **kw):
        if not self._cuda(device) or kw:
            return torch.empty(*size, dtype=dtype, device=device, **kw)
        return self._rz.alloc(self._shape(size), dtype or torch.get_default_dtype(), device, "ops.empty")

    def zeros(self, *size, dtype=None, device=None, **kw):
        if not self._cuda(device) or kw:
            return torch.zeros(*size, dtype=dtype, device=device, **kw)
        return self._rz.alloc(self._shape(size), dtype or torch.get_default_dtype(), device, "ops.zeros").zero_()

    def empty_like(self, t, dtype=None, **kw):
        if not (t.is_cuda and t.is_contiguous()) or kw:
            return torch.empty_like(t, dtype=dtype, **kw)
        return self._rz.alloc(t.shape, dtype or t.dtype, t.device, "ops.empty_like")


def install(monkeypatch, test_module=None) -> RedZone:
    """Patch ``openvivqa_amd.ops`` (its allocations) and, when given, the test module's ``rnd`` (its inputs)."""
    from openvivqa_amd import ops
    rz = RedZone()
    monkeypatch.setattr(ops, "torch", _TorchProxy(rz))
    if test_module is not None and hasattr(test_module, "rnd"):
        plain = test_module.rnd

        def rnd(*a, **k):
            return rz.guard(plain(*a, **k), "rnd")
        monkeypatch.setattr(test_module, "rnd", rnd)
    return rz
