"""Tensor-level wrappers over the C ABI (one Python function per entry point).

PyTorch is used here for device memory and the current HIP stream only; all
arithmetic happens in libovqa_hip.so.  Every function requires CUDA(HIP)
tensors and raises otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, OVQA_BF16, OVQA_F32

_DT = {torch.float32: OVQA_F32, torch.bfloat16: OVQA_BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"openvivqa_amd: unsupported dtype {t.dtype} (float32 or bfloat16)") from None


def _dev(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            "openvivqa_amd kernels run only on an AMD GPU (tensor is on '%s'); there is no CPU fallback" % t.device)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rows(t: torch.Tensor):
    """Return (row_stride, n_rows) of a tensor viewed as [rows, features]."""
    if t.stride(-1) != 1:
        raise RuntimeError("last dimension must be contiguous")
    if t.dim() == 2:
        return t.stride(0), t.shape[0]
    if t.dim() == 3:
        if t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
            raise RuntimeError("batch rows must be uniformly strided")
        return t.stride(1), t.shape[0] * t.shape[1]
    raise RuntimeError("expected a 2-D or 3-D tensor")


@dataclass
class DropSpec:
    """Dropout call-site description (see ovqa_dropout in include/ovqa_hip.h)."""
    p: float
    seed: int
    site: int
    step: Optional[torch.Tensor] = None  # uint32/int32 device scalar

    def c(self):
        if self.p <= 0.0:
            return None
        return C.byref(_lib.Dropout(float(self.p), self.seed & 0xFFFFFFFF, self.site & 0xFFFFFFFF, _p(self.step)))


def _drop(d: Optional[DropSpec]):
    return None if d is None else d.c()


_workspaces = {}


def workspace(device: torch.device) -> torch.Tensor:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ws = _workspaces.get(key)
    if ws is None:
        n = _lib.load().ovqa_workspace_bytes()
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def priority_range():
    """(least, greatest) stream priority of the device (numerically lower = served first)."""
    lo, hi = C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().ovqa_stream_priority_range(C.byref(lo), C.byref(hi)), "stream_priority_range")
    return lo.value, hi.value


def make_stream(device, priority: int = 0, cu_mask=None):
    """A torch stream over ``ovqa_stream_create``: ``cu_mask`` = iterable of the compute-unit indices the stream's
    kernels may run on (hipExtStreamCreateWithCUMask) or None; otherwise ``priority``.  The HIP stream lives as long as
    the process (a training step keeps its streams)."""
    dev = torch.device(device)
    words, n = None, 0
    if cu_mask is not None:
        cus = sorted(set(int(c) for c in cu_mask))
        n = cus[-1] // 32 + 1
        arr = (C.c_uint32 * n)()
        for c in cus:
            arr[c // 32] |= 1 << (c % 32)
        words = C.cast(arr, C.c_void_p)
    h = C.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().ovqa_stream_create(C.byref(h), int(priority), words, n), "stream_create")
    return torch.cuda.ExternalStream(h.value, device=dev)


# ---------------------------------------------------------------------------
def _weight(w, what):
    """A weight operand must be a dense [N, K] matrix.  The one way a caller gets anything else: a parameter whose reduction
    length is not a multiple of 8 lives zero-padded in the arena and ``arena.compute(p)`` / ``p.data`` are its [:N, :K] corner
    view -- ``functional.linear`` / ``linear_gelu_dropout`` / ``classify_log_softmax`` / ``embed_rows`` take the padded footprint
    (``arena.padded(p)``, activation zero-padded to match); the fused attention / feed-forward / LSTM blocks require
    d_model % 8 == 0 (every shipped config: 512, 768) and say so here instead of in an assertion."""
    if not w.is_contiguous():
        raise RuntimeError(f"{what}: the weight is a non-contiguous view {tuple(w.shape)} with strides {tuple(w.stride())} -- "
                           "a parameter whose reduction length is not a multiple of 8 is zero-padded in the arena; the fused "
                           "attention / feed-forward / LSTM blocks need d_model % 8 == 0 (use functional.linear, which takes "
                           "the padded footprint, or pad the model dimension)")
    return w


def linear_fwd(x, w, bias=None, epilogue=EPI_BIAS, residual=None, want_preact=False, drop=None, out=None,
               preact_out=None):
    """y = epilogue(x w^T + bias).  x [..., K] (rows may be strided), w [N, K]."""
    _dev(x)
    lib = _lib.load()
    ldx, M = _rows(x)
    N, K = w.shape
    _weight(w, "linear_fwd")
    assert x.shape[-1] == K and w.dtype == x.dtype
    y = out if out is not None else torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    ldy, _ = _rows(y)
    preact = preact_out if preact_out is not None else (
        torch.empty(M, N, dtype=x.dtype, device=x.device) if want_preact else None)
    ldres = 0
    if residual is not None:
        ldres, mr = _rows(residual)
        assert mr == M and residual.dtype == x.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    _lib.check(lib.ovqa_linear_fwd(_dt(x), epilogue, _p(x), ldx, _p(w), _p(bias), _p(residual), ldres, _p(y), ldy,
                                   _p(preact), M, N, K, _drop(drop), _stream()), "linear_fwd")
    return (y, preact) if want_preact else y


class LnRef:
    """A LayerNorm whose fp32 output is recomputed on the fly from its fp32 input ``pre`` (see ovqa_ln_ref)."""

    def __init__(self, pre, mean, rstd, gamma, beta, eps):
        self.pre, self.mean, self.rstd, self.gamma, self.beta, self.eps = pre, mean, rstd, gamma, beta, eps

    def c(self):
        return C.byref(_lib.LnRef(_p(self.mean), _p(self.rstd), _p(self.gamma), _p(self.beta)))

    def materialize(self):
        """The fp32 LayerNorm output itself (one extra launch; used where a plain tensor is needed)."""
        y = torch.empty_like(self.pre)
        lib = _lib.load()
        D = self.pre.shape[-1]
        _lib.check(lib.ovqa_layernorm_fwd(OVQA_F32, OVQA_F32, _p(self.pre), _p(self.gamma), _p(self.beta), None, 0, _p(y),
                                          None, None, None, self.pre.numel() // D, D, float(self.eps), _stream()),
                   "layernorm_fwd")
        return y


def linear_fwd_split3(x, w, bias, outs):
    """outs[i][m, :] = x[m, :] w[i F:(i + 1) F]^T + bias[i F:(i + 1) F] for the three stacked [F, K] matrices of w, in one
    launch (``ovqa_linear_fwd_split3``); every output has its own row stride (e.g. a slot ``cache[:, n]`` of a cache)."""
    _dev(x)
    ldx, M = _rows(x)
    F3, K = w.shape
    F = F3 // 3
    _weight(w, "linear_fwd_split3")
    assert F3 == 3 * F and x.shape[-1] == K and w.dtype == x.dtype and len(outs) == 3
    lds = []
    for o in outs:
        ldo, rows = _rows(o)
        assert rows == M and o.shape[-1] == F and o.dtype == x.dtype
        lds.append(ldo)
    _lib.check(_lib.load().ovqa_linear_fwd_split3(_dt(x), _p(x), ldx, _p(w), _p(bias), _p(outs[0]), lds[0], _p(outs[1]),
                                                  lds[1], _p(outs[2]), lds[2], M, F, K, _stream()), "linear_fwd_split3")
    return outs


def linear_fwd_res32(x, w, bias, residual, drop=None):
    """pre32 = res + drop(x w^T + bias) in fp32; ``residual`` is an fp32 tensor [.., N] or an LnRef (the previous
    block's LayerNorm, recomputed in the epilogue).  x, w bf16."""
    _dev(x)
    lib = _lib.load()
    ldx, M = _rows(x)
    N, K = w.shape
    _weight(w, "linear_fwd_res32")
    assert x.shape[-1] == K and w.dtype == x.dtype == torch.bfloat16
    ln = None
    if isinstance(residual, LnRef):
        ln, residual = residual.c(), residual.pre
    assert residual.dtype == torch.float32 and residual.shape[-1] == N
    if not residual.is_contiguous() and (residual.stride(-1) != 1 or residual.dim() > 2):
        residual = residual.contiguous()
    ldres, mr = _rows(residual)
    assert mr == M
    pre = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    _lib.check(lib.ovqa_linear_fwd_res32(_p(x), ldx, _p(w), _p(bias), _p(residual), ldres, ln, _p(pre), N, M, N, K,
                                         _drop(drop), _stream()), "linear_fwd_res32")
    return pre


def linear_bwd_data(dy, w, preact=None, drop=None, out=None, addend=None):
    """dx = dy w  [* dropmask * gelu'(preact)]  [+ addend]   (addend may alias out)."""
    _dev(dy)
    lib = _lib.load()
    lddy, M = _rows(dy)
    N, K = w.shape
    assert dy.shape[-1] == N and w.dtype == dy.dtype
    dx = out if out is not None else torch.empty(*dy.shape[:-1], K, dtype=dy.dtype, device=dy.device)
    lddx, _ = _rows(dx)
    ldadd = 0
    if addend is not None:
        ldadd, ma = _rows(addend)
        assert ma == M and addend.dtype == dy.dtype and addend.shape[-1] == K
    _lib.check(lib.ovqa_linear_bwd_data(_dt(dy), _p(dy), lddy, _p(w), _p(dx), lddx, _p(preact), _p(addend), ldadd,
                                        M, N, K, _drop(drop), _stream()), "linear_bwd_data")
    return dx


def linear_bwd_data_wt_ok(dy, wt) -> bool:
    """Shapes/strides the transposed-weight dX kernel accepts (bf16, everything a multiple of 8)."""
    K, N = wt.shape
    if os.environ.get("OVQA_FORCE_SIMPLE", "0") not in ("", "0"):
        return False  # cross-check mode: everything through the VALU reference kernels
    return (dy.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and wt.stride(1) == 1 and N % 8 == 0
            and K % 8 == 0 and wt.stride(0) % 8 == 0 and _rows(dy)[0] % 8 == 0 and wt.data_ptr() % 16 == 0)


def linear_bwd_data_wt(dy, wt, preact=None, drop=None, out=None, addend=None):
    """dx = dy wt^T [* dropmask * gelu'(preact)] [+ addend] with wt = the TRANSPOSED weight copy [K, N] (rows may
    be strided: a column block of a wider transposed matrix)."""
    _dev(dy)
    lib = _lib.load()
    lddy, M = _rows(dy)
    K, N = wt.shape
    assert dy.shape[-1] == N and wt.dtype == dy.dtype and wt.stride(1) == 1
    dx = out if out is not None else torch.empty(*dy.shape[:-1], K, dtype=dy.dtype, device=dy.device)
    lddx, _ = _rows(dx)
    ldadd = 0
    if addend is not None:
        ldadd, ma = _rows(addend)
        assert ma == M and addend.dtype == dy.dtype and addend.shape[-1] == K
    _lib.check(lib.ovqa_linear_bwd_data_wt(_dt(dy), _p(dy), lddy, _p(wt), wt.stride(0), _p(dx), lddx, _p(preact),
                                           _p(addend), ldadd, M, N, K, _drop(drop), _stream()), "linear_bwd_data_wt")
    return dx


def grouped_transpose(table, n, max_tiles):
    """table: device uint8 tensor holding n ovqa_transpose_problem structs."""
    _dev(table)
    _lib.check(_lib.load().ovqa_grouped_transpose(_p(table), n, max_tiles, _stream()), "grouped_transpose")


def linear_bwd_weight(dy, x, dw, db=None, accumulate=False, accumulate_db=None):
    """dw (fp32 [N,K]) (+)= dy^T x ; db (fp32 [N]) (+)= colsum(dy)."""
    flags = int(bool(accumulate)) | (int(bool(accumulate if accumulate_db is None else accumulate_db)) << 1)
    _dev(dy)
    lib = _lib.load()
    lddy, M = _rows(dy)
    ldx, Mx = _rows(x)
    N, K = dy.shape[-1], x.shape[-1]
    assert M == Mx and dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == N * K
    assert db is None or (db.dtype == torch.float32 and db.numel() == N)
    _lib.check(lib.ovqa_linear_bwd_weight(_dt(dy), _p(dy), lddy, _p(x), ldx, _p(dw), _p(db), M, N, K,
                                          flags, _p(workspace(dy.device)), _stream()), "linear_bwd_weight")


def bias_grad(dy, db, accumulate=False):
    """db (fp32 [N]) (+)= column sums of dy (bf16 only; fp32 uses linear_bwd_weight)."""
    _dev(dy)
    lddy, M = _rows(dy)
    _lib.check(_lib.load().ovqa_bias_grad(_dt(dy), _p(dy), lddy, _p(db), M, dy.shape[-1], int(accumulate), _stream()),
               "bias_grad")


class WgradQueue:
    """Deferred weight gradients: (dy, x, dw[, db]) collected during a backward pass and computed by ONE grouped launch
    (ovqa_grouped_linear_bwd_weight) at its end, on the main stream; ``finish()`` launches them.  The queued activations
    are referenced until then.

    MEASURED alternatives, removed from the source in round 6 (profiles/README.md has the numbers): launching the tiles
    early on a side stream so that they overlap the rest of backward LOSES (5.80 / 5.77 / 5.70 ms per step at 224 / 700 /
    1300 tiles per flush against 5.43 for one launch at the very end, round 2: every kernel of the step is bound by the
    per-CU L2->LDS path, concurrent kernels steal it from the critical chain); the single launch on a forked side stream:
    no measurable effect; 256 x 256 tiles on one 16-wave workgroup per CU: 3.231 against 3.195 ms per step."""

    TILE = 128

    def __init__(self):
        self.items = []
        self.reduces = []    # deferred LayerNorm dgamma/dbeta reductions: (partial, blocks, D, out0, out1)
        self.inflight = []   # the items of launches of this pass (referenced until finish())
        self.keepalive = []  # tables/tensors of captured launches must outlive the graph
        self._cache = []
        self.defer_uploads = False  # capture mode of a harness: tables are uploaded ONCE after the capture
        self._deferred = []         # (pinned host, device, nbytes) of tables a captured launch reads
        self._producers = set()     # streams (other than the flushing one) whose kernels wrote queued operands
        # A backward pass differentiated in PHASES (train.TrainStep, N > 1) flushes the weight gradients at the end of
        # every phase -- its gradient segment goes on the wire -- but the LayerNorm dgamma / dbeta reductions write the 1-D
        # tail of the arena, which belongs to the LAST segment: held until the last phase they are one launch, not one per
        # phase (5 x 9.5 us against 15 in the MCAN step)
        self.hold_reduces = False
        self.hold_items = False     # the same for the products of a phase that releases no gradient segment
        self.armed = False          # functional._armed_queue: a flush callback is registered for the running backward call
        # The optimiser step inside the LAST launch of a pass (ovqa_grouped_linear_bwd_weight_adam; train.TrainStep at
        # world size 1): an object with ``pre_flush()`` (the step counters / learning rate of this step, launched in front
        # of the weight-gradient launch), ``targets(items) -> [AdamTarget | None]`` and ``consts() -> AdamConsts``.
        self.adam = None

    def _note_producer(self, t):
        if t.is_cuda:
            self._producers.add(torch.cuda.current_stream(t.device))

    def _join_producers(self):
        """Backward ops run on the stream their forward ran on; a block that ran on a side stream queues operands the
        deferred launches (current stream) must not read early."""
        for st in self._producers:
            cur = torch.cuda.current_stream(st.device)
            if st != cur:
                cur.wait_stream(st)
        self._producers.clear()

    def add(self, dy, x, dw, accumulate, db=None, accumulate_db=False):
        lddy, M = _rows(dy)
        ldx, Mx = _rows(x)
        assert M == Mx and dw.dtype == torch.float32 and dw.is_contiguous()
        assert db is None or (db.dtype == torch.float32 and db.numel() == dy.shape[-1])
        if self._overlaps_queued(dw) or (db is not None and self._overlaps_queued(db)):
            # the same weight (or an overlapping packed group: [fc_q|fc_k|fc_v] of a module's self-attention use and
            # [fc_k|fc_v] of its cross-attention use have different base pointers) written twice in one backward
            # pass: the two contributions must be ordered, not binned into one grouped launch
            self.finish()
        flags = int(bool(accumulate)) | (int(bool(accumulate_db)) << 1)
        N, K = dy.shape[-1], x.shape[-1]
        self._note_producer(dy)
        self.items.append((dy, x, dw, lddy, ldx, M, N, K, flags, db))

    @staticmethod
    def _span(t):
        return t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()

    def _overlaps_queued(self, t):
        lo, hi = self._span(t)
        for it in self.items + [i for grp in self.inflight for i in grp]:
            for other in (it[2], it[9]):
                if other is not None:
                    olo, ohi = self._span(other)
                    if lo < ohi and olo < hi:
                        return True
        return False

    def add_reduce(self, partial, blocks, D, out0, out1, accumulate=False):
        assert out0.dtype == torch.float32 and (out1 is None or out1.dtype == torch.float32) and partial.dtype == torch.float32
        self._note_producer(partial)
        self.reduces.append((partial, blocks, D, out0, out1, bool(accumulate)))

    def _flush_reduces(self):
        """One launch summing the row-slab partials of every queued LayerNorm backward (main stream)."""
        if not self.reduces:
            return
        red, self.reduces = self.reduces, []
        # The reduce is deterministic: plain stores, a fixed summation order.  Two problems of ONE launch must therefore
        # not share an output (a LayerNorm applied twice in a pass): later uses of an output go to a later launch, in
        # order, and add to what the earlier one stored.
        waves = []
        for item in red:
            key = (item[3].data_ptr(), item[4].data_ptr() if item[4] is not None else 0)
            for w in waves:
                if key not in w[0]:
                    w[0].add(key)
                    w[1].append(item)
                    break
            else:
                waves.append(({key}, [item]))
        for _, items in waves:
            self._launch_reduces(items)

    def _launch_reduces(self, red):
        import numpy as np
        dev = red[0][0].device
        probs = (_lib.ReduceProblem * len(red))()
        for i, (partial, blocks, D, out0, out1, acc) in enumerate(red):
            probs[i] = _lib.ReduceProblem(_p(partial), _p(out0), _p(out1), blocks, D, int(acc), 0)
        raw = np.frombuffer(bytes(probs), dtype=np.uint8)
        capturing = torch.cuda.is_current_stream_capturing()
        entry = self._buffers(raw.size, dev, capturing)
        host, devbuf = entry[0], entry[1]
        host[:raw.size] = torch.from_numpy(raw.copy())
        self._upload(host, devbuf, raw.size, capturing)
        _lib.check(_lib.load().ovqa_grouped_partial_reduce(devbuf.data_ptr(), len(red), max(r[1] for r in red),
                                                           max(r[2] for r in red), _stream()),
                   "grouped_partial_reduce")
        self._used(entry, torch.cuda.current_stream(dev), capturing)
        if capturing:
            self.keepalive.append((host, devbuf, red))

    def flush(self, final=False):
        """Launch everything queued so far.  ``final``: the last launch of the pass (``finish``): the one that may carry the
        optimiser step (``self.adam``)."""
        if not self.items:
            return
        import numpy as np
        items, self.items = self.items, []
        dev = items[0][0].device
        probs = (_lib.WgradProblem * len(items))()
        per_problem = []
        fast = all(it[5] % 64 == 0 and it[3] % 8 == 0 and it[4] % 8 == 0 and it[0].data_ptr() % 16 == 0
                   and it[1].data_ptr() % 16 == 0 for it in items)
        tile = self.TILE
        for i, (dy, x, dw, lddy, ldx, M, N, K, acc, db) in enumerate(items):
            probs[i] = _lib.WgradProblem(_p(dy), _p(x), _p(dw), _p(db), lddy, ldx, M, N, K, acc)
            tn, tk = (N + tile - 1) // tile, (K + tile - 1) // tile
            per_problem.append((M * tn * tk, M, [(i, c, r, 0) for c in range(tn) for r in range(tk)]))
        # XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD
        # group), each XCD has a private L2, and all tiles of one problem stream the same dY / X panels.
        # Whole problems are therefore binned per XCD (longest-processing-time first) and the bins are
        # interleaved; without this every XCD fetches every panel (measured: 2.9 GB of L2 misses per launch).
        # A problem far above an XCD's fair share is split (whole column blocks of tiles stay together: they share the dY
        # panel): in the ONE launch of a plain step no problem is (61 problems, 336 tiles per XCD), but a launch per backward
        # phase (N > 1) can consist of little more than the hoisted K / V projection -- 192 tiles that one XCD then walked
        # alone while seven idled (73 us for 385 light tiles).  The other operand's panel is fetched once per XCD that
        # takes a part: M x K x 2 bytes, small beside the idle time it removes.
        fair = sum(w for w, _, _ in per_problem) / 8.0
        parts = []
        for work, M, tl in per_problem:
            n_parts = min(8, int(work / max(fair, 1.0) + 0.75)) if work > 1.25 * fair else 1
            if n_parts <= 1:
                parts.append((work, M, tl))
                continue
            per = (len(tl) + n_parts - 1) // n_parts
            for i in range(0, len(tl), per):
                chunk = tl[i:i + per]
                parts.append((work * len(chunk) / len(tl), M, chunk))
        bins = [[0, []] for _ in range(8)]
        for work, M, tl in sorted(parts, key=lambda t: (-t[1], -t[0])):
            b = min(bins, key=lambda bb: bb[0])
            b[0] += work
            b[1].extend(tl)
        # (round 5, with the optimiser step in the epilogue: interleaving long and short tiles -- 1, 2, 3 or 5 long ones per short
        # one -- so that the HBM-bound epilogues do not come in rounds measured 614-756 us against 599 longest-first: kept as is)
        depth = max(len(b[1]) for b in bins)
        tile_arr = np.full((depth * 8, 4), -1, dtype=np.int32)  # -1 = padding entry (kernel returns)
        for xcd, b in enumerate(bins):
            if b[1]:
                tile_arr[xcd:xcd + 8 * len(b[1]):8] = np.array(b[1], dtype=np.int32)
        tiles = tile_arr
        prob_bytes = np.frombuffer(bytes(probs), dtype=np.uint8)
        pad = (-prob_bytes.size) % 16  # keep the int4 tile table 16-byte aligned
        if pad:
            prob_bytes = np.concatenate([prob_bytes, np.zeros(pad, dtype=np.uint8)])
        # the optimiser step inside this launch (the last one of the pass): one target per problem behind the tile table
        targets = None
        if final and self.adam is not None and fast and os.environ.get("OVQA_FORCE_SIMPLE", "0") in ("", "0"):
            tl = self.adam.targets(items)
            if any(t is not None for t in tl):
                arr = (_lib.AdamTarget * len(items))(*[t if t is not None else _lib.AdamTarget() for t in tl])
                targets = np.frombuffer(bytes(arr), dtype=np.uint8)
        tile_bytes = tile_arr.nbytes + ((-tile_arr.nbytes) % 16)
        nbytes = prob_bytes.size + tile_bytes + (targets.size if targets is not None else 0)
        capturing = torch.cuda.is_current_stream_capturing()
        entry = self._buffers(nbytes, dev, capturing)
        host, devbuf = entry[0], entry[1]
        host[:prob_bytes.size] = torch.from_numpy(prob_bytes.copy())
        host[prob_bytes.size:prob_bytes.size + tile_arr.nbytes] = torch.from_numpy(tile_arr.view(np.uint8).reshape(-1).copy())
        if targets is not None:
            host[prob_bytes.size + tile_bytes:nbytes] = torch.from_numpy(targets.copy())
        # (on the main stream: a fork / join in a captured graph turns every node boundary of the replay into a cross-queue
        # dependency -- scripts/boundary_bench.py: 1.6 us per dependent launch on one queue)
        main = torch.cuda.current_stream(dev)
        self._upload(host, devbuf, nbytes, capturing)
        if targets is not None:
            consts = self.adam.consts()
            _lib.check(_lib.load().ovqa_grouped_linear_bwd_weight_adam(
                OVQA_BF16, devbuf.data_ptr(), devbuf.data_ptr() + prob_bytes.size, len(tiles),
                devbuf.data_ptr() + prob_bytes.size + tile_bytes, C.addressof(consts), main.cuda_stream),
                "grouped_linear_bwd_weight_adam")
        else:
            _lib.check(_lib.load().ovqa_grouped_linear_bwd_weight(
                OVQA_BF16, devbuf.data_ptr(), devbuf.data_ptr() + prob_bytes.size, len(tiles), int(fast),
                main.cuda_stream), "grouped_linear_bwd_weight")
        self._used(entry, main, capturing)
        self.inflight.append(items)
        if capturing:
            self.keepalive.append((host, devbuf, items))

    def _upload(self, host, devbuf, nbytes, capturing):
        """Host table -> device.  Inside a harness capture (defer_uploads) the copy is NOT recorded as a memcpy node:
        the captured launches read device addresses that never change between replays, so ``upload_deferred``
        copies each table once, right after the capture."""
        if capturing and self.defer_uploads:
            self._deferred.append((host, devbuf, nbytes))
        else:
            devbuf[:nbytes].copy_(host[:nbytes], non_blocking=True)

    def upload_deferred(self):
        for host, devbuf, nbytes in self._deferred:
            devbuf[:nbytes].copy_(host[:nbytes], non_blocking=True)
        self._deferred = []
        torch.cuda.synchronize()

    def abandon(self):
        """Drop everything queued and every table upload deferred by an ABORTED stream capture (nothing of it ran):
        the next pass starts from an empty queue."""
        self.items, self.reduces = [], []
        self.inflight, self._deferred = [], []
        self._producers.clear()
        self.defer_uploads = False
        self.hold_reduces = False
        self.hold_items = False
        self.armed = False

    def finish(self):
        """End of the backward pass: the LayerNorm parameter reductions and the one grouped weight-gradient launch."""
        self.armed = False  # (the flush callback of this backward call has run: the next call registers its own)
        self._join_producers()
        if not self.hold_reduces:
            self._flush_reduces()
        if self.hold_items:
            return  # (the queued operands stay referenced; the next phase's flush launches them with its own)
        if self.adam is not None and self.items:
            self.adam.pre_flush()  # this step's counters and learning rate, in front of the launch that applies them
        self.flush(final=True)
        self.inflight = []

    def _buffers(self, nbytes, dev, capturing):
        """Cache entry [pinned host, device, event, owned-by-a-graph] for a table upload.  Pinned allocations
        are not permitted while a stream is capturing, so buffers are created in the eager warm-up pass and
        handed to the capture; a buffer is never reused while a previous launch may still read it (captured
        ones never; eager ones after the event the caller records with ``_used`` once its launch is queued)."""
        for entry in self._cache:
            host, devbuf, ev, owned = entry
            if owned or host.numel() < nbytes or devbuf.device != dev:
                continue
            if ev is not None:
                if capturing:
                    continue
                ev.synchronize()
            if capturing:
                entry[3] = True  # the graph owns it from now on
            return entry
        if capturing:
            raise RuntimeError(
                f"WgradQueue: no idle table buffer of {nbytes} bytes is available while a stream is capturing (pinned "
                "memory cannot be allocated during capture); run the backward pass eagerly once and call "
                "wgrad_queue().reserve(n) before the capture")
        self._max_nbytes = max(getattr(self, "_max_nbytes", 0), nbytes)
        size = max(nbytes * 2, 1 << 16)
        entry = [torch.empty(size, dtype=torch.uint8).pin_memory(), torch.empty(size, dtype=torch.uint8, device=dev),
                 None, capturing]
        self._cache.append(entry)
        return entry

    @staticmethod
    def _used(entry, stream, capturing):
        if not capturing:
            entry[2] = torch.cuda.Event()
            entry[2].record(stream)

    def reserve(self, n):
        """Pre-create ``n`` idle table buffers (call after the eager warm-up passes and a device synchronise, before
        graph capture).  They are sized from the largest table the warm-up passes built; buffers of finished eager
        launches are recycled (their events have fired once the device is idle)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        size = max(1 << 17, 2 * getattr(self, "_max_nbytes", 0))
        for e in self._cache:
            if not e[3] and e[2] is not None and e[2].query():
                e[2] = None
        while sum(1 for e in self._cache if not e[3] and e[2] is None and e[0].numel() >= size) < n:
            self._cache.append([torch.empty(size, dtype=torch.uint8).pin_memory(),
                                torch.empty(size, dtype=torch.uint8, device=dev), None, False])


def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=None, pos=None, save_stats=True, want_f32=False):
    """Returns (y, mean, rstd), or (y, y_f32, mean, rstd) with ``want_f32`` (the unrounded fp32 result next to y)."""
    _dev(x)
    lib = _lib.load()
    assert x.is_contiguous()
    D = x.shape[-1]
    M = x.numel() // D
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    pos_rows = 0
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[-1] == D
        pos_rows = pos.shape[0]
    _lib.check(lib.ovqa_layernorm_fwd(_DT[out_dtype], _dt(x), _p(x), _p(gamma), _p(beta), _p(pos), pos_rows, _p(y),
                                      _p(y32), _p(mean), _p(rstd), M, D, float(eps), _stream()), "layernorm_fwd")
    return (y, y32, mean, rstd) if want_f32 else (y, mean, rstd)


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, drop=None, dx_dtype=None, accumulate=False, defer=None):
    """Returns (dx, dx_dropped) -- dx_dropped is dx when dropout is off.

    ``defer`` (a WgradQueue): dgamma/dbeta are not reduced now; the kernel's row-slab partials go to a buffer
    of their own and the queue sums the partials of every LayerNorm of the backward pass in one launch."""
    _dev(dy)
    lib = _lib.load()
    assert dy.is_contiguous() and x.is_contiguous()
    D = x.shape[-1]
    M = x.numel() // D
    dx_dtype = dx_dtype or dy.dtype
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    has_drop = drop is not None and drop.p > 0.0
    dxd = torch.empty(x.shape, dtype=dy.dtype, device=x.device) if has_drop else None
    if defer is not None and M > 0:
        blocks = lib.ovqa_layernorm_bwd_blocks(M, D)
        ws = torch.empty(blocks * 2 * D, dtype=torch.float32, device=x.device)
        gout, bout = None, None
    else:
        ws, gout, bout = workspace(dy.device), dgamma, dbeta
    _lib.check(lib.ovqa_layernorm_bwd(_dt(dy), _DT[dx_dtype], _p(dy), _p(x), _dt(x), _p(gamma), _p(mean), _p(rstd),
                                      _p(dx), _p(dxd), _p(gout), _p(bout), M, D, int(accumulate), _drop(drop),
                                      _p(ws), _stream()), "layernorm_bwd")
    if defer is not None and M > 0:
        defer.add_reduce(ws, blocks, D, dgamma, dbeta, accumulate)  # (= or +=: the reduce stores, no pre-zeroing)
    return dx, (dxd if has_drop else dx)


def _mask_strides(mask, B, H, nq, nk):
    """Broadcast strides (in elements) of an additive fp32 mask of shape (b|1, h|1, q|1, nk)."""
    if mask is None:
        return None, 0, 0, 0
    assert mask.dtype == torch.float32 and mask.dim() == 4 and mask.shape[-1] == nk and mask.stride(-1) == 1, \
        f"mask must be fp32 (B|1, H|1, nq|1, nk); got {tuple(mask.shape)} {mask.dtype}"
    sb = mask.stride(0) if mask.shape[0] > 1 else 0
    sh = mask.stride(1) if mask.shape[1] > 1 else 0
    sq = mask.stride(2) if mask.shape[2] > 1 else 0
    assert mask.shape[0] in (1, B) and mask.shape[1] in (1, H) and mask.shape[2] in (1, nq)
    return mask, sb, sh, sq


def _lo_buffer(lo_out, like):
    """``lo_out``: a list the caller passes to receive the bf16 rounding residual of the attention output (what
    attention_bwd takes as ``o_lo``: delta = dO . (o + o_lo)); bf16 only -- fp32 outputs have none."""
    if lo_out is None or like.dtype != torch.bfloat16:
        return None
    lo = torch.empty_like(like)
    lo_out.append(lo)
    return lo


def attention_fwd(q, k, v, mask, H, scale=None, need_att=False, save_lse=True, att_drop=None, lo_out=None):
    """q [B,nq,H*dk], k [B,nk,H*dk], v [B,nk,H*dv] (row-strided views allowed) -> o [B,nq,H*dv], lse, att.
    ``att_drop`` (DropSpec): dropout on the attention probabilities (BERT-style; VALU kernels)."""
    _dev(q)
    lib = _lib.load()
    B, nq = q.shape[0], q.shape[1]
    nk = k.shape[1]
    dk, dv = q.shape[2] // H, v.shape[2] // H
    ldq, _ = _rows(q)
    ldk, _ = _rows(k)
    ldv, _ = _rows(v)
    scale = (1.0 / math.sqrt(dk)) if scale is None else scale
    o = torch.empty(B, nq, H * dv, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, nq, dtype=torch.float32, device=q.device) if save_lse else None
    att = torch.empty(B, H, nq, nk, dtype=q.dtype, device=q.device) if need_att else None
    mask, sb, sh, sq = _mask_strides(mask, B, H, nq, nk)
    lo = _lo_buffer(lo_out, o)
    _lib.check(lib.ovqa_attention_fwd(_dt(q), _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(mask), sb, sh, sq, _p(o), H * dv,
                                      _p(lse), _p(att), _p(lo), B, H, nq, nk, dk, dv, float(scale), _drop(att_drop),
                                      _stream()), "attention_fwd")
    return o, lse, att


def attention_fwd_prefix_lm_ok(q, H):
    """Shapes ``attention_fwd_prefix_lm`` covers: bf16 on the GPU, heads of 64 / 96 / 128 features, at most 256 (192 for the
    wider heads) positions, 16-byte aligned rows."""
    if os.environ.get("OVQA_FORCE_SIMPLE", "0") == "1" or os.environ.get("OVQA_NO_PREFIX_LM", "0") == "1":
        return False
    d = q.shape[2] // H
    return (q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 3 and d in (64, 96, 128)
            and q.shape[1] <= (256 if d == 64 else 192) and q.data_ptr() % 16 == 0 and _rows(q)[0] % 8 == 0)


def attention_fwd_prefix_lm(q, k, v, key_mask, causal_tail, H, scale=None):
    """Inference-only self-attention under a prefix-LM mask given by its structure (``ovqa_attention_fwd_prefix_lm``): the
    key-mask row (b|1, h|1, 1, n) or None, and among the last ``causal_tail`` positions query i does not see keys j > i.
    Returns o [B, n, H*d]."""
    _dev(q)
    lib = _lib.load()
    B, n = q.shape[0], q.shape[1]
    d = q.shape[2] // H
    assert k.shape[1] == n and v.shape[1] == n and v.shape[2] == q.shape[2]
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    o = torch.empty(B, n, H * d, dtype=q.dtype, device=q.device)
    key_mask, sb, sh, sq = _mask_strides(key_mask, B, H, n, n)
    assert sq == 0, "a key mask: one row per (b, h)"
    _lib.check(lib.ovqa_attention_fwd_prefix_lm(_dt(q), _p(q), _rows(q)[0], _p(k), _rows(k)[0], _p(v), _rows(v)[0],
                                                _p(key_mask), sb, sh, int(causal_tail), _p(o), H * d, None, B, H, n, d,
                                                float(scale), _stream()), "attention_fwd_prefix_lm")
    return o


def attention_qkv_fwd(x, w, bias, mask, H, scale=None, save_lse=True, lo_out=None):
    """Self-attention forward with the packed projections inside (``ovqa_attention_qkv_fwd``): x [B,n,d_model],
    w [3*H*d, d_model] (fc_q | fc_k | fc_v rows), bias fp32 [3*H*d] -> (qkv [B,n,3*H*d], o [B,n,H*d], lse).
    ``mask``: key mask (b|1, h|1, 1, n) or None."""
    _dev(x)
    lib = _lib.load()
    B, n, Dm = x.shape
    d = w.shape[0] // (3 * H)
    _weight(w, "attention_qkv_fwd")
    assert w.shape[0] == 3 * H * d and w.shape[1] == Dm and w.dtype == x.dtype
    ldx, _ = _rows(x)
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    qkv = torch.empty(B, n, 3 * H * d, dtype=x.dtype, device=x.device)
    o = torch.empty(B, n, H * d, dtype=x.dtype, device=x.device)
    lse = torch.empty(B, H, n, dtype=torch.float32, device=x.device) if save_lse else None
    mask, sb, sh, sq = _mask_strides(mask, B, H, n, n)
    assert sq == 0, "attention_qkv_fwd takes a key mask (one row per (b, h))"
    lo = _lo_buffer(lo_out, o)
    _lib.check(lib.ovqa_attention_qkv_fwd(_dt(x), _p(x), ldx, _p(w), _p(bias), _p(qkv), 3 * H * d, _p(mask), sb, sh,
                                          _p(o), H * d, _p(lse), _p(lo), B, H, n, Dm, d, float(scale), _stream()),
               "attention_qkv_fwd")
    return qkv, o, lse


def attention_q_fwd(x, w, bias, k, v, mask, H, scale=None, save_lse=True, lo_out=None):
    """Cross / guided attention forward with the query projection inside (``ovqa_attention_q_fwd``): x [B,nq,d_model],
    w = fc_q weight [H*d, d_model], bias fp32 [H*d]; k, v [B,nk,H*d] already projected (strided views of a packed buffer
    are fine) -> (q [B,nq,H*d], o [B,nq,H*d], lse).  ``mask``: key mask (b|1, h|1, 1, nk) or None."""
    _dev(x)
    lib = _lib.load()
    B, nq, Dm = x.shape
    nk = k.shape[1]
    d = w.shape[0] // H
    _weight(w, "attention_q_fwd")
    assert w.shape[0] == H * d and w.shape[1] == Dm and w.dtype == x.dtype
    assert k.shape[0] == B and v.shape[:2] == k.shape[:2] and k.shape[2] == H * d == v.shape[2] and k.dtype == x.dtype
    ldx, _ = _rows(x)
    ldk, _ = _rows(k)
    ldv, _ = _rows(v)
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    q = torch.empty(B, nq, H * d, dtype=x.dtype, device=x.device)
    o = torch.empty(B, nq, H * d, dtype=x.dtype, device=x.device)
    lse = torch.empty(B, H, nq, dtype=torch.float32, device=x.device) if save_lse else None
    mask, sb, sh, sq = _mask_strides(mask, B, H, nq, nk)
    assert sq == 0, "attention_q_fwd takes a key mask (one row per (b, h))"
    lo = _lo_buffer(lo_out, o)
    _lib.check(lib.ovqa_attention_q_fwd(_dt(x), _p(x), ldx, _p(w), _p(bias), _p(q), H * d, _p(k), ldk, _p(v), ldv, _p(mask),
                                        sb, sh, _p(o), H * d, _p(lse), _p(lo), B, H, nq, nk, Dm, d, float(scale),
                                        _stream()), "attention_q_fwd")
    return q, o, lse


def attention_decode(q, k_cache, v_cache, n, H, mask=None, group=1, scale=None, out=None):
    """One decoding step's attention (``ovqa_attention_decode``): q [R, 1, H*d] or [R, H*d]; k_cache / v_cache
    [R // group, Lmax, H*d] in-place caches of which the first ``n`` keys are live (the ``group`` beams of a sample
    share a cache row); mask: additive fp32 [R, >= n] or None.  Returns o with q's shape."""
    _dev(q)
    lib = _lib.load()
    shape = q.shape
    R, F = shape[0], shape[-1]
    d = F // H
    assert q.numel() == R * F and q.stride(-1) == 1 and k_cache.dim() == 3 and v_cache.shape == k_cache.shape
    assert k_cache.stride(2) == 1 and v_cache.stride(2) == 1 and k_cache.shape[2] == F and 1 <= n <= k_cache.shape[1]
    assert k_cache.shape[0] * group == R, (tuple(k_cache.shape), group, R)
    assert k_cache.stride() == v_cache.stride() and k_cache.dtype == q.dtype == v_cache.dtype
    o = out if out is not None else torch.empty(shape, dtype=q.dtype, device=q.device)
    ldq = q.stride(0)
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.dim() == 2 and mask.shape[0] == R and mask.shape[1] >= n
        assert mask.stride(1) == 1
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    _lib.check(lib.ovqa_attention_decode(_dt(q), _p(q), ldq, _p(k_cache), k_cache.stride(1), _p(v_cache),
                                         v_cache.stride(1), k_cache.stride(0), group, _p(mask),
                                         0 if mask is None else mask.stride(0), _p(o), o.stride(0), R, H, n, d,
                                         float(scale), _stream()), "attention_decode")
    return o


def topk_rows(x, k):
    """(values, indices) of the k <= 8 largest entries of every row of the fp32 tensor x [..., V], best first (ties: the
    smaller index): what torch.topk(x, k, dim=-1) returns, in one small launch (``ovqa_topk_rows``)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and 1 <= k <= min(8, x.shape[-1])
    x2 = x.reshape(-1, x.shape[-1])
    vals = torch.empty(x2.shape[0], k, dtype=torch.float32, device=x.device)
    idx = torch.empty(x2.shape[0], k, dtype=torch.int64, device=x.device)
    _lib.check(_lib.load().ovqa_topk_rows(_p(x2), x2.stride(0), x2.shape[0], x2.shape[1], k, _p(vals), _p(idx),
                                          _stream()), "topk_rows")
    return vals.view(*x.shape[:-1], k), idx.view(*x.shape[:-1], k)


def decode_embed(tokens, emb, pos, seq, pad_idx, mask_value, mask, col, out_dtype, x32=None, x=None):
    """One decoding step's inputs (``ovqa_decode_embed``): seq += 1 in place; x32 / x [R, D] = emb[tokens] + pos[seq] in
    fp32 / ``out_dtype``; mask[:, col] = mask_value where tokens == pad_idx else 0 (mask fp32 [R, >= col + 1] or None).
    Returns (x32, x); either may be None when not asked for (x is None iff out_dtype is fp32 and x32 is given)."""
    _dev(emb)
    R, D = tokens.numel(), emb.shape[1]
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and seq.dtype == torch.int64 and seq.is_contiguous()
    assert seq.numel() == R and emb.dtype == torch.float32 and pos.dtype == torch.float32
    assert emb.stride(1) == 1 and pos.stride(1) == 1 and pos.shape[1] == D
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.dim() == 2 and mask.shape[0] == R and mask.stride(1) == 1
    if x32 is None:
        x32 = torch.empty(R, D, dtype=torch.float32, device=emb.device)
    if x is None and out_dtype != torch.float32:
        x = torch.empty(R, D, dtype=out_dtype, device=emb.device)
    _lib.check(_lib.load().ovqa_decode_embed(_dt(x) if x is not None else OVQA_F32, _p(tokens), _p(emb), emb.stride(0),
                                             emb.shape[0], _p(pos), pos.stride(0), pos.shape[0], _p(seq), int(pad_idx),
                                             float(mask_value), _p(mask), 0 if mask is None else mask.stride(0), int(col),
                                             _p(x32), _p(x), R, D, _stream()), "decode_embed")
    return x32, x


def beam_candidates(logits, seq_logprob, seq_mask, prev_words, eos, k):
    """(vals, idx, wl) [R, k] of ``ovqa_beam_candidates``: log-softmax of the logits [R, V], candidate scores of
    beam_search.py:41-57 and the k best per row; seq_mask [R] is updated in place (prev_words == eos)."""
    _dev(logits)
    R, V = logits.shape
    assert logits.stride(1) == 1 and seq_logprob.numel() == R and seq_mask.numel() == R
    assert seq_logprob.dtype == torch.float32 and seq_mask.dtype == torch.float32
    assert seq_logprob.is_contiguous() and seq_mask.is_contiguous()
    if prev_words is not None:
        assert prev_words.dtype == torch.int64 and prev_words.numel() == R and prev_words.is_contiguous()
    vals = torch.empty(R, k, dtype=torch.float32, device=logits.device)
    wl = torch.empty(R, k, dtype=torch.float32, device=logits.device)
    idx = torch.empty(R, k, dtype=torch.int64, device=logits.device)
    _lib.check(_lib.load().ovqa_beam_candidates(_dt(logits), _p(logits), logits.stride(0), R, V, k, _p(seq_logprob),
                                                _p(seq_mask), _p(prev_words), int(eos), _p(vals), _p(idx), _p(wl),
                                                _stream()), "beam_candidates")
    return vals, idx, wl


def beam_commit(vals, idx, wl, seq_mask_in, hist_in, hist_out, seq_logprob_out, seq_mask_out, selected_beam, words,
                b_s, cur, k, beam, t):
    """``ovqa_beam_commit``: hist_in / hist_out = (outputs int64, log_probs fp32) [b_s, cur | beam, T] pairs."""
    _dev(vals)
    T = hist_out[0].shape[-1]
    for a in (vals, idx, wl, seq_mask_in, *hist_out, seq_logprob_out, seq_mask_out, selected_beam, words):
        assert a.is_contiguous()
    assert selected_beam.dtype == torch.int32 and words.dtype == torch.int64 and hist_out[0].dtype == torch.int64
    assert vals.numel() == b_s * cur * k and seq_mask_in.numel() == b_s * cur and hist_out[0].numel() == b_s * beam * T
    _lib.check(_lib.load().ovqa_beam_commit(_p(vals), _p(idx), _p(wl), _p(seq_mask_in),
                                            _p(hist_in[0]) if t > 0 else None, _p(hist_in[1]) if t > 0 else None,
                                            _p(hist_out[0]), _p(hist_out[1]), _p(seq_logprob_out), _p(seq_mask_out),
                                            _p(selected_beam), _p(words), b_s, cur, k, beam, t, T, _stream()),
               "beam_commit")


def attention_bwd(d_o, q, k, v, o, lse, mask, H, scale=None, dq=None, dk=None, dv=None, d_att=None, d_lse=None,
                  att_drop=None, o_lo=None):
    _dev(q)
    lib = _lib.load()
    B, nq = q.shape[0], q.shape[1]
    nk = k.shape[1]
    dkk, dvv = q.shape[2] // H, v.shape[2] // H
    scale = (1.0 / math.sqrt(dkk)) if scale is None else scale
    dq = dq if dq is not None else torch.empty(B, nq, H * dkk, dtype=q.dtype, device=q.device)
    dk = dk if dk is not None else torch.empty(B, nk, H * dkk, dtype=q.dtype, device=q.device)
    dv = dv if dv is not None else torch.empty(B, nk, H * dvv, dtype=q.dtype, device=q.device)
    delta = torch.empty(B, H, nq, dtype=torch.float32, device=q.device)
    mask, sb, sh, sq = _mask_strides(mask, B, H, nq, nk)
    if d_att is not None:
        assert d_att.is_contiguous() and d_att.dtype == q.dtype and d_att.shape == (B, H, nq, nk)
    if d_lse is not None:
        assert d_lse.is_contiguous() and d_lse.dtype == torch.float32 and d_lse.shape == (B, H, nq)
    if o_lo is not None:
        assert o_lo.dtype == o.dtype == torch.bfloat16 and o_lo.shape == o.shape and _rows(o_lo)[0] == _rows(o)[0]
    _lib.check(lib.ovqa_attention_bwd(
        _dt(q), _p(d_o), _rows(d_o)[0], _p(q), _rows(q)[0], _p(k), _rows(k)[0], _p(v), _rows(v)[0], _p(o), _rows(o)[0],
        _p(o_lo), _p(d_att), _p(lse), _p(mask), sb, sh, sq, _p(dq), _rows(dq)[0], _p(dk), _rows(dk)[0], _p(dv), _rows(dv)[0], _p(delta),
        _p(d_lse), B, H, nq, nk, dkk, dvv, float(scale), _drop(att_drop), _stream()), "attention_bwd")
    return dq, dk, dv


def attention_bwd_do_ok(dy, wt, q, k, mask, H):
    """Shapes the fused form of ``attention_bwd_do`` covers (``ovqa_attention_bwd_do``): bf16, d = 64, <= 32 keys, 65-128
    queries (guided attention) or <= 32 queries with an even head count (the 20 x 20 question self-attention), or 97-128
    queries x 97-128 keys (the image self-attention, round 5); key mask or none."""
    nq = q.shape[1] if q.dim() == 3 else 0
    if os.environ.get("OVQA_FORCE_SIMPLE", "0") == "1" or os.environ.get("OVQA_NO_FUSED_QKV", "0") == "1":
        return False  # (the library's A/B switches that turn the fused attention forms off)
    nk = k.shape[1]
    small_k = (64 < nq <= 128 or (1 <= nq <= 32 and H % 2 == 0)) and nk <= 32
    roles = 96 < nq <= 128 and 96 < nk <= 128  # image self-attention
    return (dy.is_cuda and dy.dtype == torch.bfloat16 and wt is not None and wt.dtype == torch.bfloat16 and q.dim() == 3
            and q.shape[2] == H * 64 and (small_k or roles)
            and dy.shape[-1] % 64 == 0
            and (mask is None or mask.shape[2] == 1) and wt.shape[0] == H * 64 and wt.shape[1] == dy.shape[-1])


def attention_bwd_do(dy, wt, q, k, v, o, lse, mask, H, scale=None, dq=None, dk=None, dv=None, o_lo=None):
    """Attention backward with the fc_o dX product inside (``ovqa_attention_bwd_do``): dy [B,nq,d_model] = gradient
    w.r.t. fc_o's output, wt [H*d, d_model] = the transposed weight copy; returns (dq, dk, dv).  The caller checks
    ``attention_bwd_do_ok`` (there is no scratch buffer for the two-kernel form here)."""
    _dev(q)
    lib = _lib.load()
    B, nq = q.shape[0], q.shape[1]
    nk, d = k.shape[1], q.shape[2] // H
    Dm = dy.shape[-1]
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    dq = dq if dq is not None else torch.empty(B, nq, H * d, dtype=q.dtype, device=q.device)
    dk = dk if dk is not None else torch.empty(B, nk, H * d, dtype=q.dtype, device=q.device)
    dv = dv if dv is not None else torch.empty(B, nk, H * d, dtype=q.dtype, device=q.device)
    delta = torch.empty(B, H, nq, dtype=torch.float32, device=q.device)
    mask, sb, sh, sq = _mask_strides(mask, B, H, nq, nk)
    assert sq == 0 and wt.stride(1) == 1 and dy.stride(-1) == 1
    if o_lo is not None:
        assert o_lo.dtype == o.dtype == torch.bfloat16 and o_lo.shape == o.shape and _rows(o_lo)[0] == _rows(o)[0]
    _lib.check(lib.ovqa_attention_bwd_do(
        _dt(q), _p(dy), _rows(dy)[0], _p(wt), wt.stride(0), None, 0, _p(q), _rows(q)[0], _p(k), _rows(k)[0], _p(v),
        _rows(v)[0], _p(o), _rows(o)[0], _p(o_lo), _p(lse), _p(mask), sb, sh, _p(dq), _rows(dq)[0], _p(dk), _rows(dk)[0],
        _p(dv), _rows(dv)[0], _p(delta), B, H, nq, nk, Dm, d, float(scale), _stream()), "attention_bwd_do")
    return dq, dk, dv


def pointer_score(q, k, scale, add_mask=None, key_fill=None, query_fill=None):
    """scores[b,t,n] = q[b,t].k[b,n]*scale (+ add_mask[b,n]) (-inf where key_fill[b,n] / query_fill[b,t])."""
    _dev(q)
    lib = _lib.load()
    assert q.is_contiguous() and k.is_contiguous()
    B, T, D = q.shape
    Nk = k.shape[1]
    s = torch.empty(B, T, Nk, dtype=torch.float32, device=q.device)
    _lib.check(lib.ovqa_pointer_score(_dt(q), _p(q), _p(k), _p(add_mask), _p(key_fill), _p(query_fill), _p(s),
                                      B, T, Nk, D, float(scale), _stream()), "pointer_score")
    return s


def batched_gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, out_dtype=None):
    """C[i] = alpha * op(A[i]) op(B[i]) for contiguous 3-D tensors."""
    _dev(a)
    lib = _lib.load()
    assert a.is_contiguous() and b.is_contiguous() and a.dim() == 3 and b.dim() == 3
    batch = a.shape[0]
    M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    N = b.shape[1] if trans_b else b.shape[2]
    out_dtype = out_dtype or a.dtype
    c = torch.empty(batch, M, N, dtype=out_dtype, device=a.device)
    _lib.check(lib.ovqa_batched_gemm(_dt(a), _DT[out_dtype], int(trans_a), int(trans_b), _p(a), a.shape[2],
                                     a.shape[1] * a.shape[2], _p(b), b.shape[2], b.shape[1] * b.shape[2], _p(c), N,
                                     M * N, batch, M, N, K, float(alpha), _stream()), "batched_gemm")
    return c


def adam_step(param, grad, exp_avg, exp_avg_sq, shadow, lr, step, lr_scale=None, betas=(0.9, 0.98), eps=1e-8,
              weight_decay=0.0, grad_scale=1.0):
    _dev(param)
    lib = _lib.load()
    assert grad.numel() == param.numel()
    _lib.check(lib.ovqa_adam_step(_p(param), _p(grad), _dt(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), param.numel(),
                                  float(lr), _p(lr_scale), float(betas[0]), float(betas[1]), float(eps),
                                  float(weight_decay), float(grad_scale), _p(step), _stream()), "adam_step")


def adam_step_tiled(param, grad, exp_avg, exp_avg_sq, shadow, shadow_t, tiles, n_tiles, flat_lo, flat_hi, lr, step,
                    lr_scale=None, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """Adam over the whole arena with the transposed bf16 shadow written in the same pass (``tiles``: device table of
    ovqa_adam_tile from ParamArena.adam_tiles())."""
    _dev(param)
    lib = _lib.load()
    assert grad.numel() == param.numel()
    _lib.check(lib.ovqa_adam_step_tiled(_p(param), _p(grad), _dt(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), _p(shadow_t),
                                        _p(tiles), int(n_tiles), int(flat_lo), int(flat_hi), float(lr), _p(lr_scale),
                                        float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                        float(grad_scale), _p(step), _stream()), "adam_step_tiled")


def increment_step(step, also=None):
    """step += 1 (and ``also`` += 1 in the same launch)."""
    _dev(step)
    if also is None:
        _lib.check(_lib.load().ovqa_increment_step(_p(step), _stream()), "increment_step")
    else:
        _lib.check(_lib.load().ovqa_increment_steps(_p(step), _p(also), _stream()), "increment_steps")


def begin_step(step, also, lr_table, lr_out):
    """``ovqa_begin_step``: lr_out = lr_table[step % len(lr_table)]; step += 1; also += 1 (if given)."""
    _dev(step)
    assert lr_table.dtype == torch.float32 and lr_out.dtype == torch.float32 and lr_table.is_contiguous()
    _lib.check(_lib.load().ovqa_begin_step(_p(step), _p(also), _p(lr_table), lr_table.numel(), _p(lr_out), _stream()),
               "begin_step")


def cast(src, dst):
    _dev(src)
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _lib.check(_lib.load().ovqa_cast(_dt(src), _dt(dst), _p(src), _p(dst), src.numel(), _stream()), "cast")
    return dst


def gelu_bwd(dy, u, drop=None):
    """dy * dropout_mask/(1-p) * gelu'(u) (flat dropout index, the forward epilogue's site)."""
    _dev(dy)
    assert dy.is_contiguous() and u.is_contiguous() and dy.dtype == u.dtype and dy.numel() == u.numel()
    du = torch.empty_like(dy)
    _lib.check(_lib.load().ovqa_gelu_bwd(_dt(dy), _p(dy), _p(u), _p(du), dy.numel(), _drop(drop), _stream()),
               "gelu_bwd")
    return du


def row_padding_mask(x, pad_value=0.0):
    """(B, 1, 1, N) additive fp32 mask of the rows of x [B, N, D] whose features sum to pad_value * D."""
    _dev(x)
    assert x.dim() == 3 and x.is_contiguous()
    B, N, D = x.shape
    mask = torch.empty(B, 1, 1, N, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ovqa_row_padding_mask(_dt(x), _p(x), _p(mask), B * N, D, float(pad_value), _stream()),
               "row_padding_mask")
    return mask


def grouped_row_gather(states, selected_beam, b_s, cur_beam, beam, outs=None):
    """Beam reorder of many state buffers in ONE launch: for every tensor s of ``states`` (leading dimension
    b_s * cur_beam) returns a tensor with leading dimension b_s * beam whose row b*beam + j is row
    b*cur_beam + selected_beam[b, j] of s (beam_search.py:19-34).  A state is contiguous, or a live prefix
    ``cache[:, :n]`` of an in-place cache (rows contiguous, uniformly strided); ``outs[i]`` (optional) is the
    destination to fill -- e.g. the prefix of the alternate cache buffer -- instead of a new tensor."""
    import numpy as np
    dev = states[0].device
    _dev(states[0])
    sel = selected_beam.reshape(-1).to(device=dev, dtype=torch.int32).contiguous()
    assert sel.numel() == b_s * beam

    def row_layout(t):  # (bytes per row, row stride in bytes): rows must be dense
        inner = 1
        for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
            assert size == 1 or stride == inner, "rows of a gathered state must be dense"
            inner *= size
        return inner * t.element_size(), (t.stride(0) if t.shape[0] > 1 else inner) * t.element_size()
    res, probs = [], (_lib.GatherProblem * len(states))()
    for i, s in enumerate(states):
        assert s.is_cuda and s.shape[0] == b_s * cur_beam, (tuple(s.shape), b_s, cur_beam)
        o = outs[i] if outs is not None and outs[i] is not None else \
            torch.empty((b_s * beam,) + tuple(s.shape[1:]), dtype=s.dtype, device=dev)
        assert o.shape == (b_s * beam,) + tuple(s.shape[1:]) and o.dtype == s.dtype
        row_bytes, sstride = row_layout(s)
        orow, ostride = row_layout(o)
        assert orow == row_bytes
        probs[i] = _lib.GatherProblem(_p(s), _p(o), row_bytes, sstride, ostride)
        res.append(o)
    # (the table is a HOST array: it travels in the kernel arguments -- no upload, capturable into a hipGraph)
    _lib.check(_lib.load().ovqa_grouped_row_gather(C.cast(probs, C.c_void_p), len(states), _p(sel), b_s, cur_beam, beam,
                                                   _stream()), "grouped_row_gather")
    return res


def dropout_keep_mask(drop: DropSpec, n: int, device) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.uint8, device=device)
    _dev(out)
    d = _lib.Dropout(float(drop.p), drop.seed & 0xFFFFFFFF, drop.site & 0xFFFFFFFF, _p(drop.step))
    _lib.check(_lib.load().ovqa_dropout_keep_mask(C.byref(d), _p(out), n, _stream()), "dropout_keep_mask")
    return out


def sq_loss_fwd_bwd(x, loss, want_grad=True, accumulate=False, target=None):
    """loss (fp32 device scalar) (+)= mean((x-target)^2); returns d loss / d x (same dtype as x)."""
    _dev(x)
    assert x.is_contiguous()
    if target is not None:
        assert target.is_contiguous() and target.dtype == x.dtype and target.numel() == x.numel()
    dx = torch.empty_like(x) if want_grad else None
    _lib.check(_lib.load().ovqa_sq_loss_fwd_bwd(_dt(x), _p(x), _p(target), _p(dx), _p(loss), x.numel(),
                                                int(accumulate), _stream()), "sq_loss")
    return dx


# ---------------------------------------------------------------------------
# LSTM recurrence (ovqa_lstm_fwd / ovqa_lstm_bwd)
def lstm_status(raise_on_error: bool = True) -> int:
    """The process-lifetime status word of the persistent LSTM kernels (ovqa_lstm_status): synchronises the current
    stream, returns the OR of the give-up codes since the last call (0 = every hand-off wait was served) and clears
    it.  A non-zero word means a workgroup of a launch never ran beside the others (its sample group's outputs are NaN
    from that step on): raised as a RuntimeError unless ``raise_on_error`` is False.  Not callable under stream capture."""
    import ctypes as C
    st = C.c_uint32(0)
    _lib.check(_lib.load().ovqa_lstm_status(C.byref(st), _stream()), "lstm_status")
    if st.value and raise_on_error:
        raise RuntimeError(f"persistent LSTM kernel: a hand-off wait gave up (status {st.value}: "
                           f"{'counter wait' if st.value & 1 else ''}{' / ' if st.value == 3 else ''}"
                           f"{'sentinel sweep' if st.value & 2 else ''}); a workgroup of the launch was not resident "
                           "beside the others -- the outputs of its sample group are NaN")
    return int(st.value)


def _lstm_check():
    """Eager calls (no stream capture) read the status word right away: the price is one stream synchronisation per call
    (OVQA_LSTM_CHECK=0 skips it; TrainStep.check_device_status() covers captured steps)."""
    if os.environ.get("OVQA_LSTM_CHECK", "1") != "0" and not torch.cuda.is_current_stream_capturing():
        lstm_status()


class LstmChunks(list):
    """``saved`` of a batch that ran as several (padded) sample-group chunks: [(b0, b1, bp, saved bytes)]."""


def _lstm_chunks(lib, dtype, B, I, H):
    """[(b0, b1, bp)] when the batch has to be padded to whole sample groups of 16 and / or split into chunks the persistent
    kernels can keep co-resident; None = one plain call (fits as it is, or the per-step route anyway)."""
    if dtype != torch.bfloat16 or H != 512 or I != 512 or B < 1:
        return None
    maxb = int(lib.ovqa_lstm_persistent_max_batch())
    if maxb < 16 or (B % 16 == 0 and B <= maxb):
        return None
    out, b0 = [], 0
    while b0 < B:
        b1 = min(B, b0 + maxb)
        out.append((b0, b1, (b1 - b0 + 15) // 16 * 16))
        b0 = b1
    return out


def _lstm_fwd_call(lib, x_tb, w_ih, w_hh, b_ih, b_hh, B, T, want_lp):
    H4, I = w_ih.shape
    H = H4 // 4
    dev = x_tb.device
    y = torch.empty(B, T, H, dtype=torch.float32, device=dev)
    y_lp = torch.empty(B, T, H, dtype=x_tb.dtype, device=dev) if want_lp else None
    hseq = torch.empty((T + 1) * B, H, dtype=x_tb.dtype, device=dev)
    saved = torch.empty(lib.ovqa_lstm_saved_bytes(B, T, H), dtype=torch.uint8, device=dev)
    scratch = torch.empty(lib.ovqa_lstm_scratch_bytes(B, T, H), dtype=torch.uint8, device=dev)
    _lib.check(lib.ovqa_lstm_fwd(_dt(x_tb), _p(x_tb), x_tb.stride(0), _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh), _p(y),
                                 _p(y_lp), _p(hseq), _p(saved), _p(scratch), B, T, I, H, _stream()), "lstm_fwd")
    return y, hseq, saved, scratch, y_lp


def lstm_fwd(x_tb, w_ih, w_hh, b_ih, b_hh, B, T, want_lp=False):
    """x_tb [T*B, I] TIME-MAJOR rows (row t*B + b) -> (y fp32 [B, T, H], hseq [(T+1)*B, H] time-major with block 0 = 0 and
    block t+1 = h_t, saved (opaque, for ``lstm_bwd``), scratch[, y_lp = y in x's dtype with ``want_lp``]).  Gate order
    i, f, g, o; zero initial state.  A bf16 batch that is not a multiple of 16 samples, or larger than the persistent
    kernels can keep co-resident on this device, is padded with zero samples / split into chunks HERE (the padded
    samples' rows are dropped again; their gradients are zero), so that every batch size takes the MFMA route."""
    _dev(x_tb)
    lib = _lib.load()
    H4, I = w_ih.shape
    H = H4 // 4
    assert x_tb.dim() == 2 and x_tb.shape == (T * B, I) and x_tb.stride(1) == 1 and w_hh.shape == (H4, H)
    _weight(w_ih, "lstm_fwd (w_ih)")
    _weight(w_hh, "lstm_fwd (w_hh)")
    assert w_ih.dtype == w_hh.dtype == x_tb.dtype
    assert b_ih.dtype == torch.float32 and b_hh.dtype == torch.float32 and b_ih.numel() == H4 == b_hh.numel()
    chunks = _lstm_chunks(lib, x_tb.dtype, B, I, H)
    if chunks is None:
        y, hseq, saved, scratch, y_lp = _lstm_fwd_call(lib, x_tb, w_ih, w_hh, b_ih, b_hh, B, T, want_lp)
    else:
        dev = x_tb.device
        y = torch.empty(B, T, H, dtype=torch.float32, device=dev)
        y_lp = torch.empty(B, T, H, dtype=x_tb.dtype, device=dev) if want_lp else None
        hseq = torch.empty((T + 1) * B, H, dtype=x_tb.dtype, device=dev)
        x3, h3 = x_tb.view(T, B, I) if x_tb.is_contiguous() else x_tb.reshape(T, B, I), hseq.view(T + 1, B, H)
        saved, scratch = LstmChunks(), None
        for b0, b1, bp in chunks:
            nb = b1 - b0
            xc = torch.zeros(T, bp, I, dtype=x_tb.dtype, device=dev)
            xc[:, :nb] = x3[:, b0:b1]
            yc, hc, sc, scratch, ylc = _lstm_fwd_call(lib, xc.view(T * bp, I), w_ih, w_hh, b_ih, b_hh, bp, T, want_lp)
            y[b0:b1] = yc[:nb]
            h3[:, b0:b1] = hc.view(T + 1, bp, H)[:, :nb]
            if want_lp:
                y_lp[b0:b1] = ylc[:nb]
            saved.append((b0, b1, bp, sc))
    _lstm_check()
    return (y, hseq, saved, scratch, y_lp) if want_lp else (y, hseq, saved, scratch)


def _lstm_bwd_call(lib, dy, w_hh, w_hh_t, ldwt, saved, B, T, I):
    H4, H = w_hh.shape
    dgates = torch.empty(T * B, H4, dtype=w_hh.dtype, device=dy.device)
    scratch = torch.empty(lib.ovqa_lstm_scratch_bytes(B, T, H), dtype=torch.uint8, device=dy.device)
    _lib.check(lib.ovqa_lstm_bwd(_dt(w_hh), _p(dy), _dt(dy), _p(w_hh), _p(w_hh_t), ldwt, _p(saved), _p(dgates), _p(scratch), B, T,
                                 I, H, _stream()), "lstm_bwd")
    return dgates, scratch


def lstm_bwd(dy, w_hh, w_hh_t, saved, B, T, I):
    """dy [B, T, H] (fp32 or bf16) -> dgates [T*B, 4H] (time-major rows, columns gate*H + unit) in w_hh's dtype; ``w_hh_t`` =
    the transposed bf16 copy [H, 4H] (rows may be strided) or None in fp32 mode.  ``saved`` as ``lstm_fwd`` returned it (the
    chunks of a padded / split batch included)."""
    _dev(dy)
    lib = _lib.load()
    H4, H = w_hh.shape
    assert dy.is_contiguous() and dy.shape == (B, T, H) and w_hh.is_contiguous()
    ldwt = 0
    if w_hh_t is not None:
        assert w_hh_t.shape == (H, H4) and w_hh_t.stride(1) == 1 and w_hh_t.dtype == w_hh.dtype
        ldwt = w_hh_t.stride(0)
    if not isinstance(saved, LstmChunks):
        out = _lstm_bwd_call(lib, dy, w_hh, w_hh_t, ldwt, saved, B, T, I)
    else:
        dgates = torch.empty(T * B, H4, dtype=w_hh.dtype, device=dy.device)
        g3, scratch = dgates.view(T, B, H4), None
        for b0, b1, bp, sc in saved:
            nb = b1 - b0
            dyc = torch.zeros(bp, T, H, dtype=dy.dtype, device=dy.device)
            dyc[:nb] = dy[b0:b1]
            gc, scratch = _lstm_bwd_call(lib, dyc, w_hh, w_hh_t, ldwt, sc, bp, T, I)
            g3[:, b0:b1] = gc.view(T, bp, H4)[:, :nb]
        out = (dgates, scratch)
    _lstm_check()
    return out


# ---------------------------------------------------------------------------
# The two ends of the model (csrc/model_ends.hip)
def embed_gather(tokens, table, time_major=False, want_mask=False, padding_idx=-1):
    """rows [B*T, W] = table[tokens] in the table's dtype (W = table.shape[1], zero padding of a ragged table included); row
    r = t*B + b (time_major) or b*T + t.  ``want_mask``: also the (B, 1, 1, T) additive padding mask of the token ids."""
    _dev(table)
    assert tokens.dtype == torch.int64 and tokens.dim() == 2 and tokens.is_contiguous() and tokens.is_cuda
    assert table.dim() == 2 and table.stride(1) == 1
    B, T = tokens.shape
    rows = torch.empty(B * T, table.shape[1], dtype=table.dtype, device=table.device)
    mask = torch.empty(B, 1, 1, T, dtype=torch.float32, device=table.device) if want_mask else None
    _lib.check(_lib.load().ovqa_embed_gather(_dt(table), _p(tokens), _p(table), table.stride(0), table.shape[0], _p(rows),
                                             rows.stride(0), B, T, table.shape[1], int(time_major), _p(mask), int(padding_idx),
                                             _stream()), "embed_gather")
    return (rows, mask) if want_mask else rows


def embed_scatter(tokens, drows, dtable, time_major=False, padding_idx=-1, accumulate=False):
    """dtable (fp32 [V, W], every row) (=|+=) the sum of drows [B*T, W] per token, in row order (deterministic)."""
    _dev(drows)
    B, T = tokens.shape
    assert drows.dim() == 2 and drows.shape[0] == B * T and drows.stride(1) == 1 and dtable.dtype == torch.float32
    assert dtable.dim() == 2 and dtable.stride(1) == 1 and dtable.shape[1] == drows.shape[1] and tokens.is_contiguous()
    _lib.check(_lib.load().ovqa_embed_scatter(_dt(drows), _p(tokens), _p(drows), drows.stride(0), _p(dtable),
                                              dtable.stride(0), dtable.shape[0], B, T, drows.shape[1], int(time_major),
                                              int(padding_idx), int(bool(accumulate)), _stream()), "embed_scatter")


def decoder_inputs(tokens, emb, pos_table, padding_idx):
    """(emb + pos_table[seq], self_mask [B, 1, T, T]) of a teacher-forced decoder pass in one launch
    (``ovqa_decoder_inputs``; decoders.py:50-60,66).  tokens int64 [B, T]; emb fp32 [B, T, D]; pos_table fp32 [>= T + 1, D]."""
    _dev(emb)
    B, T = tokens.shape
    D = emb.shape[-1]
    assert emb.dtype == pos_table.dtype == torch.float32 and emb.is_contiguous() and pos_table.is_contiguous()
    assert emb.shape == (B, T, D) and pos_table.shape[1] == D and tokens.is_contiguous() and tokens.dtype == torch.int64
    out = torch.empty_like(emb)
    mask = torch.empty(B, 1, T, T, dtype=torch.float32, device=emb.device)
    _lib.check(_lib.load().ovqa_decoder_inputs(_p(tokens), _p(emb), _p(pos_table), pos_table.shape[0], _p(out), _p(mask),
                                               B, T, D, int(padding_idx), _stream()), "decoder_inputs")
    return out, mask


def dropout_apply(x, drop):
    """x * keep / (1 - p) with the counter-hash mask of ``drop`` (flat element index); x itself when dropout is off."""
    if drop is None or drop.p <= 0.0:
        return x
    _dev(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.load().ovqa_dropout_apply(_dt(x), _p(x), _p(y), x.numel(), _drop(drop), _stream()), "dropout_apply")
    return y


def pool_fwd(feat, hpre, w2, b2, drop=None):
    """Attention pooling forward (``ovqa_pool_fwd``): feat [B, N, D] (fp32 or the compute dtype), hpre [B*N, D] = fc1(feat);
    w2 fp32 [D], b2 fp32 [1] or None -> (att fp32 [B, N], pooled [B, D] in hpre's dtype)."""
    _dev(feat)
    B, N, D = feat.shape
    assert feat.is_contiguous() and hpre.is_contiguous() and hpre.numel() == B * N * D
    assert w2.dtype == torch.float32 and w2.numel() == D and w2.is_contiguous()
    att = torch.empty(B, N, dtype=torch.float32, device=feat.device)
    pooled = torch.empty(B, D, dtype=hpre.dtype, device=feat.device)
    _lib.check(_lib.load().ovqa_pool_fwd(_dt(feat), _dt(hpre), _p(feat), _p(hpre), _p(w2), _p(b2), _p(att), _p(pooled), None,
                                         B, N, D, _drop(drop), _stream()), "pool_fwd")
    return att, pooled


def pool_bwd(feat, hpre, w2, att, dpooled, drop=None):
    """-> (dh [B*N, D], dfeat [B*N, D] (= att * dpooled), dw2_part fp32 [B, 2*D], db2_part fp32 [B, 16]): the two partial
    blocks are rows for the grouped partial reduce (fc2's weight gradient in columns 0..D, its bias gradient in column 0)."""
    _dev(feat)
    B, N, D = feat.shape
    assert dpooled.is_contiguous() and dpooled.dtype == hpre.dtype and dpooled.numel() == B * D
    dh = torch.empty(B * N, D, dtype=hpre.dtype, device=feat.device)
    dfeat = torch.empty(B * N, D, dtype=hpre.dtype, device=feat.device)
    part = torch.empty(B, 2 * D, dtype=torch.float32, device=feat.device)
    bpart = torch.empty(B, 16, dtype=torch.float32, device=feat.device)
    _lib.check(_lib.load().ovqa_pool_bwd(_dt(feat), _dt(hpre), _p(feat), _p(hpre), _p(w2), _p(att), _p(dpooled), _p(dh),
                                         _p(dfeat), _p(part), _p(bpart), B, N, D, _drop(drop), _stream()), "pool_bwd")
    return dh, dfeat, part, bpart


def log_softmax_fwd(x, n=None):
    """fp32 [M, n] = log_softmax over the first n columns of every row of x [M, ld]."""
    _dev(x)
    assert x.dim() == 2 and x.stride(1) == 1
    n = x.shape[1] if n is None else n
    out = torch.empty(x.shape[0], n, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ovqa_log_softmax_fwd(_dt(x), _p(x), x.stride(0), _p(out), x.shape[0], n, _stream()),
               "log_softmax_fwd")
    return out


def log_softmax_bwd(g, logp, ld, dtype):
    """dx [M, ld] of ``dtype`` = g - exp(logp) * rowsum(g) in the first n columns, zeros in the padded ones."""
    _dev(g)
    assert g.dtype == torch.float32 and logp.dtype == torch.float32 and g.is_contiguous() and logp.is_contiguous()
    M, n = logp.shape
    dx = torch.empty(M, ld, dtype=dtype, device=g.device)
    _lib.check(_lib.load().ovqa_log_softmax_bwd(_DT[dtype], _p(g), _p(logp), _p(dx), ld, M, n, _stream()), "log_softmax_bwd")
    return dx


def nll_loss(logp, target, ignore_index=-100, loss=None, want_grad=False, gscale=None, accumulate=False):
    """NLLLoss (mean over the rows whose target != ignore_index) of fp32 log-probabilities [M, n]: writes ``loss`` (fp32
    device scalar, (=|+=)) if given and returns the dense gradient [M, n] (scaled by the device scalar ``gscale``) if asked."""
    _dev(logp)
    assert logp.dtype == torch.float32 and logp.dim() == 2 and logp.is_contiguous()
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == logp.shape[0]
    dlogp = torch.empty_like(logp) if want_grad else None
    _lib.check(_lib.load().ovqa_nll_loss(_p(logp), _p(target), _p(loss), _p(dlogp), _p(gscale), logp.shape[0], logp.shape[1],
                                         int(ignore_index), int(bool(accumulate)), _stream()), "nll_loss")
    return dlogp
