"""Host-side runtime: compute dtype, flat parameter arenas, dropout bookkeeping.

Data layout in HBM:
  * every block's parameters live in one flat fp32 *master* buffer; the
    ``nn.Parameter`` objects keep their reference names/shapes (checkpoint
    contract) but their ``.data`` are views into it;
  * fc_q/fc_k/fc_v weights (and biases) of an attention are adjacent, so the
    fused QKV projection reads ONE [3*H*dk, D] matrix without any copy;
  * a same-layout fp32 *grad* buffer receives weight gradients straight from the
    kernels (one flat all-reduce for data parallelism);
  * in bf16 mode a same-layout bf16 *shadow* is what the MFMA kernels read; the
    fused Adam kernel rewrites master and shadow together.
"""
from __future__ import annotations

import itertools
from typing import Dict, Iterable, List, Optional, Sequence

import torch
from torch import nn

from . import ops

_state = {"compute_dtype": torch.bfloat16, "seed": 0x5EED1234, "call": 0}
_site_counter = itertools.count(1)
_step_tensors: Dict[int, torch.Tensor] = {}

ALIGN = 64  # elements; keeps every group start 256-byte aligned in fp32 and 128-byte in bf16
PAD = 8     # ragged parameters are laid out with rows / columns (or length) rounded up to this: 16-byte bf16 rows


def _footprint(p) -> tuple:
    """Shape a parameter occupies in the arena: a matrix whose rows or columns are not multiples of 8 (a 353-way
    classifier, a 300-wide word embedding: mcan.yaml) gets zero padding up to the next multiple, so that the MFMA GEMMs,
    the direct-to-LDS weight gradients and the tiled Adam see 16-byte aligned rows; ``p.data`` is the [:rows, :cols] view
    (same name and shape for state_dict), the padding holds zeros for ever (zero gradient, zero Adam update)."""
    if p.dim() == 2:
        rows, cols = p.shape
        # a ragged reduction length above one K step goes up to whole 64-deep steps (300 -> 320): the direct-to-LDS
        # GEMM forms take over from the register-staged one (13 + 16 us -> 6 + 9 for the 1280 x 512 x 300 products)
        cpad = 64 if cols % PAD and cols > 64 else PAD
        # ... and so does a LONG output dimension (a 4000-word vocabulary projection, decoders.py:44: 4000 -> 4032): it is
        # the reduction length of that layer's dX product (76 -> 12 us at 1280 positions on the direct-to-LDS form)
        rpad = 64 if rows % 64 and rows > 1024 else PAD
        return ((rows + rpad - 1) // rpad * rpad, (cols + cpad - 1) // cpad * cpad)
    if p.dim() == 1:
        # (the same rule as the rows of a matrix: the bias of a padded layer has to be as long as its weight is high)
        n = p.shape[0]
        pad = 64 if n % 64 and n > 1024 else PAD
        return ((n + pad - 1) // pad * pad,)
    return tuple(p.shape)


def set_compute_dtype(dtype: torch.dtype) -> None:
    """torch.float32: exact-fp32 HIP kernels (parity <= 1e-3 vs the reference);
    torch.bfloat16 (default): bf16 storage + MFMA kernels (parity <= 1e-2)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _state["compute_dtype"] = dtype


def get_compute_dtype() -> torch.dtype:
    return _state["compute_dtype"]


def manual_seed(seed: int) -> None:
    _state["seed"] = int(seed) & 0xFFFFFFFF
    _state["call"] = 0


def capture_error_mode() -> str:
    """`capture_error_mode` for torch.cuda.graph: "thread_local" while a torch.distributed process group is alive.  Its
    watchdog thread polls the events of eagerly issued collectives every ~100 ms until it has seen them complete; such a
    poll during a capture in the default GLOBAL mode invalidates the capture and terminates the process from the watchdog
    thread (measured: a collective issued just before a capture is enough)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return "thread_local"
    except Exception:  # noqa: BLE001
        pass
    return "global"


def new_dropout_site() -> int:
    return next(_site_counter)


def step_tensor(device: torch.device) -> torch.Tensor:
    """Device-resident uint32 step counter mixed into every dropout key (lets a
    captured graph draw fresh masks on each replay)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _step_tensors.get(idx)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _step_tensors[idx] = t
    return t


def dropout_spec(p: float, site: int, training: bool, device) -> Optional[ops.DropSpec]:
    if not training or p <= 0.0:
        return None
    _state["call"] += 1
    seed = (_state["seed"] + 0x9E3779B9 * _state["call"]) & 0xFFFFFFFF
    return ops.DropSpec(p=float(p), seed=seed, site=site, step=step_tensor(device))


# ---- backward milestones -------------------------------------------------------------------------------
# A training harness that overlaps the data-parallel gradient exchange with the rest of backward needs
# points at which "everything above here has been differentiated".  Stacks mark their layer boundaries
# with grad_milestone(x); it is the identity unless a harness installed a sink (train.TrainStep with
# world_size > 1), which may cut the autograd graph there and run backward in phases.
_milestone_sink = None


def set_milestone_sink(sink) -> None:
    global _milestone_sink
    _milestone_sink = sink


def grad_milestone(x: torch.Tensor, barrier: bool = False) -> torch.Tensor:
    """Mark ``x`` as a point where backward may be split: everything computed from the returned tensor is
    differentiated before anything ``x`` was computed from.  ``barrier=True`` is for a tensor consumed by
    SEVERAL later milestone-delimited sections (the question features every guided layer attends to): the
    graph must be cut here whenever it is cut at any later milestone."""
    sink = _milestone_sink
    if sink is None or not torch.is_grad_enabled() or not x.requires_grad:
        return x
    return sink.cut(x, barrier)


class ParamArena:
    """Flat fp32 master / fp32 grad / (bf16 shadow) buffers for a set of parameters."""

    def __init__(self, groups: Sequence[Sequence[nn.Parameter]], device: torch.device, compute_dtype: torch.dtype):
        self.device = device
        self.compute_dtype = compute_dtype
        self.params: List[nn.Parameter] = []
        self.offsets: Dict[int, int] = {}
        off = 0
        # matrices first, then every 1-D parameter (biases, LayerNorm affine): the 1-D tail is what the
        # training harness zeroes with ONE memset per step (their gradients are reduced with atomics)
        groups = sorted(groups, key=lambda g: 0 if g[0].dim() >= 2 else 1)
        self.small_lo = None
        self._group_of: Dict[int, tuple] = {}  # id(param) -> (group offset, first row, group rows, cols) for 2-D groups
        self._groups2d: List[tuple] = []
        self.foot: Dict[int, tuple] = {}  # id(param) -> padded shape in the arena (== p.shape unless ragged)
        for g in groups:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            if g[0].dim() < 2 and self.small_lo is None:
                self.small_lo = off
            # only a group of its own is padded: members of an adjacency group ([fc_q | fc_k | fc_v]) must stay dense
            foots = [_footprint(p) if len(g) == 1 else tuple(p.shape) for p in g]
            if g[0].dim() == 2 and all(p.dim() == 2 and f[1] == foots[0][1] for p, f in zip(g, foots)):
                rows, r0 = sum(f[0] for f in foots), 0
                for p, f in zip(g, foots):
                    self._group_of[id(p)] = (off, r0, rows, foots[0][1])
                    r0 += f[0]
                self._groups2d.append((off, rows, foots[0][1]))
            for p, f in zip(g, foots):
                if id(p) in self.offsets:
                    raise RuntimeError("parameter appears twice in an arena")
                self.offsets[id(p)] = off
                self.foot[id(p)] = f
                self.params.append(p)
                n = 1
                for d in f:
                    n *= d
                off += n
        self.numel = (off + ALIGN - 1) // ALIGN * ALIGN
        if self.small_lo is None:
            self.small_lo = self.numel
        self.master = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.shadow = (torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
                       if compute_dtype == torch.bfloat16 else None)
        # transposed bf16 copy of every matrix GROUP ([cols, group rows] at the group's offset): what the dX GEMMs
        # read (a row-major weight tile; the [N, K] form must be staged k-major, +0.44 ms per MCAN step)
        self.shadow_t = (torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
                         if compute_dtype == torch.bfloat16 and device.type == "cuda" and self._groups2d else None)
        self._tr_table = None
        self.overwrite_grads = False  # harness mode: backward always overwrites the grad buffer
        self.kernel_written = set()   # ids of parameters whose gradient the HIP kernels produce
        self._written_pass = set()    # harness mode: ids written so far in the CURRENT backward pass
        with torch.no_grad():
            for p in self.params:
                view = self._slice(self.master, p)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p._ovqa_arena = self
        self._versions = None
        self.refresh_shadow()

    def layout(self, names: Optional[Dict[int, str]] = None) -> list:
        """Signature of the flat layout: ``[key, shape, offset]`` per parameter in arena order, key = the parameter's
        name when ``names`` (id(param) -> name) is given, else its position.  A checkpoint of flat moments is only
        meaningful for the layout it was written with (FlatAdam.load_state_dict compares)."""
        return [[names.get(id(p), str(i)) if names else str(i), list(p.shape), int(self.offsets[id(p)])]
                for i, p in enumerate(self.params)]

    # -- views -------------------------------------------------------------
    def span(self, p) -> int:
        """Elements ``p`` occupies in the arena (its padded footprint)."""
        n = 1
        for d in self.foot[id(p)]:
            n *= d
        return n

    def _slice(self, buf, p):
        """View of ``buf`` with p's shape: dense, or the [:rows, :cols] corner of the padded footprint."""
        o, f = self.offsets[id(p)], self.foot[id(p)]
        if f == tuple(p.shape):
            return buf[o:o + p.numel()].view(p.shape)
        full = buf[o:o + self.span(p)].view(f)
        return full[:p.shape[0], :p.shape[1]] if p.dim() == 2 else full[:p.shape[0]]

    def is_padded(self, p) -> bool:
        return self.foot[id(p)] != tuple(p.shape)

    def padded(self, p, buf: str = "compute") -> torch.Tensor:
        """The WHOLE footprint of ``p`` (contiguous [rows_p, cols_p] or [n_p], zeros beyond p's own shape): what the GEMM
        kernels take as the weight / bias of a ragged layer."""
        base = {"compute": self.shadow if self.shadow is not None else self.master, "master": self.master,
                "grad": self.grad}[buf]
        o = self.offsets[id(p)]
        return base[o:o + self.span(p)].view(self.foot[id(p)])

    def compute(self, p) -> torch.Tensor:
        """Weight in the compute dtype (bf16 shadow or the fp32 master itself)."""
        return self._slice(self.shadow if self.shadow is not None else self.master, p)

    def master_of(self, p) -> torch.Tensor:
        return self._slice(self.master, p)

    def grad_of(self, p) -> torch.Tensor:
        return self._slice(self.grad, p)

    def packed(self, ps: Sequence[nn.Parameter], buf: str = "compute") -> torch.Tensor:
        """One [sum(rows), cols] (or [sum(n)]) view over adjacent parameters."""
        if len(ps) == 1:
            return self.padded(ps[0], buf)  # (a ragged parameter: its whole zero-padded footprint)
        o0 = self.offsets[id(ps[0])]
        n, o = 0, o0
        for p in ps:
            if self.offsets[id(p)] != o or self.is_padded(p):
                raise RuntimeError("parameters are not adjacent in the arena")
            o += p.numel()
            n += p.numel()
        base = {"compute": self.shadow if self.shadow is not None else self.master,
                "master": self.master, "grad": self.grad}[buf]
        flat = base[o0:o0 + n]
        if ps[0].dim() == 2:
            return flat.view(n // ps[0].shape[1], ps[0].shape[1])
        return flat

    # -- consistency -------------------------------------------------------
    def owns(self, ps: Iterable[nn.Parameter]) -> bool:
        for p in ps:
            if getattr(p, "_ovqa_arena", None) is not self or id(p) not in self.offsets:
                return False
            o = self.offsets[id(p)]
            if p.data.data_ptr() != self.master.data_ptr() + 4 * o or p.device != self.master.device:
                return False
        return True

    def refresh_shadow(self) -> None:
        if self.shadow is not None:
            ops.cast(self.master, self.shadow)
            self.refresh_transposed()
        self._versions = [p._version for p in self.params]

    def refresh_transposed(self) -> None:
        """shadow_t <- transpose of every matrix group of the bf16 shadow: one grouped launch."""
        if self.shadow_t is None:
            return
        if self._tr_table is None:
            import numpy as np
            from . import _lib
            probs = (_lib.TransposeProblem * len(self._groups2d))()
            for i, (off, rows, cols) in enumerate(self._groups2d):
                probs[i] = _lib.TransposeProblem(self.shadow.data_ptr() + 2 * off, self.shadow_t.data_ptr() + 2 * off,
                                                 cols, rows, rows, cols)
            raw = torch.from_numpy(np.frombuffer(bytes(probs), dtype=np.uint8).copy())
            tiles = max(((r + 63) // 64) * ((c + 63) // 64) for _, r, c in self._groups2d)
            self._tr_table = (raw.to(self.device), len(self._groups2d), tiles)
        ops.grouped_transpose(*self._tr_table)

    def adam_tiles(self):
        """(device table of ovqa_adam_tile, number of tiles, flat_lo, flat_hi) for ops.adam_step_tiled, or None when the
        arena does not have the shape that kernel assumes: every element in front of the 1-D tail belongs to a matrix
        group (up to alignment padding) whose rows and columns are multiples of 8."""
        if self.shadow_t is None:
            return None
        if getattr(self, "_adam_tiles", None) is None:
            self._adam_tiles = False
            end = 0
            ok = True
            for off, rows, cols in sorted(self._groups2d):
                ok = ok and off == (end + ALIGN - 1) // ALIGN * ALIGN and rows % 8 == 0 and cols % 8 == 0
                end = off + rows * cols
            ok = ok and (end + ALIGN - 1) // ALIGN * ALIGN == self.small_lo
            if ok:
                import numpy as np
                from . import _lib
                entries = []
                for off, rows, cols in sorted(self._groups2d):
                    for r0 in range(0, rows, 64):
                        for c0 in range(0, cols, 64):
                            entries.append(_lib.AdamTile(off, rows, cols, r0, c0))
                arr = (_lib.AdamTile * len(entries))(*entries)
                raw = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
                self._adam_tiles = (raw.to(self.device), len(entries), self.small_lo, self.numel)
                # per matrix group: (offset, end, first tile, number of tiles) -- for updates of a sub-range of the arena
                self._adam_groups, t0 = [], 0
                for off, rows, cols in sorted(self._groups2d):
                    nt = ((rows + 63) // 64) * ((cols + 63) // 64)
                    self._adam_groups.append((off, off + rows * cols, t0, nt))
                    t0 += nt
        return self._adam_tiles or None

    def adam_tiles_of(self, group_offsets):
        """(device table, number of tiles) of the matrix groups that START at the given arena offsets -- any subset, in one
        table (the entries are self-describing), cached per subset: what is left for the separate Adam launch when the
        optimiser step of the other groups ran inside the weight-gradient launch (train._FusedAdam)."""
        if self.adam_tiles() is None:
            return None
        key = tuple(sorted(group_offsets))
        cache = self.__dict__.setdefault("_adam_subsets", {})
        if key not in cache:
            import numpy as np
            from . import _lib
            entries = []
            for off, rows, cols in sorted(self._groups2d):
                if off in key:
                    for r0 in range(0, rows, 64):
                        for c0 in range(0, cols, 64):
                            entries.append(_lib.AdamTile(off, rows, cols, r0, c0))
            if entries:
                arr = (_lib.AdamTile * len(entries))(*entries)
                raw = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
            else:
                raw = torch.empty(0, dtype=torch.uint8, device=self.device)
            cache[key] = (raw, len(entries))
        return cache[key]

    def adam_tiles_in(self, lo: int, hi: int):
        """The part of ``adam_tiles()`` inside the arena range [lo, hi): (table view, number of tiles, flat_lo, flat_hi),
        or None when a matrix group straddles the range's bounds (or there is no table)."""
        full = self.adam_tiles()
        if full is None:
            return None
        table, _, small_lo, numel = full
        inside = [g for g in self._adam_groups if g[0] < hi and g[1] > lo]
        if any(g[0] < lo or g[1] > hi for g in inside):
            return None
        import ctypes
        from . import _lib
        sz = ctypes.sizeof(_lib.AdamTile)
        if inside:
            t0, t1 = inside[0][2], inside[-1][2] + inside[-1][3]
            if t1 - t0 != sum(g[3] for g in inside):
                return None
        else:
            t0 = t1 = 0
        flat_lo, flat_hi = max(lo, small_lo), min(hi, numel)
        if flat_hi <= flat_lo:
            flat_lo = flat_hi = 0
        if flat_lo % 4 or flat_hi % 4:
            return None
        return table[t0 * sz:t1 * sz] if t1 > t0 else table[:0], t1 - t0, flat_lo, flat_hi

    def transposed(self, ps: Sequence[nn.Parameter]):
        """[cols, sum(rows)] bf16 view (row stride = rows of the whole adjacency group) of the transposed copy of
        the adjacent matrices ``ps``, or None when there is none (fp32 mode, CPU, parameters of different groups)."""
        if self.shadow_t is None:
            return None
        info = [self._group_of.get(id(p)) for p in ps]
        if any(i is None for i in info) or any(i[0] != info[0][0] for i in info):
            return None
        off, r0, rows, cols = info[0]
        r = r0
        for p, i in zip(ps, info):
            if i[1] != r:
                return None
            r += self.foot[id(p)][0]
        return self.shadow_t[off:off + rows * cols].view(cols, rows)[:, r0:r]

    def sync_if_stale(self) -> None:
        """Re-cast the bf16 shadow if torch mutated a parameter in place
        (load_state_dict, a torch optimiser step...)."""
        if self.shadow is None:
            return
        if self._versions != [p._version for p in self.params]:
            self.refresh_shadow()

    def foreign_ranges(self):
        """Merged [start, end) element ranges of the grad buffer NOT written by the kernels during the
        last backward: parameters that live in plain torch modules (autograd accumulates into their
        ``.grad``) or that receive no gradient at all (dead cross-attention).  The training
        harness zeroes exactly these before each backward; everything else is overwritten."""
        spans = sorted((self.offsets[id(p)], self.offsets[id(p)] + self.span(p)) for p in self.params
                       if id(p) not in self.kernel_written)
        merged = []
        for s, e in spans:
            if merged and s <= merged[-1][1] + ALIGN:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        return [(s, e) for s, e in merged]

    def begin_backward_pass(self) -> None:
        """Harness mode: a new forward+backward starts (every matrix gradient is overwritten by its first product)."""
        self._written_pass = set()

    def end_backward_pass(self) -> None:
        """Harness mode: zero the gradient of every kernel-owned parameter that this pass did NOT write (a branch that
        was skipped this time), so that a stale gradient of an earlier step never reaches the optimiser."""
        for p in self.params:
            if id(p) in self.kernel_written and id(p) not in self._written_pass:
                self.grad_of(p).zero_()

    def attach_grads(self) -> None:
        for p in self.params:
            v = self.grad_of(p)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def grad_views(self, ps: Sequence[nn.Parameter]):
        """(packed fp32 grad view over ``ps``, accumulate flag) for a backward pass,
        following autograd's convention: a parameter whose ``.grad`` is None gets its
        gradient written, an existing ``.grad`` is accumulated into."""
        ps = list(ps)
        views = [self.grad_of(p) for p in ps]
        self.kernel_written.update(id(p) for p in ps)
        if self.overwrite_grads:
            # harness mode: the FIRST product of a pass overwrites a gradient, later products of the same pass (a
            # weight applied twice in one forward) accumulate.  Round 4: the 1-D tail (biases, LayerNorm parameters)
            # follows the same rule -- its producers store deterministically (fused column sums of the grouped dW,
            # the fixed-order LayerNorm reduce) instead of adding atomically into a region zeroed once per step
            for p, v in zip(ps, views):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
            seen = [id(p) in self._written_pass for p in ps]
            if any(seen) and not all(seen):  # a packed group of which only some members were written before
                for p, v, was in zip(ps, views, seen):
                    if not was:
                        v.zero_()
            self._written_pass.update(id(p) for p in ps)
            return self.packed(ps, "grad"), any(seen)
        if all(p.grad is None for p in ps):
            for p, v in zip(ps, views):
                p.grad = v
            return self.packed(ps, "grad"), False
        for p, v in zip(ps, views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        return self.packed(ps, "grad"), True


def collect_groups(module: nn.Module) -> List[List[nn.Parameter]]:
    """Adjacency groups first (declared by modules via ``_ovqa_param_groups``), then the rest."""
    order = {id(p): i for i, p in enumerate(module.parameters())}
    seen, groups, key = set(), [], {}
    for m in module.modules():
        fn = getattr(m, "_ovqa_param_groups", None)
        if fn is None:
            continue
        first = next(iter(m.parameters()), None)
        for g in fn():
            g = [p for p in g if id(p) not in seen]
            if g:
                groups.append(g)
                seen.update(id(p) for p in g)
                # a group declared by a CONTAINER (the guided stack's hoisted K/V weights) is used where the
                # container starts, i.e. before any of its layers
                key[id(g[0])] = min(order[id(g[0])], order[id(first)] if first is not None else order[id(g[0])])
    for p in module.parameters():
        if id(p) not in seen:
            groups.append([p])
            seen.add(id(p))
            key[id(p)] = order[id(p)]
    # layer-contiguous layout in REVERSE module order: backward differentiates the last modules first, so "the
    # gradients that are final after the first k sections of backward" is a PREFIX of the matrix region, and the
    # segment that is final last (the first modules of the model) sits right in front of the 1-D tail -- every
    # segment of the overlapped data-parallel exchange is one contiguous range (one cast, one collective)
    groups.sort(key=lambda g: -key[id(g[0])])
    return groups


def build_arena(module: nn.Module, device=None, compute_dtype=None) -> ParamArena:
    params = list(module.parameters())
    if not params:
        raise RuntimeError("module has no parameters")
    device = torch.device(device) if device is not None else params[0].device
    if device.type != "cuda":
        raise RuntimeError("openvivqa_amd: parameters must be on an AMD GPU ('cuda' device) before the forward pass; "
                           "there is no CPU fallback")
    return ParamArena(collect_groups(module), device, compute_dtype or get_compute_dtype())


def ensure_arena(block: nn.Module) -> ParamArena:
    """Arena holding ``block``'s parameters (lazily created per block; a call to
    ``prepare(model)`` replaces the per-block arenas by one model-wide arena)."""
    params = list(block.parameters())
    arena = getattr(params[0], "_ovqa_arena", None)
    if arena is None or arena.compute_dtype != get_compute_dtype() or not arena.owns(params):
        arena = build_arena(block)
    arena.sync_if_stale()
    return arena


def prepare(model: nn.Module, device=None, compute_dtype=None) -> ParamArena:
    """Put ALL parameters of ``model`` in one arena (one flat gradient buffer for
    the data-parallel all-reduce and one fused optimiser launch)."""
    if device is not None:
        model.to(device)
    if compute_dtype is not None:
        set_compute_dtype(compute_dtype)
    arena = build_arena(model)
    model._ovqa_arena = arena
    return arena
