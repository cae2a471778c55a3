"""Region / grid / OCR features from the reference's on-disk layout straight into HBM (second half).

The reference stores one pickled dict of arrays per image (``{image_id}.npy``, read with
``np.load(path, allow_pickle=True)[()]``: data_utils/datasets/base_dataset.py:27-34) and collates a batch by zero-padding
every array field to the longest sample and concatenating (utils/instance.py:31-54,155-170).  ``FeatureCollator`` does the
same into REUSED pinned host buffers and copies them to the device on a side stream, two buffers deep, so that the staging
of batch i + 1 overlaps the step of batch i and a hipGraph-captured step can read fixed device addresses (``pad_to``
fixes the padded length as well; the zero rows it adds are padding positions to ``FeatureEmbedding``'s row mask, exactly
like the reference's own padding: models/utils.py:44-58).

Host-side plumbing in front of the hot path: numpy memcpy + one asynchronous H2D copy per field and batch.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def load_features(path: str) -> Dict[str, object]:
    """The dict stored in an ``{image_id}.npy`` file (base_dataset.py:27-34); arrays stay numpy (no per-sample tensor)."""
    obj = np.load(path, allow_pickle=True)[()]
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a pickled dict of features, got {type(obj).__name__}")
    return obj


class FeatureCollator:
    """``collate(samples) -> {key: device tensor (B, L, D)}`` for the array fields named in ``keys``.

    L = the longest sample of the batch (the reference's padding, utils/instance.py:155-170) or ``pad_to``; shorter
    samples are zero-padded at the end.  ``depth`` buffers per key, reused strictly alternately and grown to the largest
    batch seen, host buffers pinned when the target is a GPU: the tensors of ``depth`` consecutive collates are distinct
    memory, whatever their shapes.  ``collate`` returns after QUEUING the copies on its side
    stream; ``wait()`` makes the current stream wait for them (call it before the step that consumes the batch).  A copy
    starts once the work queued on the current stream AT THE TIME OF THE CALL is done (that work may still read the
    device buffer being refilled), so to overlap staging with compute collate batch i + 1 BEFORE queuing step i:

        nxt = col.collate(samples[i + 1]); step(cur); col.wait(); cur = nxt"""

    def __init__(self, keys: Sequence[str], device, pad_to: Optional[Dict[str, int]] = None, dtype=torch.float32,
                 depth: int = 2):
        self.keys, self.device = list(keys), torch.device(device)
        self.pad_to = dict(pad_to or {})
        self.dtype, self.depth = dtype, depth
        self._slots: Dict[str, list] = {}   # key -> `depth` slots [flat host staging, flat device buffer, last-copy event]
        self._next: Dict[str, int] = {}     # key -> slot of its NEXT collate (round-robin per key, whatever the shapes)
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._event = None

    def _buffers(self, key, shape):
        """Slot of this collate for ``key``: strictly alternating per key, so that the tensors of two consecutive
        collates never share memory whatever their shapes (a slot index shared by all keys and shapes, with slots created
        on first use, could hand two consecutive batches of one shape the same buffer).  A slot is ONE flat buffer per
        side, sized for the largest batch it has held and viewed as (B, L, D): memory is bounded by depth x the largest
        batch per key instead of growing with every distinct padded length."""
        slots = self._slots.setdefault(key, [None] * self.depth)
        i = self._next.get(key, 0)
        self._next[key] = (i + 1) % self.depth
        numel = int(np.prod(shape))
        slot = slots[i]
        if slot is not None and slot[2] is not None:
            slot[2].synchronize()  # the H2D copy that last read this host buffer is done before the CPU rewrites it
        if slot is None or slot[0].numel() < numel:
            host = torch.zeros(numel, dtype=self.dtype, pin_memory=self.device.type == "cuda")
            dev = host if self.device.type == "cpu" else torch.empty(numel, dtype=self.dtype, device=self.device)
            slot = slots[i] = [host, dev, None]
        return slot, slot[0][:numel].view(shape), slot[1][:numel].view(shape)

    def collate(self, samples: List[Dict[str, object]]) -> Dict[str, torch.Tensor]:
        if not samples:
            raise ValueError("empty batch")
        out = {}
        np_dtype = torch.empty(0, dtype=self.dtype).numpy().dtype
        for key in self.keys:
            arrs = [np.asarray(s[key]) for s in samples]
            arrs = [a.reshape(a.shape[0], -1) if a.ndim != 2 else a for a in arrs]
            D = arrs[0].shape[1]
            if any(a.shape[1] != D for a in arrs):
                raise ValueError(f"field {key!r}: samples disagree on the feature size")
            longest = max(a.shape[0] for a in arrs)
            L = self.pad_to.get(key, longest)
            if longest > L:
                raise ValueError(f"field {key!r}: a sample has {longest} rows, pad_to allows {L}")
            slot, host, dev = self._buffers(key, (len(arrs), L, D))
            hv = host.numpy()
            for i, a in enumerate(arrs):
                n = a.shape[0]
                hv[i, :n] = a.astype(np_dtype, copy=False)
                hv[i, n:] = 0  # (the buffer is reused: clear what an earlier, longer sample left)
            if self._stream is not None:
                cur = torch.cuda.current_stream(self.device)
                self._stream.wait_stream(cur)  # the previous consumer of this device buffer is done before we overwrite it
                with torch.cuda.stream(self._stream):
                    dev.copy_(host, non_blocking=True)
                    slot[2] = self._stream.record_event()
            out[key] = dev
        if self._stream is not None:
            self._event = self._stream.record_event()
        return out

    def wait(self) -> None:
        if self._event is not None:
            torch.cuda.current_stream(self.device).wait_event(self._event)
