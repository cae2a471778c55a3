"""Batched beam search over a stateful decoder: the control flow of models/modules/beam_search.py:36-118 (select by
sorting the candidates, mask finished sequences, gather the outputs) with ONE change: the per-step reorder of every
state buffer goes through ``Module.reorder_states`` (one grouped gather launch, caches shared between the beams of a
sample untouched) instead of ``apply_to_states(_expand_state(...))`` (beam_search.py:19-34,61).

It is host-side plumbing around the hot path -- what ``bench.py --workload decode`` and the decode tests drive the
``Decoder`` with; ``step(t, prev_words) -> (b_s * cur_beam, 1, |V|)`` log-probabilities is the reference's
``model.step`` (models/base_transformer.py:31-44).
"""
from __future__ import annotations

from typing import Callable

import torch


class BeamSearch:
    """``reorder="fused"``: Module.reorder_states; ``"reference"``: the reference's own
    ``apply_to_states(_expand_state(selected_beam, cur_beam_size))`` (beam_search.py:19-34,61), for comparison."""

    def __init__(self, module, step: Callable, b_s: int, max_len: int, eos_idx: int, beam_size: int, device,
                 reorder: str = "fused", logits_step: Callable = None):
        """``logits_step(t, prev_words) -> (b_s * cur_beam, [1,] |V|)`` raw vocabulary logits: when given (and on the
        GPU, beam <= 8) a step's selection and bookkeeping run as two kernels (``_apply_fused``)."""
        self.module, self.step = module, step
        self.b_s, self.max_len, self.eos_idx, self.beam_size, self.device = b_s, max_len, eos_idx, beam_size, device
        self.reorder = reorder
        self.logits_step = logits_step

    def select(self, candidate_logprob):
        """The `beam` best of the cur_beam * |V| candidates of every sample, best first: (flat index, value), what
        beam_search.py:36-39 takes from a full descending sort.  Two stages, same result: the best `beam` of every
        beam's |V| continuations first (any of the overall best `beam` is among the best `beam` of its own beam), then
        the best `beam` of those cur_beam * beam -- short slices only (a 3 x 4000-wide top-k would take torch's
        multi-block radix path: six launches and a device scan per step)."""
        b_s, cur, V = candidate_logprob.shape
        k = min(self.beam_size, V)
        if candidate_logprob.is_cuda and candidate_logprob.dtype == torch.float32 and self.beam_size <= 8:
            from . import ops
            v1, i1 = ops.topk_rows(candidate_logprob.contiguous(), k)                       # (b_s, cur, k), one launch
            v2, i2 = ops.topk_rows(v1.reshape(b_s, cur * k), self.beam_size)
        else:
            v1, i1 = torch.topk(candidate_logprob, k, dim=-1, largest=True, sorted=True)
            v2, i2 = torch.topk(v1.reshape(b_s, cur * k), self.beam_size, dim=-1, largest=True, sorted=True)
        beam_of = torch.div(i2, k, rounding_mode="trunc")
        word = torch.gather(i1.reshape(b_s, cur * k), 1, i2)
        return beam_of * V + word, v2

    def _expand_state(self, selected_beam, cur_beam_size):  # beam_search.py:19-34
        def fn(s):
            shape = [int(sh) for sh in s.shape]
            beam = selected_beam
            for _ in shape[1:]:
                beam = beam.unsqueeze(-1)
            s = torch.gather(s.view(*([self.b_s, cur_beam_size] + shape[1:])), 1,
                             beam.expand(*([self.b_s, self.beam_size] + shape[1:])))
            return s.view(*([-1] + shape[1:]))
        return fn

    def apply(self, out_size: int = 1):
        """beam_search.py:85-118 with the per-step bookkeeping on whole (b_s, beam, T) buffers: the reference keeps the
        chosen words and their scores as Python lists of (b_s, beam, 1) tensors and re-gathers every element of both
        lists at every step (2 t tiny launches at step t); one gather per buffer and step gives the same values."""
        if (self.logits_step is not None and torch.device(self.device).type == "cuda" and self.beam_size <= 8
                and self.reorder == "fused"):
            return self._apply_fused(out_size)
        b_s, beam, T = self.b_s, self.beam_size, self.max_len
        seq_mask = torch.ones((b_s, beam, 1), device=self.device)
        seq_logprob = torch.zeros((b_s, 1, 1), device=self.device)
        outputs = torch.zeros((b_s, beam, T), dtype=torch.long, device=self.device)
        log_probs = torch.zeros((b_s, beam, T), device=self.device)
        selected_words = None
        for t in range(T):
            cur = 1 if t == 0 else beam
            word_logprob = self.step(t, selected_words).view(b_s, cur, -1)
            candidate = seq_logprob + word_logprob
            if t > 0:  # beam_search.py:49-55: sequences that reached <eos> keep their score
                mask = (selected_words.view(b_s, cur) != self.eos_idx).float().unsqueeze(-1)
                seq_mask = seq_mask * mask
                word_logprob = word_logprob * seq_mask.expand_as(word_logprob)
                old = seq_logprob.expand_as(candidate).contiguous()
                old[:, :, 1:] = -999
                candidate = seq_mask * candidate + old * (1 - seq_mask)
            idx, val = self.select(candidate)
            selected_beam = torch.div(idx, candidate.shape[-1], rounding_mode="trunc")
            words = idx - selected_beam * candidate.shape[-1]
            if self.reorder == "fused":
                self.module.reorder_states(selected_beam, b_s, cur, beam)
            else:
                self.module.apply_to_states(self._expand_state(selected_beam, cur))
            seq_logprob = val.unsqueeze(-1)
            seq_mask = torch.gather(seq_mask, 1, selected_beam.unsqueeze(-1))
            this = torch.gather(word_logprob.reshape(b_s, -1), 1, idx)  # word_logprob[b, selected_beam, word]
            if t > 0:  # histories follow their beams (columns >= t are still zero)
                sel3 = selected_beam.unsqueeze(-1).expand(b_s, beam, T)
                outputs, log_probs = torch.gather(outputs, 1, sel3), torch.gather(log_probs, 1, sel3)
            outputs[:, :, t] = words
            log_probs[:, :, t] = this
            selected_words = words.view(-1, 1)
        seq_logprob, order = torch.sort(seq_logprob, 1, descending=True)
        outputs = torch.gather(outputs, 1, order.expand(b_s, beam, T))
        log_probs = torch.gather(log_probs, 1, order.expand(b_s, beam, T))
        outputs, log_probs = outputs.contiguous()[:, :out_size], log_probs.contiguous()[:, :out_size]
        if out_size == 1:
            outputs, log_probs = outputs.squeeze(1), log_probs.squeeze(1)
        return outputs, log_probs

    def _apply_fused(self, out_size: int = 1):
        """The same search with a step's selection and bookkeeping in two launches: ``ovqa_beam_candidates`` (log-softmax
        of the logits, candidate scores of beam_search.py:41-57, the k best per beam) and ``ovqa_beam_commit`` (the best
        `beam` per sample, scores / masks / histories of beam_search.py:58-83, the gather index), then ONE grouped gather
        of every state buffer -- 3 launches where the loop above spends ~45 small torch kernels."""
        from . import ops
        b_s, beam, T, dev = self.b_s, self.beam_size, self.max_len, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        hist = [(torch.zeros((b_s, beam, T), dtype=torch.long, device=dev), torch.zeros((b_s, beam, T), **f32))
                for _ in range(2)]
        seq_logprob, seq_mask = torch.zeros(b_s, **f32), torch.ones(b_s, **f32)
        sl = [torch.empty(b_s * beam, **f32) for _ in range(2)]
        sm = [torch.empty(b_s * beam, **f32) for _ in range(2)]
        wd = [torch.empty((b_s * beam, 1), dtype=torch.long, device=dev) for _ in range(2)]
        sel = torch.empty((b_s, beam), dtype=torch.int32, device=dev)
        words = None
        for t in range(T):
            cur, o = (1 if t == 0 else beam), t & 1
            logits = self.logits_step(t, words)
            logits = logits.reshape(b_s * cur, logits.shape[-1])
            k = min(beam, logits.shape[-1])
            vals, idx, wl = ops.beam_candidates(logits, seq_logprob, seq_mask, None if t == 0 else words.view(-1),
                                                self.eos_idx, k)
            ops.beam_commit(vals, idx, wl, seq_mask, hist[1 - o], hist[o], sl[o], sm[o], sel, wd[o], b_s, cur, k, beam, t)
            self.module.reorder_states(sel, b_s, cur, beam)
            seq_logprob, seq_mask, words = sl[o], sm[o], wd[o]
        outputs, log_probs = hist[(T - 1) & 1]
        seq_logprob, order = torch.sort(seq_logprob.view(b_s, beam, 1), 1, descending=True)
        outputs = torch.gather(outputs, 1, order.expand(b_s, beam, T))
        log_probs = torch.gather(log_probs, 1, order.expand(b_s, beam, T))
        outputs, log_probs = outputs.contiguous()[:, :out_size], log_probs.contiguous()[:, :out_size]
        if out_size == 1:
            outputs, log_probs = outputs.squeeze(1), log_probs.squeeze(1)
        return outputs, log_probs


class GraphedBeamSearch:
    """The WHOLE decode of a batch -- every decoder step, the candidate selection and the reorder of every state buffer,
    T times -- captured once into a hipGraph and replayed per batch.  A decoding step is ~45 launches of 2-20 us each
    and the selection another ~25 tiny torch kernels; launched eagerly from Python the decode is host-bound (1.7-1.8 ms
    per decoding step measured for BASELINE configs[4] at B=64), replayed from a graph it runs at the kernels' own pace.

    Everything inside is capturable by construction: the K / V caches are pre-allocated and appended in place, the
    grouped state gather takes its table in the kernel arguments (no upload), state defaults live on the device, the
    selection is torch.topk / gather / arithmetic without a host sync (finished sequences are masked, not skipped:
    beam_search.py:49-55 has no early exit either).  Inputs are copied into static buffers before each replay.

    ``decoder(prev_tokens, encoder_features, encoder_attention_mask) -> log-probabilities`` is the reference's
    ``Decoder.forward`` as ``model.step`` calls it (base_transformer.py:31-44)."""

    def __init__(self, decoder, b_s: int, max_len: int, bos_idx: int, eos_idx: int, beam_size: int, out_size: int = 1,
                 fused: bool = True):
        """``fused``: selection and bookkeeping as two kernels per step (BeamSearch._apply_fused); off = the torch ops."""
        self.decoder, self.b_s, self.max_len, self.bos, self.eos, self.beam = decoder, b_s, max_len, bos_idx, eos_idx, beam_size
        self.out_size = out_size
        self.fused = fused
        self.graph = None
        self.static_in = None
        self.static_out = None

    def _decode(self, enc, mask):
        b_s, beam, dev = self.b_s, self.beam, enc.device

        def step(t, prev, **kw):
            if t == 0:
                prev = torch.full((b_s, 1), self.bos, dtype=torch.long, device=dev)
            # from t = 1 on a sample's encoder features serve all its beams: the reference keeps them as a model state
            # and gathers a copy per beam (base_transformer.py:21-22, beam_search.py:61); here the decoder is TOLD the
            # relation (encoder_group) and gets the per-sample tensors, so its encoder attention projects K / V once
            return self.decoder(prev, enc, mask, encoder_group=(1 if t == 0 else beam), **kw)
        logits_step = (lambda t, prev: step(t, prev, return_logits=True)) if self.fused else None
        with torch.no_grad(), self.decoder.statefulness(b_s):
            return BeamSearch(self.decoder, step, b_s, self.max_len, self.eos, beam, dev,
                              logits_step=logits_step).apply(self.out_size)

    def __call__(self, enc, mask, use_graph: bool = True):
        if not use_graph or not enc.is_cuda:
            return self._decode(enc, mask)
        if self.graph is None:
            self.static_in = (enc.clone(), mask.clone())
            side = torch.cuda.Stream(device=enc.device)
            side.wait_stream(torch.cuda.current_stream(enc.device))
            with torch.cuda.stream(side):  # warm-up outside the capture: lazy initialisations, arena packing, defaults
                for _ in range(2):
                    self._decode(*self.static_in)
            torch.cuda.current_stream(enc.device).wait_stream(side)
            torch.cuda.synchronize(enc.device)
            self.graph = torch.cuda.CUDAGraph()
            from . import runtime as rt
            with torch.cuda.graph(self.graph, capture_error_mode=rt.capture_error_mode()):
                self.static_out = self._decode(*self.static_in)
        assert enc.shape == self.static_in[0].shape and mask.shape == self.static_in[1].shape, "static input shapes"
        if enc.data_ptr() != self.static_in[0].data_ptr():
            self.static_in[0].copy_(enc, non_blocking=True)
        if mask.data_ptr() != self.static_in[1].data_ptr():
            self.static_in[1].copy_(mask, non_blocking=True)
        self.graph.replay()
        return self.static_out
