"""M4C multimodal transformer body on the HIP path ("next" row 3 of SURVEY 8f).

``BertEncoder`` here is a drop-in for ``transformers.models.bert.modeling_bert.BertEncoder`` as the reference
calls it (models/mmf_m4c.py:262-263,287,349-351: ``encoder(hidden, additive_mask, head_mask=[None]*L)[0]``):
same constructor argument (a BertConfig-like object), same ``state_dict`` keys
(``layer.{i}.attention.self.query.weight`` ...), so HF / reference checkpoints load.  A layer is exactly the two
fused blocks of the hot path -- MHA block (QKV GEMM, attention with an additive (B,1,S,S) prefix-LM mask, output
GEMM + dropout + residual, LayerNorm eps 1e-12) and FFN block (GEMM+GELU, GEMM + dropout + residual, LayerNorm).
Heads of 96 features (768/8) run on the MFMA attention kernels (256-byte LDS image rows; two-kernel backward).

Dropout on the attention PROBABILITIES (attention_probs_dropout_prob, training only) runs inside the VALU attention
kernels (ovqa_attention_fwd/bwd ``att_drop``); evaluation / decoding (BASELINE config 4) has none.

``PrevPredEmbeddings`` / ``MMT``: mmf_m4c.py:399-459 / 282-364.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from .. import functional as Fn
from .. import runtime as rt
from ..utils import generate_sequential_mask


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.query = nn.Linear(config.hidden_size, config.hidden_size)
        self.key = nn.Linear(config.hidden_size, config.hidden_size)
        self.value = nn.Linear(config.hidden_size, config.hidden_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _ovqa_param_groups(self):
        return [[self.query.weight, self.key.weight, self.value.weight],
                [self.query.bias, self.key.bias, self.value.bias]]


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        if getattr(config, "hidden_act", "gelu") != "gelu":
            raise NotImplementedError("only the exact-erf GELU of BERT is fused")
        self.heads = config.num_attention_heads
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._sites = [rt.new_dropout_site() for _ in range(3)]

    def forward(self, hidden_states, attention_mask=None):
        arena = rt.ensure_arena(self)
        x = Fn.to_compute(hidden_states, arena.compute_dtype)
        a = self.attention
        mask = attention_mask
        if mask is not None:
            mask = mask.to(torch.float32)
            if mask.dim() == 2:  # (B, S) additive
                mask = mask[:, None, None, :]
        att = SimpleNamespace(fc_q=a.self.query, fc_k=a.self.key, fc_v=a.self.value, fc_o=a.output.dense, h=self.heads)
        ln1 = a.output.LayerNorm
        st = dict(arena=arena, att=att, ln=ln1,
                  params=[a.self.query.weight, a.self.query.bias, a.self.key.weight, a.self.key.bias,
                          a.self.value.weight, a.self.value.bias, a.output.dense.weight, a.output.dense.bias,
                          ln1.weight, ln1.bias],
                  drop=rt.dropout_spec(a.output.dropout.p, self._sites[0], self.training, x.device),
                  # dropout on the attention probabilities (HF BertSelfAttention): training only, VALU attention kernels
                  att_drop=rt.dropout_spec(a.self.dropout.p, self._sites[2], self.training, x.device))
        x = Fn.mha_block(x, x, x, mask, st)
        ffn = SimpleNamespace(fc1=self.intermediate.dense, fc2=self.output.dense, layer_norm=self.output.LayerNorm)
        st = dict(arena=arena, mod=ffn,
                  params=[ffn.fc1.weight, ffn.fc1.bias, ffn.fc2.weight, ffn.fc2.bias, ffn.layer_norm.weight,
                          ffn.layer_norm.bias],
                  drop1=None, drop2=rt.dropout_spec(self.output.dropout.p, self._sites[1], self.training, x.device))
        return Fn.ffn_block(x, st)


class BertEncoder(nn.Module):
    """``encoder(hidden_states, attention_mask, head_mask=None) -> (hidden_states,)`` (head masks must be None)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, head_mask=None, **kwargs):
        if head_mask is not None and any(h is not None for h in head_mask):
            raise NotImplementedError("head masks are not supported (the reference passes [None] * L)")
        dtype = hidden_states.dtype
        for layer in self.layer:
            hidden_states = layer(hidden_states, attention_mask)
        return (Fn.finalize(hidden_states, dtype),)


class PrevPredEmbeddings(nn.Module):
    """Embeddings of the previous decoding steps' predictions (mmf_m4c.py:399-446): gather from
    [LN(fixed answer table); LN(OCR embeddings)] + LN(position + token-type embedding)."""

    def __init__(self, config):
        super().__init__()
        hidden, eps = config.hidden_size, config.layer_norm_eps
        self.position_embeddings = nn.Embedding(100, hidden)
        self.token_type_embeddings = nn.Embedding(5, hidden)
        self.ans_layer_norm = nn.LayerNorm(hidden, eps=eps)
        self.ocr_layer_norm = nn.LayerNorm(hidden, eps=eps)
        self.emb_layer_norm = nn.LayerNorm(hidden, eps=eps)
        self.emb_dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, ans_emb, ocr_emb, prev_inds, cache=None):
        """``cache`` (a dict, evaluation only): the two LayerNorm results do not depend on ``prev_inds`` -- the decoding loop
        computes them on its first pass and keeps them."""
        assert prev_inds.dim() == 2 and prev_inds.dtype == torch.long and ans_emb.dim() == 2
        arena = rt.ensure_arena(self)
        T = arena.compute_dtype
        B, steps = prev_inds.shape
        ans_num = ans_emb.size(0)
        if cache is not None and "ans" in cache and not torch.is_grad_enabled():
            ans, ocr = cache["ans"], cache["ocr"]
        else:
            ans = Fn.prologue(ans_emb.float().unsqueeze(0), self.ans_layer_norm, None, arena, T)[0]
            ocr = Fn.prologue(ocr_emb.float(), self.ocr_layer_norm, None, arena, T)
            if cache is not None and not torch.is_grad_enabled():
                cache["ans"], cache["ocr"] = ans, ocr
        # the reference concatenates the (expanded) answer table with the OCR embeddings per sample and gathers from that
        # (mmf_m4c.py:425-428): B x (5000 + 50) x 768 values written to pick B x steps rows -- 496 MB and 428 of the 2430 us
        # of a decoding pass at configs[3] size.  The same rows from the two sources directly:
        is_ocr = prev_inds.ge(ans_num)
        a_rows = ans[prev_inds.clamp(max=ans_num - 1)]
        o_rows = torch.gather(ocr, 1, (prev_inds - ans_num).clamp(min=0).unsqueeze(-1).expand(-1, -1, ocr.size(-1)))
        raw = torch.where(is_ocr.unsqueeze(-1), o_rows, a_rows)
        pos = self.position_embeddings(torch.arange(steps, device=ocr_emb.device).unsqueeze(0).expand(B, steps))
        typ = self.token_type_embeddings(prev_inds.ge(ans_num).long())
        emb = Fn.prologue((pos + typ).float(), self.emb_layer_norm, None, arena, T)
        return raw + self.emb_dropout(emb)


class MMT(nn.Module):
    """Multimodal transformer of M4C (mmf_m4c.py:282-364): [txt; obj; ocr; dec] under a prefix-LM mask."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.prev_pred_embeddings = PrevPredEmbeddings(config)
        self.encoder = BertEncoder(config)

    def forward(self, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, fixed_ans_emb, prev_inds, cache=None):
        """``cache`` (a dict, evaluation only): the decoding loop calls this once per pass with the SAME txt / obj / ocr
        embeddings and masks -- only ``prev_inds`` changes.  With a cache the [txt; obj; ocr] rows of the input and the
        whole (B, 1, S, S) mask are written on the first pass and kept; later passes write the decoding rows only (the
        per-pass concatenation of the four blocks, the mask's repeat and its causal corner: ~40 us of stock launches)."""
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds, cache=cache)
        steps = dec_emb.size(1)
        if cache is not None and not torch.is_grad_enabled():
            if "x" not in cache:
                nt, no, nc = txt_emb.size(1), obj_emb.size(1), ocr_emb.size(1)
                x = torch.empty(dec_emb.size(0), nt + no + nc + steps, dec_emb.size(2), dtype=dec_emb.dtype,
                                device=dec_emb.device)
                x[:, :nt], x[:, nt:nt + no], x[:, nt + no:nt + no + nc] = txt_emb, obj_emb, ocr_emb
                cache["x"], cache["ext"] = x, self._extended_mask(txt_mask, obj_mask, ocr_mask, steps, dec_emb)
            x, ext = cache["x"], cache["ext"]
            x[:, -steps:] = dec_emb
        else:
            x = torch.cat([txt_emb.to(dec_emb.dtype), obj_emb.to(dec_emb.dtype), ocr_emb.to(dec_emb.dtype), dec_emb], dim=1)
            ext = self._extended_mask(txt_mask, obj_mask, ocr_mask, steps, dec_emb)
        out = self.encoder(x, ext, head_mask=[None] * len(self.encoder.layer))[0]
        nt, no, nc = txt_mask.size(-1), obj_mask.size(-1), ocr_mask.size(-1)
        return {"mmt_seq_output": out, "mmt_txt_output": out[:, :nt], "mmt_ocr_output": out[:, nt + no:nt + no + nc],
                "mmt_dec_output": out[:, -steps:]}

    @staticmethod
    def _extended_mask(txt_mask, obj_mask, ocr_mask, steps, dec_emb):
        dec_mask = torch.zeros(dec_emb.size(0), 1, 1, steps, dtype=torch.float32, device=dec_emb.device)
        mask = torch.cat([txt_mask, obj_mask, ocr_mask, dec_mask], dim=-1).float()
        S = mask.size(-1)
        ext = mask.repeat(1, 1, S, 1)
        ext[:, :, -steps:, -steps:] = generate_sequential_mask(steps, device=ext.device)
        # what this tensor IS, for the attention kernel (inference): the key row + a causal corner of `steps` positions --
        # ovqa_attention_fwd_prefix_lm then reads S mask values per (b, h) instead of S x S (functional._project_and_attend)
        if mask.shape[1] == 1 and mask.shape[2] == 1:
            ext._ovqa_prefix_lm = (mask, steps)
        return ext


class M4CDecodingHead(nn.Module):
    """Output scorer and greedy decoder of M4C (mmf_m4c.py:221-256): scores = [classifier(dec) | OcrPtrNet(dec, ocr)],
    and at evaluation time ``max_iter`` passes of the multimodal transformer, each feeding the arg-max of the previous
    pass back as ``prev_inds`` (fixed-vocabulary index, or ``num_choices + i`` for OCR token i), with the reference's
    early exit once every sample has produced ``eos_idx``.

    No key/value reuse across the passes: in the reference the encoding positions attend to the decoding positions
    too (``dec_mask`` is all zeros and the masks are ADDITIVE, mmf_m4c.py:310-313,333-340), so every hidden state of
    every layer changes from pass to pass; caching the [txt; obj; ocr] prefix would change the results."""

    def __init__(self, hidden_size: int, num_choices: int):
        super().__init__()
        from .pointer import OcrPtrNet
        self.classifier = nn.Linear(hidden_size, num_choices)
        self.ocr_ptr_net = OcrPtrNet(hidden_size)

    def scores(self, mmt_results, ocr_mask):
        """mmf_m4c.py:221-229 (_forward_output)."""
        dec, ocr = mmt_results["mmt_dec_output"], mmt_results["mmt_ocr_output"]
        arena = rt.ensure_arena(self.classifier)
        fixed = Fn.linear(dec.to(arena.compute_dtype), self.classifier, arena).float()
        dynamic = self.ocr_ptr_net(dec, ocr, ocr_mask)
        return torch.cat([fixed, dynamic], dim=-1)

    @torch.no_grad()
    def greedy_decode(self, mmt, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, max_iter: int, bos_idx: int,
                      eos_idx: int):
        """mmf_m4c.py:236-256.  Returns (scores of the last pass (B, max_iter, num_choices + n_ocr), prev_inds,
        number of passes run)."""
        B, dev = txt_emb.shape[0], txt_emb.device
        prev_inds = torch.zeros((B, max_iter), dtype=torch.long, device=dev)
        prev_inds[:, 0] = bos_idx
        last_ids = torch.zeros((B,), device=dev)
        scores, passes = None, 0
        cache = {} if isinstance(mmt, MMT) else None  # (a foreign transformer keeps the reference's call)
        for ith in range(max_iter):
            kw = {"cache": cache} if cache is not None else {}
            res = mmt(txt_emb=txt_emb, txt_mask=txt_mask, obj_emb=obj_emb, obj_mask=obj_mask, ocr_emb=ocr_emb,
                      ocr_mask=ocr_mask, fixed_ans_emb=self.classifier.weight, prev_inds=prev_inds, **kw)
            scores = self.scores(res, ocr_mask)
            passes += 1
            argmax_inds = scores.argmax(dim=-1)
            prev_inds[:, 1:] = argmax_inds[:, :-1]
            last_ids = torch.where(last_ids == eos_idx, last_ids, argmax_inds[:, ith].to(last_ids.dtype))
            if last_ids.mean() == eos_idx:  # one host sync per pass, as in the reference
                break
        return scores, prev_inds, passes



class GraphedGreedyDecode:
    """``M4CDecodingHead.greedy_decode`` with every PASS of the multimodal transformer replayed from one hipGraph: a pass is
    ~70 launches of 3-100 us; launched eagerly from Python the GPU waits ~0.3 ms per pass for the host (1.62 ms of kernels in
    a 1.91 ms pass at configs[3] size).  What is captured: ``MMT.forward`` on static copies of the inputs (prefix rows, mask
    and the two input LayerNorms kept from the first pass), the output scores, their arg-max and the in-place update of
    ``prev_inds``.  What stays on the host, as in the reference (mmf_m4c.py:236-256): the early exit -- one check per pass.
    Inputs must keep their shapes from call to call (they are copied into the static buffers)."""

    def __init__(self, head, mmt, max_iter: int, bos_idx: int, eos_idx: int):
        self.head, self.mmt, self.max_iter, self.bos, self.eos = head, mmt, max_iter, bos_idx, eos_idx
        self.graph = None

    def _pass(self):
        txt, tm, obj, om, ocr, cm = self.static_in
        res = self.mmt(txt_emb=txt, txt_mask=tm, obj_emb=obj, obj_mask=om, ocr_emb=ocr, ocr_mask=cm,
                       fixed_ans_emb=self.head.classifier.weight, prev_inds=self.prev, cache=self.cache)
        scores = self.head.scores(res, cm)
        am = scores.argmax(dim=-1)
        self.prev[:, 1:] = am[:, :-1]
        return scores, am

    @torch.no_grad()
    def __call__(self, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, use_graph: bool = True):
        ins = (txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask)
        if not use_graph or not txt_emb.is_cuda:
            return self.head.greedy_decode(self.mmt, *ins, self.max_iter, self.bos, self.eos)
        B, dev = txt_emb.shape[0], txt_emb.device
        if self.graph is None:
            self.static_in = tuple(t.clone() for t in ins)
            self.prev = torch.zeros((B, self.max_iter), dtype=torch.long, device=dev)
        else:
            assert all(a.shape == b.shape for a, b in zip(ins, self.static_in)), "static input shapes"
            for dst, src in zip(self.static_in, ins):
                dst.copy_(src, non_blocking=True)
        self.prev.zero_()
        self.prev[:, 0] = self.bos
        self.cache = {}  # (the first pass of every call is eager: it rebuilds prefix rows, mask and input LayerNorms)
        last_ids = torch.zeros((B,), device=dev)
        scores, passes = None, 0
        for ith in range(self.max_iter):
            if ith == 0:
                scores, am = self._pass()
                if self.graph is None:
                    self._capture()
                else:  # the graph reads the buffers the cache held at capture time: refresh THEM
                    for k in ("x", "ext", "ans", "ocr"):
                        self.cache0[k].copy_(self.cache[k], non_blocking=True)
                    h0, h1 = (getattr(c["ext"], "_ovqa_prefix_lm", None) for c in (self.cache0, self.cache))
                    if h0 is not None and h1 is not None:  # (the key row the attention kernel reads instead of `ext`)
                        h0[0].copy_(h1[0], non_blocking=True)
                    self.cache = self.cache0
            else:
                self.graph.replay()
                scores, am = self.static_out
            passes += 1
            last_ids = torch.where(last_ids == self.eos, last_ids, am[:, ith].to(last_ids.dtype))
            if last_ids.mean() == self.eos:
                break
        return scores, self.prev, passes

    def _capture(self):
        from .. import runtime as _rt
        self.cache0 = self.cache
        keep = self.prev.clone()
        side = torch.cuda.Stream(device=self.prev.device)
        side.wait_stream(torch.cuda.current_stream(self.prev.device))
        with torch.cuda.stream(side):  # warm-up outside the capture
            self._pass()
        torch.cuda.current_stream(self.prev.device).wait_stream(side)
        torch.cuda.synchronize(self.prev.device)
        self.prev.copy_(keep)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=_rt.capture_error_mode()):
            self.static_out = self._pass()
        self.prev.copy_(keep)  # (capture does not execute, the warm-up pass did)
