"""Plain-PyTorch CPU restatement of the reference hot path (test oracle).

Every class keeps the reference's attribute names so that ``state_dict`` keys
are interchangeable with the reference modules and with the HIP-backed modules
in ``openvivqa_amd``.  Citations are ``file:line`` under ``/root/reference``.

This file is the *checker*: it is imported only by tests, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``.
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from typing import Optional

import torch
from torch import nn
from torch.nn import functional as F

MASK_VALUE = -10e4  # models/utils.py:56,64,71  (== -100000.0)

__all__ = [
    "emulate_bf16", "MASK_VALUE", "padding_mask", "sequential_mask", "self_attention_masks",
    "sinusoid_positions", "sinusoid_table", "sdpa_core",
    "OracleSDPA", "OracleMemorySDPA", "OracleGeometrySDPA", "OracleAdaptiveSDPA", "box_relational_embedding", "OracleMHA", "OraclePWFF", "OracleEncoderLayer",
    "OracleGuidedEncoderLayer", "OracleCrossModalityEncoderLayer",
    "OracleEncoder", "OracleGuidedAttentionEncoder", "OracleCoAttentionEncoder",
    "OracleCrossModalityEncoder", "OracleDecoderLayer", "OracleDecoder",
    "OracleUsualEmbedding", "OracleOcrPtrNet", "OracleDynamicPointerNetwork",
    "OracleFeatureEmbedding", "OracleLSTMTextEmbedding", "lstm_recurrence", "OracleMLP", "OracleMCAN",
    "OracleBertEncoder", "OraclePrevPredEmbeddings", "OracleMMT", "OracleM4CDecodingHead", "batch_gather",
    "noam_lambda", "oracle_train_step", "build_oracle_encoder", "oracle_beam_search", "oracle_generate",
]


# --------------------------------------------------------------------------
# bf16 emulation mode (bug detector for the HIP bf16 path, not a second reference)
# --------------------------------------------------------------------------
# Inside ``with emulate_bf16():`` the oracle rounds to bf16 exactly where the HIP bf16 path STORES bf16 (DESIGN.md
# section 3): nn.Linear weights, every GEMM input operand, the projected q / k / v, the attention output o, the
# un-normalised softmax numerators that feed the P.V matrix product, the FFN hidden activation h.  Everything the HIP
# path keeps in fp32 stays fp32 here: accumulators, biases, softmax statistics, the residual stream (pre-LayerNorm sums
# and LayerNorm outputs), LayerNorm statistics.  The roundings are straight-through for autograd.  The gap between
# the HIP path and this mode is what the KERNELS add (accumulation order, fast exp / erf); the gap between this mode
# and the plain fp32 oracle is the price of bf16 storage.
_EMU = {"on": False}


@contextmanager
def emulate_bf16(on: bool = True):
    prev = _EMU["on"]
    _EMU["on"] = bool(on)
    try:
        yield
    finally:
        _EMU["on"] = prev


class _StoreBf16(torch.autograd.Function):
    """A tensor the HIP path stores in bf16: its value is rounded in forward, and so is its gradient in backward
    (the HIP backward kernels hand gradients from kernel to kernel in bf16 as well)."""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _GradBf16(torch.autograd.Function):
    """Value untouched (kept in fp32 by the HIP path), gradient rounded to bf16 (LayerNorm backward writes bf16)."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _r(t: torch.Tensor) -> torch.Tensor:
    """bf16 storage round trip of a value and of its gradient (identity outside emulate_bf16)."""
    if not _EMU["on"]:
        return t
    return _StoreBf16.apply(t) if t.requires_grad else t.bfloat16().float()


def _gr(t: torch.Tensor) -> torch.Tensor:
    """fp32 value whose GRADIENT the HIP path stores in bf16 (the pre-LayerNorm sums of the residual stream)."""
    if not _EMU["on"] or not t.requires_grad:
        return t
    return _GradBf16.apply(t)


def _lin(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear as the MFMA GEMM sees it: bf16 input and weight, fp32 accumulate, fp32 bias."""
    if not _EMU["on"]:
        return lin(x)
    w = lin.weight
    return F.linear(_r(x), w + (w.bfloat16().float() - w).detach(), lin.bias)  # weight gradients stay fp32


# --------------------------------------------------------------------------
# masks / positions                                  models/utils.py:32-73
# --------------------------------------------------------------------------
def padding_mask(seq: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """Additive key-padding mask, (B,1,1,N), values in {-0.0, -1e5}.

    models/utils.py:44-57: a row is padding iff its feature sum equals
    ``padding_idx * D`` (token tensors are treated as D == 1)."""
    s = seq.unsqueeze(-1) if seq.dim() == 2 else seq
    hit = s.sum(dim=-1) == (padding_idx * s.shape[-1])
    return (hit.long() * MASK_VALUE)[:, None, None, :]


def sequential_mask(t: int) -> torch.Tensor:
    """Causal additive mask (1,1,T,T).  models/utils.py:59-66."""
    return (torch.ones(t, t).triu(1) * MASK_VALUE)[None, None]


def self_attention_masks(pad: torch.Tensor, seq: torch.Tensor) -> torch.Tensor:
    """OR of padding and causal masks -> (B,1,T,T).  models/utils.py:68-73."""
    return ((pad != 0) | (seq != 0)).long() * MASK_VALUE


def sinusoid_positions(n: int, d: int, temperature: float = 10000.0) -> torch.Tensor:
    """(n, d) fp32 table of the encoder positional embedding.

    models/modules/pos_embeddings.py:58-72 called with mask=None: positions are
    cumsum(ones) = 1..n, channel c uses temperature**(2*(c//2)/d), even
    channels take sin and odd channels cos (interleaved)."""
    pos = torch.arange(1, n + 1, dtype=torch.float32)[:, None]
    c = torch.arange(d, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(c, 2, rounding_mode="floor") / d)
    ang = pos / dim_t
    out = torch.empty(n, d, dtype=torch.float32)
    out[:, 0::2] = ang[:, 0::2].sin()
    out[:, 1::2] = ang[:, 1::2].cos()
    return out


def sinusoid_table(max_len: int, d: int, padding_idx: Optional[int] = None) -> torch.Tensor:
    """Decoder position table (max_len, d).  models/utils.py:21-38: positions
    0..max_len-1, channel pair j uses 10000**(2j/d); row padding_idx zeroed."""
    pos = torch.arange(max_len, dtype=torch.float32)[:, None]
    j = torch.arange(d // 2, dtype=torch.float32)[None, :]
    ang = pos / 10000 ** (2 * j / d)
    out = torch.zeros(max_len, d)
    out[:, 0::2] = ang.sin()
    out[:, 1::2] = ang.cos()
    if padding_idx is not None:
        out[padding_idx] = 0
    return out


# --------------------------------------------------------------------------
# attention                                  models/modules/attentions.py
# --------------------------------------------------------------------------
class _EmuCore(torch.autograd.Function):
    """The attention core in bf16-emulation mode, backward included: the matrix cores take bf16 operands, so the
    backward rounds exactly what the HIP kernels feed them -- the probabilities P (for dV) and dS = P (dP - delta)
    (for dQ, dK) -- and nothing else; delta = rowsum(P dP) = dO . (P V) with the unrounded P (the kernels read that
    O as o + o_lo, 16 significant bits), sums are fp32.  Plain autograd over fp32 matmuls would keep dS in fp32, which no bf16 MFMA
    kernel can: with near-uniform attention dQ / dK are cancellations 100-3000x smaller than their terms, and that
    one rounding is most of their error."""

    @staticmethod
    def forward(ctx, q, k, v, mask, d_k):
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d_k)
        if mask is not None:
            s = s + mask
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        den = e.sum(dim=-1, keepdim=True)
        p = e / den
        o = torch.matmul(e.bfloat16().float(), v) / den  # bf16 numerators into P.V, fp32 row sum
        ctx.save_for_backward(q, k, v, p, o)
        ctx.scale = 1.0 / math.sqrt(d_k)
        return o, p

    @staticmethod
    def backward(ctx, d_o, d_p):
        q, k, v, p, o = ctx.saved_tensors
        dp = torch.matmul(d_o, v.transpose(-1, -2))
        delta = (p * dp).sum(-1, keepdim=True)  # = dO . (P V) with the unrounded P: what o + o_lo gives the kernels
        if d_p is not None:  # the returned weights are differentiable too (attentions.py:56,60): VALU kernels, fp32
            dp = dp + d_p
            delta = delta + (p * d_p).sum(-1, keepdim=True)
            ds, pb = p * (dp - delta), p
        else:
            ds, pb = (p * (dp - delta)).bfloat16().float(), p.bfloat16().float()
        dq = torch.matmul(ds, k) * ctx.scale
        dk = torch.matmul(ds.transpose(-1, -2), q) * ctx.scale
        dv = torch.matmul(pb.transpose(-1, -2), d_o)
        return dq, dk, dv, None, None


def sdpa_core(q, k, v, mask, d_k):
    """softmax(q k^T / sqrt(d_k) + mask) v on (B,H,n,d) tensors.
    models/modules/attentions.py:53-57.  Returns (out, att)."""
    if _EMU["on"]:
        if q.requires_grad or k.requires_grad or v.requires_grad:
            return _EmuCore.apply(q, k, v, mask, d_k)
        att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d_k)
        if mask is not None:
            att = att + mask
        # the kernel feeds bf16 exp(s - max) to the second matrix product and divides by the fp32 row sum
        e = torch.exp(att - att.max(dim=-1, keepdim=True).values)
        den = e.sum(dim=-1, keepdim=True)
        return torch.matmul(e.bfloat16().float(), v) / den, e / den
    att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d_k)
    if mask is not None:
        att = att + mask
    att = torch.softmax(att, dim=-1)
    return torch.matmul(att, v), att


class OracleSDPA(nn.Module):
    """ScaledDotProductAttention.  models/modules/attentions.py:10-60."""

    def __init__(self, cfg):
        super().__init__()
        self.d_model, self.h = cfg.D_MODEL, cfg.HEAD
        self.d_k, self.d_v = cfg.D_KEY, cfg.D_VALUE
        self.fc_q = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_k = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_v = nn.Linear(self.d_model, self.h * self.d_v)
        self.fc_o = nn.Linear(self.h * self.d_v, self.d_model)
        for lin in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):  # :36-44
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, queries, keys, values, attention_mask=None, **kw):
        b, nq, nk = queries.shape[0], queries.shape[1], keys.shape[1]
        q = _r(_lin(self.fc_q, queries)).view(b, nq, self.h, self.d_k).transpose(1, 2)
        k = _r(_lin(self.fc_k, keys)).view(b, nk, self.h, self.d_k).transpose(1, 2)
        v = _r(_lin(self.fc_v, values)).view(b, nk, self.h, self.d_v).transpose(1, 2)
        o, att = sdpa_core(q, k, v, attention_mask, self.d_k)
        o = _r(o.transpose(1, 2).reshape(b, nq, self.h * self.d_v))
        return _lin(self.fc_o, o), att


class OracleMemorySDPA(nn.Module):
    """AugmentedMemoryScaledDotProductAttention.  models/modules/attentions.py:129-205: m learned memory slots are
    appended to the projected keys/values (scaled by sqrt(d_k) / sqrt(m)); the mask is added to the real keys only."""

    def __init__(self, cfg):
        super().__init__()
        self.d_model, self.h, self.d_k, self.d_v, self.m = cfg.D_MODEL, cfg.HEAD, cfg.D_KEY, cfg.D_VALUE, cfg.MEMORY
        self.fc_q = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_k = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_v = nn.Linear(self.d_model, self.h * self.d_v)
        self.fc_o = nn.Linear(self.h * self.d_v, self.d_model)
        self.m_k = nn.Parameter(torch.randn(1, self.m, self.h * self.d_k) / self.d_k)
        self.m_v = nn.Parameter(torch.randn(1, self.m, self.h * self.d_v) / self.m)

    def forward(self, queries, keys, values, attention_mask=None, **kw):
        b, nq, nk = queries.shape[0], queries.shape[1], keys.shape[1]
        m_k = math.sqrt(self.d_k) * self.m_k.expand(b, -1, -1)
        m_v = math.sqrt(self.m) * self.m_v.expand(b, -1, -1)
        q = _r(_lin(self.fc_q, queries)).view(b, nq, self.h, self.d_k).transpose(1, 2)
        k = torch.cat([_r(_lin(self.fc_k, keys)), _r(m_k)], 1).view(b, nk + self.m, self.h, self.d_k).transpose(1, 2)
        v = torch.cat([_r(_lin(self.fc_v, values)), _r(m_v)], 1).view(b, nk + self.m, self.h, self.d_v).transpose(1, 2)
        att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.d_k)
        if attention_mask is not None:
            att = torch.cat([att[..., :nk] + attention_mask, att[..., nk:]], dim=-1)
        if _EMU["on"]:  # as sdpa_core: bf16 numerators into the second product, fp32 row sum
            e = torch.exp(att - att.max(dim=-1, keepdim=True).values)
            den = e.sum(dim=-1, keepdim=True)
            o, att = torch.matmul(_r(e), v) / den, e / den
        else:
            att = torch.softmax(att, dim=-1)
            o = torch.matmul(att, v)
        o = _r(o.transpose(1, 2).reshape(b, nq, self.h * self.d_v))
        return _lin(self.fc_o, o), att


def box_relational_embedding(f_g, dim_g=64, wave_len=1000, trignometric_embedding=True):
    """Pairwise box geometry (B, N, N, dim_g).  models/utils.py:102-162."""
    bs = f_g.size(0)
    x_min, y_min, x_max, y_max = torch.chunk(f_g, chunks=4, dim=-1)
    cx, cy = (x_min + x_max) * 0.5, (y_min + y_max) * 0.5
    w, h = (x_max - x_min) + 1.0, (y_max - y_min) + 1.0
    delta_x = torch.log(torch.clamp(torch.abs((cx - cx.view(bs, 1, -1)) / w), min=1e-3))  # :126-128
    delta_y = torch.log(torch.clamp(torch.abs((cy - cy.view(bs, 1, -1)) / h), min=1e-3))  # :130-132
    delta_w = torch.log(w / w.view(bs, 1, -1))  # :134
    delta_h = torch.log(h / h.view(bs, 1, -1))  # :135
    n = delta_h.size(1)
    position_mat = torch.cat([d.view(bs, n, n, 1) for d in (delta_x, delta_y, delta_w, delta_h)], -1)  # :143
    if not trignometric_embedding:
        return position_mat
    feat_range = torch.arange(dim_g / 8).to(f_g.device)  # :146-148
    dim_mat = 1.0 / torch.pow(wave_len, feat_range / (dim_g / 8))
    mul_mat = (100.0 * position_mat.view(bs, n, n, 4, -1)) * dim_mat.view(1, 1, 1, -1)  # :150-154
    mul_mat = mul_mat.view(bs, n, n, -1)
    return torch.cat((torch.sin(mul_mat), torch.cos(mul_mat)), -1)  # :156-158


class OracleGeometrySDPA(nn.Module):
    """AugmentedGeometryScaledDotProductAttention, models/modules/attentions.py:62-137, as INTENDED: upstream's
    forward raises NameError on every call (``att`` is read but never assigned, :124,134-137); the computation it
    spells out up to there -- mn = softmax(log(clamp(relu(g), 1e-6)) + QK^T/sqrt(d_k) [+ mask]), out = fc_o(mn V) --
    is what is restated, returning (out, mn).  Constructor and parameter names follow :68-110."""

    def __init__(self, cfg):
        super().__init__()
        self.d_model, self.h, self.d_k, self.d_v = cfg.D_MODEL, cfg.HEAD, cfg.D_KEY, cfg.D_VALUE
        self.trignometric_embedding = cfg.TRIGNOMETRIC_EMBEDDING
        self.d_g = self.d_model // self.h if self.trignometric_embedding else 4
        self.fc_q = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_k = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_v = nn.Linear(self.d_model, self.h * self.d_v)
        self.fc_o = nn.Linear(self.h * self.d_v, self.d_model)
        self.fc_gs = nn.ModuleList([nn.Linear(self.d_g, 1) for _ in range(self.h)])

    def forward(self, queries, keys, values, boxes, attention_mask=None, **kw):
        emb = box_relational_embedding(boxes, dim_g=self.d_g, trignometric_embedding=self.trignometric_embedding)
        bs, nk = emb.shape[:2]
        flat = emb.view(-1, self.d_g)
        g = F.relu(torch.cat([fc(flat).view(bs, 1, nk, nk) for fc in self.fc_gs], dim=1))  # :113-118
        b, nq = queries.shape[:2]
        q = self.fc_q(queries).view(b, nq, self.h, self.d_k).permute(0, 2, 1, 3)
        k = self.fc_k(keys).view(b, nk, self.h, self.d_k).permute(0, 2, 3, 1)
        v = self.fc_v(values).view(b, nk, self.h, self.d_v).permute(0, 2, 1, 3)
        a = torch.matmul(q, k) / math.sqrt(self.d_k)  # :125
        if attention_mask is not None:
            a = a + attention_mask
        mn = torch.softmax(torch.log(torch.clamp(g, min=1e-6)) + a, dim=-1)  # :129-131
        out = torch.matmul(mn, v).permute(0, 2, 1, 3).contiguous().view(b, nq, self.h * self.d_v)
        return self.fc_o(out), mn


class OracleAdaptiveSDPA(nn.Module):
    """AdaptiveScaledDotProductAttention.  models/modules/attentions.py:210-291: per query one extra softmax column,
    its score against its own projected language signal, whose value is that signal."""

    def __init__(self, cfg):
        super().__init__()
        self.d_model, self.h, self.d_k, self.d_v = cfg.D_MODEL, cfg.HEAD, cfg.D_KEY, cfg.D_VALUE
        self.fc_q = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_k = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_v = nn.Linear(self.d_model, self.h * self.d_v)
        self.fc_s = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_o = nn.Linear(self.h * self.d_v, self.d_model)
        self.dropout = nn.Dropout(cfg.DROPOUT)

    def forward(self, queries, keys, values, language_signals, attention_mask=None):
        b, nq, nk = queries.shape[0], queries.shape[1], keys.shape[1]
        q = _r(_lin(self.fc_q, queries)).view(b, nq, self.h, self.d_k).permute(0, 2, 1, 3)
        s = _r(_lin(self.fc_s, language_signals)).view(b, nq, self.h, self.d_k).permute(0, 2, 1, 3)  # :262
        k = _r(_lin(self.fc_k, keys)).view(b, nk, self.h, self.d_k).permute(0, 2, 3, 1)
        v = _r(_lin(self.fc_v, values)).view(b, nk, self.h, self.d_v).permute(0, 2, 1, 3)
        attn = torch.matmul(q, k) / math.sqrt(self.d_k)
        if attention_mask is not None:
            attn = attn + attention_mask
        lang = (q * s).sum(-1) / math.sqrt(self.d_k)  # the diagonal of q s^T, :271-272
        if _EMU["on"]:
            # the HIP path's closed form (same function): the kernel's output o over the nk keys (bf16 numerators,
            # stored bf16) merged with the signal column through sigma = sigmoid(lse - lang), in fp32
            mx = attn.max(dim=-1, keepdim=True).values
            e = torch.exp(attn - mx)
            den = e.sum(dim=-1, keepdim=True)
            o = _r(torch.matmul(_r(e), v) / den)
            sigma = torch.sigmoid((mx + den.log()).squeeze(-1) - lang).unsqueeze(-1)
            out = _r((sigma * o + (1 - sigma) * s).permute(0, 2, 1, 3).contiguous().view(b, nq, self.h * self.d_v))
            comb = torch.cat([e / den * sigma, 1 - sigma], dim=-1)
            return _lin(self.fc_o, out), [comb[:, :, i:i + 1] for i in range(nq)]
        comb = torch.softmax(torch.cat([attn, lang.unsqueeze(-1)], dim=-1), dim=-1)  # :274-275, row by row
        out = torch.matmul(comb[..., :nk], v) + comb[..., nk:] * s  # :277-281
        out = out.permute(0, 2, 1, 3).contiguous().view(b, nq, self.h * self.d_v)
        return self.fc_o(out), [comb[:, :, i:i + 1] for i in range(nq)]


class _Stateful(nn.Module):
    """State-buffer machinery.  models/modules/containers.py:4-70."""

    def __init__(self):
        super().__init__()
        self._is_stateful = False
        self._state_names = []
        self._state_defaults = {}

    def register_state(self, name, default):
        self._state_names.append(name)
        self._state_defaults[name] = None if default is None else default.clone().detach()
        self.register_buffer(name, default)

    def _walk(self):
        yield self
        for m in self.children():
            if isinstance(m, _Stateful):
                yield from m._walk()
            elif isinstance(m, nn.ModuleList):
                for mm in m:
                    if isinstance(mm, _Stateful):
                        yield from mm._walk()

    def apply_to_states(self, fn):
        for m in self._walk():
            for name in m._state_names:
                m._buffers[name] = fn(m._buffers[name])

    def enable_statefulness(self, batch_size):
        for m in self._walk():
            for name in m._state_names:
                d = m._state_defaults[name]
                if d is None:
                    m._buffers[name] = None
                else:
                    t = d.clone().to(m._buffers[name].device).unsqueeze(0)
                    m._buffers[name] = t.expand([batch_size] + list(t.shape[1:])).contiguous()
            m._is_stateful = True

    def disable_statefulness(self):
        for m in self._walk():
            for name in m._state_names:
                d = m._state_defaults[name]
                m._buffers[name] = None if d is None else d.clone().to(m._buffers[name].device)
            m._is_stateful = False

    @contextmanager
    def statefulness(self, batch_size):
        self.enable_statefulness(batch_size)
        try:
            yield
        finally:
            self.disable_statefulness()


class OracleMHA(_Stateful):
    """MultiHeadAttention: post-LN residual wrapper, optional AoA gate and
    running K/V state.  models/modules/attentions.py:293-339."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.D_MODEL
        self.use_aoa = cfg.USE_AOA
        if self.use_aoa:
            self.informative_attention = nn.Linear(2 * d, d)
            self.gated_attention = nn.Linear(2 * d, d)
        self.attention = OracleSDPA(cfg)
        self.dropout = nn.Dropout(p=cfg.DROPOUT)
        self.layer_norm = nn.LayerNorm(d)
        self.can_be_stateful = cfg.CAN_BE_STATEFUL
        if self.can_be_stateful:
            self.register_state("running_keys", torch.zeros((0, d)))
            self.register_state("running_values", torch.zeros((0, d)))

    def forward(self, queries, keys, values, attention_mask, **kw):
        if self.can_be_stateful and self._is_stateful:  # :320-325
            self.running_keys = torch.cat([self.running_keys, keys], 1)
            self.running_values = torch.cat([self.running_values, values], 1)
            keys, values = self.running_keys, self.running_values
        out, _ = self.attention(queries, keys, values, attention_mask, **kw)
        out = self.layer_norm(_gr(queries + self.dropout(out)))  # :330-331
        if self.use_aoa:  # :333-337
            z = torch.cat([queries, out], dim=-1)
            out = self.informative_attention(z) * torch.sigmoid(self.gated_attention(z))
        return out


class OraclePWFF(nn.Module):
    """LN(x + drop(fc2(drop(gelu(fc1 x))))).  positionwise_feed_forward.py:5-28."""

    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.D_MODEL, cfg.D_FF)
        self.fc2 = nn.Linear(cfg.D_FF, cfg.D_MODEL)
        self.dropout_1 = nn.Dropout(p=cfg.DROPOUT)
        self.dropout_2 = nn.Dropout(p=cfg.DROPOUT)
        self.layer_norm = nn.LayerNorm(cfg.D_MODEL)

    def forward(self, x):
        h = _r(self.dropout_1(F.gelu(_lin(self.fc1, x))))
        return self.layer_norm(_gr(x + self.dropout_2(_lin(self.fc2, h))))


# --------------------------------------------------------------------------
# encoder layers / encoders                    models/modules/encoders.py
# --------------------------------------------------------------------------
class OracleEncoderLayer(nn.Module):
    """encoders.py:9-19."""

    def __init__(self, cfg):
        super().__init__()
        self.mhatt = OracleMHA(cfg)
        self.pwff = OraclePWFF(cfg)

    def forward(self, queries, keys, values, attention_mask, **kw):
        return self.pwff(self.mhatt(queries, keys, values, attention_mask, **kw))


class OracleGuidedEncoderLayer(nn.Module):
    """MCAN SGA unit.  encoders.py:74-99."""

    def __init__(self, cfg):
        super().__init__()
        self.self_mhatt = OracleMHA(cfg)
        self.guided_mhatt = OracleMHA(cfg)
        self.pwff = OraclePWFF(cfg)

    def forward(self, queries, keys, values, self_attention_mask, guided_attention_mask, **kw):
        x = self.self_mhatt(queries, queries, queries, self_attention_mask, **kw)
        x = self.guided_mhatt(x, keys, values, guided_attention_mask, **kw)
        return self.pwff(x)


class OracleCrossModalityEncoderLayer(nn.Module):
    """encoders.py:21-72.  The two cross-attention results are computed and
    then overwritten by the self-attention results (lines 39-66), so the
    cross-attention parameters never influence the output."""

    def __init__(self, cfg, compute_dead_cross_attention: bool = True):
        super().__init__()
        self.vision_language_mhattn = OracleMHA(cfg.VISION_LANGUAGE_ATTENTION)
        self.language_vision_mhattn = OracleMHA(cfg.LANGUAGE_VISION_ATTENTION)
        self.vision_mhattn = OracleMHA(cfg.VISION_SELF_ATTENTION)
        self.language_mhattn = OracleMHA(cfg.LANGUAGE_SELF_ATTENTION)
        self.vision_pff = OraclePWFF(cfg.VISION_SELF_ATTENTION)
        self.language_pff = OraclePWFF(cfg.LANGUAGE_SELF_ATTENTION)
        self.compute_dead = compute_dead_cross_attention

    def forward(self, vision_features, vision_padding_mask, language_features, language_padding_mask, **kw):
        if self.compute_dead:
            self.vision_language_mhattn(vision_features, language_features, language_features, language_padding_mask)
            self.language_vision_mhattn(language_features, vision_features, vision_features, vision_padding_mask)
        v = self.vision_mhattn(vision_features, vision_features, vision_features, vision_padding_mask)
        l = self.language_mhattn(language_features, language_features, language_features, language_padding_mask)
        return self.vision_pff(v), self.language_pff(l)


class _PosLN(nn.Module):
    def _prologue(self, ln, x):
        # encoders.py:113,154,192-193,243-244: LN(x) + fp32 sinusoid table
        return ln(x) + sinusoid_positions(x.shape[1], x.shape[2]).to(x.device)[None]


class OracleEncoder(_PosLN):
    """Self-attention stack.  encoders.py:101-117."""

    def __init__(self, cfg):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.d_model = cfg.D_MODEL
        self.layers = nn.ModuleList([OracleEncoderLayer(cfg.SELF_ATTENTION) for _ in range(cfg.LAYERS)])

    def forward(self, features, padding_mask):
        out = self._prologue(self.layer_norm, features)
        for layer in self.layers:
            out = layer(out, out, out, padding_mask)
        return out


class OracleGuidedAttentionEncoder(_PosLN):
    """MCAN guided stack; both MHAs come from GUIDED_ATTENTION.  encoders.py:137-164."""

    def __init__(self, cfg):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.d_model = cfg.D_MODEL
        self.guided_attn_layers = nn.ModuleList(
            [OracleGuidedEncoderLayer(cfg.GUIDED_ATTENTION) for _ in range(cfg.LAYERS)])

    def forward(self, vision_features, vision_padding_mask, language_features, language_padding_mask):
        out = self._prologue(self.layer_norm, vision_features)
        for layer in self.guided_attn_layers:
            out = layer(out, language_features, language_features, vision_padding_mask, language_padding_mask)
        return out


class OracleCoAttentionEncoder(_PosLN):
    """ViLBERT-style chained co-attention.  encoders.py:166-224."""

    def __init__(self, cfg):
        super().__init__()
        self.vision_layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.language_layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.d_model = cfg.D_MODEL
        mk = lambda c: nn.ModuleList([OracleEncoderLayer(c) for _ in range(cfg.LAYERS)])
        self.vision_language_attn_layers = mk(cfg.VISION_LANGUAGE_ATTENTION)
        self.language_vision_attn_layers = mk(cfg.LANGUAGE_VISION_ATTENTION)
        self.vision_self_attn_layers = mk(cfg.VISION_SELF_ATTENTION)
        self.language_self_attn_layers = mk(cfg.LANGUAGE_SELF_ATTENTION)

    def forward(self, vision_features, vision_padding_mask, language_features, language_padding_mask):
        v = self._prologue(self.vision_layer_norm, vision_features)
        l = self._prologue(self.language_layer_norm, language_features)
        for vl, lv, vs, ls in zip(self.vision_language_attn_layers, self.language_vision_attn_layers,
                                  self.vision_self_attn_layers, self.language_self_attn_layers):
            v = vl(v, l, l, language_padding_mask)
            l = lv(l, v, v, vision_padding_mask)
            v = vs(v, v, v, vision_padding_mask)
            l = ls(l, l, l, language_padding_mask)
        return v, l


class OracleCrossModalityEncoder(_PosLN):
    """LXMERT-style.  encoders.py:226-253."""

    def __init__(self, cfg, compute_dead_cross_attention: bool = True):
        super().__init__()
        self.vision_layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.language_layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.d_model = cfg.D_MODEL
        self.layers = nn.ModuleList(
            [OracleCrossModalityEncoderLayer(cfg, compute_dead_cross_attention) for _ in range(cfg.LAYERS)])

    def forward(self, vision_features, vision_padding_mask, language_features, language_padding_mask):
        v = self._prologue(self.vision_layer_norm, vision_features)
        l = self._prologue(self.language_layer_norm, language_features)
        for layer in self.layers:
            v, l = layer(v, vision_padding_mask, l, language_padding_mask)
        return v, l


_ENCODERS = {
    "Encoder": OracleEncoder,
    "GuidedAttentionEncoder": OracleGuidedAttentionEncoder,
    "CoAttentionEncoder": OracleCoAttentionEncoder,
    "CrossModalityEncoder": OracleCrossModalityEncoder,
}


def build_oracle_encoder(cfg):
    return _ENCODERS[cfg.ARCHITECTURE](cfg)


# --------------------------------------------------------------------------
# decoder                                      models/modules/decoders.py
# --------------------------------------------------------------------------
class OracleUsualEmbedding(nn.Module):
    """Token embedding without pretrained vectors.  text_embeddings.py:56-80."""

    def __init__(self, cfg, vocab):
        super().__init__()
        self.padding_idx = vocab.padding_idx
        self.components = nn.Embedding(len(vocab), cfg.D_MODEL, vocab.padding_idx)

    def forward(self, tokens):
        pm = padding_mask(tokens, self.padding_idx)
        sm = sequential_mask(tokens.shape[-1])
        return self.components(tokens), (pm, sm)


class OracleDecoderLayer(_Stateful):
    """self-MHA -> enc-MHA -> FFN(ENC_ATTENTION cfg).  decoders.py:13-27."""

    def __init__(self, cfg):
        super().__init__()
        self.self_attn = OracleMHA(cfg.SELF_ATTENTION)
        self.enc_attn = OracleMHA(cfg.ENC_ATTENTION)
        self.pwff = OraclePWFF(cfg.ENC_ATTENTION)

    def forward(self, queries, keys, values, self_attention_mask, enc_attention_mask, **kw):
        x = self.self_attn(queries, queries, queries, self_attention_mask)
        x = self.enc_attn(x, keys, values, enc_attention_mask)
        return self.pwff(x)


class OracleDecoder(_Stateful):
    """decoders.py:29-76 (teacher-forced and stateful single-step)."""

    def __init__(self, cfg, vocab):
        super().__init__()
        self.d_model = cfg.D_MODEL
        self.max_len = vocab.max_answer_length
        self.padding_idx = vocab.padding_idx
        self.N = cfg.LAYERS
        self.word_emb = OracleUsualEmbedding(cfg.TEXT_EMBEDDING, vocab)
        self.pos_emb = nn.Embedding.from_pretrained(
            sinusoid_table(self.max_len + 1, cfg.D_MODEL, padding_idx=0), freeze=True)
        self.layers = nn.ModuleList([OracleDecoderLayer(cfg.ATTENTION) for _ in range(cfg.LAYERS)])
        self.fc = nn.Linear(cfg.D_MODEL, len(vocab), bias=False)
        self.register_state("running_mask_self_attention", torch.zeros((1, 1, 0)).bool())
        self.register_state("running_seq", torch.zeros((1,)).long())

    def forward(self, answer_tokens, encoder_features, encoder_attention_mask):
        b, t = answer_tokens.shape
        pm = padding_mask(answer_tokens, self.padding_idx)
        sam = self_attention_masks(pm, sequential_mask(t))
        if self._is_stateful:  # :55-57
            self.running_mask_self_attention = torch.cat([self.running_mask_self_attention, sam], -1)
            sam = self.running_mask_self_attention
        seq = torch.arange(1, t + 1).view(1, -1).expand(b, -1)
        seq = seq.masked_fill(pm[:, 0, 0, :] != 0, 0)
        if self._is_stateful:  # :61-63
            self.running_seq.add_(1)
            seq = self.running_seq
        emb, _ = self.word_emb(answer_tokens)
        out = emb + self.pos_emb(seq)
        for layer in self.layers:
            out = layer(out, encoder_features, encoder_features, sam, encoder_attention_mask)
        return F.log_softmax(_r(_lin(self.fc, out)), dim=-1)  # (the HIP path's vocabulary GEMM writes bf16 logits)


# --------------------------------------------------------------------------
# pointer scorers / feature embedding
# --------------------------------------------------------------------------
class OracleOcrPtrNet(nn.Module):
    """q W_q (k W_k)^T / sqrt(qk) + additive mask.  models/mmf_m4c.py:367-396."""

    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.query_key_size = query_key_size or hidden_size
        self.query = nn.Linear(hidden_size, self.query_key_size)
        self.key = nn.Linear(hidden_size, self.query_key_size)

    def forward(self, query_inputs, key_inputs, attention_mask):
        m = attention_mask.squeeze(1)
        q = _r(_lin(self.query, query_inputs))
        two_d = q.dim() == 2
        if two_d:
            q = q.unsqueeze(1)
        s = torch.matmul(q, _r(_lin(self.key, key_inputs)).transpose(-1, -2)) / math.sqrt(self.query_key_size) + m
        return s.squeeze(1) if two_d else s


class OracleDynamicPointerNetwork(nn.Module):
    """Bilinear scorer with a boolean -inf fill.  ``axis='key'`` follows
    models/m4c.py:19-33, ``axis='query'`` follows models/iterative_m4c.py:18-32."""

    def __init__(self, cfg, axis="key"):
        super().__init__()
        self.query = nn.Linear(cfg.D_MODEL, cfg.D_MODEL)
        self.key = nn.Linear(cfg.D_MODEL, cfg.D_MODEL)
        self.d_model = cfg.D_MODEL
        self.axis = axis

    def forward(self, query_inputs, key_inputs, attention_mask):
        s = torch.matmul(_r(_lin(self.query, query_inputs)),
                         _r(_lin(self.key, key_inputs)).transpose(-1, -2)) / math.sqrt(self.d_model)
        if self.axis == "key":
            return s.masked_fill(attention_mask.squeeze(1), float("-inf"))
        return s.masked_fill(attention_mask.squeeze(1).squeeze(1).unsqueeze(-1), float("-inf"))


class OracleFeatureEmbedding(nn.Module):
    """Linear + GELU + dropout and zero-row padding mask.
    models/modules/vision_embeddings.py:10-25 (restated from text: the module
    does not import under transformers 5)."""

    def __init__(self, cfg):
        super().__init__()
        self.proj = nn.Linear(cfg.D_FEATURE, cfg.D_MODEL)
        self.gelu = nn.GELU()
        self.dropout = nn.Dropout(cfg.DROPOUT)

    def forward(self, features):
        return _r(self.dropout(self.gelu(_lin(self.proj, features)))), padding_mask(features, 0)


def lstm_recurrence(x, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM (one layer, batch_first, zero initial state: text_embeddings.py:236,243) restated step by step:
    gates_t = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh, chunks (i, f, g, o) along the features;
    c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g);  h_t = sigmoid(o) tanh(c_t);  returns all h_t, (B, T, H).
    Padded positions are ordinary steps (the reference does not pack the sequence).
    emulate_bf16: weights, x and the recurrent operand h_{t-1} are bf16 (MFMA operands), accumulation / gates / cell
    state fp32, the returned h_t fp32; the gradient w.r.t. the pre-activations is stored in bf16 (it is the operand of
    the dx / dW products and of the recurrent product of the step before)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    if _EMU["on"]:
        w_ih = w_ih + (w_ih.bfloat16().float() - w_ih).detach()
        w_hh = w_hh + (w_hh.bfloat16().float() - w_hh).detach()
    xg = F.linear(_r(x), w_ih, b_ih)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    ys = []
    for t in range(T):
        h_op = h + (h.bfloat16().float() - h).detach() if _EMU["on"] else h  # (its gradient dh stays fp32)
        gates = _gr(xg[:, t] + F.linear(h_op, w_hh, b_hh))
        i, f, g, o = gates.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys, dim=1)


class OracleLSTMTextEmbedding(nn.Module):
    """Embedding -> Linear -> dropout -> LSTM, plus (padding, sequential) masks.
    models/modules/text_embeddings.py:222-246 (WORD_EMBEDDING: null form).  ``self.lstm`` only holds the parameters
    under torch's names (``lstm.weight_ih_l0`` ...); the recurrence is ``lstm_recurrence`` above, pinned against the
    reference class (which calls torch's own LSTM) by tests/golden/G17_lstm_text_embedding.npz."""

    def __init__(self, cfg, vocab):
        super().__init__()
        self.embedding = nn.Embedding(len(vocab), cfg.D_EMBEDDING, padding_idx=vocab.padding_idx)
        self.padding_idx = vocab.padding_idx
        self.proj = nn.Linear(cfg.D_EMBEDDING, cfg.D_MODEL)
        self.dropout = nn.Dropout(cfg.DROPOUT)
        self.lstm = nn.LSTM(input_size=cfg.D_MODEL, hidden_size=cfg.D_MODEL, batch_first=True)

    def forward(self, tokens):
        pad = padding_mask(tokens, self.padding_idx)
        seq = sequential_mask(tokens.shape[-1])
        x = self.dropout(_r(_lin(self.proj, self.embedding(tokens))))  # (HIP path: bf16 GEMM, bf16 output)
        m = self.lstm
        x = lstm_recurrence(x, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0)
        return x, (pad, seq)


class OracleMLP(nn.Module):
    """Attention-pooling scorer of MCAN: fc2(dropout(relu(fc1 x))) -> one logit per position.  models/mcan.py:12-25."""

    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.D_MODEL, cfg.D_MODEL)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(cfg.DROPOUT)
        self.fc2 = nn.Linear(cfg.D_MODEL, 1)

    def forward(self, x):
        # (HIP path: fc1 through the bf16 GEMM with a bf16 output, the D -> 1 product in fp32)
        return self.fc2(self.dropout(self.relu(_r(_lin(self.fc1, x)))))


class OracleMCAN(nn.Module):
    """models/mcan.py:27-81: embeddings -> SA stack -> guided stack -> softmax attention pooling over each sequence
    (dim=1, padded positions included, as the reference) -> LN(proj_v + proj_t) -> classifier -> log_softmax."""

    def __init__(self, cfg, vocab):
        super().__init__()
        self.d_model = cfg.D_MODEL
        self.text_embedding = {"LSTMTextEmbedding": OracleLSTMTextEmbedding,
                               "UsualEmbedding": OracleUsualEmbedding}[cfg.TEXT_EMBEDDING.ARCHITECTURE](
            cfg.TEXT_EMBEDDING, vocab)
        self.vision_embedding = OracleFeatureEmbedding(cfg.VISION_EMBEDDING)
        self.self_encoder = build_oracle_encoder(cfg.SELF_ENCODER)
        self.guided_encoder = build_oracle_encoder(cfg.GUIDED_ENCODER)
        self.vision_attr_reduce = OracleMLP(cfg.VISION_ATTR_REDUCE)
        self.text_attr_reduce = OracleMLP(cfg.TEXT_ATTR_REDUCE)
        self.vision_proj = nn.Linear(cfg.D_MODEL, cfg.D_MODEL)
        self.text_proj = nn.Linear(cfg.D_MODEL, cfg.D_MODEL)
        self.layer_norm = nn.LayerNorm(cfg.D_MODEL)
        self.classify = nn.Linear(cfg.D_MODEL, vocab.total_answers)

    def forward(self, input_features):
        v, vmask = self.vision_embedding(input_features.region_features)
        t, (tmask, _) = self.text_embedding(input_features.question_tokens)
        t = self.self_encoder(features=t, padding_mask=tmask)
        v = self.guided_encoder(vision_features=v, vision_padding_mask=vmask, language_features=t,
                                language_padding_mask=tmask)
        av = torch.softmax(self.vision_attr_reduce(v), dim=1)
        at = torch.softmax(self.text_attr_reduce(t), dim=1)
        # (HIP path: both stacks hand their outputs over in bf16 -- the image stack's input was bf16, the question stack takes
        # the LSTM's bf16 twin --, so the pooled sums are taken over bf16 features)
        wv, wt = (_r(v) * av).sum(dim=1), (_r(t) * at).sum(dim=1)
        # (HIP path: the two projections and the classifier are bf16 GEMMs with bf16 outputs; the second projection adds
        # the first in its epilogue, so the SUM is what is stored in bf16; LayerNorm statistics in fp32)
        out = self.layer_norm(_r(_r(_lin(self.vision_proj, wv)) + _lin(self.text_proj, wt)))
        return torch.log_softmax(_r(_lin(self.classify, _r(out))), dim=-1)


# --------------------------------------------------------------------------
# M4C multimodal transformer ("next" row 3)     models/mmf_m4c.py:282-364, 399-459
# --------------------------------------------------------------------------
# The layer body is THIRD-PARTY arithmetic: Hugging Face `transformers` BertEncoder (mmf_m4c.py:7-12,287; the
# reference pins no version -- installed here: 5.15.0).  Restated from its published algorithm (post-LN BERT layer)
# with HF's parameter names, and pinned against the installed library (tests/golden/G13_*, generated by running HF
# BertEncoder and the reference's own MMT.forward / PrevPredEmbeddings).
class _BertSelf(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.value = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.dropout = nn.Dropout(cfg.attention_probs_dropout_prob)


class _BertDenseLN(nn.Module):
    def __init__(self, d_in, cfg):
        super().__init__()
        self.dense = nn.Linear(d_in, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)


class _BertAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _BertSelf(cfg)
        self.output = _BertDenseLN(cfg.hidden_size, cfg)


class _BertIntermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class _BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.heads = cfg.num_attention_heads
        self.attention = _BertAttention(cfg)
        self.intermediate = _BertIntermediate(cfg)
        self.output = _BertDenseLN(cfg.intermediate_size, cfg)

    def forward(self, x, mask):
        a, B, S, H = self.attention, x.shape[0], x.shape[1], self.heads
        d = x.shape[-1] // H
        q = _r(_lin(a.self.query, x)).view(B, S, H, d).transpose(1, 2)
        k = _r(_lin(a.self.key, x)).view(B, S, H, d).transpose(1, 2)
        v = _r(_lin(a.self.value, x)).view(B, S, H, d).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * d ** -0.5
        if mask is not None:
            s = s + mask
        if _EMU["on"] and not (self.training and a.self.dropout.p > 0):
            e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
            ctx = torch.matmul(_r(e), v) / e.sum(dim=-1, keepdim=True)
        else:
            p = a.self.dropout(torch.softmax(s, dim=-1))
            ctx = torch.matmul(p, v)
        ctx = _r(ctx.transpose(1, 2).reshape(B, S, H * d))
        x = a.output.LayerNorm(_gr(a.output.dropout(_lin(a.output.dense, ctx)) + x))
        h = _r(torch.nn.functional.gelu(_lin(self.intermediate.dense, x)))
        return self.output.LayerNorm(_gr(self.output.dropout(_lin(self.output.dense, h)) + x))


class OracleBertEncoder(nn.Module):
    """transformers BertEncoder as MMT / TextBert call it: encoder(hidden, additive_mask (B,1,S,S) or (B,1,1,S),
    head_mask=[None]*L) -> (hidden,)."""

    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        for layer in self.layer:
            hidden_states = layer(hidden_states, attention_mask)
        return (hidden_states,)


def batch_gather(x, inds):
    """x (B,L,D), inds (B,T) -> (B,T,D).  mmf_m4c.py:448-459."""
    B, L, D = x.shape
    flat = (torch.arange(B, device=inds.device) * L).unsqueeze(-1) + inds
    return torch.nn.functional.embedding(flat, x.reshape(B * L, D))


class OraclePrevPredEmbeddings(nn.Module):
    """mmf_m4c.py:399-446."""

    def __init__(self, cfg):
        super().__init__()
        self.position_embeddings = nn.Embedding(100, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(5, cfg.hidden_size)
        self.ans_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.ocr_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.emb_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.emb_dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, ans_emb, ocr_emb, prev_inds):
        B, T = prev_inds.shape
        ans_num = ans_emb.size(0)
        cat = torch.cat([self.ans_layer_norm(ans_emb).unsqueeze(0).expand(B, -1, -1), self.ocr_layer_norm(ocr_emb)], 1)
        raw = batch_gather(cat, prev_inds)
        pos = self.position_embeddings(torch.arange(T, device=ocr_emb.device).unsqueeze(0).expand(B, T))
        typ = self.token_type_embeddings(prev_inds.ge(ans_num).long())
        return raw + self.emb_dropout(self.emb_layer_norm(pos + typ))


class OracleMMT(nn.Module):
    """mmf_m4c.py:282-364: [txt; obj; ocr; dec] through the BERT encoder under a prefix-LM mask (every position sees
    the encoding steps per their padding masks; decoding steps see each other causally)."""

    def __init__(self, cfg):
        super().__init__()
        self.prev_pred_embeddings = OraclePrevPredEmbeddings(cfg)
        self.encoder = OracleBertEncoder(cfg)

    def forward(self, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, fixed_ans_emb, prev_inds):
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds)
        T = dec_emb.size(1)
        dec_mask = torch.zeros(dec_emb.size(0), 1, 1, T, dtype=torch.float32, device=dec_emb.device)
        x = torch.cat([txt_emb, obj_emb, ocr_emb, dec_emb], dim=1)
        mask = torch.cat([txt_mask, obj_mask, ocr_mask, dec_mask], dim=-1)
        S = mask.size(-1)
        ext = mask.repeat(1, 1, S, 1)
        ext[:, :, -T:, -T:] = sequential_mask(T).to(ext.device)
        out = self.encoder(x, ext, head_mask=None)[0]
        nt, no, nc = txt_mask.size(-1), obj_mask.size(-1), ocr_mask.size(-1)
        return {"mmt_seq_output": out, "mmt_txt_output": out[:, :nt], "mmt_ocr_output": out[:, nt + no:nt + no + nc],
                "mmt_dec_output": out[:, -T:]}


class OracleM4CDecodingHead(nn.Module):
    """classifier || OcrPtrNet scores and the greedy decoding loop.  mmf_m4c.py:221-256."""

    def __init__(self, hidden_size, num_choices):
        super().__init__()
        self.classifier = nn.Linear(hidden_size, num_choices)
        self.ocr_ptr_net = OracleOcrPtrNet(hidden_size)

    def scores(self, mmt_results, ocr_mask):  # :221-229
        dec, ocr = mmt_results["mmt_dec_output"], mmt_results["mmt_ocr_output"]
        return torch.cat([self.classifier(dec), self.ocr_ptr_net(dec, ocr, ocr_mask)], dim=-1)

    @torch.no_grad()
    def greedy_decode(self, mmt, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, max_iter, bos_idx, eos_idx):
        B = txt_emb.shape[0]
        prev_inds = torch.zeros((B, max_iter)).long()  # :241-242
        prev_inds[:, 0] = bos_idx
        last_ids = torch.zeros((B,))  # :245
        scores, passes, trace = None, 0, []
        for ith in range(max_iter):  # :246-256
            trace.append(prev_inds.clone())
            res = mmt(txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, self.classifier.weight, prev_inds)
            scores = self.scores(res, ocr_mask)
            passes += 1
            argmax_inds = scores.argmax(dim=-1)
            prev_inds[:, 1:] = argmax_inds[:, :-1]
            last_ids = torch.where(last_ids == eos_idx, last_ids, argmax_inds[:, ith].to(last_ids.dtype))
            if last_ids.mean() == eos_idx:
                break
        return scores, prev_inds, passes, trace


# --------------------------------------------------------------------------
# beam search (row f1 / N1)   models/modules/beam_search.py:19-118, models/base_transformer.py:31-54
# --------------------------------------------------------------------------
def oracle_beam_search(step, reorder, b_s: int, max_len: int, eos_idx: int, beam_size: int, out_size: int = 1,
                       trace: Optional[list] = None):
    """The reference's batched beam search, restated (pinned by tests/golden/G16_beam_search.npz = the reference's own
    ``BeamSearch`` over its own ``Decoder``).

    ``step(t, prev_words) -> (b_s * cur_beam, 1, |V|)`` log-probabilities is ``model.step`` (base_transformer.py:31-44);
    ``reorder(fn)`` is ``model.apply_to_states`` (beam_search.py:61).  Per step (beam_search.py:41-83): candidates =
    running score + word log-probabilities; a sequence that has produced <eos> keeps its score on word 0 and gets -999
    on every other word (:49-55); the ``beam`` best of the cur_beam * |V| candidates by a full descending sort (:36-39);
    beam = index // |V|, word = index % |V| (:58-59); every state buffer, the mask and both histories follow their beams
    (:61-66,78-81); the recorded word score is the MASKED word log-probability, i.e. 0 once a sequence is finished
    (:52,74-77).  Finally the beams are sorted by score and the best ``out_size`` returned (:99-118).  ``trace`` (a list)
    receives per step what the search saw and chose: the step's log-probabilities, the source beam and the word of every
    new beam -- so that a test can drive another decoder through the SAME choices."""
    seq_mask = torch.ones(b_s, beam_size, 1)
    seq_logprob = torch.zeros(b_s, 1, 1)
    words_hist, score_hist, prev = [], [], None
    for t in range(max_len):
        cur = 1 if t == 0 else beam_size
        word_lp = word_lp_raw = step(t, prev).view(b_s, cur, -1)
        n_words = word_lp.shape[-1]
        cand = seq_logprob + word_lp
        if t > 0:
            alive = (prev.view(b_s, cur) != eos_idx).float().unsqueeze(-1)
            seq_mask = seq_mask * alive
            word_lp = word_lp * seq_mask
            frozen = seq_logprob.expand(b_s, cur, n_words).clone()
            frozen[:, :, 1:] = -999
            cand = seq_mask * cand + frozen * (1 - seq_mask)
        best, flat = torch.sort(cand.view(b_s, -1), -1, descending=True)
        best, flat = best[:, :beam_size], flat[:, :beam_size]
        from_beam = torch.div(flat, n_words, rounding_mode="trunc")
        word = flat - from_beam * n_words

        def follow(s, from_beam=from_beam, cur=cur):  # beam_search.py:19-34
            tail = list(s.shape[1:])
            idx = from_beam.view(b_s, beam_size, *([1] * len(tail))).expand(b_s, beam_size, *tail)
            return torch.gather(s.view(b_s, cur, *tail), 1, idx).view(-1, *tail)
        reorder(follow)
        if trace is not None:
            trace.append(dict(t=t, cur=cur, step_logp=word_lp_raw, from_beam=from_beam.clone(), word=word.clone()))
        seq_logprob = best.unsqueeze(-1)
        seq_mask = torch.gather(seq_mask, 1, from_beam.unsqueeze(-1))
        pick = from_beam.unsqueeze(-1)
        words_hist = [torch.gather(w, 1, pick) for w in words_hist] + [word.unsqueeze(-1)]
        chosen = torch.gather(word_lp.expand(b_s, cur, n_words), 1, pick.expand(b_s, beam_size, n_words))
        chosen = torch.gather(chosen, 2, word.unsqueeze(-1))
        score_hist = [torch.gather(s, 1, pick) for s in score_hist] + [chosen]
        prev = word.reshape(-1, 1)
    _, order = torch.sort(seq_logprob, 1, descending=True)
    order = order.expand(b_s, beam_size, max_len)
    tokens = torch.gather(torch.cat(words_hist, -1), 1, order)[:, :out_size]
    scores = torch.gather(torch.cat(score_hist, -1), 1, order)[:, :out_size]
    if out_size == 1:
        tokens, scores = tokens.squeeze(1), scores.squeeze(1)
    return tokens, scores


def oracle_generate(decoder, encoder_features, encoder_mask, bos_idx: int, eos_idx: int, beam_size: int,
                    max_len: Optional[int] = None, out_size: int = 1, trace: Optional[list] = None):
    """``BaseTransformer.beam_search`` (base_transformer.py:31-54) over a stateful decoder: inside ``statefulness(b_s)``
    the encoder features and their mask are STATES of the model, so they are expanded to the beams by the same reorder
    as the decoder's caches (beam_search.py:61); step 0 feeds <bos>, later steps the previously selected words."""
    b_s = encoder_features.shape[0]
    held = {"enc": encoder_features, "mask": encoder_mask}

    def step(t, prev):
        tokens = torch.full((b_s, 1), bos_idx, dtype=torch.long) if t == 0 else prev
        return decoder(tokens, held["enc"], held["mask"])

    def reorder(fn):
        held["enc"], held["mask"] = fn(held["enc"]), fn(held["mask"])
        decoder.apply_to_states(fn)
    with torch.no_grad(), decoder.statefulness(b_s):
        return oracle_beam_search(step, reorder, b_s, max_len or decoder.max_len, eos_idx, beam_size, out_size, trace)


# --------------------------------------------------------------------------
# training step (row T)      tasks/classification_task.py:120-139, base_task.py:46-76
# --------------------------------------------------------------------------
def noam_lambda(step: int, d_model: int, warmup: int) -> float:
    """tasks/base_task.py:73-76: d^-.5 * min(s^-.5, s * warmup^-1.5), s = step+1."""
    s = step + 1
    return (d_model ** -0.5) * min(s ** -0.5, s * warmup ** -1.5)


def oracle_train_step(params, loss_fn, optim, scheduler=None):
    """One reference-ordered step: forward -> zero_grad -> backward -> Adam step
    -> loss.item() -> scheduler.step()."""
    loss = loss_fn()
    optim.zero_grad()
    loss.backward()
    optim.step()
    val = loss.item()
    if scheduler is not None:
        scheduler.step()
    return val
