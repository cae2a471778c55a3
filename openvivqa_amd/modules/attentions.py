"""Attention modules backed by the HIP kernels (drop-in for models/modules/attentions.py).

``ScaledDotProductAttention`` keeps the reference's parameters (fc_q/fc_k/fc_v/
fc_o, xavier weights, zero biases -- attentions.py:16-44) and returns
``(out, att)``; ``MultiHeadAttention`` keeps dropout -> residual -> post-LN,
the optional AoA gate and the running K/V state (attentions.py:293-339) but runs
as ONE fused block: packed QKV projection -> attention core -> output
projection with bias+dropout+residual epilogue -> LayerNorm.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fn
from .. import runtime as rt
from ..builders.attention_builder import META_ATTENTION, build_attention
from .containers import Module


def _as_mask(mask):
    if mask is None:
        return None
    if mask.dtype != torch.float32:
        mask = mask.float()
    while mask.dim() < 4:
        mask = mask.unsqueeze(0)
    return mask


@META_ATTENTION.register()
class ScaledDotProductAttention(nn.Module):
    """softmax(Q K^T / sqrt(d_k) + mask) V with learned projections."""

    def __init__(self, config):
        super().__init__()
        d_model, h, d_k, d_v = config.D_MODEL, config.HEAD, config.D_KEY, config.D_VALUE
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.init_weights()

    def init_weights(self):
        for lin in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0)

    def _ovqa_param_groups(self):
        # adjacency in the arena => the fused QKV projection reads one [3*H*d, D] matrix
        return [[self.fc_q.weight, self.fc_k.weight, self.fc_v.weight],
                [self.fc_q.bias, self.fc_k.bias, self.fc_v.bias]]

    def forward(self, queries, keys, values, attention_mask=None, **kwargs):
        """Returns (out, att) like the reference; ``att`` (B,H,nq,nk) is produced by the
        kernel on request and is not differentiable."""
        arena = rt.ensure_arena(self)
        T, in_dtype = arena.compute_dtype, queries.dtype
        same_kv = Fn.same_tensor(keys, values)
        same_all = same_kv and Fn.same_tensor(queries, keys)
        queries = queries.to(T)
        keys = queries if same_all else keys.to(T)
        values = keys if same_kv else values.to(T)
        q = Fn.linear(queries, self.fc_q, arena)
        k = Fn.linear(keys, self.fc_k, arena)
        v = Fn.linear(values, self.fc_v, arena)
        o, att = Fn.attention_core(q, k, v, _as_mask(attention_mask), self.h, need_att=True)
        out = Fn.linear(o, self.fc_o, arena)
        return out.to(in_dtype), att.to(in_dtype)


@META_ATTENTION.register()
class AugmentedMemoryScaledDotProductAttention(nn.Module):
    """Attention over [keys; m learned memory slots] (attentions.py:129-205, M2-transformer style):
    K' = [fc_k(keys); sqrt(d_k) m_k], V' = [fc_v(values); sqrt(m) m_v]; the additive mask covers the real keys
    only.  Same attention kernel, nk + m key columns."""

    def __init__(self, config):
        super().__init__()
        d_model, h, d_k, d_v, m = config.D_MODEL, config.HEAD, config.D_KEY, config.D_VALUE, config.MEMORY
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.m_k = nn.Parameter(torch.empty(1, m, h * d_k))
        self.m_v = nn.Parameter(torch.empty(1, m, h * d_v))
        self.d_model, self.d_k, self.d_v, self.h, self.m = d_model, d_k, d_v, h, m
        self.init_weights()

    def init_weights(self):
        for lin in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0)
        nn.init.normal_(self.m_k, 0, 1 / self.d_k)
        nn.init.normal_(self.m_v, 0, 1 / self.m)

    def forward(self, queries, keys, values, attention_mask=None, **kwargs):
        arena = rt.ensure_arena(self)
        T, in_dtype = arena.compute_dtype, queries.dtype
        B, nk = queries.shape[0], keys.shape[1]
        q = Fn.linear(queries.to(T), self.fc_q, arena)
        k = Fn.linear(keys.to(T), self.fc_k, arena)
        v = Fn.linear(values.to(T), self.fc_v, arena)
        m_k = (self.d_k ** 0.5 * self.m_k).to(T).expand(B, self.m, self.h * self.d_k)
        m_v = (self.m ** 0.5 * self.m_v).to(T).expand(B, self.m, self.h * self.d_v)
        k, v = torch.cat([k, m_k], 1), torch.cat([v, m_v], 1)
        mask = _as_mask(attention_mask)
        if mask is not None:  # memory slots are never masked
            mask = torch.cat([mask, mask.new_zeros(*mask.shape[:-1], self.m)], dim=-1).contiguous()
        o, att = Fn.attention_core(q, k, v, mask, self.h, need_att=True)
        out = Fn.linear(o, self.fc_o, arena)
        return out.to(in_dtype), att.to(in_dtype)


def _split_extras(args, kwargs, key, is_signal):
    """The reference's variant attentions take their extra signal as the 4th POSITIONAL argument
    (``forward(q, k, v, boxes, attention_mask=None)``) while MultiHeadAttention passes the mask there
    (attentions.py:326): accept both orders -- an extra positional tensor is the signal if ``is_signal`` says so,
    else the mask."""
    signal, mask = kwargs.pop(key, None), kwargs.pop("attention_mask", None)
    for a in args:
        if a is None:
            continue
        if signal is None and is_signal(a):
            signal = a
        else:
            mask = a
    return signal, mask


@META_ATTENTION.register()
class AugmentedGeometryScaledDotProductAttention(nn.Module):
    """Geometry-aware self-attention over boxes (attentions.py:62-137, models/utils.py:102-162):
    softmax(log(clamp(relu(fc_g(geometry)), 1e-6)) + Q K^T / sqrt(d_k) + mask) V, one fc_g per head.

    Built WORKING: upstream's forward ends in two lines that read an undefined ``att`` (attentions.py:126-137,
    NameError on every call) and adds the mask to that same undefined name; the intended computation (the
    log-geometry-biased softmax ``mn`` applied to V, returned with its weights) is what runs here."""

    def __init__(self, config):
        super().__init__()
        d_model, h, d_k, d_v = config.D_MODEL, config.HEAD, config.D_KEY, config.D_VALUE
        self.trignometric_embedding = config.TRIGNOMETRIC_EMBEDDING
        self.d_g = d_model // h if self.trignometric_embedding else 4
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.fc_gs = nn.ModuleList([nn.Linear(self.d_g, 1) for _ in range(h)])
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.init_weights()

    def init_weights(self):
        for lin in (self.fc_q, self.fc_k, self.fc_v, self.fc_o, *self.fc_gs):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0)

    def forward(self, queries, keys, values, *args, **kwargs):
        from ..utils import box_relational_embedding
        boxes, attention_mask = _split_extras(args, kwargs, "boxes", lambda t: t.dim() == 3 and t.shape[-1] == 4)
        arena = rt.ensure_arena(self)
        T, in_dtype = arena.compute_dtype, queries.dtype
        emb = box_relational_embedding(boxes.float(), dim_g=self.d_g, trignometric_embedding=self.trignometric_embedding)
        wg = torch.cat([g.weight for g in self.fc_gs], 0).float()  # (H, d_g): the H heads' 1-output linears at once
        bg = torch.cat([g.bias for g in self.fc_gs], 0).float()
        geo = torch.relu(emb @ wg.t() + bg).permute(0, 3, 1, 2)  # (B, H, nk, nk)
        bias = torch.log(torch.clamp(geo, min=1e-6))
        mask = _as_mask(attention_mask)
        if mask is not None:
            bias = bias + mask
        q = Fn.linear(queries.to(T), self.fc_q, arena)
        k = Fn.linear(keys.to(T), self.fc_k, arena)
        v = Fn.linear(values.to(T), self.fc_v, arena)
        o, att = Fn.attention_core_with_bias(q, k, v, bias.contiguous(), self.h)
        out = Fn.linear(o, self.fc_o, arena)
        return out.to(in_dtype), att.to(in_dtype)


@META_ATTENTION.register()
class AdaptiveScaledDotProductAttention(nn.Module):
    """Adaptive attention (attentions.py:210-291): every query i sees the nk keys plus ONE extra key/value of its
    own, the projected language signal s_i = fc_s(language_signals)_i (score q_i . s_i / sqrt(d_k), value s_i).

    The extra column differs per query, so it cannot be appended to K/V; instead the attention kernel runs over the
    keys and returns its output, probabilities and log-sum-exp (all differentiable), and the extra column is merged
    in closed form: with sigma_i = sigmoid(lse_i - e_i), out_i = sigma_i o_i + (1 - sigma_i) s_i and the combined
    weights are [sigma_i p_i, 1 - sigma_i].  Returns (out, list of nq (B,H,1,nk+1) weight tensors) like upstream."""

    def __init__(self, config):
        super().__init__()
        d_model, h, d_k, d_v = config.D_MODEL, config.HEAD, config.D_KEY, config.D_VALUE
        assert d_k == d_v, "the language signal is used as key AND value (attentions.py:262,274): d_k must equal d_v"
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_s = nn.Linear(d_model, h * d_k)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.dropout = nn.Dropout(config.DROPOUT)  # constructed and never applied upstream (attentions.py:228)
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.init_weights()

    def init_weights(self):
        for lin in (self.fc_q, self.fc_k, self.fc_v, self.fc_o, self.fc_s):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0)

    def forward(self, queries, keys, values, *args, **kwargs):
        d_model = self.d_model
        signals, attention_mask = _split_extras(args, kwargs, "language_signals",
                                                lambda t: t.dim() == 3 and t.shape[-1] == d_model)
        arena = rt.ensure_arena(self)
        T, in_dtype = arena.compute_dtype, queries.dtype
        B, nq = queries.shape[:2]
        q = Fn.linear(queries.to(T), self.fc_q, arena)
        k = Fn.linear(keys.to(T), self.fc_k, arena)
        v = Fn.linear(values.to(T), self.fc_v, arena)
        s = Fn.linear(signals.to(T), self.fc_s, arena)
        o, att, lse = Fn.attention_core(q, k, v, _as_mask(attention_mask), self.h, need_att=True, need_lse=True)
        qh = q.float().view(B, nq, self.h, self.d_k)
        sh = s.float().view(B, nq, self.h, self.d_k)
        e = (qh * sh).sum(-1) / (self.d_k ** 0.5)              # (B, nq, H): score of query i against ITS signal
        sigma = torch.sigmoid(lse.float().transpose(1, 2) - e)  # weight kept by the nk keys
        out = sigma.unsqueeze(-1) * o.float().view(B, nq, self.h, self.d_v) + (1 - sigma).unsqueeze(-1) * sh
        out = Fn.linear(out.reshape(B, nq, self.h * self.d_v).to(T), self.fc_o, arena)
        sig = sigma.transpose(1, 2).unsqueeze(-1)               # (B, H, nq, 1)
        combined = torch.cat([att.float() * sig, 1 - sig], dim=-1).to(in_dtype)
        return out.to(in_dtype), [combined[:, :, i:i + 1] for i in range(nq)]


class MultiHeadAttention(Module):
    """Multi-head attention block with dropout, residual connection and post-LayerNorm."""

    def __init__(self, config):
        super().__init__()
        d_model = config.D_MODEL
        self.use_aoa = config.USE_AOA
        if self.use_aoa:
            self.informative_attention = nn.Linear(2 * d_model, d_model)
            self.gated_attention = nn.Linear(2 * d_model, d_model)
        self.attention = build_attention(config)  # registry call, attentions.py:309
        self.dropout = nn.Dropout(p=config.DROPOUT)
        self.layer_norm = nn.LayerNorm(d_model)
        self.can_be_stateful = config.CAN_BE_STATEFUL
        if self.can_be_stateful:
            self.register_state("running_keys", torch.zeros((0, d_model)))
            self.register_state("running_values", torch.zeros((0, d_model)))
        self._site = rt.new_dropout_site()
        self._drop_decode_caches()

    def forward(self, queries, keys, values, attention_mask, projected_kv=None, encoder_group: int = 1, **kwargs):
        """``encoder_group`` (an addition, default 1): query rows r*g .. r*g+g-1 attend to keys / values row r -- the
        beams of a sample over that sample's encoder features, handed over UN-expanded (the reference gathers a copy
        per beam, beam_search.py:19-34,61).  The relation is stated by the caller, never inferred."""
        arena = rt.ensure_arena(self)
        T = arena.compute_dtype
        if projected_kv is not None and (type(self.attention) is not ScaledDotProductAttention
                                         or (self.can_be_stateful and self._is_stateful)):
            projected_kv = None
        same_kv = Fn.same_tensor(keys, values)
        same_all = same_kv and Fn.same_tensor(queries, keys)
        queries = Fn.to_compute(queries, T)
        mask = _as_mask(attention_mask)
        if (self._is_stateful and not self.can_be_stateful and not torch.is_grad_enabled() and same_kv and not same_all
                and type(self.attention) is ScaledDotProductAttention and queries.shape[1] == 1 and queries.is_cuda
                and self.attention.d_k == self.attention.d_v and self.attention.d_k in (32, 64, 128)
                and keys.shape[1] <= 512 and (mask is None or (mask.shape[-2] == 1 and mask.shape[1] == 1))
                and queries.shape[0] == keys.shape[0] * encoder_group
                and (mask is None or mask.shape[0] in (1, queries.shape[0]))):
            return self._encoder_step(arena, queries, keys, mask, encoder_group)  # (keys as given: the cache is keyed on that tensor)
        if encoder_group > 1:  # any other path: the reference's per-beam copies
            keys = keys.repeat_interleave(encoder_group, 0)
            values = keys if same_kv else values.repeat_interleave(encoder_group, 0)
            if mask is not None and mask.shape[0] * encoder_group == queries.shape[0]:
                mask = mask.repeat_interleave(encoder_group, 0)
        keys = queries if same_all else keys.to(T)
        values = keys if same_kv else values.to(T)
        if self.can_be_stateful and self._is_stateful:
            if (type(self.attention) is ScaledDotProductAttention and not torch.is_grad_enabled()
                    and self.attention.h * self.attention.d_k == self.attention.h * self.attention.d_v == keys.shape[-1]):
                return self._stateful_step(arena, queries, keys, values, mask)
            # reference behaviour (attentions.py:320-325): cache the raw inputs, re-project the whole prefix
            self.running_keys = torch.cat([self.running_keys.to(T), keys], 1)
            keys = self.running_keys
            self.running_values = torch.cat([self.running_values.to(T), values], 1)
            values = self.running_values
        if type(self.attention) is ScaledDotProductAttention:
            params = list(self.attention.parameters()) + list(self.layer_norm.parameters())
            st = dict(arena=arena, att=self.attention, ln=self.layer_norm, params=params,
                      drop=rt.dropout_spec(self.dropout.p, self._site, self.training, queries.device))
            out = Fn.mha_block(queries, keys, values, mask, st, projected_kv=projected_kv)
        else:  # other registered attention kernels: unfused composition
            out, _ = self.attention(queries, keys, values, mask, **kwargs)
            out = Fn.prologue(queries + self.dropout(out.to(T)), self.layer_norm, None, arena, T)
        return self._aoa(arena, queries, out)

    def _aoa(self, arena, queries, out):
        if self.use_aoa:  # attentions.py:333-337
            z = torch.cat([queries, out], dim=-1)
            i = Fn.linear(z, self.informative_attention, arena)
            g = Fn.linear(z, self.gated_attention, arena)
            out = i * torch.sigmoid(g)
        return out

    # ------------------------------------------------------------------ autoregressive decoding
    DECODE_CAPACITY = 32  # keys an in-place cache is first sized for (Decoder sets max_len + 1); doubled when full

    def enable_statefulness(self, batch_size: int) -> None:
        super().enable_statefulness(batch_size)
        self._drop_decode_caches()

    def disable_statefulness(self) -> None:
        super().disable_statefulness()
        self._drop_decode_caches()

    def _drop_decode_caches(self):
        self._kv = None       # self-attention: [k cache, v cache] (R, capacity, H*d) written in place, + spare pair
        self._kv_spare = None
        self._enc = None      # encoder attention: (source identity, packed K|V (rows, nk, 2*H*d)) of the encoder features

    def _seat_cache(self, like_q):
        """The in-place K / V caches behind ``running_keys / running_values``.  The state buffers are live-prefix VIEWS
        of the caches, so the reference's protocol keeps working: ``apply_to_states(fn)`` hands fn the (R, t, H*d)
        prefix and stores what it returns -- which is then no longer our view and is copied into a cache again here."""
        rk, rv = self._buffers["running_keys"], self._buffers["running_values"]
        kv = getattr(self, "_kv", None)
        n = rk.shape[1] if rk.dim() == 3 else 0
        R, F = like_q.shape[0], like_q.shape[-1]
        if (kv is not None and kv[0].dtype == like_q.dtype and kv[0].shape[0] == R and n < kv[0].shape[1]
                and rk.dim() == 3 and rk.shape[0] == R and rk.data_ptr() == kv[0].data_ptr() and rv.data_ptr() == kv[1].data_ptr()):
            return kv, n
        cap = max(int(getattr(self, "decode_capacity", self.DECODE_CAPACITY)), 2 * (n + 1))
        new = [torch.empty(R, cap, F, dtype=like_q.dtype, device=like_q.device) for _ in range(2)]
        if n > 0:
            assert rk.shape[0] == R, "the state buffers' batch dimension must follow the queries'"
            new[0][:, :n].copy_(rk)
            new[1][:, :n].copy_(rv)
        self._kv, self._kv_spare = new, None
        return new, n

    def _cache_destination(self, name, rows):
        """reorder_states: where the gathered rows of ``running_keys / running_values`` go -- the live prefix of the
        SPARE cache buffer (allocated once per decode), so that a beam reorder moves the live keys once and the next
        step appends in place again.  None when the state is not backed by a cache."""
        kv = getattr(self, "_kv", None)
        if kv is None or name not in ("running_keys", "running_values"):
            return None
        i = 0 if name == "running_keys" else 1
        s = self._buffers[name]
        if s.dim() != 3 or s.data_ptr() != kv[i].data_ptr():
            return None
        spare = getattr(self, "_kv_spare", None)
        if spare is None or spare[0].shape[0] != rows or spare[0].shape[1:] != kv[0].shape[1:] or spare[0].dtype != kv[0].dtype:
            spare = [torch.empty((rows,) + tuple(kv[0].shape[1:]), dtype=kv[0].dtype, device=kv[0].device) for _ in range(2)]
            self._kv_spare = spare
        return spare[i][:, :s.shape[1]]

    def _caches_reordered(self):
        rk, spare = self._buffers.get("running_keys"), getattr(self, "_kv_spare", None)
        if spare is not None and rk is not None and rk.dim() == 3 and rk.data_ptr() == spare[0].data_ptr():
            self._kv, self._kv_spare = spare, self._kv  # the gathered prefix lives in the spare pair: swap roles

    def _finish_step(self, arena, queries, o):
        """fc_o + residual + LayerNorm (+ AoA) of a decoding step, as the fused block computes them."""
        from .. import ops
        from .._lib import EPI_BIAS_RESIDUAL
        a, ln = self.attention, self.layer_norm
        if o.dtype == torch.bfloat16:  # fp32 residual stream, as in the fused block
            pre = ops.linear_fwd_res32(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias),
                                       Fn.residual_of(queries))
            gamma, beta = arena.master_of(ln.weight), arena.master_of(ln.bias)
            out, mean, rstd = ops.layernorm_fwd(pre, gamma, beta, ln.eps, out_dtype=o.dtype)
            Fn.attach_residual(out, ops.LnRef(pre, mean, rstd, gamma, beta, ln.eps))
        else:
            pre = ops.linear_fwd(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias), EPI_BIAS_RESIDUAL,
                                 residual=queries.contiguous())
            out, _, _ = ops.layernorm_fwd(pre, arena.master_of(ln.weight), arena.master_of(ln.bias), ln.eps,
                                          save_stats=False)
        return self._aoa(arena, queries, out)

    @staticmethod
    def _row_mask(mask, R, n):
        """(R, >= n) fp32 rows of an additive (R | 1, 1, 1, n) mask, or None."""
        if mask is None:
            return None
        m = mask.reshape(mask.shape[0], mask.shape[-1])
        if m.shape[0] != R:
            m = m.expand(R, m.shape[1])
        if m.stride(1) == 1 and m.stride(0) >= m.shape[1]:  # dense rows (the live prefix of an in-place mask buffer)
            return m
        return m.contiguous()

    def _stateful_step(self, arena, queries, keys, values, mask):
        """Autoregressive decode step with a PROJECTED, IN-PLACE K/V cache.

        The reference appends the raw d_model inputs to ``running_keys/values`` and re-applies fc_k / fc_v to the whole
        prefix at every step (attentions.py:320-325: O(T^2) projections per sequence, a torch.cat per buffer and step).
        Here only the new position is projected, straight into its slot of a pre-allocated (R, capacity, H*d) cache
        (no concatenation), the state buffers are live-prefix views of the caches -- same names, same (R, t, H*d) shape,
        so ``apply_to_states`` and ``reorder_states`` work -- and the attention is the single-query kernel
        (ovqa_attention_decode: one wave per (row, head), K / V streamed once).  Same outputs."""
        from .. import ops
        a = self.attention
        nq = queries.shape[1]
        single = (nq == 1 and keys.shape[1] == 1 and queries.is_cuda and a.d_k == a.d_v and a.d_k in (32, 64, 128)
                  and (mask is None or (mask.shape[-2] == 1 and mask.shape[1] == 1)))
        if single and Fn.same_tensor(queries, keys) and Fn.same_tensor(keys, values):
            # self-attention over the new position: fc_q / fc_k / fc_v in ONE launch, k and v straight into their cache slots
            try:
                w3 = arena.packed([a.fc_q.weight, a.fc_k.weight, a.fc_v.weight])
                b3 = arena.packed([a.fc_q.bias, a.fc_k.bias, a.fc_v.bias], "master")
            except RuntimeError:  # (not adjacent in this arena: the three launches below)
                w3 = None
            if w3 is not None and w3.dtype == queries.dtype:
                x = queries.reshape(queries.shape[0], -1)
                q = torch.empty(x.shape[0], a.h * a.d_k, dtype=x.dtype, device=x.device)
                (kc, vc), n = self._seat_cache(q)
                ops.linear_fwd_split3(x, w3, b3, (q, kc[:, n], vc[:, n]))
                n += 1
                self._buffers["running_keys"], self._buffers["running_values"] = kc[:, :n], vc[:, :n]
                o = ops.attention_decode(q.view(x.shape[0], 1, -1), kc, vc, n, a.h, mask=self._row_mask(mask, x.shape[0], n))
                return self._finish_step(arena, queries, o)
        q = ops.linear_fwd(queries.contiguous(), arena.compute(a.fc_q.weight), arena.master_of(a.fc_q.bias))
        if not single:  # several new positions at once (teacher-forced prefixes): the general kernel on a grown cache
            k_new = ops.linear_fwd(keys.contiguous(), arena.compute(a.fc_k.weight), arena.master_of(a.fc_k.bias))
            v_new = ops.linear_fwd(values.contiguous(), arena.compute(a.fc_v.weight), arena.master_of(a.fc_v.bias))
            self.running_keys = torch.cat([self.running_keys.to(q.dtype), k_new], 1)
            self.running_values = torch.cat([self.running_values.to(q.dtype), v_new], 1)
            self._kv = None
            o, _, _ = ops.attention_fwd(q, self.running_keys, self.running_values, mask, a.h, save_lse=False)
            return self._finish_step(arena, queries, o)
        (kc, vc), n = self._seat_cache(q)
        ops.linear_fwd(keys.reshape(keys.shape[0], -1), arena.compute(a.fc_k.weight), arena.master_of(a.fc_k.bias),
                       out=kc[:, n])
        ops.linear_fwd(values.reshape(values.shape[0], -1), arena.compute(a.fc_v.weight), arena.master_of(a.fc_v.bias),
                       out=vc[:, n])
        n += 1
        self._buffers["running_keys"], self._buffers["running_values"] = kc[:, :n], vc[:, :n]
        o = ops.attention_decode(q, kc, vc, n, a.h, mask=self._row_mask(mask, q.shape[0], n))
        return self._finish_step(arena, queries, o)

    def _encoder_step(self, arena, queries, keys, mask, group=1):
        """Decoding step of an attention over the ENCODER positions (the non-stateful enc_attn of a DecoderLayer inside
        a stateful decoder, decoders.py:21-27): the reference re-applies fc_k / fc_v to all encoder positions at every
        step and for every beam.  The projections are computed ONCE per tensor of encoder features: cached under the
        identity (address, shape, version) of the tensor they were computed from and recomputed whenever a different one
        arrives.  ``group`` beams of a sample share that sample's projected row through the attention kernel's ``group``
        -- only when the caller says so (``encoder_group``) and hands over the un-expanded per-sample features."""
        from .. import ops
        a = self.attention
        q = ops.linear_fwd(queries.contiguous(), arena.compute(a.fc_q.weight), arena.master_of(a.fc_q.bias))
        R, nk, F = queries.shape[0], keys.shape[1], a.h * a.d_k
        assert keys.shape[0] * group == R
        ident = (keys.data_ptr(), tuple(keys.shape), keys._version, keys.dtype)
        enc = getattr(self, "_enc", None)
        if enc is not None and enc[0] == ident and enc[1].dtype == q.dtype:
            kv = enc[1]
        else:
            kv = ops.linear_fwd(keys.to(q.dtype).contiguous(), arena.packed([a.fc_k.weight, a.fc_v.weight]),
                                arena.packed([a.fc_k.bias, a.fc_v.bias], "master"))
            self._enc = (ident, kv)
        o = ops.attention_decode(q, kv[..., :F], kv[..., F:], nk, a.h, mask=self._row_mask(mask, R, nk), group=group)
        return self._finish_step(arena, queries, o)
