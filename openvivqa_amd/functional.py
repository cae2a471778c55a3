"""Block-level autograd functions over the HIP kernels.

One ``torch.autograd.Function`` per reference *block* (not per op), with a
hand-written backward that calls the backward kernels in sequence and writes
weight gradients straight into the parameter arena:

  prologue      LN(x) + sinusoid table            encoders.py:113 / pos_embeddings.py:58-72
  mha_block     fc_q/k/v -> attention core -> fc_o -> dropout -> +residual -> LN
                                                  attentions.py:46-60, 319-331
  ffn_block     fc1 -> GELU -> drop -> fc2 -> drop -> +residual -> LN
                                                  positionwise_feed_forward.py:23-28
  linear        nn.Linear (pointer heads, vocabulary projection, AoA gates)
  attention_core  softmax(qk^T/sqrt(d)+mask)v on projected tensors

Parameters are passed to ``apply`` only so that autograd schedules the node;
their gradients are produced by the kernels into ``arena.grad`` (``p.grad`` is
a view of it) and the functions return ``None`` for them.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.autograd import Function

from . import ops
from ._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL


def _c(t: Optional[torch.Tensor]):
    return t if t is None or t.is_contiguous() else t.contiguous()


_wgrad_queue = None
wgrad_observer = None  # optional callable(grad_view): told about every weight-gradient product of a backward pass


def _flush_wgrads():
    if _wgrad_queue is not None:
        _wgrad_queue.finish()


def wgrad_queue():
    """The process-wide deferred weight-gradient queue (created on first use)."""
    global _wgrad_queue
    if _wgrad_queue is None:
        _wgrad_queue = ops.WgradQueue()
    return _wgrad_queue


def _wgrad(arena, dy, x, w_params, b_params):
    """dW/db of a (possibly packed) linear into the arena's grad buffer.

    bf16: weight AND bias gradient are queued and computed at the END of the backward pass by one
    grouped launch over all linears (each product alone has only 16-64 output tiles, far fewer than
    the chip's 256 CUs); the bias gradient falls out of the same kernel (ones-fragment MFMA).  The queued ``dy``/``x``
    are never written afterwards (backward never updates a gradient tensor in place)."""
    global _wgrad_queue
    gw, acc_w = arena.grad_views(w_params)
    gb, acc_b = arena.grad_views(b_params) if b_params else (None, False)
    if wgrad_observer is not None:
        wgrad_observer(gw)
    defer = (dy.dtype == torch.bfloat16 and dy.is_cuda and dy.shape[-1] % 8 == 0 and x.shape[-1] % 8 == 0
             and os.environ.get("OVQA_DEFER_WGRAD", "1") != "0")
    if not defer:
        ops.linear_bwd_weight(dy, x, gw, gb, accumulate=acc_w, accumulate_db=acc_b)
        return
    q = _armed_queue()
    q.add(dy, x, gw, acc_w, gb, acc_b)  # bias gradient = fused column sums of dy


def _dx(arena, dy, ps, **kw):
    """dX of a (packed) linear: dy @ [W of ps].  bf16: from the arena's transposed weight copy (row-major weight
    tile for the GEMM); otherwise from the [N, K] weights."""
    wt = arena.transposed(ps)
    if wt is not None and ops.linear_bwd_data_wt_ok(dy, wt):
        return ops.linear_bwd_data_wt(dy, wt, **kw)
    return ops.linear_bwd_data(dy, arena.packed(ps), **kw)


def _pad_cols(x, n):
    """``x`` with its feature dimension zero-padded to ``n`` columns (a ragged layer's padded footprint in the arena:
    runtime._footprint); the common case -- nothing to pad -- returns x itself."""
    return x if x.shape[-1] == n else torch.nn.functional.pad(x, (0, n - x.shape[-1]))


def _lin_operands(arena, lin):
    """(weight [N_p, K_p], fp32 bias [N_p] or None) of an nn.Linear as the kernels see it: the zero-padded footprint."""
    w = arena.packed([lin.weight])
    b = arena.packed([lin.bias], "master") if lin.bias is not None else None
    return w, b


def _armed_queue():
    """The deferred-work queue, with its flush registered to run when the current backward pass ends."""
    q = wgrad_queue()
    # armed = a flush callback is registered for the running backward call.  An EMPTY queue re-arms as well: a backward pass
    # that raised never ran its callback (a second registration is harmless, the flush of an empty queue is a no-op); the
    # flag alone would not do, a phased backward holds LayerNorm reductions in the queue across its phases.
    if not getattr(q, "armed", False) or not (q.items or q.inflight or q.reduces):
        torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
        q.armed = True
    return q


def _ln_bwd(arena, dy, x, gamma, beta, mean, rstd, drop=None, dx_dtype=None):
    """LayerNorm backward; dgamma/dbeta go to the arena.  bf16 on the GPU: their cross-row reduction is
    deferred and done for ALL LayerNorms of the backward pass by one launch at its end (each of the 32
    reductions of an MCAN step would otherwise be a ~5 us dependent launch on the critical chain)."""
    gg, acc = arena.grad_views([gamma])
    gb, _ = arena.grad_views([beta])
    defer = (dy.dtype == torch.bfloat16 and dy.is_cuda and os.environ.get("OVQA_DEFER_WGRAD", "1") != "0")
    return ops.layernorm_bwd(dy, x, arena.master_of(gamma), mean, rstd, gg, gb, drop=drop, dx_dtype=dx_dtype,
                             accumulate=acc, defer=_armed_queue() if defer else None)


# ------------------------------------------------------------------ fp32 residual stream (bf16 mode)
# The post-LN residual chain (attentions.py:330-331, positionwise_feed_forward.py:25-26) is kept in fp32 between
# blocks: a block's bf16 output (the next GEMM's operand) carries, as the attribute ``_ovqa_res``, either its fp32
# twin (prologue) or an ops.LnRef from which the next block's epilogue recomputes it (block outputs: the fp32 values
# are never written to HBM).  The attribute is plumbing, not an autograd edge: the gradient of both uses flows through
# the bf16 tensor, exactly as before.  A tensor without the attribute falls back to its own values cast to fp32.
def attach_residual(y, res):
    """Attach ``res`` (fp32 twin or ops.LnRef) to the bf16 tensor ``y`` together with y's identity at this moment
    (version counter, storage address, shape): an IN-PLACE change of y afterwards (masked_fill_, mul_ by a padding mask,
    a slice assignment by the caller of a drop-in module) bumps the version and invalidates the twin."""
    y._ovqa_res = res
    y._ovqa_res_tag = (y._version, y.data_ptr(), tuple(y.shape))
    return y


def _twin(x):
    res = getattr(x, "_ovqa_res", None)
    if res is None:
        return None
    if getattr(x, "_ovqa_res_tag", None) != (x._version, x.data_ptr(), tuple(x.shape)):
        return None  # x was modified in place since the twin was attached: the fp32 values are stale
    return res


def residual_of(x):
    res = _twin(x)
    if res is None:
        return x.float()
    return res


def to_compute(x, dtype):
    """``x`` in the compute dtype; an fp32 input of the bf16 mode stays available as the fp32 twin of its bf16 cast
    (embedding sums entering a decoder, fp32 features entering an attention block directly)."""
    if x.dtype == dtype:
        return x
    y = x.to(dtype)
    if dtype == torch.bfloat16 and x.dtype == torch.float32:
        attach_residual(y, x.detach().contiguous())
    return y


def _attach(y, st):
    res = st.pop("_res_out", None)
    if res is not None:
        attach_residual(y, res)
    return y


def carry_residual(dst, src):
    """Hand ``src``'s fp32 twin to ``dst`` (a detached / re-rooted alias of the same values)."""
    res = _twin(src)
    if res is not None:
        attach_residual(dst, res)
    return dst


class _WithValue(Function):
    """forward: ``value`` (the fp32 twin of x); backward: the gradient goes to x."""

    @staticmethod
    def forward(ctx, x, value):
        ctx.dt = x.dtype
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt), None


class _DecoderInputs(Function):
    """(emb + pos_table[seq], self-attention mask) of a teacher-forced decoder pass (``ops.decoder_inputs``); the gradient of
    the sum goes to the word embeddings unchanged (the position table is frozen, decoders.py:41-42)."""

    @staticmethod
    def forward(ctx, emb, tokens, pos_table, padding_idx):
        out, mask = ops.decoder_inputs(tokens, _c(emb), pos_table, padding_idx)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, g, _gm):
        return g, None, None, None


def decoder_inputs(emb, tokens, pos_table, padding_idx):
    return _DecoderInputs.apply(emb, tokens, pos_table, padding_idx)


def finalize(out, dtype):
    """Stack output in the caller's dtype: an fp32 caller of the bf16 mode gets the unrounded fp32 stream."""
    res = _twin(out)
    if dtype == torch.float32 and out.dtype != torch.float32 and res is not None:
        y32 = res.materialize() if isinstance(res, ops.LnRef) else res
        return _WithValue.apply(out, y32) if out.requires_grad else y32
    return out.to(dtype)


def _ln_out(pre, ln, arena, st, save_stats=True):
    """LayerNorm of the fp32 pre-LN sum -> bf16 operand; leaves the lazy fp32 twin in st["_res_out"]."""
    gamma, beta = arena.master_of(ln.weight), arena.master_of(ln.bias)
    y, mean, rstd = ops.layernorm_fwd(pre, gamma, beta, ln.eps, out_dtype=arena.compute_dtype)
    st["_res_out"] = ops.LnRef(pre, mean, rstd, gamma, beta, ln.eps)
    return y, mean, rstd


# ------------------------------------------------------------------ prologue
class _Prologue(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, st):
        arena, T = st["arena"], st["dtype"]
        x = _c(x)
        ctx.dx_dtype = x.dtype
        x32 = st.pop("x32", None)
        if x32 is not None:  # the unrounded values of a bf16 input (its fp32 twin): the LayerNorm reads those
            x = _c(x32)
        if T == torch.bfloat16:
            y, y32, mean, rstd = ops.layernorm_fwd(x, arena.master_of(gamma), arena.master_of(beta), st["eps"],
                                                   out_dtype=T, pos=st["pos"], want_f32=True)
            st["_res_out"] = y32
        else:
            y, mean, rstd = ops.layernorm_fwd(x, arena.master_of(gamma), arena.master_of(beta), st["eps"], out_dtype=T,
                                              pos=st["pos"])
        ctx.st = st
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        arena = st["arena"]
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = st["gamma"], st["beta"]
        dx, _ = _ln_bwd(arena, _c(dy), x, gamma, beta, mean, rstd, dx_dtype=ctx.dx_dtype)
        return dx if ctx.needs_input_grad[0] else None, None, None, None


def prologue(x, layer_norm, pos, arena, dtype):
    st = dict(arena=arena, dtype=dtype, eps=layer_norm.eps, pos=pos, gamma=layer_norm.weight, beta=layer_norm.bias)
    tw = _twin(x)
    if isinstance(tw, torch.Tensor) and tw.dtype == torch.float32 and tw.shape == x.shape and x.dtype == torch.bfloat16:
        st["x32"] = tw
    return _attach(_Prologue.apply(x, layer_norm.weight, layer_norm.bias, st), st)


# ------------------------------------------------------------------ hoisted K/V projection
class _KVProjectAll(Function):
    """fc_k / fc_v of SEVERAL attention modules applied to the same keys in one GEMM: kv[..., s*(nk+nv):...] is
    module s's [K | V] projection.  In MCAN's guided stack every layer attends to the same final question
    features (encoders.py:156-162), so the 6 per-layer K/V projections (6 forward GEMMs, 6 dX GEMMs and 5
    gradient adds on 1280 rows, each a ~5-15 us dependent launch) become one forward and one dX GEMM."""

    @staticmethod
    def forward(ctx, keys, st, *params):
        arena = st["arena"]
        keys = _c(keys)
        kv = ops.linear_fwd(keys, arena.packed(st["weights"]), arena.packed(st["biases"], "master"))
        ctx.st = st
        ctx.save_for_backward(keys)
        return kv

    @staticmethod
    def backward(ctx, dkv):
        st = ctx.st
        arena = st["arena"]
        (keys,) = ctx.saved_tensors
        dkv = _c(dkv)
        _wgrad(arena, dkv, keys, st["weights"], st["biases"])
        dkeys = _dx(arena, dkv, st["weights"]) if ctx.needs_input_grad[0] else None
        return dkeys, None, *([None] * (len(st["weights"]) + len(st["biases"])))


def kv_project_all(keys, attentions, arena):
    """[K_0 | V_0 | K_1 | V_1 | ...] projections of ``keys`` for the given ScaledDotProductAttention modules, or
    None when their fc_k / fc_v parameters are not adjacent in the arena (then callers project per module)."""
    weights = [w for a in attentions for w in (a.fc_k.weight, a.fc_v.weight)]
    biases = [b for a in attentions for b in (a.fc_k.bias, a.fc_v.bias)]
    try:
        arena.packed(weights)
        arena.packed(biases, "master")
    except (RuntimeError, KeyError):
        return None
    st = dict(arena=arena, weights=weights, biases=biases)
    if torch.is_grad_enabled():
        return _KVProjectAll.apply(keys, st, *weights, *biases)
    return ops.linear_fwd(_c(keys), arena.packed(weights), arena.packed(biases, "master"))


# ------------------------------------------------------------------ MHA block
def same_tensor(a, b) -> bool:
    return a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
                      and a.dtype == b.dtype)


def _canon(queries, keys, values, same):
    """Make the aliasing declared in ``same`` ("all" | "kv" | "none") explicit by identity."""
    queries = _c(queries)
    if same == "all":
        return queries, queries, queries
    keys = _c(keys)
    if same == "kv":
        return queries, keys, keys
    return queries, keys, _c(values)


def _project_qkv(st, queries, keys, values):
    """Returns q, k, v (strided views of packed projection buffers) and the mode."""
    arena, a = st["arena"], st["att"]
    wq, wk, wv = a.fc_q.weight, a.fc_k.weight, a.fc_v.weight
    bq, bk, bv = a.fc_q.bias, a.fc_k.bias, a.fc_v.bias
    nqk, nv = wq.shape[0], wv.shape[0]
    if st.get("pre_kv") is not None:  # keys IS the hoisted projection buffer (kv_project_all), slot = this module
        slot = st["pre_kv"][0]
        base = slot * (wk.shape[0] + nv)
        q = ops.linear_fwd(queries, arena.compute(wq), arena.master_of(bq))
        return q, keys[..., base:base + wk.shape[0]], keys[..., base + wk.shape[0]:base + wk.shape[0] + nv], "pre", (q,)
    if queries is keys and keys is values:
        qkv = ops.linear_fwd(queries, arena.packed([wq, wk, wv]), arena.packed([bq, bk, bv], "master"))
        return qkv[..., :nqk], qkv[..., nqk:2 * nqk], qkv[..., 2 * nqk:], "self", (qkv,)
    q = ops.linear_fwd(queries, arena.compute(wq), arena.master_of(bq))
    if keys is values:
        kv = ops.linear_fwd(keys, arena.packed([wk, wv]), arena.packed([bk, bv], "master"))
        return q, kv[..., :nqk], kv[..., nqk:], "cross", (q, kv)
    k = ops.linear_fwd(keys, arena.compute(wk), arena.master_of(bk))
    v = ops.linear_fwd(values, arena.compute(wv), arena.master_of(bv))
    return q, k, v, "general", (q, k, v)


def _project_and_attend(st, queries, keys, values, mask, save_lse=True, lo_out=None):
    """Projections + attention core of a MultiHeadAttention call -> (o, lse, mode, saved projection buffers).
    Self-attention with a key mask (or none) and no probability dropout goes through ``ovqa_attention_qkv_fwd``:
    one kernel for the packed projection and the attention where the shape allows, the two separate kernels
    (inside the library) otherwise."""
    a, arena = st["att"], st["arena"]
    if (st.get("pre_kv") is None and queries is keys and keys is values and st.get("att_drop") is None
            and (mask is None or mask.shape[2] == 1) and a.fc_q.weight.shape[0] == a.fc_v.weight.shape[0]):
        wq, wk, wv = a.fc_q.weight, a.fc_k.weight, a.fc_v.weight
        qkv, o, lse = ops.attention_qkv_fwd(queries, arena.packed([wq, wk, wv]),
                                            arena.packed([a.fc_q.bias, a.fc_k.bias, a.fc_v.bias], "master"), mask, a.h,
                                            save_lse=save_lse, lo_out=lo_out)
        return o, lse, "self", (qkv,)
    if (st.get("pre_kv") is not None and st.get("att_drop") is None and (mask is None or mask.shape[2] == 1)
            and a.fc_q.weight.shape[0] == a.fc_v.weight.shape[0] and queries.dim() == 3
            and os.environ.get("OVQA_NO_FUSED_Q", "0") != "1"):
        # guided / cross attention on hoisted K | V projections: the query projection inside the attention kernel
        wk, nv = a.fc_k.weight, a.fc_v.weight.shape[0]
        base = st["pre_kv"][0] * (wk.shape[0] + nv)
        k, v = keys[..., base:base + wk.shape[0]], keys[..., base + wk.shape[0]:base + wk.shape[0] + nv]
        q, o, lse = ops.attention_q_fwd(queries, arena.compute(a.fc_q.weight), arena.master_of(a.fc_q.bias), k, v, mask,
                                        a.h, save_lse=save_lse, lo_out=lo_out)
        return o, lse, "pre", (q,)
    if (st.get("pre_kv") is None and keys is values and queries is not keys and st.get("att_drop") is None
            and (mask is None or mask.shape[2] == 1) and a.fc_q.weight.shape[0] == a.fc_k.weight.shape[0] == a.fc_v.weight.shape[0]
            and queries.dim() == 3 and keys.dim() == 3 and os.environ.get("OVQA_NO_FUSED_Q", "0") != "1"):
        # cross attention (keys is values): the packed K | V projection, then the query projection inside the attention
        wq, wk, wv = a.fc_q.weight, a.fc_k.weight, a.fc_v.weight
        kv = ops.linear_fwd(keys, arena.packed([wk, wv]), arena.packed([a.fc_k.bias, a.fc_v.bias], "master"))
        nqk = wq.shape[0]
        q, o, lse = ops.attention_q_fwd(queries, arena.compute(wq), arena.master_of(a.fc_q.bias), kv[..., :nqk],
                                        kv[..., nqk:], mask, a.h, save_lse=save_lse, lo_out=lo_out)
        return o, lse, "cross", (q, kv)
    q, k, v, mode, bufs = _project_qkv(st, queries, keys, values)
    # a dense (B, 1, n, n) mask that its builder has described as "key mask row + causal corner of the last T positions"
    # (modules/mmt.py, inference): the kernel computes the corner and reads the row out of LDS instead of n x n mask values
    hint = getattr(mask, "_ovqa_prefix_lm", None)
    if (hint is not None and not save_lse and lo_out is None and st.get("att_drop") is None and st.get("same") == "all"
            and not torch.is_grad_enabled() and ops.attention_fwd_prefix_lm_ok(q, a.h)):
        return ops.attention_fwd_prefix_lm(q, k, v, hint[0], hint[1], a.h), None, mode, bufs
    o, lse, _ = ops.attention_fwd(q, k, v, mask, a.h, save_lse=save_lse, att_drop=st.get("att_drop"), lo_out=lo_out)
    return o, lse, mode, bufs


class _MHABlock(Function):
    """LN(queries + dropout(fc_o(attention(fc_q(queries), fc_k(keys), fc_v(values)))))."""

    @staticmethod
    def forward(ctx, queries, keys, values, mask, st, *params):
        arena, a, ln = st["arena"], st["att"], st["ln"]
        queries, keys, values = _canon(queries, keys, values, st["same"])
        lo = []  # bf16: the rounding residual of o, for the backward's delta (ops.attention_bwd o_lo)
        o, lse, mode, bufs = _project_and_attend(st, queries, keys, values, mask, lo_out=lo)
        drop = st["drop"]
        if queries.dtype == torch.bfloat16:  # fp32 residual stream: fp32 pre-LN sum, bf16 operand out
            pre = ops.linear_fwd_res32(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias), st.pop("res"),
                                       drop=drop)
            y, mean, rstd = _ln_out(pre, ln, arena, st)
        else:
            pre = ops.linear_fwd(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias), EPI_BIAS_RESIDUAL,
                                 residual=queries, drop=drop)
            y, mean, rstd = ops.layernorm_fwd(pre, arena.master_of(ln.weight), arena.master_of(ln.bias), ln.eps)
        ctx.st, ctx.mode = st, mode
        ctx.mask = mask
        ctx.save_for_backward(queries, keys, values, o, lse, pre, mean, rstd, (lo[0] if lo else None), *bufs)
        return y

    @staticmethod
    def backward(ctx, dy):
        st, mode = ctx.st, ctx.mode
        arena, a, ln, drop = st["arena"], st["att"], st["ln"], st["drop"]
        queries, keys, values, o, lse, pre, mean, rstd, o_lo, *bufs = ctx.saved_tensors
        dpre, dpre_d = _ln_bwd(arena, _c(dy), pre, ln.weight, ln.bias, mean, rstd, drop=drop)
        # fc_o
        _wgrad(arena, dpre_d, o, [a.fc_o.weight], [a.fc_o.bias])
        nqk = a.fc_q.weight.shape[0]
        # guided attention and the 20 x 20 question self-attention: the fc_o dX product runs inside the attention backward
        # kernel (ovqa_attention_bwd_do), dO never travels through HBM
        fuse_do = False
        if mode in ("pre", "self") and st.get("att_drop") is None and os.environ.get("OVQA_NO_FUSED_DO", "0") != "1":
            wt_o = arena.transposed([a.fc_o.weight])
            q_chk = bufs[0] if mode == "pre" else bufs[0][..., :nqk]
            fuse_do = ops.attention_bwd_do_ok(dpre_d, wt_o, q_chk, keys if mode == "pre" else q_chk, ctx.mask, a.h)
        d_o = None if fuse_do else _dx(arena, dpre_d, [a.fc_o.weight])
        wq, wk, wv = a.fc_q.weight, a.fc_k.weight, a.fc_v.weight
        bq, bk, bv = a.fc_q.bias, a.fc_k.bias, a.fc_v.bias
        B, nq, nk = queries.shape[0], queries.shape[1], keys.shape[1]
        T, dev = queries.dtype, queries.device
        if mode == "self":
            (qkv,) = bufs
            q, k, v = qkv[..., :nqk], qkv[..., nqk:2 * nqk], qkv[..., 2 * nqk:]
            dqkv = torch.empty_like(qkv)
            if fuse_do:
                ops.attention_bwd_do(dpre_d, wt_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, dq=dqkv[..., :nqk],
                                     dk=dqkv[..., nqk:2 * nqk], dv=dqkv[..., 2 * nqk:])
            else:
                ops.attention_bwd(d_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, att_drop=st.get("att_drop"),
                                  dq=dqkv[..., :nqk], dk=dqkv[..., nqk:2 * nqk], dv=dqkv[..., 2 * nqk:])
            _wgrad(arena, dqkv, queries, [wq, wk, wv], [bq, bk, bv])
            dx = _dx(arena, dqkv, [wq, wk, wv], addend=dpre)
            return dx, None, None, None, None, *([None] * len(st["params"]))
        if mode == "pre":
            (q,) = bufs
            slot, shared = st["pre_kv"]
            base = slot * (wk.shape[0] + wv.shape[0])
            k, v = keys[..., base:base + wk.shape[0]], keys[..., base + wk.shape[0]:base + wk.shape[0] + wv.shape[0]]
            if shared.get("dkv") is None:  # one gradient buffer for the whole hoisted projection; every module
                shared["dkv"] = torch.empty_like(keys)  # fills its slot, module 0 hands it to autograd
            dkv = shared["dkv"]
            dq = torch.empty_like(q)
            if fuse_do:
                ops.attention_bwd_do(dpre_d, wt_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, dq=dq,
                                     dk=dkv[..., base:base + wk.shape[0]],
                                     dv=dkv[..., base + wk.shape[0]:base + wk.shape[0] + wv.shape[0]])
            else:
                ops.attention_bwd(d_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, att_drop=st.get("att_drop"), dq=dq,
                                  dk=dkv[..., base:base + wk.shape[0]],
                                  dv=dkv[..., base + wk.shape[0]:base + wk.shape[0] + wv.shape[0]])
            _wgrad(arena, dq, queries, [wq], [bq])
            dx = _dx(arena, dq, [wq], addend=dpre)
            return dx, (dkv if slot == 0 else None), None, None, None, *([None] * len(st["params"]))
        if mode == "cross":
            q, kv = bufs
            k, v = kv[..., :nqk], kv[..., nqk:]
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            ops.attention_bwd(d_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, att_drop=st.get("att_drop"), dq=dq, dk=dkv[..., :nqk], dv=dkv[..., nqk:])
            _wgrad(arena, dq, queries, [wq], [bq])
            dx = _dx(arena, dq, [wq], addend=dpre)
            _wgrad(arena, dkv, keys, [wk, wv], [bk, bv])
            dkeys = _dx(arena, dkv, [wk, wv]) if ctx.needs_input_grad[1] or ctx.needs_input_grad[2] else None
            return dx, dkeys, None, None, None, *([None] * len(st["params"]))
        q, k, v = bufs
        dq, dk, dv = ops.attention_bwd(d_o, q, k, v, o, lse, ctx.mask, a.h, o_lo=o_lo, att_drop=st.get("att_drop"))
        _wgrad(arena, dq, queries, [wq], [bq])
        dx = _dx(arena, dq, [wq], addend=dpre)
        _wgrad(arena, dk, keys, [wk], [bk])
        _wgrad(arena, dv, values, [wv], [bv])
        dkeys = _dx(arena, dk, [wk]) if ctx.needs_input_grad[1] else None
        dvalues = _dx(arena, dv, [wv]) if ctx.needs_input_grad[2] else None
        return dx, dkeys, dvalues, None, None, *([None] * len(st["params"]))


def mha_block(queries, keys, values, mask, st, projected_kv=None):
    """Forward of the fused MHA block; uses autograd only when needed.  ``projected_kv = (kv_all, slot, shared)``
    (from kv_project_all) replaces keys/values by already projected K/V."""
    st = dict(st)
    if projected_kv is not None:
        keys = values = projected_kv[0]
        st["pre_kv"] = (projected_kv[1], projected_kv[2])
        st["same"] = "kv"
    else:
        st["same"] = ("all" if same_tensor(queries, keys) and same_tensor(keys, values)
                      else "kv" if same_tensor(keys, values) else "none")
    bf16 = queries.dtype == torch.bfloat16
    if bf16:
        st["res"] = residual_of(queries)
    if torch.is_grad_enabled():
        return _attach(_MHABlock.apply(queries, keys, values, mask, st, *st["params"]), st)
    arena, a, ln = st["arena"], st["att"], st["ln"]
    queries, keys, values = _canon(queries, keys, values, st["same"])
    o, _, _, _ = _project_and_attend(st, queries, keys, values, mask, save_lse=False)
    if bf16:
        pre = ops.linear_fwd_res32(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias), st.pop("res"),
                                   drop=st["drop"])
        y, _, _ = _ln_out(pre, ln, arena, st)
        return _attach(y, st)
    pre = ops.linear_fwd(o, arena.compute(a.fc_o.weight), arena.master_of(a.fc_o.bias), EPI_BIAS_RESIDUAL,
                         residual=queries, drop=st["drop"])
    y, _, _ = ops.layernorm_fwd(pre, arena.master_of(ln.weight), arena.master_of(ln.bias), ln.eps, save_stats=False)
    return y


# ------------------------------------------------------------------ FFN block
class _FFNBlock(Function):
    @staticmethod
    def forward(ctx, x, st, *params):
        arena, m = st["arena"], st["mod"]
        x = _c(x)
        h, u = ops.linear_fwd(x, arena.compute(m.fc1.weight), arena.master_of(m.fc1.bias), EPI_BIAS_GELU,
                              want_preact=True, drop=st["drop1"])
        ln = m.layer_norm
        if x.dtype == torch.bfloat16:
            pre = ops.linear_fwd_res32(h, arena.compute(m.fc2.weight), arena.master_of(m.fc2.bias), st.pop("res"),
                                       drop=st["drop2"])
            y, mean, rstd = _ln_out(pre, ln, arena, st)
        else:
            pre = ops.linear_fwd(h, arena.compute(m.fc2.weight), arena.master_of(m.fc2.bias), EPI_BIAS_RESIDUAL,
                                 residual=x, drop=st["drop2"])
            y, mean, rstd = ops.layernorm_fwd(pre, arena.master_of(ln.weight), arena.master_of(ln.bias), ln.eps)
        ctx.st = st
        ctx.save_for_backward(x, h, u, pre, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        arena, m = st["arena"], st["mod"]
        ln = m.layer_norm
        x, h, u, pre, mean, rstd = ctx.saved_tensors
        dpre, dpre_d = _ln_bwd(arena, _c(dy), pre, ln.weight, ln.bias, mean, rstd, drop=st["drop2"])
        _wgrad(arena, dpre_d, h, [m.fc2.weight], [m.fc2.bias])
        du = _dx(arena, dpre_d, [m.fc2.weight], preact=u, drop=st["drop1"])
        _wgrad(arena, du, x, [m.fc1.weight], [m.fc1.bias])
        dx = _dx(arena, du, [m.fc1.weight], addend=dpre)
        return dx, None, *([None] * len(st["params"]))


def ffn_block(x, st):
    st = dict(st)
    bf16 = x.dtype == torch.bfloat16
    if bf16:
        st["res"] = residual_of(x)
    if torch.is_grad_enabled():
        return _attach(_FFNBlock.apply(x, st, *st["params"]), st)
    arena, m = st["arena"], st["mod"]
    x = _c(x)
    h = ops.linear_fwd(x, arena.compute(m.fc1.weight), arena.master_of(m.fc1.bias), EPI_BIAS_GELU, drop=st["drop1"])
    ln = m.layer_norm
    if bf16:
        pre = ops.linear_fwd_res32(h, arena.compute(m.fc2.weight), arena.master_of(m.fc2.bias), st.pop("res"),
                                   drop=st["drop2"])
        y, _, _ = _ln_out(pre, ln, arena, st)
        return _attach(y, st)
    pre = ops.linear_fwd(h, arena.compute(m.fc2.weight), arena.master_of(m.fc2.bias), EPI_BIAS_RESIDUAL, residual=x,
                         drop=st["drop2"])
    y, _, _ = ops.layernorm_fwd(pre, arena.master_of(ln.weight), arena.master_of(ln.bias), ln.eps, save_stats=False)
    return y


# ------------------------------------------------------------------ plain linear
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, st, *params):
        arena, lin = st["arena"], st["lin"]
        w, bias = _lin_operands(arena, lin)
        ctx.k_in = x.shape[-1]  # (the caller may hand rows that carry the footprint's zero padding already)
        x = _pad_cols(_c(x), w.shape[1])
        y = ops.linear_fwd(x, w, bias)
        ctx.st = st
        ctx.save_for_backward(x)
        N = lin.weight.shape[0]
        return y if y.shape[-1] == N else y[..., :N].contiguous()

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        arena, lin = st["arena"], st["lin"]
        (x,) = ctx.saved_tensors
        dy = _pad_cols(_c(dy), arena.foot[id(lin.weight)][0])
        _wgrad(arena, dy, x, [lin.weight], [lin.bias] if lin.bias is not None else [])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx(arena, dy, [lin.weight])
            if dx.shape[-1] != ctx.k_in:
                dx = dx[..., :ctx.k_in].contiguous()
        return dx, None, *([None] * len(st["params"]))


class _LinearGeluDrop(Function):
    """dropout(gelu(lin(x))): FeatureEmbedding (vision_embeddings.py:20-23).  Forward is one GEMM with the fused
    bias+GELU+dropout epilogue; backward regenerates the dropout mask and applies gelu' elementwise."""

    @staticmethod
    def forward(ctx, x, st, *params):
        arena, lin = st["arena"], st["lin"]
        w, bias = _lin_operands(arena, lin)
        x = _pad_cols(_c(x), w.shape[1])
        y, u = ops.linear_fwd(x, w, bias, EPI_BIAS_GELU, want_preact=True, drop=st["drop"])
        ctx.st = st
        ctx.save_for_backward(x, u)
        N = lin.weight.shape[0]
        return y if y.shape[-1] == N else y[..., :N].contiguous()

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        arena, lin = st["arena"], st["lin"]
        x, u = ctx.saved_tensors
        dy = _pad_cols(_c(dy), u.shape[-1])
        du = ops.gelu_bwd(dy.reshape(u.shape), u, drop=st["drop"])
        _wgrad(arena, du, x.reshape(-1, x.shape[-1]), [lin.weight], [lin.bias])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx(arena, du, [lin.weight]).reshape(x.shape)
            K = lin.weight.shape[1]
            if dx.shape[-1] != K:
                dx = dx[..., :K].contiguous()
        return dx, None, *([None] * len(st["params"]))


def linear_gelu_dropout(x, lin, arena, drop):
    params = [lin.weight, lin.bias]
    st = dict(arena=arena, lin=lin, params=params, drop=drop)
    if torch.is_grad_enabled():
        return _LinearGeluDrop.apply(x, st, *params)
    w, bias = _lin_operands(arena, lin)
    y = ops.linear_fwd(_pad_cols(_c(x), w.shape[1]), w, bias, EPI_BIAS_GELU, drop=drop)
    return y if y.shape[-1] == lin.weight.shape[0] else y[..., :lin.weight.shape[0]].contiguous()


def linear(x, lin, arena):
    """y = lin(x) through the HIP GEMM (x is cast to the arena's compute dtype by the caller)."""
    params = [lin.weight] + ([lin.bias] if lin.bias is not None else [])
    st = dict(arena=arena, lin=lin, params=params)
    if torch.is_grad_enabled():
        return _Linear.apply(x, st, *params)
    w, bias = _lin_operands(arena, lin)
    y = ops.linear_fwd(_pad_cols(_c(x), w.shape[1]), w, bias)
    return y if y.shape[-1] == lin.weight.shape[0] else y[..., :lin.weight.shape[0]].contiguous()


# ------------------------------------------------------------------ attention core on projected tensors
class _AttentionCore(Function):
    """(o, att, lse) = attention(q, k, v); all three outputs are differentiable (the reference's second return value
    is; the log-sum-exp lets a caller merge extra softmax columns of its own, e.g. the adaptive attention)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, h, need_att):
        lo = []
        o, lse, att = ops.attention_fwd(q, k, v, mask, h, need_att=need_att, lo_out=lo)
        ctx.h, ctx.mask, ctx.need_att = h, mask, need_att
        ctx.save_for_backward(q, k, v, o, lse, (lo[0] if lo else None))
        return o, att, lse

    @staticmethod
    def backward(ctx, d_o, d_att, d_lse):
        q, k, v, o, lse, o_lo = ctx.saved_tensors
        if d_o is None:
            d_o = torch.zeros_like(o)
        if d_att is not None:
            d_att = d_att.to(q.dtype).contiguous()
        if d_lse is not None:
            d_lse = d_lse.float().contiguous()
        dq, dk, dv = ops.attention_bwd(_c(d_o), q, k, v, o, lse, ctx.mask, ctx.h, d_att=d_att, d_lse=d_lse, o_lo=o_lo)
        return dq, dk, dv, None, None, None


def attention_core(q, k, v, mask, h, need_att=False, need_lse=False):
    """Returns (o, att) -- or (o, att, lse) with ``need_lse`` -- of the attention core on projected tensors."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        o, att, lse = _AttentionCore.apply(q, k, v, mask, h, need_att)
    else:
        o, lse, att = ops.attention_fwd(q, k, v, mask, h, need_att=need_att, save_lse=need_lse)
    return (o, att, lse) if need_lse else (o, att)


class _BiasedAttentionCore(Function):
    """attention(q, k, v) with a LEARNED additive score bias (B, H, nq, nk) -- the log-geometry term of the
    geometry-aware attention.  The kernels take the bias as their mask; its gradient is dS = P (dP - rowsum(P dP)),
    assembled here from the returned probabilities with stock batched ops (the bias path is small: H x nq x nk)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, h):
        lo = []
        o, lse, att = ops.attention_fwd(q, k, v, bias, h, need_att=True, lo_out=lo)
        ctx.h = h
        ctx.save_for_backward(q, k, v, o, lse, att, bias, (lo[0] if lo else None))
        return o, att

    @staticmethod
    def backward(ctx, d_o, d_att):
        q, k, v, o, lse, att, bias, o_lo = ctx.saved_tensors
        h = ctx.h
        if d_o is None:
            d_o = torch.zeros_like(o)
        d_o = _c(d_o)
        if d_att is not None:
            d_att = d_att.to(q.dtype).contiguous()
        dq, dk, dv = ops.attention_bwd(d_o, q, k, v, o, lse, bias, h, d_att=d_att, o_lo=o_lo)
        dbias = None
        if ctx.needs_input_grad[3]:
            B, nq, nk = q.shape[0], q.shape[1], k.shape[1]
            dv_ = v.shape[-1] // h
            g = d_o.float().view(B, nq, h, dv_).transpose(1, 2)
            vh = v.float().reshape(B, nk, h, dv_).transpose(1, 2)
            dp = g @ vh.transpose(-1, -2)
            if d_att is not None:
                dp = dp + d_att.float()
            p = att.float()
            dbias = p * (dp - (p * dp).sum(-1, keepdim=True))
        return dq, dk, dv, dbias, None


def attention_core_with_bias(q, k, v, bias, h):
    """(o, att) with a differentiable additive score bias (fp32, contiguous (B, H, nq, nk))."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad or bias.requires_grad):
        return _BiasedAttentionCore.apply(q, k, v, bias, h)
    o, _, att = ops.attention_fwd(q, k, v, bias, h, need_att=True, save_lse=False)
    return o, att


# ------------------------------------------------------------------ pointer scores
class _PointerScore(Function):
    @staticmethod
    def forward(ctx, q, k, scale, add_mask, key_fill, query_fill):
        q, k = _c(q), _c(k)
        s = ops.pointer_score(q, k, scale, add_mask, key_fill, query_fill)
        ctx.scale = scale
        ctx.save_for_backward(q, k)
        return s

    @staticmethod
    def backward(ctx, ds):
        q, k = ctx.saved_tensors
        ds = torch.where(torch.isfinite(ds), ds, torch.zeros_like(ds)).to(q.dtype).contiguous()
        dq = ops.batched_gemm(ds, k, alpha=ctx.scale)                 # [B,T,N] x [B,N,D]
        dk = ops.batched_gemm(ds, q, trans_a=True, alpha=ctx.scale)   # [B,N,T] x [B,T,D]
        return dq, dk, None, None, None, None


def pointer_score(q, k, scale, add_mask=None, key_fill=None, query_fill=None):
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad):
        return _PointerScore.apply(q, k, scale, add_mask, key_fill, query_fill)
    return ops.pointer_score(_c(q), _c(k), scale, add_mask, key_fill, query_fill)


# ------------------------------------------------------------------ LSTM recurrence
class _LSTM(Function):
    """y = LSTM(x) (text_embeddings.py:236,243) on TIME-MAJOR input rows x_tb [T*B, I]; y fp32 [B, T, H] and, in bf16 mode, its
    bf16 twin (second output: what a stack of this package takes as its operand without a cast launch; a gradient may
    arrive through either).  One persistent launch each way in bf16 mode; the weight / bias gradients and dx are the
    library's GEMMs on ``dgates``."""

    @staticmethod
    def forward(ctx, x_tb, st, *params):
        arena, m = st["arena"], st["mod"]
        x_tb = _c(x_tb)
        want_lp = x_tb.dtype != torch.float32
        res = ops.lstm_fwd(x_tb, arena.compute(m.weight_ih_l0), arena.compute(m.weight_hh_l0),
                           arena.master_of(m.bias_ih_l0), arena.master_of(m.bias_hh_l0), st["B"], st["T"], want_lp=want_lp)
        y, hseq, saved = res[0], res[1], res[2]
        ctx.st = st
        ctx.set_materialize_grads(False)
        if torch.is_tensor(saved):
            ctx.save_for_backward(x_tb, hseq, saved)
        else:  # (a padded / split batch: the chunks' opaque blocks, ops.LstmChunks)
            ctx.save_for_backward(x_tb, hseq)
            ctx.lstm_chunks = saved
        return (y, res[4]) if want_lp else y

    @staticmethod
    def backward(ctx, dy, dy_lp=None):
        st = ctx.st
        arena, m, B, T = st["arena"], st["mod"], st["B"], st["T"]
        if len(ctx.saved_tensors) == 3:
            x_tb, hseq, saved = ctx.saved_tensors
        else:
            (x_tb, hseq), saved = ctx.saved_tensors, ctx.lstm_chunks
        w_ih, w_hh = m.weight_ih_l0, m.weight_hh_l0
        if dy is None and dy_lp is None:
            return (None,) * (2 + len(st["params"]))
        if dy is None:
            g = _c(dy_lp)                       # bf16, read as it is by the kernel
        elif dy_lp is None:
            g = _c(dy.float())
        else:
            g = _c(dy.float() + dy_lp.float())  # (both outputs were used)
        dgates, _ = ops.lstm_bwd(g, arena.compute(w_hh), arena.transposed([w_hh]), saved, B, T, w_ih.shape[1])
        _wgrad(arena, dgates, x_tb, [w_ih], [m.bias_ih_l0])
        _wgrad(arena, dgates, hseq[:T * B], [w_hh], [m.bias_hh_l0])
        dx = _dx(arena, dgates, [w_ih]) if ctx.needs_input_grad[0] else None
        return dx, None, *([None] * len(st["params"]))


def lstm(x_tb, mod, arena, B, T):
    """``mod`` holds weight_ih_l0 / weight_hh_l0 / bias_ih_l0 / bias_hh_l0 (torch.nn.LSTM's parameter names).  Returns y fp32
    [B, T, H]; in bf16 mode its bf16 twin rides along as plumbing (``compute_twin(y)``), the fp32 values as ITS residual twin."""
    params = [mod.weight_ih_l0, mod.weight_hh_l0, mod.bias_ih_l0, mod.bias_hh_l0]
    st = dict(arena=arena, mod=mod, params=params, B=B, T=T)
    if torch.is_grad_enabled():
        out = _LSTM.apply(x_tb, st, *params)
        if isinstance(out, tuple):
            y, y_lp = out
            attach_residual(y_lp, y.detach())
            y._ovqa_compute = y_lp
            y._ovqa_compute_tag = (y._version, y.data_ptr(), tuple(y.shape))
            return y
        return out
    return ops.lstm_fwd(_c(x_tb), arena.compute(mod.weight_ih_l0), arena.compute(mod.weight_hh_l0),
                        arena.master_of(mod.bias_ih_l0), arena.master_of(mod.bias_hh_l0), B, T)[0]


def compute_twin(x, dtype):
    """The tensor a producer of this package made NEXT TO ``x`` in the compute dtype (same values, its own autograd edge):
    handing it to the next stack instead of ``x`` saves the cast launches both ways and the fp32 round trip of the stack's
    output.  ``x`` itself when there is none (or x was modified since)."""
    tw = getattr(x, "_ovqa_compute", None)
    if tw is None or tw.dtype != dtype or getattr(x, "_ovqa_compute_tag", None) != (x._version, x.data_ptr(), tuple(x.shape)):
        return x
    return tw


# ------------------------------------------------------------------ the two ends of the model (csrc/model_ends.hip)
def _defer_queue(t):
    """The deferred grouped-reduce queue for a bf16 GPU backward pass, else None (as _ln_bwd)."""
    if t.dtype == torch.bfloat16 and t.is_cuda and os.environ.get("OVQA_DEFER_WGRAD", "1") != "0":
        return _armed_queue()
    return None


class _EmbedRows(Function):
    """rows = table[tokens] (nn.Embedding lookup, text_embeddings.py:71-80,240) in the table's arena dtype, with the zero
    padding of a ragged table included; backward: a deterministic per-token sum straight into the gradient arena."""

    @staticmethod
    def forward(ctx, tokens, st, weight):
        arena = st["arena"]
        table = arena.padded(weight, st["buf"])
        out = ops.embed_gather(tokens, table, st["time_major"], st["want_mask"], st["padding_idx"])
        rows, st["_mask"] = out if st["want_mask"] else (out, None)
        ctx.st = st
        ctx.save_for_backward(tokens)
        return rows

    @staticmethod
    def backward(ctx, drows):
        st = ctx.st
        (tokens,) = ctx.saved_tensors
        gw, acc = st["arena"].grad_views([st["weight"]])
        ops.embed_scatter(tokens, _c(drows), gw, st["time_major"], st["padding_idx"], accumulate=acc)
        return None, None, None


def embed_rows(tokens, emb, arena, time_major=False, master=False, want_mask=False):
    """Rows of ``emb`` (an nn.Embedding) for int64 ``tokens`` [B, T] -> [B*T, W] (W = the table's padded width), from the
    compute-dtype shadow or (``master``) the fp32 masters; with ``want_mask`` also the (B,1,1,T) padding mask of the ids."""
    pad = emb.padding_idx if emb.padding_idx is not None else -1
    st = dict(arena=arena, weight=emb.weight, time_major=time_major, buf="master" if master else "compute",
              want_mask=want_mask, padding_idx=pad)
    tokens = _c(tokens)
    if torch.is_grad_enabled() and emb.weight.requires_grad:
        rows = _EmbedRows.apply(tokens, st, emb.weight)
        return (rows, st["_mask"]) if want_mask else rows
    return ops.embed_gather(tokens, arena.padded(emb.weight, st["buf"]), time_major, want_mask, pad)


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return ops.dropout_apply(_c(x), drop)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout_apply(_c(dy), ctx.drop), None


def dropout(x, drop):
    """nn.Dropout on the HIP path (counter-hash mask, regenerated in backward); identity when ``drop`` is None."""
    if drop is None or drop.p <= 0.0:
        return x
    return _Dropout.apply(x, drop) if torch.is_grad_enabled() and x.requires_grad else ops.dropout_apply(_c(x), drop)


class _LinearResidual(Function):
    """res + lin(x) (bias + residual epilogue of the GEMM): the sum of the two pooled projections (mcan.py:78)."""

    @staticmethod
    def forward(ctx, x, res, st, *params):
        arena, lin = st["arena"], st["lin"]
        x = _c(x)
        y = ops.linear_fwd(x, arena.compute(lin.weight), arena.master_of(lin.bias), EPI_BIAS_RESIDUAL, residual=_c(res))
        ctx.st = st
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        arena, lin = st["arena"], st["lin"]
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        _wgrad(arena, dy, x, [lin.weight], [lin.bias])
        dx = _dx(arena, dy, [lin.weight]) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.needs_input_grad[1] else None), None, *([None] * len(st["params"]))


def linear_residual(x, res, lin, arena):
    params = [lin.weight, lin.bias]
    st = dict(arena=arena, lin=lin, params=params)
    if torch.is_grad_enabled():
        return _LinearResidual.apply(x, res, st, *params)
    return ops.linear_fwd(_c(x), arena.compute(lin.weight), arena.master_of(lin.bias), EPI_BIAS_RESIDUAL, residual=_c(res))


class _AttentionPool(Function):
    """pooled[b] = sum_n softmax_n(fc2(dropout(relu(fc1 x[b,n]))))[n] x[b,n]   (mcan.py:12-25, 70-76): the fc1 GEMM, then ONE
    launch for relu + dropout + the D -> 1 product + softmax over the positions + the weighted sum; backward: one launch for
    the softmax / fc2 / relu chain, fc1's products through the library's GEMMs (the direct gradient of the features is the
    dX product's addend), fc2's weight gradient through the deferred grouped reduce."""

    @staticmethod
    def forward(ctx, feat, st, *params):
        arena, mlp = st["arena"], st["mlp"]
        T = arena.compute_dtype
        feat = _c(feat)
        x = feat if feat.dtype == T else feat.to(T)
        B, N, D = feat.shape
        hpre = ops.linear_fwd(x.reshape(B * N, D), arena.compute(mlp.fc1.weight), arena.master_of(mlp.fc1.bias))
        att, pooled = ops.pool_fwd(feat, hpre, arena.master_of(mlp.fc2.weight).reshape(-1), arena.master_of(mlp.fc2.bias),
                                   st["drop"])
        ctx.st = st
        ctx.save_for_backward(feat, x, hpre, att)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        st = ctx.st
        arena, mlp = st["arena"], st["mlp"]
        feat, x, hpre, att = ctx.saved_tensors
        B, N, D = feat.shape
        dh, dfeat, part, bpart = ops.pool_bwd(feat, hpre, arena.master_of(mlp.fc2.weight).reshape(-1), att,
                                              _c(dpooled.to(hpre.dtype)), st["drop"])
        gw2, acc_w2 = arena.grad_views([mlp.fc2.weight])
        gw2 = gw2.reshape(-1)[:D]  # (row 0 of the [8, D] footprint of the 1 x D matrix)
        gb2, acc_b2 = arena.grad_views([mlp.fc2.bias])  # (the 8-element footprint of the 1-element bias)
        q = _defer_queue(dh)
        if q is not None:
            q.add_reduce(part, B, D, gw2, None, acc_w2)
            q.add_reduce(bpart, B, 8, gb2, None, acc_b2)
        else:
            for g, acc, src in ((gw2, acc_w2, part[:, :D].sum(0)), (gb2[:1], acc_b2, bpart[:, :1].sum(0))):
                if acc:
                    g.add_(src)
                else:
                    g.copy_(src)
        _wgrad(arena, dh, x.reshape(B * N, D), [mlp.fc1.weight], [mlp.fc1.bias])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx(arena, dh, [mlp.fc1.weight], addend=dfeat).reshape(B, N, D)
        return dx, None, *([None] * len(st["params"]))


def attention_pool(feat, mlp, arena, drop):
    """``mlp``: the MCAN ``MLP`` module (fc1, fc2).  feat [B, N, D] (fp32 or the compute dtype) -> pooled [B, D] (compute dtype)."""
    params = [mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias]
    st = dict(arena=arena, mlp=mlp, params=params, drop=drop)
    if torch.is_grad_enabled():
        return _AttentionPool.apply(feat, st, *params)
    T = arena.compute_dtype
    feat = _c(feat)
    B, N, D = feat.shape
    hpre = ops.linear_fwd(feat.to(T).reshape(B * N, D), arena.compute(mlp.fc1.weight), arena.master_of(mlp.fc1.bias))
    return ops.pool_fwd(feat, hpre, arena.master_of(mlp.fc2.weight).reshape(-1), arena.master_of(mlp.fc2.bias), drop)[1]


class _ClassifyLogSoftmax(Function):
    """log_softmax(lin(x)) (mcan.py:79-81) for a classifier whose class count need not be a multiple of 8: the GEMM runs
    on the zero-padded footprint of the weights, log_softmax reads only the real classes."""

    @staticmethod
    def forward(ctx, x, st, *params):
        arena, lin = st["arena"], st["lin"]
        w, bias = _lin_operands(arena, lin)
        x = _c(x)
        y = ops.linear_fwd(x, w, bias)
        logp = ops.log_softmax_fwd(y, lin.weight.shape[0])
        ctx.st = st
        ctx.save_for_backward(x, logp)
        return logp

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        arena, lin = st["arena"], st["lin"]
        x, logp = ctx.saved_tensors
        dy = ops.log_softmax_bwd(_c(g.float()), logp, arena.foot[id(lin.weight)][0], x.dtype)
        _wgrad(arena, dy, x, [lin.weight], [lin.bias] if lin.bias is not None else [])
        dx = _dx(arena, dy, [lin.weight]) if ctx.needs_input_grad[0] else None
        return dx, None, *([None] * len(st["params"]))


def classify_log_softmax(x, lin, arena):
    """log_softmax(lin(x)) over the last dimension; x [..., K] -> fp32 log-probabilities [..., classes]."""
    lead = x.shape[:-1]
    if x.dim() != 2:
        return classify_log_softmax(x.reshape(-1, x.shape[-1]), lin, arena).view(*lead, -1)
    params = [lin.weight] + ([lin.bias] if lin.bias is not None else [])
    st = dict(arena=arena, lin=lin, params=params)
    if torch.is_grad_enabled():
        return _ClassifyLogSoftmax.apply(x, st, *params)
    w, bias = _lin_operands(arena, lin)
    return ops.log_softmax_fwd(ops.linear_fwd(_c(x), w, bias), lin.weight.shape[0])


class _NLL(Function):
    """nn.NLLLoss(ignore_index) on fp32 log-probabilities (classification_task.py:125-127): one launch each way."""

    @staticmethod
    def forward(ctx, logp, target, ignore_index):
        logp = _c(logp.float())
        loss = torch.empty(1, dtype=torch.float32, device=logp.device)
        ops.nll_loss(logp, target, ignore_index, loss=loss)
        ctx.ignore_index = ignore_index
        ctx.save_for_backward(logp, target)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logp, target = ctx.saved_tensors
        d = ops.nll_loss(logp, target, ctx.ignore_index, want_grad=True, gscale=_c(g.float()).reshape(1))
        return d, None, None


def nll_loss(logp, target, ignore_index=-100):
    return _NLL.apply(logp, _c(target), ignore_index)
