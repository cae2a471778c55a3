"""TEST-ONLY stand-in for ``openvivqa_amd.ops`` implemented with plain torch math on CPU.

Purpose: exercise the HOST logic of the product (autograd plumbing in functional.py, arena
bookkeeping, module wiring) in the CPU-only container.  It is never imported by the product;
tests monkeypatch it in.  Semantics follow include/ovqa_hip.h exactly (same arguments, same
in-place/accumulate behaviour), dropout excluded (p must be 0).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL = 0, 1, 2


@dataclass
class DropSpec:
    p: float
    seed: int
    site: int
    step: Optional[torch.Tensor] = None


def workspace(device):
    return torch.empty(1)


def _gelu_grad(u):
    return 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


def linear_fwd(x, w, bias=None, epilogue=EPI_BIAS, residual=None, want_preact=False, drop=None, out=None,
               preact_out=None):
    assert drop is None or drop.p == 0
    u = x.float() @ w.float().t()
    if bias is not None:
        u = u + bias
    pre = None
    if epilogue == EPI_BIAS_GELU:
        pre = u.to(x.dtype).reshape(-1, w.shape[0])
        y = torch.nn.functional.gelu(u)
    elif epilogue == EPI_BIAS_RESIDUAL:
        y = residual.float() + u
    else:
        y = u
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        y = out
    return (y, pre) if want_preact else y


class LnRef:
    def __init__(self, pre, mean, rstd, gamma, beta, eps):
        self.pre, self.mean, self.rstd, self.gamma, self.beta, self.eps = pre, mean, rstd, gamma, beta, eps

    def materialize(self):
        D = self.pre.shape[-1]
        x = self.pre.reshape(-1, D)
        return ((x - self.mean[:, None]) * self.rstd[:, None] * self.gamma + self.beta).reshape(self.pre.shape)


def linear_fwd_res32(x, w, bias, residual, drop=None):
    assert drop is None or drop.p == 0
    u = x.float() @ w.float().t()
    if bias is not None:
        u = u + bias
    res = residual.materialize() if isinstance(residual, LnRef) else residual
    assert res.dtype == torch.float32
    return res + u


def linear_bwd_data(dy, w, preact=None, drop=None, out=None, addend=None):
    dx = dy.float() @ w.float()
    if preact is not None:
        dx = dx * _gelu_grad(preact.float()).reshape(dx.shape)
    if addend is not None:
        dx = dx + addend.float()
    if out is not None:
        out.copy_(dx.to(out.dtype))
        return out
    return dx.to(dy.dtype)


def bias_grad(dy, db, accumulate=False):
    s = dy.reshape(-1, dy.shape[-1]).float().sum(0)
    if accumulate:
        db.add_(s)
    else:
        db.copy_(s)


class WgradQueue:
    def __init__(self):
        self.items = []
        self.inflight = []
        self.reduces = []
        self.hold_reduces = False
        self.hold_items = False

    def finish(self):
        if not self.hold_items:
            self.flush()

    def reserve(self, n):
        pass

    def abandon(self):
        self.items = []

    def add(self, dy, x, dw, accumulate, db=None, accumulate_db=False):
        self.items.append((dy, x, dw, accumulate, db, accumulate_db))

    def flush(self):
        items, self.items = self.items, []
        for dy, x, dw, acc, db, acc_b in items:
            linear_bwd_weight(dy, x, dw, db, accumulate=acc, accumulate_db=acc_b)


def linear_bwd_weight(dy, x, dw, db=None, accumulate=False, accumulate_db=None):
    d2, x2 = dy.reshape(-1, dy.shape[-1]).float(), x.reshape(-1, x.shape[-1]).float()
    g = d2.t() @ x2
    if accumulate:
        dw.add_(g.view_as(dw))
    else:
        dw.copy_(g.view_as(dw))
    if db is not None:
        s = d2.sum(0)
        if (accumulate if accumulate_db is None else accumulate_db):
            db.add_(s.view_as(db))
        else:
            db.copy_(s.view_as(db))


def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=None, pos=None, save_stats=True, want_f32=False):
    xf = x.float()
    mean = xf.mean(-1)
    var = xf.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean[..., None]) * rstd[..., None] * gamma + beta
    if pos is not None:
        y = y + pos[None]
    if want_f32:
        return y.to(out_dtype or x.dtype), y, mean.reshape(-1), rstd.reshape(-1)
    return y.to(out_dtype or x.dtype), mean.reshape(-1), rstd.reshape(-1)


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, drop=None, dx_dtype=None, accumulate=False, defer=None):
    assert defer is None
    assert drop is None or drop.p == 0
    D = x.shape[-1]
    xf, df = x.float().reshape(-1, D), dy.float().reshape(-1, D)
    xh = (xf - mean[:, None]) * rstd[:, None]
    dg, db = (df * xh).sum(0), df.sum(0)
    if accumulate:
        dgamma.add_(dg)
        dbeta.add_(db)
    else:
        dgamma.copy_(dg)
        dbeta.copy_(db)
    dyg = df * gamma
    dx = rstd[:, None] * (dyg - dyg.mean(-1, keepdim=True) - xh * (dyg * xh).mean(-1, keepdim=True))
    dx = dx.reshape(x.shape).to(dx_dtype or dy.dtype)
    return dx, dx


def _heads(t, H):
    B, n, F = t.shape
    return t.float().reshape(B, n, H, F // H).transpose(1, 2)


def attention_fwd(q, k, v, mask, H, scale=None, need_att=False, save_lse=True, att_drop=None, lo_out=None):
    assert att_drop is None or att_drop.p == 0
    qh, kh, vh = _heads(q, H), _heads(k, H), _heads(v, H)
    scale = scale or 1.0 / math.sqrt(qh.shape[-1])
    s = qh @ kh.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, -1)
    if q.dtype == torch.bfloat16:  # like the MFMA kernel: bf16 exp(s - max) into P.V, fp32 row sum
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (e.bfloat16().float() @ vh) / e.sum(-1, keepdim=True)
    else:
        o = p @ vh
    o32 = o.transpose(1, 2).reshape(q.shape[0], q.shape[1], -1)
    o = o32.to(q.dtype)
    if lo_out is not None and q.dtype == torch.bfloat16:  # o + o_lo = P V with the unrounded probabilities
        lo_out.append(((p @ vh).transpose(1, 2).reshape(q.shape[0], q.shape[1], -1) - o.float()).to(q.dtype))
    return o, torch.logsumexp(s, -1), (p.to(q.dtype) if need_att else None)


def attention_qkv_fwd(x, w, bias, mask, H, scale=None, save_lse=True, lo_out=None):
    qkv = linear_fwd(x, w, bias)
    n3 = qkv.shape[-1] // 3
    o, lse, _ = attention_fwd(qkv[..., :n3], qkv[..., n3:2 * n3], qkv[..., 2 * n3:], mask, H, scale, save_lse=save_lse,
                              lo_out=lo_out)
    return qkv, o, lse


def attention_bwd_do_ok(dy, wt, q, k, mask, H):
    return False  # (the CPU plumbing tests keep the two-kernel form)


def attention_fwd_prefix_lm_ok(q, H):
    return False  # (the dense mask on the CPU)


def attention_q_fwd(x, w, bias, k, v, mask, H, scale=None, save_lse=True, lo_out=None):
    q = linear_fwd(x, w, bias)
    o, lse, _ = attention_fwd(q, k, v, mask, H, scale, save_lse=save_lse, lo_out=lo_out)
    return q, o, lse


def attention_bwd(d_o, q, k, v, o, lse, mask, H, scale=None, dq=None, dk=None, dv=None, d_att=None, d_lse=None,
                  att_drop=None, o_lo=None):
    assert att_drop is None or att_drop.p == 0
    qh, kh, vh, gh = _heads(q, H), _heads(k, H), _heads(v, H), _heads(d_o, H)
    scale = scale or 1.0 / math.sqrt(qh.shape[-1])
    s = qh @ kh.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask
    p = torch.exp(s - lse[..., None])
    dp = gh @ vh.transpose(-1, -2)
    if d_att is not None:
        dp = dp + d_att.float()
    delta = (p * dp).sum(-1, keepdim=True)
    if d_lse is not None:
        delta = delta - d_lse[..., None].float()
    ds = p * (dp - delta)
    rdq = (ds @ kh * scale).transpose(1, 2).reshape(q.shape)
    rdk = (ds.transpose(-1, -2) @ qh * scale).transpose(1, 2).reshape(k.shape)
    rdv = (p.transpose(-1, -2) @ gh).transpose(1, 2).reshape(v.shape)
    outs = []
    for dst, r, like in ((dq, rdq, q), (dk, rdk, k), (dv, rdv, v)):
        if dst is None:
            outs.append(r.to(like.dtype))
        else:
            dst.copy_(r.to(dst.dtype))
            outs.append(dst)
    return tuple(outs)


def pointer_score(q, k, scale, add_mask=None, key_fill=None, query_fill=None):
    s = q.float() @ k.float().transpose(1, 2) * scale
    if add_mask is not None:
        s = s + add_mask[:, None, :]
    if key_fill is not None:
        s = s.masked_fill(key_fill.bool()[:, None, :], float("-inf"))
    if query_fill is not None:
        s = s.masked_fill(query_fill.bool()[:, :, None], float("-inf"))
    return s


def batched_gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, out_dtype=None):
    a2 = a.float().transpose(1, 2) if trans_a else a.float()
    b2 = b.float().transpose(1, 2) if trans_b else b.float()
    return (alpha * (a2 @ b2)).to(out_dtype or a.dtype)


def cast(src, dst):
    dst.copy_(src)
    return dst


def increment_step(step, also=None):
    step.add_(1)
    if also is not None:
        also.add_(1)


def begin_step(step, also, lr_table, lr_out):
    lr_out.copy_(lr_table[int(step.item()) % lr_table.numel()].reshape(lr_out.shape))
    step += 1
    if also is not None:
        also += 1


def adam_step(param, grad, exp_avg, exp_avg_sq, shadow, lr, step, lr_scale=None, betas=(0.9, 0.98), eps=1e-8,
              weight_decay=0.0, grad_scale=1.0):
    t = float(step.item())
    lr_eff = lr * (float(lr_scale.item()) if lr_scale is not None else 1.0)
    g = grad * grad_scale + weight_decay * param
    exp_avg.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    denom = exp_avg_sq.sqrt() / math.sqrt(1 - betas[1] ** t) + eps
    param.addcdiv_(exp_avg, denom, value=-lr_eff / (1 - betas[0] ** t))
    if shadow is not None:
        shadow.copy_(param)


def sq_loss_fwd_bwd(x, loss, want_grad=True, accumulate=False, target=None):
    if target is not None:
        x = x.float() - target.float()
    v = x.float().pow(2).mean()
    if accumulate:
        loss.add_(v)
    else:
        loss.fill_(float(v))
    return (2 * x.float() / x.numel()).to(x.dtype) if want_grad else None


def gelu_bwd(dy, u, drop=None):
    assert drop is None or drop.p == 0
    uf = u.float()
    cdf = 0.5 * (1 + torch.erf(uf * 0.7071067811865476))
    pdf = torch.exp(-0.5 * uf * uf) * 0.3989422804014327
    return (dy.float() * (cdf + uf * pdf)).to(dy.dtype)


def row_padding_mask(x, pad_value=0.0):
    B, N, D = x.shape
    return ((x.float().sum(-1) == pad_value * D).float() * -10e4).reshape(B, 1, 1, N)


def lstm_fwd(x_tb, w_ih, w_hh, b_ih, b_hh, B, T, want_lp=False):
    H = w_hh.shape[1]
    xg = (x_tb.float() @ w_ih.float().t() + b_ih).view(T, B, 4 * H)
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    hseq, gates, cs, ys = [h.to(x_tb.dtype)], [], [], []
    for t in range(T):
        g = xg[t] + hseq[-1].float() @ w_hh.float().t() + b_hh
        i, f, gg, o = g.chunk(4, -1)
        i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
        c = f * c + i * gg
        h = o * torch.tanh(c)
        gates.append(torch.cat([i, f, gg, o], -1))
        cs.append(c)
        ys.append(h)
        hseq.append(h.to(x_tb.dtype))
    saved = torch.cat([torch.stack(gates), torch.stack(cs)], -1)  # [T, B, 5H]
    y = torch.stack(ys, 1).contiguous()
    return (y, torch.cat(hseq, 0), saved, None, y.to(x_tb.dtype)) if want_lp else (y, torch.cat(hseq, 0), saved, None)


def lstm_bwd(dy, w_hh, w_hh_t, saved, B, T, I):
    H = w_hh.shape[1]
    gates, cs = saved[..., :4 * H], saved[..., 4 * H:]
    dg_next, carry, out = None, torch.zeros(B, H), [None] * T
    for t in reversed(range(T)):
        dh = dy[:, t].float()
        if dg_next is not None:
            dh = dh + dg_next.float() @ w_hh.float()
        i, f, g, o = gates[t].chunk(4, -1)
        cp = cs[t - 1] if t > 0 else torch.zeros(B, H)
        tc = torch.tanh(cs[t])
        dc = dh * o * (1 - tc * tc) + carry
        carry = dc * f
        dg_next = torch.cat([dc * g * i * (1 - i), dc * cp * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)],
                            -1).to(w_hh.dtype)
        out[t] = dg_next
    return torch.cat(out, 0), None


def _rc(B, T, time_major):
    r = torch.arange(B * T)
    return (r % B, r // B) if time_major else (r // T, r % T)


def embed_gather(tokens, table, time_major=False, want_mask=False, padding_idx=-1):
    B, T = tokens.shape
    b, t = _rc(B, T, time_major)
    rows = table[tokens[b, t]].clone()
    if want_mask:
        return rows, ((tokens == padding_idx).float() * -10e4).reshape(B, 1, 1, T)
    return rows


def embed_scatter(tokens, drows, dtable, time_major=False, padding_idx=-1, accumulate=False):
    B, T = tokens.shape
    b, t = _rc(B, T, time_major)
    tok = tokens[b, t]
    g = torch.zeros_like(dtable)
    keep = tok != padding_idx
    g.index_add_(0, tok[keep], drows.float()[keep])
    if accumulate:
        dtable.add_(g)
    else:
        dtable.copy_(g)


def dropout_apply(x, drop):
    assert drop is None or drop.p == 0
    return x


def pool_fwd(feat, hpre, w2, b2, drop=None):
    assert drop is None or drop.p == 0
    B, N, D = feat.shape
    logit = torch.relu(hpre.float()).reshape(B, N, D) @ w2.float() + (b2.float()[0] if b2 is not None else 0.0)
    att = torch.softmax(logit, dim=1)
    return att, (feat.float() * att[..., None]).sum(1).to(hpre.dtype)


def pool_bwd(feat, hpre, w2, att, dpooled, drop=None):
    assert drop is None or drop.p == 0
    B, N, D = feat.shape
    dp = dpooled.float()
    datt = (feat.float() * dp[:, None, :]).sum(-1)
    dl = att * (datt - (att * datt).sum(1, keepdim=True))
    h = hpre.float().reshape(B, N, D)
    dh = (dl[..., None] * w2.float() * (h > 0)).reshape(B * N, D).to(hpre.dtype)
    dfeat = (att[..., None] * dp[:, None, :]).reshape(B * N, D).to(hpre.dtype)
    part = torch.zeros(B, 2 * D)
    part[:, :D] = (dl[..., None] * torch.relu(h)).sum(1)
    bpart = torch.zeros(B, 16)
    bpart[:, 0] = dl.sum(1)
    return dh, dfeat, part, bpart


def log_softmax_fwd(x, n=None):
    n = x.shape[1] if n is None else n
    return torch.log_softmax(x[:, :n].float(), dim=-1)


def log_softmax_bwd(g, logp, ld, dtype):
    M, n = logp.shape
    dx = torch.zeros(M, ld, dtype=dtype)
    dx[:, :n] = (g - torch.exp(logp) * g.sum(-1, keepdim=True)).to(dtype)
    return dx


def nll_loss(logp, target, ignore_index=-100, loss=None, want_grad=False, gscale=None, accumulate=False):
    valid = target != ignore_index
    cnt = valid.sum().clamp(min=1).float()
    if loss is not None:
        v = -(logp[valid, target[valid]]).sum() / cnt
        if accumulate:
            loss.add_(v)
        else:
            loss.fill_(float(v))
    if not want_grad:
        return None
    d = torch.zeros_like(logp)
    sc = (gscale.float().reshape(-1)[0] if gscale is not None else 1.0) / cnt
    d[valid, target[valid]] = -sc
    return d
