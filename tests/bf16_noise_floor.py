"""Where does the bf16 error of a 6-layer MCAN stack come from?  (CPU experiment, oracle only.)

Emulates bf16 rounding at selected points of the fp32 oracle (weights ``w``, GEMM inputs/outputs
``act``, the pre-LayerNorm sum ``pre``, the block output ``y``) and prints the normalised max error
max|a-b|/max(1,max|b|) and the relative L2 error against the un-rounded fp32 oracle at L=6.

Measured (B=16, 100x20, D=512, L=6):   nerr     relL2
    y only                            6.3e-3   3.8e-3
    weights only                      5.9e-3   5.3e-3
    y + act + w  (fp32 pre)           1.0e-2   6.8e-3
    all four (what the HIP bf16 path stores)  1.4e-2   7.8e-3
=> ~1e-2 normalised max error at L=6 is the bf16 noise floor of the algorithm (the HIP path
measures 1.1-1.4e-2, i.e. it adds nothing on top).  Hence the stack-level bf16 bar in
tests/test_modules_gpu.py: relative L2 <= 1e-2 and normalised max <= 2e-2; single modules/blocks
keep the 1e-2 normalised-max bar.
Run:  python tests/bf16_noise_floor.py
"""
import sys, math, torch
sys.path.insert(0, "/root/repo")
import oracle as O
from openvivqa_amd.config import ConfigNode, attention_config
torch.set_num_threads(8)
def r(t, on): return t.bfloat16().float() if on else t
def mha(m, q_in, kv_in, mask, R):
    a = m.attention
    W = lambda lin: r(lin.weight, R["w"])
    q = r(torch.nn.functional.linear(q_in, W(a.fc_q), a.fc_q.bias), R["act"])
    k = r(torch.nn.functional.linear(kv_in, W(a.fc_k), a.fc_k.bias), R["act"])
    v = r(torch.nn.functional.linear(kv_in, W(a.fc_v), a.fc_v.bias), R["act"])
    B, nq, nk = q.shape[0], q.shape[1], k.shape[1]
    sp = lambda t, n: t.view(B, n, a.h, -1).transpose(1, 2)
    o, _ = O.sdpa_core(sp(q, nq), sp(k, nk), sp(v, nk), mask, a.d_k)
    o = r(o.transpose(1, 2).reshape(B, nq, -1), R["act"])
    pre = r(q_in + torch.nn.functional.linear(o, W(a.fc_o), a.fc_o.bias), R["pre"])
    return r(m.layer_norm(pre), R["y"])
def ffn(m, x, R):
    h = r(torch.nn.functional.gelu(torch.nn.functional.linear(x, r(m.fc1.weight, R["w"]), m.fc1.bias)), R["act"])
    pre = r(x + torch.nn.functional.linear(h, r(m.fc2.weight, R["w"]), m.fc2.bias), R["pre"])
    return r(m.layer_norm(pre), R["y"])
def run(te, ve, v, l, vm, lm, R):
    pos = O.sinusoid_positions
    t = r(te.layer_norm(l) + pos(20, 512), R["y"])
    for L in te.layers:
        t = ffn(L.pwff, mha(L.mhatt, t, t, lm, R), R)
    x = r(ve.layer_norm(v) + pos(100, 512), R["y"])
    for L in ve.guided_attn_layers:
        x = mha(L.self_mhatt, x, x, vm, R)
        x = mha(L.guided_mhatt, x, t, lm, R)
        x = ffn(L.pwff, x, R)
    return x, t
torch.manual_seed(77)
sa = attention_config()
te = O.OracleEncoder(ConfigNode(dict(D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa))).eval()
ve = O.OracleGuidedAttentionEncoder(ConfigNode(dict(D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa, GUIDED_ATTENTION=sa))).eval()
g = torch.Generator().manual_seed(5)
v, l = torch.randn(16, 100, 512, generator=g), torch.randn(16, 20, 512, generator=g)
v[1, 90:] = 0; l[2, 12:] = 0
vm, lm = O.padding_mask(v, 0), O.padding_mask(l, 0)
with torch.no_grad():
    ref = run(te, ve, v, l, vm, lm, dict(w=0, act=0, pre=0, y=0))
    chk = ve(v, vm, te(l, lm), lm)
    print("emul == oracle:", (ref[0]-chk).abs().max().item())
    for name, R in [("y only", dict(w=0,act=0,pre=0,y=1)), ("y+pre", dict(w=0,act=0,pre=1,y=1)),
                    ("y+act+w (fp32 pre)", dict(w=1,act=1,pre=0,y=1)), ("all (current)", dict(w=1,act=1,pre=1,y=1)),
                    ("w only", dict(w=1,act=0,pre=0,y=0)), ("act only", dict(w=0,act=1,pre=0,y=0))]:
        out = run(te, ve, v.bfloat16().float(), l.bfloat16().float(), vm, lm, R)
        for o_, r_, nm in ((out[0], ref[0], "vision"), (out[1], ref[1], "text")):
            e = (o_-r_).abs().max().item()/max(1, r_.abs().max().item())
            l2 = ((o_-r_).norm()/r_.norm()).item()
            print(f"{name:22s} {nm:6s} nerr={e:.3e} relL2={l2:.3e} max|ref|={r_.abs().max().item():.2f}")
