import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openvivqa_amd as A
import openvivqa_amd.modules as M
from openvivqa_amd.config import ConfigNode, attention_config
import openvivqa_amd.utils as U
DEV = "cuda"
A.set_compute_dtype(torch.bfloat16)
torch.manual_seed(901)
sa = attention_config()
te = M.Encoder(ConfigNode(dict(ARCHITECTURE="Encoder", D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa))).to(DEV).eval()
ve = M.GuidedAttentionEncoder(ConfigNode(dict(ARCHITECTURE="GuidedAttentionEncoder", D_MODEL=512, LAYERS=6, SELF_ATTENTION=sa, GUIDED_ATTENTION=attention_config()))).to(DEV).eval()
gen = torch.Generator().manual_seed(902)
v = torch.randn(4, 100, 512, generator=gen); l = torch.randn(4, 20, 512, generator=gen)
v[1, 90:] = 0; l[2, 12:] = 0
wv = torch.randn(4, 100, 512, generator=gen).to(DEV); wl = torch.randn(4, 20, 512, generator=gen).to(DEV)
def run(defer):
    os.environ["OVQA_DEFER_WGRAD"] = "1" if defer else "0"
    for m in (te, ve):
        for p in m.parameters(): p.grad = None
    vd, ld = v.to(DEV).requires_grad_(True), l.to(DEV).requires_grad_(True)
    vm, lm = U.generate_padding_mask(vd, 0), U.generate_padding_mask(ld, 0)
    lo = te(features=ld, padding_mask=lm)
    vo = ve(vision_features=vd, vision_padding_mask=vm, language_features=lo, language_padding_mask=lm)
    ((vo.float() * wv).mean() + (lo.float() * wl).mean()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for mod, pre in ((te, "t."), (ve, "v.")) for n, p in ((pre + k, q) for k, q in mod.named_parameters())}
g0 = run(False); g1 = run(True); g2 = run(True)
bad = 0
for k in g0:
    e = ((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item()
    e2 = ((g2[k] - g1[k]).norm() / (g1[k].norm() + 1e-30)).item()
    if e > 1e-3 or e2 > 1e-3:
        bad += 1
        print(f"{k:70s} defer-vs-direct {e:.3e}  defer-vs-defer {e2:.3e}  norms {g0[k].norm().item():.3e} {g1[k].norm().item():.3e}")
print("mismatching params:", bad, "of", len(g0))
