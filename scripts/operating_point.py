"""CPU-only: the bf16 storage format's OWN error on every gradient tensor of the MCAN L=6 stacks (emulating oracle vs fp32
oracle, no kernel involved) at candidate operating points -- how tests/test_modules_gpu.py::_sharpen was chosen."""
import sys, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle as O
from golden_cases import oracle_namespace
import test_modules_gpu as T
torch.set_num_threads(8)
def run(score, value, B=8):
    te_o, ve_o = T._mcan_pair(oracle_namespace(), 6, 41)
    with torch.no_grad():
        for m in (te_o, ve_o):
            for k,p in m.named_parameters():
                if ".fc_q." in k: p.mul_(score)
                if ".fc_v." in k: p.mul_(value)
    te_o.eval(); ve_o.eval()
    gen = torch.Generator().manual_seed(8)
    v, l = torch.randn(B,100,512,generator=gen), torch.randn(B,20,512,generator=gen)
    for i in range(B):
        v[i, 84+i%16:] = 0; l[i, 8+i%12:] = 0
    wv, wl = torch.randn(v.shape,generator=gen), torch.randn(l.shape,generator=gen)
    gs=[]
    for emu in (False, True):
        te_o.zero_grad(set_to_none=True); ve_o.zero_grad(set_to_none=True)
        with O.emulate_bf16(emu):
            lo = te_o(l, O.padding_mask(l,0)); vo = ve_o(v, O.padding_mask(v,0), lo, O.padding_mask(l,0))
            ((vo*wv).mean()+(lo*wl).mean()).backward()
        gs.append({pre+k: p.grad.clone() for pre,m in (("te.",te_o),("ve.",ve_o)) for k,p in m.named_parameters()})
        outs = (lo.detach(), vo.detach()) if not emu else outs
        if emu: oe=(lo.detach(), vo.detach())
    rel=lambda a,b: float((a-b).norm()/b.norm())
    worst=sorted(((rel(gs[1][k],gs[0][k]),k) for k in gs[0] if not k.endswith("fc_k.bias")), reverse=True)[:6]
    nerr=lambda a,b: float((a-b).abs().max()/max(1.0,float(b.abs().max())))
    print(f"score x{score} value x{value}: fwd emu-vs-fp32 {nerr(oe[0],outs[0]):.2e} {nerr(oe[1],outs[1]):.2e}")
    for e,k in worst: print(f"   {e:.3e} {k}")
    k5="ve.guided_attn_layers.5.self_mhatt.attention.fc_q.weight"
    print("   ratio L5 self fc_q/fc_v:", float(gs[0][k5].norm()/gs[0][k5.replace('fc_q','fc_v')].norm()))
for sc,va in ((1.0,1.0),(2.0,0.3),(2.0,0.1),(3.0,0.2)):
    t=time.time(); run(sc,va); print("   t", time.time()-t)
