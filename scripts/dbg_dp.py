import torch, os, sys, tempfile
sys.path.insert(0, "/root/repo/tests")
import torch.multiprocessing as mp
import dp_helpers as H

def worker(rank, rdv, q):
    import torch.distributed as dist
    H.patch_cpu_ops()
    dist.init_process_group("gloo", init_method="file://"+rdv, rank=rank, world_size=2)
    model, ts = H.make_step(0.02)
    ts.static_inputs=[t.clone() for t in H.batch(rank)]
    ts._discover_foreign()
    ts._fwd_bwd(on_phase=ts._release)
    q.put((rank, ts.arena.grad.clone()))
    dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    rdv = tempfile.mkdtemp()+"/s"
    ps=[ctx.Process(target=worker,args=(r,rdv,q)) for r in range(2)]
    [p.start() for p in ps]
    res=dict(q.get(timeout=120) for _ in range(2))
    [p.join() for p in ps]
    H.patch_cpu_ops()
    model, ts = H.make_step(0.0)
    g=torch.zeros_like(ts.arena.grad)
    ts.static_inputs=[t.clone() for t in H.batch(0)]; ts._discover_foreign()
    for r in range(2):
        ts.static_inputs=[t.clone() for t in H.batch(r)]; ts._fwd_bwd(); g+=ts.arena.grad
    d=(res[0]-g).abs()
    print("ranks equal", torch.equal(res[0],res[1]), "max diff", d.max().item())
    names={id(p):n for n,p in model.named_parameters()}
    for p in ts.arena.params:
        o=ts.arena.offsets[id(p)]; e=d[o:o+p.numel()].max().item()
        if e>1e-6: print(names[id(p)], o, e, g[o:o+p.numel()].abs().max().item())
