import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_gpu_fullsize as t
import bench
for prec in ('bf16x3','fp32'):
    model = t._model(prec)
    rays = bench.make_rays(t.N, 1, 'cuda')
    out, loss = t._step(model, rays)
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    o1 = {k: v.clone() for k, v in out.items()}
    for rep in range(6):
        out2, loss2 = t._step(model, rays)
        bad = [(n, (p.grad-g1[n]).abs().max().item(), g1[n].abs().max().item()) for n,p in model.named_parameters() if p.grad is not None and not torch.equal(p.grad,g1[n])]
        obad = [k for k in o1 if not torch.equal(o1[k], out2[k])]
        print(prec, rep, len(bad), bad[:2], 'outputs differing:', obad, flush=True)
