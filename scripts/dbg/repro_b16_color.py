import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_gpu_fullsize as t
model = t._model('bf16x3')
rn = model.rendering_network
g = torch.Generator(device='cuda').manual_seed(0)
P = 1024*98
pts = torch.randn(P,3,device='cuda',generator=g); nrm = torch.randn(P,3,device='cuda',generator=g).requires_grad_(True)
dirs = torch.nn.functional.normalize(torch.randn(1024,3,device='cuda',generator=g),dim=-1).unsqueeze(1).repeat(1,98,1).reshape(-1,3)
feat = (torch.randn(P,256,device='cuda',generator=g)*0.3).requires_grad_(True)
go = torch.randn(P,3,device='cuda',generator=g)
ref=None
for rep in range(6):
    for p in rn.parameters(): p.grad=None
    nrm.grad=None; feat.grad=None
    rgb = rn(pts, nrm, dirs[::98].contiguous(), feat, torch.arange(1024,device='cuda'), if_pixel_input=True, samples_per_ray=98)['rgb']
    (rgb*go).sum().backward()
    cur = {'rgb':rgb.detach().clone(),'g_nrm':nrm.grad.clone(),'g_feat':feat.grad.clone()}
    cur.update({n:p.grad.clone() for n,p in rn.named_parameters()})
    if ref is None: ref=cur
    else:
        d = (ref['g_feat']-cur['g_feat']).abs()
        rows = (d.amax(1) > 0).nonzero().flatten()
        print(rep, 'g_feat rows differing', rows.numel(), 'of', d.shape[0], 'first', rows[:12].tolist(), 'max diff %.3e max val %.3e' % (d.max().item(), ref['g_feat'].abs().max().item()),
              'cols of first row', (d[rows[0]] > 0).nonzero().flatten()[:20].tolist() if rows.numel() else None, flush=True)
