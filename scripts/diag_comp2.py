import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import rel_err
from oracle import monosdf_oracle as mo
from monosdf_amd import ops
g = torch.Generator().manual_seed(11)
for (N, S, white) in [(33, 98, False), (7, 17, False), (5, 130, True)]:
    z = torch.sort(torch.rand(N, S, generator=g) * 3.5, -1)[0]
    sdf0 = (torch.randn(N, S, generator=g) * 0.2)
    rgb0 = torch.rand(N, S, 3, generator=g)
    nrm0 = torch.randn(N, S, 3, generator=g)
    ds = torch.rand(N, 1, generator=g) + 0.5
    bg = [0.9, 0.8, 0.7]
    cs = (torch.randn(N, 3, generator=g), torch.randn(N, 1, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, S, generator=g) * 0.1)
    for term in range(4):
        sdf = sdf0.clone().requires_grad_(True); rgb = rgb0.clone().requires_grad_(True); nrm = nrm0.clone().requires_grad_(True)
        beta = torch.tensor(0.07, requires_grad=True)
        dens = mo.laplace_density(sdf, beta)
        w_o = mo.transmittance_weights(z, dens)[0]
        rgbv_o = (w_o.unsqueeze(-1) * rgb).sum(1)
        dep_o = ds * ((w_o * z).sum(1, keepdim=True) / (w_o.sum(1, keepdim=True) + 1e-8))
        if white: rgbv_o = rgbv_o + (1 - w_o.sum(-1, keepdim=True)) * torch.tensor(bg)
        nm_o = (w_o.unsqueeze(-1) * (nrm / (nrm.norm(2, -1, keepdim=True) + 1e-6))).sum(1)
        outs_o = [rgbv_o, dep_o, nm_o, w_o]
        g_o = torch.autograd.grad((cs[term] * outs_o[term]).sum(), [sdf, rgb, nrm, beta], allow_unused=True)
        leaf = lambda t: t.detach().cuda().requires_grad_(True)
        sdf_g, rgb_g, nrm_g, beta_g = leaf(sdf), leaf(rgb), leaf(nrm), leaf(beta)
        w, rgbv, dep, nm = ops.CompositeFunction.apply(z.cuda(), sdf_g, rgb_g, nrm_g, beta_g, ds.cuda(), white, bg)
        outs = [rgbv, dep, nm, w]
        (cs[term].cuda() * outs[term]).sum().backward()
        res = []
        for nm_, a, b in zip(['sdf', 'rgb', 'nrm', 'beta'], [sdf_g, rgb_g, nrm_g, beta_g], g_o):
            if b is None: b = torch.zeros_like(a.detach().cpu())
            res.append('%s %.2e' % (nm_, rel_err(a.grad, b)))
        print(N, S, 'term', ['rgb', 'depth', 'normal', 'weights'][term], ' | '.join(res), ' beta grad mine', beta_g.grad.item(), 'ref', g_o[3].item())
        if term == 3:
            e = (sdf_g.grad.cpu() - g_o[0]).abs()
            idx = torch.nonzero(e > 1e-3 * g_o[0].abs().max())
            print('   bad sdf grad positions (ray, sample):', idx[:10].tolist())
