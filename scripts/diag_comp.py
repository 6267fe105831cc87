import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import rel_err
from oracle import monosdf_oracle as mo, hashgrid_oracle as hg
from monosdf_amd import ops, _lib
g = torch.Generator().manual_seed(11)
for (N, S, white) in [(33, 98, False), (5, 130, True), (7, 17, False)]:
    z = torch.sort(torch.rand(N, S, generator=g) * 3.5, -1)[0]
    sdf = (torch.randn(N, S, generator=g) * 0.2)
    rgb = torch.rand(N, S, 3, generator=g)
    nrm = torch.randn(N, S, 3, generator=g)
    beta = torch.tensor(0.07)
    ds = torch.rand(N, 1, generator=g) + 0.5
    bg = [0.9, 0.8, 0.7]
    dens = mo.laplace_density(sdf, beta)
    w_o = mo.transmittance_weights(z, dens)[0]
    rgbv_o = (w_o.unsqueeze(-1) * rgb).sum(1)
    dep_o = ds * ((w_o * z).sum(1, keepdim=True) / (w_o.sum(1, keepdim=True) + 1e-8))
    if white: rgbv_o = rgbv_o + (1 - w_o.sum(-1, keepdim=True)) * torch.tensor(bg)
    nm_o = (w_o.unsqueeze(-1) * (nrm / (nrm.norm(2, -1, keepdim=True) + 1e-6))).sum(1)
    w, rgbv, dep, nm = ops.CompositeFunction.apply(z.cuda(), sdf.cuda(), rgb.cuda(), nrm.cuda(), beta.cuda(), ds.cuda(), white, bg)
    print(N, S, 'w', rel_err(w, w_o), 'rgbv', rel_err(rgbv, rgbv_o), 'dep', rel_err(dep, dep_o), 'nm', rel_err(nm, nm_o))
    print('  w row0 mine', w[0, :6].tolist(), '\n  w row0 ref ', w_o[0, :6].tolist())
    print('  w row1 mine', w[1, :6].tolist(), '\n  w row1 ref ', w_o[1, :6].tolist())
# hash per level
ic = dict(num_levels=16, level_dim=2, logmap=19, base_size=16, end_size=2048)
geo = hg.level_geometry(ic)
B, L, C = 513, geo['L'], geo['C']
x = torch.rand(B, 3, generator=g)
emb = (torch.rand(geo['n_entries'], C, generator=g) - 0.5)
out_o, dy_o = hg.encode_forward(x, emb, geo, True)
offs = torch.tensor(geo['offsets'], dtype=torch.int32).cuda()
out = torch.empty(L, B, C, device='cuda'); dy = torch.empty(B, L * 3 * C, device='cuda')
_lib.call('msdf_hash_encode_forward', _lib.ptr(x.cuda()), _lib.ptr(emb.cuda()), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy), _lib.stream_ptr())
for l in range(L):
    print('level', l, 'scale', hg.level_scale(geo, l), 'out err', rel_err(out[l], out_o[l]), 'dy err', rel_err(dy.view(B, L, 3, C)[:, l], dy_o.view(B, L, 3, C)[:, l]))
