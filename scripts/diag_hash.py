import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import rel_err
from oracle import hashgrid_oracle as hg
from monosdf_amd import _lib
g = torch.Generator().manual_seed(13)
for ic in [dict(num_levels=16, level_dim=2, logmap=19, base_size=16, end_size=2048),
           dict(num_levels=4, level_dim=2, logmap=10, base_size=16, end_size=64),
           dict(num_levels=6, level_dim=4, logmap=12, base_size=8, end_size=128)]:
    geo = hg.level_geometry(ic)
    B, L, C = 513, geo['L'], geo['C']
    x = torch.rand(B, 3, generator=g)
    x[:7] = torch.tensor([0.0, 1.0, 0.5]); x[7:11] = torch.tensor([1.2, 0.5, -0.1])
    emb = (torch.rand(geo['n_entries'], C, generator=g) - 0.5)
    out_o, dy_o = hg.encode_forward(x, emb, geo, True)
    offs = torch.tensor(geo['offsets'], dtype=torch.int32).cuda()
    xg, eg = x.cuda(), emb.cuda()
    out = torch.empty(L, B, C, device='cuda'); dy = torch.empty(B, L * 3 * C, device='cuda')
    st = _lib.stream_ptr()
    _lib.call('msdf_hash_encode_forward', _lib.ptr(xg), _lib.ptr(eg), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy), st)
    grad = torch.randn(L, B, C, generator=g); ggi = torch.randn(B, 3, generator=g)
    g2_o = hg.second_backward_embedding(grad, x, ggi, geo, geo['n_entries'])
    gg, g2 = torch.zeros(L, B, C, device='cuda'), torch.zeros_like(eg)
    _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad.cuda()), _lib.ptr(xg), _lib.ptr(eg), _lib.ptr(offs), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy), _lib.ptr(ggi.cuda()), _lib.ptr(gg), _lib.ptr(g2), st)
    print('config', ic, 'total', rel_err(g2, g2_o))
    off = geo['offsets']
    for l in range(L):
        a, b = g2[off[l]:off[l+1]].cpu(), g2_o[off[l]:off[l+1]]
        e = (a - b).abs()
        print('  level', l, 'max abs err', e.max().item(), 'max ref', b.abs().max().item(), 'n bad', int((e > 1e-4 * b.abs().max()).sum()))
    # remove the special points
    x2 = x.clone(); x2[:11] = 0.3
    g2_o2 = hg.second_backward_embedding(grad, x2, ggi, geo, geo['n_entries'])
    g2b = torch.zeros_like(eg); dy2 = torch.empty_like(dy)
    _lib.call('msdf_hash_encode_forward', _lib.ptr(x2.cuda()), _lib.ptr(eg), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy2), st)
    _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad.cuda()), _lib.ptr(x2.cuda()), _lib.ptr(eg), _lib.ptr(offs), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy2), _lib.ptr(ggi.cuda()), _lib.ptr(gg), _lib.ptr(g2b), st)
    print('  without border points:', rel_err(g2b, g2_o2))
