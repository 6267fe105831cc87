import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
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
    dy = dy_o.cuda().contiguous()
    st = _lib.stream_ptr()
    grad = torch.randn(L, B, C, generator=g); grad_g = grad.cuda()
    ggi = torch.randn(B, 3, generator=g); ggi_g = ggi.cuda()
    g2_o = hg.second_backward_embedding(grad.double(), x.double(), ggi.double(), geo, geo['n_entries']) if False else hg.second_backward_embedding(grad, x, ggi, geo, geo['n_entries'])
    n = geo['n_entries']
    gg, g2 = torch.zeros(L, B, C, device='cuda'), torch.zeros_like(eg)
    _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad_g), _lib.ptr(xg), _lib.ptr(eg), _lib.ptr(offs), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy), _lib.ptr(ggi_g), _lib.ptr(gg), _lib.ptr(g2), st)
    nbytes = _lib.load().msdf_hash_scatter_workspace_bytes(B, C, L, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    g2b = torch.zeros_like(eg)
    _lib.call('msdf_hash_encode_second_backward_ws', _lib.ptr(grad_g), _lib.ptr(xg), _lib.ptr(eg), _lib.ptr(offs), B, 3, C, L, geo['S'], geo['H'], 1, _lib.ptr(dy), _lib.ptr(ggi_g), _lib.ptr(gg), _lib.ptr(g2b), n, _lib.ptr(ws), nbytes, st)
    a, b, o = g2.cpu(), g2b.cpu(), g2_o
    print('config', ic, 'max ref', o.abs().max().item())
    for nm, u, v in (('atomic-oracle', a, o), ('binned-oracle', b, o), ('binned-atomic', b, a)):
        d = (u - v).abs()
        i = int(d.max(1)[0].argmax())
        lvl = max(l for l in range(L) if geo['offsets'][l] <= i)
        print('  %-14s max %.3e at entry %d (level %d): %s vs %s ; support equal %s' % (nm, d.max().item(), i, lvl, u[i].tolist(), v[i].tolist(), bool(torch.equal(u != 0, v != 0))))
