"""Self-consistency + hand-computed index KATs of the hash-grid restatement.

The reference has no runnable hash-grid code here (CUDA only) and no golden vectors, so this part of
the oracle is "parity unpinned" by the reference's outputs; these tests pin it to the CUDA source read
as text (index arithmetic, cu:35-72) and to its own calculus (finite differences, adjointness)."""
import numpy as np
import torch

from oracle import hashgrid_oracle as hg

IC = dict(num_levels=16, level_dim=2, logmap=19, base_size=16, end_size=2048)


def test_level_geometry_matches_survey_probe():
    geo = hg.level_geometry(IC)
    assert geo['n_entries'] == 6098108
    sizes = np.diff(geo['offsets'])
    assert list(sizes[:5]) == [4096, 12167, 29791, 79507, 205379]   # dense levels: res^3, res = 16,23,31,43,59
    assert all(s == 524288 for s in sizes[5:])
    assert abs(np.float32(geo['S']) - np.float32(0.46666667)) < 1e-7
    assert hg.level_scale(geo, 0) == (15.0, 16)
    assert hg.level_scale(geo, 15)[1] == 2048


def test_index_known_answers():
    # hashed level: (1*1) ^ (2*2654435761 mod 2^32) ^ (3*805459861 mod 2^32), then mod 2^19   (cu:35-51,67-71)
    p = torch.tensor([[1, 2, 3]], dtype=torch.int64)
    h = 1 ^ ((2 * 2654435761) & 0xFFFFFFFF) ^ ((3 * 805459861) & 0xFFFFFFFF)
    assert int(hg.grid_index(p, 524288, 2048)[0]) == h % 524288
    # dense level 0 (res 16, 4096 entries): stride uses `resolution`, so the +1 corner at x=1 aliases (cu:62-63)
    p = torch.tensor([[15, 15, 15], [16, 15, 15], [16, 16, 16], [0, 0, 0]], dtype=torch.int64)
    idx = hg.grid_index(p, 4096, 16)
    assert idx.tolist() == [15 + 15 * 16 + 15 * 256, (16 + 15 * 16 + 15 * 256) % 4096,
                            (16 + 16 * 16 + 16 * 256) % 4096, 0]
    # a level whose res^3 exceeds the table switches to the hash after the stride overflows
    p = torch.tensor([[7, 9, 11]], dtype=torch.int64)
    h = 7 ^ ((9 * 2654435761) & 0xFFFFFFFF) ^ ((11 * 805459861) & 0xFFFFFFFF)
    assert int(hg.grid_index(p, 524288, 81)[0]) == h % 524288


def _small():
    ic = dict(num_levels=5, level_dim=2, logmap=9, base_size=4, end_size=40)
    geo = hg.level_geometry(ic)
    g = torch.Generator().manual_seed(0)
    emb = torch.rand(geo['n_entries'], 2, generator=g, dtype=torch.float64) - 0.5
    x = torch.rand(40, 3, generator=g, dtype=torch.float64) * 0.9 + 0.05
    return geo, emb, x


def test_dy_dx_is_the_derivative_of_the_forward():
    geo, emb, x = _small()
    out, dy = hg.encode_forward(x, emb, geo, True)
    B, L, C = x.shape[0], geo['L'], geo['C']
    dy = dy.view(B, L, 3, C)
    eps = 1e-7
    for d in range(3):
        xp, xm = x.clone(), x.clone()
        xp[:, d] += eps
        xm[:, d] -= eps
        fd = (hg.encode_forward(xp, emb, geo, False)[0] - hg.encode_forward(xm, emb, geo, False)[0]) / (2 * eps)
        assert torch.allclose(fd.permute(1, 0, 2), dy[:, :, d, :], rtol=1e-4, atol=1e-5)


def test_backward_is_the_adjoint_and_second_order_terms():
    geo, emb, x = _small()
    B, L, C = x.shape[0], geo['L'], geo['C']
    g = torch.Generator().manual_seed(1)
    out, dy = hg.encode_forward(x, emb, geo, True)
    grad = torch.randn(L, B, C, generator=g, dtype=torch.float64)
    # <grad, enc(emb)> is linear in emb: its gradient is the scatter
    ge = hg.encode_backward_grid(grad, x, geo, emb.shape[0])
    assert torch.allclose((ge * emb).sum(), (grad * out).sum(), rtol=1e-10)
    gi = hg.encode_backward_input(grad, dy, geo)
    ggi = torch.randn(B, 3, generator=g, dtype=torch.float64)
    # s = <ggi, gi> ;  ds/dgrad = second_backward_grad ; ds/demb = second_backward_embedding
    gg = hg.second_backward_grad(ggi, dy, geo)
    assert torch.allclose((gg * grad).sum(), (ggi * gi).sum(), rtol=1e-10)
    g2 = hg.second_backward_embedding(grad, x, ggi, geo, emb.shape[0])
    assert torch.allclose((g2 * emb).sum(), (ggi * gi).sum(), rtol=1e-9)


def test_out_of_range_inputs_give_zeros():
    geo, emb, x = _small()
    x = x.clone()
    x[0, 1] = 1.5
    x[1, 0] = -0.2
    out, dy = hg.encode_forward(x, emb, geo, True)
    assert out[:, :2].abs().max() == 0 and dy[:2].abs().max() == 0
    grad = torch.ones(geo['L'], x.shape[0], geo['C'], dtype=torch.float64)
    only = torch.zeros_like(grad)
    only[:, :2] = 1
    assert hg.encode_backward_grid(only, x, geo, emb.shape[0]).abs().max() == 0


def test_autograd_wiring_drops_the_same_terms_as_the_reference():
    geo, emb, x = _small()
    emb = emb.clone().requires_grad_(True)
    xx = (x * 2 - 1).clone().requires_grad_(True)
    enc = hg.hash_encode_autograd(xx, emb, geo)
    w = torch.randn(enc.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    gx, = torch.autograd.grad((enc * w).sum(), xx, create_graph=True)
    # second order: d/d emb exists, d/d x is dropped (hashgrid.py:101 returns None for inputs)
    g_emb, = torch.autograd.grad(gx.pow(2).sum(), emb, retain_graph=True)
    assert g_emb.abs().max() > 0
    g_x = torch.autograd.grad(gx.pow(2).sum(), xx, allow_unused=True)[0]
    assert g_x is None or g_x.abs().max() == 0
