"""CPU restatement of the reference's multi-resolution hash-grid encoder.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference implements this only as CUDA (code/hashencoder/src/hashencoder.cu)
which cannot be built or run here (no nvcc, no GPU) and it ships no tests or
golden vectors for it, so THE ARITHMETIC BELOW IS "PARITY UNPINNED" by the
reference: it is pinned by (i) the CUDA source read as text, (ii) the hand-computed
index known-answer tests in tests/test_hashgrid_oracle.py (computed from cu:35-72),
(iii) self-consistency (finite differences of the forward against dy_dx, linearity
of the scatter, adjointness of forward/backward).  The Python-side wiring
(offsets, init range, [-1,1]->[0,1] map, which double-backward terms are dropped)
follows code/hashencoder/hashgrid.py and IS exercised through the real reference
classes in make_golden.py by installing this module as the fake ``_backend``.

All tensors are torch CPU tensors; integer work is done in int64 and masked to
uint32 where the CUDA code relies on 32-bit wrap-around.
"""
import math

import numpy as np
import torch

_U32 = 0xFFFFFFFF
_PRIMES = (1, 2654435761, 805459861)          # hashencoder.cu:42


def level_geometry(ic):
    """Offsets / scale parameters of HashEncoder.__init__ (hashgrid.py:107-136).

    ``ic`` is the implicit_network conf dict.  Returns a dict with ints and the
    per-level scale exponent ``S`` exactly as the Python side hands it to the
    kernel (a Python float that the C++ signature narrows to float32).
    """
    D = 3
    L = int(ic.get('num_levels', 16))
    C = int(ic.get('level_dim', 2))
    H = int(ic.get('base_size', 16))
    desired = ic.get('end_size', 2048)
    log2_hash = int(ic.get('logmap', 19))
    per_level_scale = np.exp2(np.log2(desired / H) / (L - 1))
    max_params = 2 ** log2_hash
    offsets, off = [], 0
    for i in range(L):
        res = int(np.ceil(H * per_level_scale ** i))
        offsets.append(off)
        off += min(max_params, res ** D)
    offsets.append(off)
    return dict(D=D, L=L, C=C, H=H, S=float(np.log2(per_level_scale)),
                per_level_scale=float(per_level_scale), offsets=offsets, n_entries=off)


def level_scale(geo, level):
    """scale = exp2f(level*S)*H - 1 and resolution = ceil(scale)+1 in float32 (cu:152-153)."""
    # exp2f evaluated as the correctly rounded float (double pow, then one rounding): numpy's float32
    # exp2 and device exp2f are both only accurate to ~1 ulp and disagree with each other on some levels,
    # and one ulp of scale moves a fine-level sample by ~1e-4 of a cell.
    s32 = np.float32(geo['S'])
    arg = np.float32(np.float32(level) * s32)
    e = np.float32(2.0 ** float(arg))
    scale = np.float32(np.float32(e * np.float32(geo['H'])) - np.float32(1.0))
    return float(scale), int(math.ceil(float(scale))) + 1


def grid_index(pos, hashmap_size, resolution):
    """get_grid_index<D,C>(ch=0, ...) / C  -> entry index (cu:54-72), pos: int64 [B, D]."""
    D = pos.shape[1]
    stride = 1
    index = torch.zeros(pos.shape[0], dtype=torch.int64)
    d = 0
    while d < D and stride <= hashmap_size:
        index = (index + pos[:, d] * stride) & _U32
        stride = (stride * resolution) & _U32
        d += 1
    if stride > hashmap_size:
        index = torch.zeros(pos.shape[0], dtype=torch.int64)
        for k in range(D):
            index = index ^ ((pos[:, k] * _PRIMES[k]) & _U32)
    return index % hashmap_size


def _locate(x01, geo, level):
    """Cell, smoothstep weights and their derivative for one level (cu:156-167)."""
    scale, res = level_scale(geo, level)
    pos = x01 * torch.tensor(scale, dtype=x01.dtype)
    cell = torch.floor(pos)
    frac = pos - cell
    dsm = 6 * frac * (1.0 - frac)
    sm = frac * frac * (3.0 - 2.0 * frac)
    return scale, res, cell.to(torch.int64), sm, dsm


def _in_range(x01):
    return ~((x01 < 0) | (x01 > 1)).any(dim=1)


def encode_forward(x01, emb, geo, want_dy_dx):
    """kernel_grid (cu:103-254): returns outputs [L,B,C] and dy_dx [B, L*D*C] (or None)."""
    B, D = x01.shape
    L, C, off = geo['L'], geo['C'], geo['offsets']
    out = torch.zeros(L, B, C, dtype=x01.dtype)
    dy = torch.zeros(B, L, D, C, dtype=x01.dtype) if want_dy_dx else None
    ok = _in_range(x01).unsqueeze(1).to(x01.dtype)
    for l in range(L):
        hsize = off[l + 1] - off[l]
        table = emb[off[l]:off[l + 1]]
        scale, res, cell, sm, dsm = _locate(x01, geo, l)
        acc = torch.zeros(B, C, dtype=x01.dtype)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=x01.dtype)
            p = cell.clone()
            for d in range(D):
                if corner & (1 << d):
                    w = w * sm[:, d]
                    p[:, d] += 1
                else:
                    w = w * (1 - sm[:, d])
            acc = acc + w.unsqueeze(1) * table[grid_index(p, hsize, res)]
        out[l] = acc * ok
        if want_dy_dx:
            for gd in range(D):
                g = torch.zeros(B, C, dtype=x01.dtype)
                others = [d for d in range(D) if d != gd]
                for combo in range(1 << (D - 1)):
                    w = torch.full((B,), scale, dtype=x01.dtype)
                    p = cell.clone()
                    for nd, d in enumerate(others):
                        if combo & (1 << nd):
                            w = w * sm[:, d]
                            p[:, d] += 1
                        else:
                            w = w * (1 - sm[:, d])
                    left = table[grid_index(p, hsize, res)]
                    p[:, gd] += 1
                    right = table[grid_index(p, hsize, res)]
                    g = g + w.unsqueeze(1) * (right - left) * dsm[:, gd:gd + 1]
                dy[:, l, gd, :] = g * ok
    return out, (dy.reshape(B, L * D * C) if want_dy_dx else None)


def encode_backward_grid(grad, x01, geo, n_entries):
    """kernel_grid_backward (cu:257-343): scatter w*grad into the 8 corners. grad [L,B,C]."""
    B, D = x01.shape
    L, C, off = geo['L'], geo['C'], geo['offsets']
    g_emb = torch.zeros(n_entries, C, dtype=grad.dtype)
    ok = _in_range(x01).unsqueeze(1).to(grad.dtype)
    for l in range(L):
        hsize = off[l + 1] - off[l]
        _, res, cell, sm, _ = _locate(x01, geo, l)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=grad.dtype)
            p = cell.clone()
            for d in range(D):
                if corner & (1 << d):
                    w = w * sm[:, d]
                    p[:, d] += 1
                else:
                    w = w * (1 - sm[:, d])
            idx = grid_index(p, hsize, res) + off[l]
            g_emb.index_add_(0, idx, w.unsqueeze(1) * grad[l] * ok)
    return g_emb


def encode_backward_input(grad, dy_dx, geo):
    """kernel_input_backward (cu:346-372): grad_x[b,d] = sum_{l,c} grad[l,b,c] dy_dx[b,l,d,c]."""
    L, C, D = geo['L'], geo['C'], geo['D']
    B = grad.shape[1]
    return torch.einsum('lbc,bldc->bd', grad, dy_dx.reshape(B, L, D, C))


def second_backward_grad(gg_x, dy_dx, geo):
    """kernel_grid_second_backward_grad (cu:375-428): [L,B,C] = sum_d gg_x[b,d] dy_dx[b,l,d,c]."""
    L, C, D = geo['L'], geo['C'], geo['D']
    B = gg_x.shape[0]
    return torch.einsum('bd,bldc->lbc', gg_x, dy_dx.reshape(B, L, D, C))


def second_backward_embedding(grad, x01, gg_x, geo, n_entries):
    """kernel_grid_second_backward_embedding (cu:431-595)."""
    B, D = x01.shape
    L, C, off = geo['L'], geo['C'], geo['offsets']
    g2 = torch.zeros(n_entries, C, dtype=grad.dtype)
    ok = _in_range(x01).unsqueeze(1).to(grad.dtype)
    for l in range(L):
        hsize = off[l + 1] - off[l]
        scale, res, cell, sm, dsm = _locate(x01, geo, l)
        cache = [torch.zeros(B, C, dtype=grad.dtype) for _ in range(1 << D)]
        for gd in range(D):
            others = [d for d in range(D) if d != gd]
            for combo in range(1 << (D - 1)):
                w = torch.full((B,), scale, dtype=grad.dtype)
                local = 0
                for nd, d in enumerate(others):
                    if combo & (1 << nd):
                        w = w * sm[:, d]
                        local |= (1 << d)
                    else:
                        w = w * (1 - sm[:, d])
                term = (w * gg_x[:, gd] * dsm[:, gd]).unsqueeze(1) * grad[l]
                cache[local | (1 << gd)] = cache[local | (1 << gd)] + term
                cache[local] = cache[local] - term
        for corner in range(1 << D):
            p = cell.clone()
            for d in range(D):
                if corner & (1 << d):
                    p[:, d] += 1
            idx = grid_index(p, hsize, res) + off[l]
            g2.index_add_(0, idx, cache[corner] * ok)
    return g2


# ----------------------------------------------------------------------------
# autograd wiring (hashgrid.py:14-101) -- which terms exist and which are dropped
# ----------------------------------------------------------------------------
class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x01, emb, geo, want_dx):
        out, dy_dx = encode_forward(x01.detach(), emb.detach(), geo, want_dx)
        ctx.geo, ctx.want_dx = geo, want_dx
        ctx.save_for_backward(x01, emb, dy_dx if want_dx else torch.zeros(1))
        B = x01.shape[0]
        return out.permute(1, 0, 2).reshape(B, geo['L'] * geo['C'])

    @staticmethod
    def backward(ctx, grad):
        x01, emb, dy_dx = ctx.saved_tensors
        geo = ctx.geo
        B = x01.shape[0]
        g = grad.view(B, geo['L'], geo['C']).permute(1, 0, 2).contiguous()
        gx, gemb = _EncodeBackward.apply(g, x01, emb, dy_dx, geo, ctx.want_dx)
        return (gx if ctx.want_dx else None), gemb, None, None


class _EncodeBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad, x01, emb, dy_dx, geo, want_dx):
        ctx.geo = geo
        ctx.save_for_backward(grad, x01, emb, dy_dx)
        gemb = encode_backward_grid(grad.detach(), x01.detach(), geo, emb.shape[0])
        gx = encode_backward_input(grad.detach(), dy_dx, geo) if want_dx else torch.zeros_like(x01)
        return gx, gemb

    @staticmethod
    def backward(ctx, gg_x, gg_emb):
        # gg_emb is ignored and no term flows to x (hashgrid.py:87,101): reproduced on purpose
        grad, x01, emb, dy_dx = ctx.saved_tensors
        geo = ctx.geo
        gg = second_backward_grad(gg_x, dy_dx, geo)
        g2 = second_backward_embedding(grad, x01, gg_x, geo, emb.shape[0])
        return gg, None, g2, None, None, None


def hash_encode_autograd(x, emb, geo):
    """HashEncoder.forward (hashgrid.py:154-166): x in [-1,1] -> [B, L*C]."""
    x01 = (x + 1) / 2
    return _Encode.apply(x01, emb, geo, x.requires_grad)
