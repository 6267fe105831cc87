"""CPU restatement of the reference's training loss (code/model/loss.py).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/loss_*.npz, which
oracle/make_golden_loss.py records from the REAL reference class (values and gradients).
Only the pixel-batch mode the runner uses is restated: with depth_alpha > 0 the reference's
image mode hits `assert False, 'Rui: disabled'` (loss.py:167-168).
"""
import math

import torch


def gamma2(x):
    """loss.py:209-215."""
    lo = x <= 0.0031308
    return torch.where(lo, 12.92 * x, 1.055 * x.clamp_min(1e-30).pow(1 / 2.4) - 0.055)


def scale_and_shift_1d(pred, target, mask):
    """compute_scale_and_shift_1D (loss.py:29-49): per-row least squares of  s*pred + h ~ target."""
    a00 = torch.sum(mask * pred * pred, 1)
    a01 = torch.sum(mask * pred, 1)
    a11 = torch.sum(mask, 1)
    b0 = torch.sum(mask * pred * target, 1)
    b1 = torch.sum(mask * target, 1)
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    x0 = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(det))
    x1 = torch.where(ok, (-a01 * b0 + a00 * b1) / safe, torch.zeros_like(det))
    return x0, x1


def depth_loss(depth_pred, depth_gt, mask, scale_invariant=True):
    """get_depth_loss -> ScaleAndShiftInvariantLoss -> mse_loss(pixel) -> reduction_batch_based
    (loss.py:236-243, 156-171, 75-87, 52-61)."""
    pred = depth_pred.reshape(1, -1)
    m = mask.reshape(1, -1).to(pred.dtype)
    if scale_invariant:
        target = (depth_gt * 50 + 0.5).reshape(1, -1)
        s, h = scale_and_shift_1d(pred, target, m)
        pred = s.view(1, -1) * pred + h.view(1, -1)
    else:
        target = depth_gt.reshape(1, -1)
    M = torch.sum(m, 1)
    res = pred - target
    image_loss = torch.sum(m * res * res, 1)
    divisor = torch.sum(2 * M)
    if divisor == 0:
        return torch.zeros((), dtype=pred.dtype)
    return torch.sum(image_loss) / divisor


def monosdf_loss(out, gt, weights, step=0, end_step=-1, if_gamma_loss=False, scale_invariant=True):
    """MonoSDFLoss.forward (loss.py:252-311) for if_pixel_input=True and rgb_loss = torch.nn.L1Loss.

    weights: dict eikonal / smooth / depth / normal_l1 / normal_cos (ctor arguments, loss.py:181-190).
    """
    rgb, rgb_gt = out['rgb_values'], gt['rgb'].reshape(-1, 3)
    if if_gamma_loss:
        rgb, rgb_gt = gamma2(rgb), gamma2(rgb_gt)
    rgb_loss = (rgb - rgb_gt).abs().mean()
    if 'grad_theta' in out:
        eik = ((out['grad_theta'].norm(2, dim=1) - 1) ** 2).mean()
    else:
        eik = torch.zeros(())
    sdf = out['sdf']
    fg = ((sdf > 0.).any(dim=-1) & (sdf < 0.).any(dim=-1))[None, :, None]
    mask = (gt['mask'] > 0.5) & fg
    d = depth_loss(out['depth_values'], gt['depth'], mask, scale_invariant)
    n_gt = torch.nn.functional.normalize(gt['normal'], p=2, dim=-1)
    n_pr = torch.nn.functional.normalize(out['normal_map'][None] * mask, p=2, dim=-1)
    n_l1 = torch.abs(n_pr - n_gt).sum(dim=-1).mean()
    n_cos = (1. - torch.sum(n_pr * n_gt, dim=-1)).mean()
    g1, g2 = out['grad_theta'], out['grad_theta_nei']
    u1 = g1 / (g1.norm(2, dim=1).unsqueeze(-1) + 1e-5)
    u2 = g2 / (g2.norm(2, dim=1).unsqueeze(-1) + 1e-5)
    smooth = torch.norm(u1 - u2, dim=-1).mean()
    decay = math.exp(-step / end_step * 10.) if end_step > 0 else 1.0
    loss = rgb_loss + weights['eikonal'] * eik + weights['smooth'] * smooth + decay * weights['depth'] * d + \
        decay * weights['normal_l1'] * n_l1 + decay * weights['normal_cos'] * n_cos
    return {'loss': loss, 'rgb_loss': rgb_loss, 'eikonal_loss': eik, 'smooth_loss': smooth, 'depth_loss': d,
            'normal_l1': n_l1, 'normal_cos': n_cos}
