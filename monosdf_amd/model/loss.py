"""MonoSDFLoss with the reference's interface (reference: code/model/loss.py:180-311), computed by one
HIP launch (csrc/loss.hip: msdf_monosdf_loss) that also returns the gradients of `loss`.

Point `train.loss_class` at `monosdf_amd.model.loss.MonoSDFLoss` (reference: training/monosdf_train.py:206).
Covered: the pixel-batch mode the runner trains in, `rgb_loss = torch.nn.L1Loss`.  The reference's image mode
with depth_alpha > 0 ends in `assert False, 'Rui: disabled'` (loss.py:167-168) and raises the same here.
Unlike the reference, nothing is printed per step (loss.py:164).
"""
import math

import torch
from torch import nn

from .. import ops


class MonoSDFLoss(nn.Module):
    def __init__(self, rgb_loss, eikonal_weight, smooth_weight=0.005, depth_weight=0.1, depth_alpha=0.5,
                 normal_l1_weight=0.05, normal_cos_weight=0.05, if_gamma_loss=False, if_scale_invariant_depth=True,
                 end_step=-1):
        super().__init__()
        if rgb_loss not in ('torch.nn.L1Loss',):
            raise NotImplementedError('monosdf_amd: rgb_loss %r (the confs of the reference use torch.nn.L1Loss)'
                                      % (rgb_loss,))
        self.eikonal_weight = eikonal_weight
        self.smooth_weight = smooth_weight
        self.depth_weight = depth_weight
        self.depth_alpha = depth_alpha
        self.normal_l1_weight = normal_l1_weight
        self.normal_cos_weight = normal_cos_weight
        self.if_scale_invariant_depth = if_scale_invariant_depth
        self.if_gamma_loss = if_gamma_loss
        self.step = 0
        self.end_step = end_step

    def forward(self, model_outputs, ground_truth, if_pixel_input=False):
        if self.depth_alpha > 0 and not if_pixel_input:
            raise AssertionError('Rui: disabled')        # loss.py:167-168, same condition
        decay = math.exp(-self.step / self.end_step * 10.) if self.end_step > 0 else 1.0
        self.step += 1
        weights = (self.eikonal_weight, self.smooth_weight, decay * self.depth_weight, decay * self.normal_l1_weight,
                   decay * self.normal_cos_weight)
        g1 = model_outputs.get('grad_theta')
        g2 = model_outputs.get('grad_theta_nei') if g1 is not None else None
        if g1 is None:
            # loss.py:226-234 reads grad_theta unconditionally and fails without it
            raise KeyError('grad_theta')
        res = ops.MonoSdfLossFunction.apply(
            model_outputs['rgb_values'], model_outputs['depth_values'], model_outputs['normal_map'], g1, g2,
            model_outputs['sdf'], ground_truth['rgb'], ground_truth['depth'], ground_truth['normal'],
            ground_truth['mask'], weights, self.if_gamma_loss, self.if_scale_invariant_depth)
        det = res.detach()
        return {'loss': res[0], 'rgb_loss': det[1], 'eikonal_loss': det[2], 'smooth_loss': det[3],
                'depth_loss': det[4], 'normal_l1': det[5], 'normal_cos': det[6]}
