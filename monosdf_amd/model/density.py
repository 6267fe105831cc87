"""SDF -> density (reference: code/model/density.py:5-30).  ``LaplaceDensity.forward`` is used by
callers outside the fused path (its own HIP kernel, forward and backward); inside
MonoSDFNetwork.forward the density is evaluated by the compositor / sampler kernels from ``get_beta()``."""
import torch
import torch.nn as nn

from .. import ops


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for p in params_init:
            setattr(self, p, nn.Parameter(torch.tensor(params_init[p])))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.register_buffer('beta_min', torch.tensor(beta_min), persistent=False)
        self.beta_min_f = float(beta_min)          # the same constant for kernel arguments (no device read)

    def density_func(self, sdf, beta=None):
        # Device tensors only, first order only (ops.LaplaceDensityFunction raises on a CPU tensor): this package has
        # no CPU or eager-PyTorch path on purpose -- a silent fallback would make "the HIP kernels ran" unverifiable.
        # The reference's expression (density.py:21-27) also runs on the CPU and to any order of differentiation.
        if beta is None:
            beta = self.get_beta()
        if not torch.is_tensor(beta):
            beta = torch.tensor(float(beta), device=sdf.device)
        return ops.LaplaceDensityFunction.apply(sdf, beta)

    def get_beta(self):
        return self.beta.abs() + self.beta_min
