"""Multi-resolution hash-grid encoder module (reference: code/hashencoder/hashgrid.py:107-166).

Same constructor, parameter / buffer names (``embeddings``, ``offsets``) and initialisation as the
reference's HashEncoder; the kernels behind it are csrc/hashgrid.hip through the C ABI
(msdf_hash_encode_*), wired with the same two autograd Functions (first and second backward,
including the terms the reference drops).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.max_params = 2 ** log2_hashmap_size
        offsets, offset = [], 0
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            offsets.append(offset)
            offset += min(self.max_params, resolution ** input_dim)
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return ('HashEncoder: input_dim=%d num_levels=%d level_dim=%d base_resolution=%d per_level_scale=%s '
                'params=%s' % (self.input_dim, self.num_levels, self.level_dim, self.base_resolution,
                               self.per_level_scale, tuple(self.embeddings.shape)))

    @property
    def log2_scale(self):
        return float(np.log2(self.per_level_scale))

    def forward(self, inputs, size=1, calc_grad_inputs=None):
        """inputs in [-size, size] -> [..., num_levels * level_dim]."""
        inputs = (inputs + size) / (2 * size)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        if calc_grad_inputs is None:
            calc_grad_inputs = inputs.requires_grad
        out = ops.HashEncodeFunction.apply(inputs, self.embeddings, self.offsets, self.log2_scale,
                                           self.base_resolution, calc_grad_inputs)
        return out.view(prefix + [self.output_dim])

    def encode_with_jacobian(self, inputs01):
        """Features [B, L*C] plus the handle needed to apply d enc / d x later (fused-MLP path)."""
        inputs01 = inputs01.detach().contiguous()
        B, D = inputs01.shape
        L, C = self.num_levels, self.level_dim
        feats, dy_dx = ops.HashEncodeWithJacobian.apply(inputs01, self.embeddings, self.offsets, self.log2_scale,
                                                        self.base_resolution)
        dims = (B, D, C, L, self.log2_scale, int(self.base_resolution))
        return feats, (inputs01, dy_dx, dims)

    def input_gradient(self, handle, d_out):
        """sum_{l,c} d_out[b, l*C+c] * d enc[b,l,c] / d x01 (differentiable: second-order terms flow)."""
        inputs01, dy_dx, dims = handle
        B, D, C, L, S, H = dims
        g = d_out.reshape(B, L, C).permute(1, 0, 2).contiguous()
        g_in, _ = ops.HashEncodeBackwardFunction.apply(g, inputs01, self.embeddings, self.offsets, dy_dx, dims, True,
                                                       False)
        return g_in
