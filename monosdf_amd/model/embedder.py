"""Positional encoding (reference: code/model/embedder.py).  On the hot path the encoding is
computed inside the fused kernels (csrc/mlp_core.h pe_slot); this host version serves callers
that ask for the embedding itself."""
import torch


class Embedder:
    def __init__(self, input_dims, num_freqs, include_input=True):
        self.input_dims, self.num_freqs, self.include_input = input_dims, num_freqs, include_input
        self.out_dim = input_dims * (2 * num_freqs + (1 if include_input else 0))

    def embed(self, x):
        parts = [x] if self.include_input else []
        for k in range(self.num_freqs):
            f = float(2.0 ** k)
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    eo = Embedder(input_dims, multires)
    return eo.embed, eo.out_dim
