"""monosdf_amd -- MI355X (gfx950) implementation of MonoSDF's SDF volume-rendering hot path.

Drop-in for the reference's ``model.network.MonoSDFNetwork`` (same constructor,
``forward`` outputs, sub-module API and state-dict keys); the arithmetic runs in
hand-written HIP kernels behind the C ABI of include/monosdf_hip.h.
"""
__all__ = ['model', 'ops', 'plan', 'conf']
