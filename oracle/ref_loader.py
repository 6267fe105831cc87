"""Container-only loader for the real reference (TEST INFRASTRUCTURE).

Imports ``/root/reference/code/model/*`` on CPU by registering import-time stubs
for packages the image lacks (SURVEY.md 8(c) recipe) and neutralising ``.cuda()``.
Used ONLY by ``oracle/make_golden.py`` to generate ``tests/golden/*.npz``; the
reference never travels to the GPU box, and nothing here is imported by tests,
bench.py or the product at run time.

The reference's hash-grid backend is CUDA-only; ``FakeHashBackend`` plugs the
restated kernels of ``hashgrid_oracle`` into the reference's own autograd
wiring (hashencoder/hashgrid.py) so ``ImplicitNetworkGrid`` can run on CPU.
"""
import contextlib
import os
import sys
import types

import torch

REF_ROOT = '/root/reference/code'


def available():
    return os.path.isdir(REF_ROOT)


class FakeHashBackend:
    """Drop-in for the pybind module `_hash_encoder` (hashencoder/src/bindings.cpp:5-8)."""

    @staticmethod
    def _geo(offsets, D, C, L, S, H):
        return dict(D=D, L=L, C=C, H=H, S=float(S), offsets=[int(v) for v in offsets.tolist()])

    @classmethod
    def hash_encode_forward(cls, inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
                            calc_grad_inputs, dy_dx):
        from . import hashgrid_oracle as hg
        out, dy = hg.encode_forward(inputs, embeddings, cls._geo(offsets, D, C, L, S, H),
                                    bool(calc_grad_inputs))
        outputs.copy_(out)
        if calc_grad_inputs:
            dy_dx.copy_(dy)

    @classmethod
    def hash_encode_backward(cls, grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L,
                             S, H, calc_grad_inputs, dy_dx, grad_inputs):
        from . import hashgrid_oracle as hg
        geo = cls._geo(offsets, D, C, L, S, H)
        grad_embeddings.add_(hg.encode_backward_grid(grad, inputs, geo, embeddings.shape[0]))
        if calc_grad_inputs:
            grad_inputs.copy_(hg.encode_backward_input(grad, dy_dx, geo))

    @classmethod
    def hash_encode_second_backward(cls, grad, inputs, embeddings, offsets, B, D, C, L, S, H,
                                    calc_grad_inputs, dy_dx, grad_grad_inputs, grad_grad,
                                    grad2_embeddings):
        from . import hashgrid_oracle as hg
        geo = cls._geo(offsets, D, C, L, S, H)
        grad_grad.copy_(hg.second_backward_grad(grad_grad_inputs, dy_dx, geo))
        grad2_embeddings.add_(hg.second_backward_embedding(grad, inputs, grad_grad_inputs, geo,
                                                           embeddings.shape[0]))


_loaded = None


def load():
    """Returns the reference's ``model.network`` module (imported once)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference not present (expected on the GPU box)')

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    for name in ('imageio', 'skimage', 'cv2', 'matplotlib', 'matplotlib.pyplot'):
        if name not in sys.modules:
            stub(name)
    stub('tkinter')
    stub('tkinter.messagebox', NO='no')
    stub('cachetools', cached=lambda *a, **k: (lambda f: f))
    stub('hashencoder.backend', _backend=FakeHashBackend)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF_ROOT)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import model.network as net        # noqa: E402  (reference module)
    _loaded = net
    return net


def load_plots():
    """The reference's ``utils.plots`` module with the meshing / imaging packages stubbed; returns (module,
    captured) where ``captured`` collects every volume handed to ``measure.marching_cubes`` (utils/plots.py:199-205)
    -- that array is what the SDF-volume evaluator has to reproduce."""
    load()
    import numpy as np
    captured = []

    def marching_cubes(volume=None, level=0, spacing=(1., 1., 1.), **kw):
        captured.append({'volume': np.array(volume, copy=True), 'spacing': tuple(float(v) for v in spacing)})
        tri = np.eye(3)               # one dummy triangle: the caller prints verts.min() / max()
        return tri, np.array([[0, 1, 2]]), tri, np.zeros((3,))

    class _Mesh:
        def __init__(self, *a, **k):
            pass

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    measure = stub('skimage.measure', marching_cubes=marching_cubes)
    sys.modules['skimage'].measure = measure
    tv = stub('torchvision', utils=types.SimpleNamespace(make_grid=None))
    tv.transforms = stub('torchvision.transforms', ToPILImage=lambda *a, **k: None)
    stub('trimesh', Trimesh=_Mesh, util=types.SimpleNamespace(concatenate=lambda meshes: meshes))
    stub('termcolor', colored=lambda s, *a, **k: s)
    import utils.plots as plots        # noqa: E402  (reference module)
    return plots, captured


def build_model(conf, state=None, training=False, if_hdr=False):
    """Instantiate the reference MonoSDFNetwork on CPU and load ``state`` into it."""
    net = load()
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = net.MonoSDFNetwork(conf=conf, if_hdr=if_hdr)
    if state is not None:
        model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model.train(training)
    return model


@contextlib.contextmanager
def record_rng():
    """Records the random draws the reference makes during one forward, in call order."""
    log = []
    saved = (torch.rand, torch.randperm, torch.randint, torch.rand_like, torch.Tensor.uniform_)

    def wrap(tag, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            log.append((tag, out.detach().clone()))
            return out
        return inner

    torch.rand = wrap('rand', saved[0])
    torch.randperm = wrap('randperm', saved[1])
    torch.randint = wrap('randint', saved[2])
    torch.rand_like = wrap('rand_like', saved[3])
    torch.Tensor.uniform_ = wrap('uniform_', saved[4])
    try:
        yield log
    finally:
        torch.rand, torch.randperm, torch.randint, torch.rand_like, torch.Tensor.uniform_ = saved


@contextlib.contextmanager
def inject_rng(noise, n_rays, conf):
    """The opposite of record_rng: the six random draws of one training forward of the reference come from
    ``noise`` (oracle/synth.make_noise_table) instead of torch's generators, matched by call site shape
    (SURVEY 8(a) RNG note: ray_sampler.py:79,213,244,254; network.py:587,593)."""
    sc = conf['ray_sampler']
    saved = (torch.rand, torch.randperm, torch.randint, torch.rand_like, torch.Tensor.uniform_)

    def rand(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        if shape == (n_rays, sc['N_samples_eval']):
            return noise['jitter'].clone()
        if shape == (n_rays, sc['N_samples']):
            return noise['final_u'].clone()
        raise AssertionError('unexpected torch.rand%r' % (shape,))

    def randperm(m, **kw):
        row = noise['extra_idx'][m // sc['N_samples_eval'] - 1]
        rest = torch.tensor([i for i in range(m) if i not in set(row.tolist())], dtype=torch.int64)
        return torch.cat([row, rest])

    def randint(high, size, **kw):
        assert tuple(size) == (n_rays,) and high == sc['N_samples'] + sc['N_samples_extra'] + 2
        return noise['eik_idx'].clone()

    def rand_like(t, **kw):
        assert tuple(t.shape) == (2 * n_rays, 3)
        return noise['nei_rand'].clone()

    def uniform_(self, a=0.0, b=1.0, **kw):
        assert tuple(self.shape) == (n_rays, 3)
        return self.copy_(noise['eik_uniform'])

    torch.rand, torch.randperm, torch.randint, torch.rand_like, torch.Tensor.uniform_ = \
        rand, randperm, randint, rand_like, uniform_
    try:
        yield
    finally:
        torch.rand, torch.randperm, torch.randint, torch.rand_like, torch.Tensor.uniform_ = saved


@contextlib.contextmanager
def record_sampler(model):
    """Records what the reference's ErrorBoundSampler.get_z_vals computes in each round, without touching its
    source: the values are picked up at the functions it calls (reference: model/ray_sampler.py) --
    get_error_bound (first call of a round: sorted z, merged sdf, d*, and the error at beta0; 141-157),
    model.density called from get_z_vals itself (the beta the bisection ended with; 168), torch.searchsorted
    (cdf and u; 216) and torch.sort (its input holds the new samples behind the old ones; 233, 251).
    Yields a list with one dict per round."""
    rounds = []
    state = {'in_eb': False, 'active': False}
    sampler, density = model.ray_sampler, model.density
    orig_eb, orig_dens, orig_gz = sampler.get_error_bound, density.forward, sampler.get_z_vals
    orig_ss, orig_sort = torch.searchsorted, torch.sort

    def eb(beta, mdl, sdf, z_vals, dists, d_star):
        state['in_eb'] = True
        try:
            out = orig_eb(beta, mdl, sdf, z_vals, dists, d_star)
        finally:
            state['in_eb'] = False
        if not rounds or 'beta' in rounds[-1]:
            rounds.append({'z': z_vals.detach().clone(), 'sdf': sdf.detach().reshape(z_vals.shape).clone(),
                           'dstar': d_star.detach().clone(), 'err0': out.detach().clone()})
        return out

    def dens(sdf, beta=None):
        out = orig_dens(sdf, beta=beta)
        if state['active'] and not state['in_eb'] and beta is not None:
            rounds[-1]['beta'] = beta.detach().reshape(-1).clone()
        return out

    def searchsorted(cdf, u, **kw):
        if state['active']:
            rounds[-1]['cdf'] = cdf.detach().clone()
            rounds[-1]['u'] = u.detach().clone()
        return orig_ss(cdf, u, **kw)

    def sort(t, *a, **kw):
        if state['active'] and 'cdf' in rounds[-1] and 'samples' not in rounds[-1]:
            n_new = rounds[-1]['u'].shape[1]
            m = rounds[-1]['z'].shape[1]
            # continuing round: cat([z_vals, samples]); last round: cat([z_samples, near, far, extra])
            rounds[-1]['samples'] = (t[:, m:m + n_new] if t.shape[1] == m + n_new else t[:, :n_new]).detach().clone()
        return orig_sort(t, *a, **kw)

    def get_z_vals(*a, **kw):
        state['active'] = True
        try:
            return orig_gz(*a, **kw)
        finally:
            state['active'] = False

    sampler.get_error_bound, density.forward, sampler.get_z_vals = eb, dens, get_z_vals
    torch.searchsorted, torch.sort = searchsorted, sort
    try:
        yield rounds
    finally:
        del sampler.get_error_bound, density.forward, sampler.get_z_vals
        torch.searchsorted, torch.sort = orig_ss, orig_sort


def noise_from_log(log, n_extra=32):
    """Map the recorded draws onto the oracle's ``noise`` keys (SURVEY.md 8(a) RNG order)."""
    rands = [t for tag, t in log if tag == 'rand']
    noise = {}
    if rands:
        noise['jitter'] = rands[0]
    if len(rands) > 1:
        noise['final_u'] = rands[1]
    for tag, t in log:
        if tag == 'randperm':
            noise['extra_idx'] = t[:n_extra]
        elif tag == 'randint':
            noise['eik_idx'] = t
        elif tag == 'uniform_':
            noise['eik_uniform'] = t
        elif tag == 'rand_like':
            noise['nei_rand'] = t
    return noise
