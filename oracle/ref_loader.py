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
