"""Which parts of one training step make PyTorch launch kernels of its own (tuning aid): the step is cut into named
ranges (torch.profiler.record_function around the model's stages, the loss, backward, the optimiser) and every ATen op
with device time of its own is listed under the innermost range that contains it.

    python scripts/aten_sources.py [grid]
"""
import collections
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, record_function, ProfilerActivity  # noqa: E402
from monosdf_amd import ops  # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork  # noqa: E402

torch.manual_seed(0)
grid = len(sys.argv) > 1 and sys.argv[1] == 'grid'
model = MonoSDFNetwork(bench.model_conf(grid=grid)).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
rays = bench.make_rays(1024, 1, 'cuda')
idx = torch.arange(1024, device='cuda')


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        with record_function(label):
            return fn(*a, **k)
    setattr(obj, name, inner)


wrap(model.ray_sampler, 'sample', 'R:sampler')
wrap(model.implicit_network, 'evaluate', 'R:sdf_evaluate')
wrap(model.implicit_network, 'share', 'R:weights(share/pack)')
wrap(model.rendering_network, 'share', 'R:weights(share/pack)')
wrap(model.rendering_network, 'forward', 'R:colour')
wrap(model.density, 'get_beta', 'R:get_beta')
for cls in ('CompositeFunction', 'SdfMlpFunction', 'ColorMlpFunction', 'GridSdfFunction', 'SplitRowsFunction'):
    c = getattr(ops, cls, None)
    if c is not None:
        for m in ('backward',):
            f = getattr(c, m)

            def mk(f, lab):
                def inner(*a, **k):
                    with record_function(lab):
                        return f(*a, **k)
                return staticmethod(inner)
            setattr(c, m, mk(f, 'R:bwd ' + cls))


def step():
    with record_function('R:zero_grad'):
        opt.zero_grad(set_to_none=True)
    with record_function('R:forward(other)'):
        out = model(rays, idx, if_pixel_input=True)
    with record_function('R:loss'):
        loss = ops.probe_loss(out)
    with record_function('R:backward(other)'):
        loss.backward()
    with record_function('R:optimizer'):
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = list(prof.events())
ranges = [e for e in evs if e.key.startswith('R:')]
seen = collections.Counter()
for e in evs:
    if not e.key.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    inside = [r for r in ranges if r.time_range.start <= e.time_range.start and e.time_range.end <= r.time_range.end
              and r.thread == e.thread]
    lab = min(inside, key=lambda r: r.time_range.end - r.time_range.start).key if inside else 'R:?(autograd thread)'
    seen[(lab, e.key, str(e.input_shapes)[:60])] += 1
for (lab, k, shp), n in sorted(seen.items()):
    print('%-26s %-20s x%d %s' % (lab, k, n, shp))
print(sum(seen.values()), 'ATen launches in the step')
