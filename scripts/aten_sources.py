"""Which Python lines make PyTorch launch kernels inside one training step (tuning aid): every ATen op with device
time of its own, with its shapes and the nearest frame inside this repository.

    python scripts/aten_sources.py [grid]
"""
import os
import sys
import collections

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from monosdf_amd import ops  # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork  # noqa: E402

torch.manual_seed(0)
grid = len(sys.argv) > 1 and sys.argv[1] == 'grid'
model = MonoSDFNetwork(bench.model_conf(grid=grid)).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
rays = bench.make_rays(1024, 1, 'cuda')
idx = torch.arange(1024, device='cuda')


def step():
    opt.zero_grad(set_to_none=True)
    out = model(rays, idx, if_pixel_input=True)
    ops.probe_loss(out).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
seen = collections.Counter()
for e in prof.events():
    if not e.key.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    frame = next((f for f in (e.stack or []) if 'monosdf_amd' in f or 'bench.py' in f or 'aten_sources' in f), '?')
    seen[(e.key, str(e.input_shapes)[:70], frame.replace(R + '/', '')[:90])] += 1
for (k, shp, fr), n in sorted(seen.items(), key=lambda kv: kv[0][2]):
    print('%-22s x%d %-70s %s' % (k, n, shp, fr))
print(sum(seen.values()), 'ATen launches in the step')
