"""Counts the ATen ops (and their input shapes) PyTorch itself runs in one training step (tuning aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from torch.profiler import profile, ProfilerActivity
from monosdf_amd import ops
from monosdf_amd.model.network import MonoSDFNetwork
torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
rays = bench.make_rays(1024, 1, 'cuda'); idx = torch.arange(1024, device='cuda')
def step():
    opt.zero_grad(set_to_none=True)
    out = model(rays, idx, if_pixel_input=True)
    ops.probe_loss(out).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::mul',
                     'aten::sum', 'aten::cat', 'aten::slice_backward', 'aten::zeros_like', 'aten::contiguous',
                     'aten::clone', 'aten::abs', 'aten::sgn', 'aten::sign', 'aten::select_backward', 'aten::neg',
                     'aten::div', 'aten::sub', 'aten::reshape', 'aten::_to_copy')]
for e in sorted(rows, key=lambda e: (e.key, str(e.input_shapes))):
    print('%-24s x%-3d %s' % (e.key, e.count, str(e.input_shapes)[:110]))
