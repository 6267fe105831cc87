"""Host-side cost of one training step (time to ENQUEUE it, no device sync) next to the device time (tuning aid)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from monosdf_amd import ops
from monosdf_amd.model.network import MonoSDFNetwork
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train().set_precision(prec)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
rays = bench.make_rays(1024, 1, 'cuda'); idx = torch.arange(1024, device='cuda')
def step():
    opt.zero_grad(set_to_none=True)
    out = model(rays, idx, if_pixel_input=True)
    ops.probe_loss(out).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
# host time: a tiny batch makes the device side negligible, the Python work is identical
small = bench.make_rays(64, 1, 'cuda'); sidx = torch.arange(64, device='cuda')
def small_step():
    opt.zero_grad(set_to_none=True)
    out = model(small, sidx, if_pixel_input=True)
    ops.probe_loss(out).backward()
    opt.step()
for _ in range(5): small_step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(30): small_step()
torch.cuda.synchronize()
print(prec, '64-ray step (host-bound): %.3f ms' % ((time.time() - t0) / 30 * 1e3))
t0 = time.time()
for _ in range(30): step()
torch.cuda.synchronize()
print(prec, '1024-ray step: %.3f ms' % ((time.time() - t0) / 30 * 1e3))
