"""How long does the HOST need to enqueue one training step (configs[1])?  The GPU needs ~8.6 ms; if the host needs
more than that, the step is launch-bound.  Measured by timing the Python side of K steps between two device
synchronisations and subtracting nothing: host time = wall time until the last launch returns."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                     # noqa: E402
from monosdf_amd import ops                      # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork   # noqa: E402

torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4, fused=True)
batches = [bench.make_rays(1024, 1 + b, 'cuda') for b in range(8)]
idx = torch.arange(1024, device='cuda')


def step(i):
    opt.zero_grad(set_to_none=True)
    out = model(batches[i % 8], idx, if_pixel_input=True)
    loss = ops.probe_loss(out)
    loss.backward()
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 30
t0 = time.time()
for i in range(K):
    step(i)
t_host = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
print('host enqueue %.2f ms/step, wall incl. device %.2f ms/step (the host blocks once per step on the sampler flags)'
      % (1e3 * t_host / K, 1e3 * t_all / K))
# the same with MSDF_SPECULATE_ROUNDS=all: no read-back at all, the host runs ahead freely
model.speculate_rounds = 'all'
for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.time()
for i in range(K):
    step(i)
t_host = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
print("speculate 'all': host enqueue %.2f ms/step, wall %.2f ms/step" % (1e3 * t_host / K, 1e3 * t_all / K))
# the host's own cost: the same step on 16 rays (the device finishes long before the host has enqueued it)
small = [bench.make_rays(16, 100 + b, 'cuda') for b in range(8)]
idx16 = torch.arange(16, device='cuda')
batches, idx = small, idx16
model.ray_sampler._hist[0] = []
for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.time()
for i in range(K):
    step(i)
torch.cuda.synchronize()
print("16 rays, speculate 'all': %.2f ms/step wall = what the host needs per step" % (1e3 * (time.time() - t0) / K))
