"""Time the SDF forward+gradient kernel (8x256 network, training mode) for point counts around whole rounds
of workgroups: how long does a partly filled round take?  (diagnostic; prints ms per launch)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                     # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork   # noqa: E402

torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
net = model.implicit_network
net.share('cuda')
for P in (64 * 96, 64 * 256, 64 * 512, 64 * 608, 64 * 1024, 64 * 1536, 64 * 1632, 64 * 2048):
    x = (torch.rand(P, 3, device='cuda') - 0.5)
    for _ in range(3):
        net.evaluate(x, P, P, save=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.evaluate(x, P, P, save=True)
    e1.record()
    torch.cuda.synchronize()
    print('%6d workgroups  %.3f ms' % (P // 64, e0.elapsed_time(e1) / 10))
