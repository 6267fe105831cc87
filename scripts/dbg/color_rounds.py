"""Time the colour-network forward kernel for point counts around whole rounds of workgroups (512 workgroup slots on
the chip): what does the partly filled fourth round of the 100,352-point launch cost?  (diagnostic; ms per launch)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                     # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork   # noqa: E402

torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
rn = model.rendering_network
model.implicit_network.share('cuda')
rn.share('cuda')
S = 64
for wgs in (1, 32, 96, 256, 512, 1024, 1536, 1568, 2048):
    P = 64 * wgs
    N = P // S
    g = torch.Generator(device='cuda').manual_seed(0)
    pts = torch.randn(P, 3, device='cuda', generator=g)
    nrm = torch.randn(P, 3, device='cuda', generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda', generator=g), dim=-1)
    feat = torch.randn(P, 256, device='cuda', generator=g) * 0.3
    idx = torch.arange(N, device='cuda')
    fn = lambda: rn(pts, nrm, dirs, feat, idx, if_pixel_input=True, samples_per_ray=S)['rgb']
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
    print('%6d workgroups  %.4f ms' % (wgs, e0.elapsed_time(e1) / 20), flush=True)
