"""Time the SDF kernels of the configs[2] network (hash grid + 2x256 MLP; `mlp` as argument: the 8x256 network) for point
counts around whole rounds of workgroups: per entry point, ms per launch from HIP events (diagnostic).
MSDF_SIDE_STREAM=0 is set so that nothing runs beside the timed kernels."""
import os
import sys
os.environ.setdefault('MSDF_SIDE_STREAM', '0')
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                     # noqa: E402
from monosdf_amd import _lib                     # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork   # noqa: E402

grid = 'mlp' not in sys.argv[1:]
torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf(grid=grid)).cuda().train()
net = model.implicit_network
NAMES = ('msdf_sdf_fwd_grad', 'msdf_sdf_backward', 'msdf_wgrad', 'msdf_hash_node_forward', 'msdf_hash_node_scatter',
         'msdf_hash_node_second_grad', 'msdf_hash_node_input_gradient', 'msdf_hash_transpose')
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64 * 96, 64 * 512, 64 * 1024, 64 * 1536, 64 * 1632, 64 * 2048]
for P in sizes:
    x = (torch.rand(P, 3, device='cuda') - 0.5)
    ns = P - P // 25
    g = [torch.randn(ns, 1, device='cuda'), torch.randn(P, 256, device='cuda') * 0.01, torch.randn(ns, 3, device='cuda'),
         torch.randn(P - ns, 3, device='cuda')]

    def once():
        for p in net.parameters():
            p.grad = None
        sdf, feat, nrm, nrm_b = net.evaluate(x, P, P, save=True, split=ns)
        torch.autograd.backward([sdf, feat, nrm, nrm_b], g)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    _lib.PROFILE, _lib.PROFILE_NAMES = {}, set(NAMES)
    for _ in range(8):
        once()
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    row = {n: float(np.mean([a.elapsed_time(b) for a, b in ev])) for n, ev in prof.items()}
    print('%6d workgroups ' % (P // 64) + '  '.join('%s %.3f' % (n.replace('msdf_', ''), row[n]) for n in NAMES if n in row),
          flush=True)
