import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from monosdf_amd.utils import render
import bench
from monosdf_amd.model.network import MonoSDFNetwork
torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().eval()
import sys as _s
fn = (lambda p: model.implicit_network.raw_sdf(p)) if 'raw' in _s.argv[1:] else (lambda p: model.implicit_network(p)[:, 0])
with torch.no_grad():
    fn(torch.zeros(64, 3, device='cuda'))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        orig = render._to_host
        tt = {}
        def th(t):
            torch.cuda.synchronize(); tt['gpu_done'] = time.time()
            r = orig(t); tt['host_done'] = time.time(); return r
        render._to_host = th
        blocks = list(render.sdf_volume(fn, resolution=512, grid_boundary=(-1.1, 1.1), shard=False))
        render._to_host = orig
        t1 = time.time()
        print('rep %d total %.3f s: device part %.3f, to host %.3f' % (rep, t1 - t0, tt['gpu_done'] - t0, tt['host_done'] - tt['gpu_done']))
