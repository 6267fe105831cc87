"""bf16x3 vs fp32 SDF kernels: accuracy against the CPU oracle and speed of the sampler forward (tuning aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import Case, rel_err
from oracle import monosdf_oracle as mo
from monosdf_amd.conf import ConfigTree
from monosdf_amd.model.network import MonoSDFNetwork
from monosdf_amd import ops


def model(name, precision):
    c = Case(name)
    m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state)
    return c, m.cuda().eval().set_precision(precision)


for name in ['mlp_w64_eval', 'mlp_w256_eval']:
    errs = []
    for prec in ops.PRECISIONS:
        c, m = model(name, prec)
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(4096 + 13, 3, generator=g) * 2 - 1)
        ref = mo.get_sdf_vals(c.state, c.conf, x)
        with torch.no_grad():
            v = m.implicit_network.get_sdf_vals(x.cuda())
        errs.append('%s err %.2e' % (prec, rel_err(v, ref)))
    print(name, '  '.join(errs))

x = (torch.rand(131072, 3, device='cuda') * 2 - 1)
for prec in ops.PRECISIONS:
    c, m = model('mlp_w256_eval', prec)
    net = m.implicit_network
    with torch.no_grad():
        fused, fw, fb, wpack, bpack = net.packed(torch.device('cuda'))
        args = (fused, wpack, bpack, x, None, 1.1, 1.0)
        for _ in range(3): ops.sdf_forward_nograd(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.sdf_forward_nograd(*args)
        e1.record(); torch.cuda.synchronize()
        print(prec, 'forward, 131072 points: %.3f ms' % (e0.elapsed_time(e1) / 20))
