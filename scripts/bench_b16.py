"""bf16x3 vs fp32 forward kernel: accuracy against the CPU oracle and speed (tuning aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import Case, rel_err
from oracle import monosdf_oracle as mo
from monosdf_amd.conf import ConfigTree
from monosdf_amd.model.network import MonoSDFNetwork
from monosdf_amd import ops, _lib

for name in ['mlp_w64_eval', 'mlp_w256_eval', 'gridless_w128_train']:
    c = Case(name)
    m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state); m = m.cuda().eval()
    net = m.implicit_network
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(4096 + 13, 3, generator=g) * 2 - 1) * 1.0
    ref = mo.get_sdf_vals(c.state, c.conf, x)
    with torch.no_grad():
        fused, fw, fb, wpack, bpack = net.packed(x.device if False else torch.device('cuda'))
        v32 = ops.sdf_forward_nograd(fused, wpack, bpack, x.cuda(), None, 1.1, 1.0)
        w16, b16 = fused.pack_b16(fw, fb)
        v16 = ops.sdf_forward_nograd(fused, w16, b16, x.cuda(), None, 1.1, 1.0, b16=True)
    print(name, 'fp32 err %.2e   bf16x3 err %.2e' % (rel_err(v32, ref), rel_err(v16, ref)))

# speed at the sampler size
c = Case('mlp_w256_eval')
m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state); m = m.cuda().eval()
net = m.implicit_network
x = (torch.rand(131072, 3, device='cuda') * 2 - 1)
with torch.no_grad():
    fused, fw, fb, wpack, bpack = net.packed(torch.device('cuda'))
    w16, b16 = fused.pack_b16(fw, fb)
    for label, args, kw in [('fp32', (fused, wpack, bpack), {}), ('bf16x3', (fused, w16, b16), {'b16': True})]:
        for _ in range(3): ops.sdf_forward_nograd(*args, x, None, 1.1, 1.0, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.sdf_forward_nograd(*args, x, None, 1.1, 1.0, **kw)
        e1.record(); torch.cuda.synchronize()
        print(label, '131072 points: %.3f ms' % (e0.elapsed_time(e1) / 10))
