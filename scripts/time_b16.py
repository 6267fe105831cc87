"""Times the bf16x3 forward kernel only (tuning aid; MONOSDF_HIP_LIB selects an experimental build)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import Case
from monosdf_amd.conf import ConfigTree
from monosdf_amd.model.network import MonoSDFNetwork
from monosdf_amd import ops

c = Case('mlp_w256_eval')
m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state); m = m.cuda().eval()
net = m.implicit_network
x = (torch.rand(131072, 3, device='cuda') * 2 - 1)
with torch.no_grad():
    fused, fw, fb, wpack, bpack = net.packed(torch.device('cuda'))
    w16, b16 = fused.pack_b16(fw, fb)
    for _ in range(3): ops.sdf_forward_nograd(fused, w16, b16, x, None, 1.1, 1.0, b16=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.sdf_forward_nograd(fused, w16, b16, x, None, 1.1, 1.0, b16=True)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get('MONOSDF_HIP_LIB', 'default'), 'bf16x3 131072 points: %.3f ms' % (e0.elapsed_time(e1) / 20))
