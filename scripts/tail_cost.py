"""How much the partly filled last round of workgroups costs the two 104,448-point kernels: time of forward+gradient
and backward for point counts around multiples of 512 workgroups x 64 points (tuning aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from monosdf_amd import _lib
from monosdf_amd.model.network import MonoSDFNetwork
torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
net = model.implicit_network
for P in (65536, 98304, 100352, 104448, 114688, 131072):
    x = (torch.rand(P, 3, device='cuda') * 2 - 1) * 0.9
    ca = torch.randn(P, 1, device='cuda'); cc = torch.randn(P, 3, device='cuda')
    def run():
        model.zero_grad(set_to_none=True)
        sdf, feat, grad = net.get_outputs(x)
        ((ca * sdf).sum() + feat.sum() + (cc * grad).sum()).backward()
    for _ in range(2): run()
    torch.cuda.synchronize()
    _lib.PROFILE = {}; _lib.PROFILE_NAMES = {'msdf_sdf_fwd_grad', 'msdf_sdf_backward'}
    for _ in range(5): run()
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in prof.items()}
    print('P %6d  WGs %4d = %.2f rounds   fwd_grad %.3f ms (%.2f us/kpt)   backward %.3f ms (%.2f us/kpt)' % (
        P, P // 64, P / 64 / 512, t['msdf_sdf_fwd_grad'], 1e3 * t['msdf_sdf_fwd_grad'] / (P / 1e3),
        t['msdf_sdf_backward'], 1e3 * t['msdf_sdf_backward'] / (P / 1e3)))
