"""One configuration of the SDF weight-gradient launch, for rocprofv3 --pmc runs (tuning aid): python scripts/time_wgrad.py [fp32|bf16x3] [S]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from monosdf_amd import plan as planlib, ops, _lib
prec = ops.PRECISIONS.index(sys.argv[1]) if len(sys.argv) > 1 else 0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
mp = planlib.build_sdf_plan(shapes, [4], 6, 0, False, 256)
dev = torch.device('cuda')
P_pad = 104448
woff, total = planlib.sdf_workspace(mp, P_pad)
ws = torch.randn(total, device=dev) * 0.01
prog = planlib.build_sdf_wgrad(mp, P_pad, lambda w: S)
items = torch.from_numpy(prog.items_bytes()).to(dev)
wg_map = torch.from_numpy(prog.wg_map()).to(dev)
part = torch.empty(prog.part_f + 64, device=dev)
for _ in range(6):
    _lib.call('msdf_wgrad', _lib.ptr(items), _lib.ptr(wg_map), wg_map.numel() // 2, _lib.ptr(part), P_pad, prec, _lib.ptr(ws), None, _lib.stream_ptr())
torch.cuda.synchronize()
