"""Micro-benchmark of the weight-gradient launch on a synthetic workspace (tuning aid)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import numpy as np
from monosdf_amd import plan as planlib, ops, _lib

shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
mp = planlib.build_sdf_plan(shapes, [4], 6, 0, False, 256)
dev = torch.device('cuda')
fm = ops.FusedMlp(mp, dev)
P_pad = 104448
woff, total = planlib.sdf_workspace(mp, P_pad)
ws = torch.randn(total, device=dev) * 0.01
flops = None
PREC = 0

def run(split_fn, label, iters=5):
    prog = planlib.build_sdf_wgrad(mp, P_pad, split_fn)
    items = torch.from_numpy(prog.items_bytes({'ws': ws.data_ptr()})).to(dev)
    wg_map = torch.from_numpy(prog.wg_map()).to(dev)
    part = torch.empty(prog.part_f + 64, device=dev)
    st = _lib.stream_ptr()
    macs = sum(it['wx'] * it['wy'] for it in prog.items) * P_pad
    def once():
        _lib.call('msdf_wgrad', _lib.ptr(items), _lib.ptr(wg_map), wg_map.numel() // 2, _lib.ptr(part), P_pad, PREC, st)
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): once()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('%-34s WGs %4d  %.3f ms  %.1f TFLOP/s' % (label, wg_map.numel() // 2, ms, 2 * macs / ms / 1e9))

prog = planlib.balanced_program(planlib.build_sdf_wgrad, mp, P_pad)
print('balanced: WGs', len(prog.wg_map()) // 2, sorted(set((it['weight'], it['n_splits']) for it in prog.items)))
for PREC in (0, 1):
    for S in (48, 56, 59, 60, 61, 64, 72, 73, 85):
        run(lambda w, S=S: S, '%s all items S=%d' % (ops.PRECISIONS[PREC], S))
