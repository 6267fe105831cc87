"""Per-level cost of the hash-grid scatter kernel (tuning aid): calls the backward entry point with L = 1 on each
level's table slice (H' ~ the level's resolution), ray-ordered points as the training step produces them."""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import ctypes as C
import numpy as np
import torch
from monosdf_amd import _lib
from oracle import hashgrid_oracle as hg

geo = hg.level_geometry(dict(num_levels=16, level_dim=2, base_size=16, end_size=2048, logmap=19))
N, S = 1024, 102
rng = np.random.default_rng(0)
o = rng.uniform(-0.2, 0.2, (N, 1, 3)); d = rng.normal(size=(N, 1, 3)); d /= np.linalg.norm(d, axis=2, keepdims=True)
z = np.sort(rng.uniform(0.0, 1.5, (N, S, 1)), axis=1)
x = ((o + z * d) / 1.1 + 1) / 2
x01 = torch.from_numpy(np.clip(x, 0, 1).reshape(-1, 3).astype(np.float32)).cuda()
B = x01.shape[0]
lib = _lib.load()
tot = 0.0
for l in range(16):
    hs = geo['offsets'][l + 1] - geo['offsets'][l]
    off = torch.tensor([0, hs], dtype=torch.int32).cuda()
    Hl = int(round(16 * 2 ** (l * geo['S'])))
    grad = torch.randn(1, B, 2, device='cuda')
    table = torch.zeros(hs, 2, device='cuda')
    dummy = torch.zeros(1, device='cuda')
    def run():
        rc = lib.msdf_hash_encode_backward(_lib.ptr(grad), _lib.ptr(x01), _lib.ptr(table), _lib.ptr(off), _lib.ptr(table),
                                           B, 3, 2, 1, C.c_float(0.0), Hl, 0, _lib.ptr(dummy), _lib.ptr(dummy),
                                           _lib.stream_ptr())
        assert rc == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    print('level %2d  res %5d  entries %7d  %.4f ms' % (l, Hl, hs, ms))
print('sum %.3f ms' % tot)
