"""Times msdf_hash_encode_forward at configs[2] size: the main pass (104,448 points, with dy_dx) and one sampler
evaluation (131,072 points, no dy_dx).  MSDF_HASH_FORWARD=level | xcd[:wd:wh] selects the launch form.

    python scripts/bench_hash_forward.py            # prints one JSON line
"""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
from monosdf_amd import _lib  # noqa: E402
from monosdf_amd.hashencoder.hashgrid import HashEncoder  # noqa: E402


def main():
    N, S_ = bench.N_RAYS, 98
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048).cuda()
    rays = bench.make_rays(N, 1, 'cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    res = {'mode': os.environ.get('MSDF_HASH_FORWARD', 'default')}
    for name, S_n, calc in (('main_dy_dx', 98, 2), ('sampler', 128, 0)):
        z = torch.sort(torch.rand(N, S_n, device='cuda', generator=g) * 2.0, 1)[0]
        pts = (rays['ray_cam_loc'].unsqueeze(1) + z.unsqueeze(2) * rays['ray_dirs'].unsqueeze(1)).reshape(-1, 3)
        if calc:
            pts = torch.cat([pts, (torch.rand(4 * N, 3, device='cuda', generator=g) * 2 - 1) * 1.1])
        x = ((pts / 1.1 + 1) / 2).clamp(0, 1).contiguous()
        B, L, C = x.shape[0], 16, 2
        emb, offs = enc.embeddings.detach(), enc.offsets
        st = _lib.stream_ptr()
        dy = torch.empty(B, L * 3 * C, device='cuda')
        out = torch.empty(L, B, C, device='cuda')
        fn = lambda: _lib.call('msdf_hash_encode_forward', _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(out), B, 3,
                               C, L, enc.log2_scale, int(enc.base_resolution), calc, _lib.ptr(dy), st)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        nbytes = (1548.0 if calc else 1164.0) * B
        res[name] = {'points': B, 'ms': ms, 'algorithmic_GBps': nbytes / ms / 1e6, 'checksum': float(out.double().sum())}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
