"""Times the embedding-gradient scatter of the hash grid at configs[2] size: one atomic per corner vs the binned
scatter vs both gradients fused into one binned scatter.  Points: the ray samples of the bench workload.

    python scripts/bench_hash_scatter.py            # prints one JSON line
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
    # ray-major sample points like the training step's (sorted depths along each ray), plus 4N eikonal points
    z = torch.sort(torch.rand(N, S_, device='cuda', generator=g) * 2.0, 1)[0]
    pts = (rays['ray_cam_loc'].unsqueeze(1) + z.unsqueeze(2) * rays['ray_dirs'].unsqueeze(1)).reshape(-1, 3)
    pts = torch.cat([pts, (torch.rand(4 * N, 3, device='cuda', generator=g) * 2 - 1) * 1.1])
    x = ((pts / 1.1 + 1) / 2).clamp(0, 1).contiguous()
    if os.environ.get('PTS') == 'random':         # no spatial coherence at all (diagnostic)
        x = torch.rand(x.shape, device='cuda', generator=g)
    B, L, C = x.shape[0], 16, 2
    grad = torch.randn(L, B, C, device='cuda', generator=g)
    grad2 = torch.randn(L, B, C, device='cuda', generator=g)
    gg = torch.randn(B, 3, device='cuda', generator=g)
    emb, offs = enc.embeddings.detach(), enc.offsets
    n = emb.shape[0]
    Sc, H = enc.log2_scale, int(enc.base_resolution)
    st = _lib.stream_ptr()
    dy = torch.empty(B, L * 3 * C, device='cuda')
    out = torch.empty(L, B, C, device='cuda')
    _lib.call('msdf_hash_encode_forward', _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, Sc, H, 1,
              _lib.ptr(dy), st)
    nbytes = _lib.load().msdf_hash_scatter_workspace_bytes(B, C, L, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    acc = torch.zeros_like(emb)
    ggrad = torch.empty(L, B, C, device='cuda')
    P = _lib.ptr
    calls = {
        'atomic_first': lambda: _lib.call('msdf_hash_encode_backward', P(grad), P(x), P(emb), P(offs), P(acc), B, 3, C, L,
                                          Sc, H, 0, P(dy), None, st),
        'atomic_second': lambda: _lib.call('msdf_hash_encode_second_backward', P(grad2), P(x), P(emb), P(offs), B, 3, C, L,
                                           Sc, H, 1, P(dy), P(gg), P(ggrad), P(acc), st),
        'binned_first': lambda: _lib.call('msdf_hash_encode_backward_ws', P(grad), P(x), P(emb), P(offs), P(acc), B, 3, C,
                                          L, Sc, H, 0, P(dy), None, n, P(ws), nbytes, st),
        'binned_second': lambda: _lib.call('msdf_hash_encode_second_backward_ws', P(grad2), P(x), P(emb), P(offs), B, 3, C,
                                           L, Sc, H, 1, P(dy), P(gg), P(ggrad), P(acc), n, P(ws), nbytes, st),
        'binned_fused': lambda: _lib.call('msdf_hash_encode_backward_fused', P(grad), P(grad2), P(x), P(offs), P(acc), B, 3,
                                          C, L, Sc, H, P(gg), n, P(ws), nbytes, st),
        'binned_fused_out': lambda: _lib.call('msdf_hash_encode_backward_fused_out', P(grad), P(grad2), P(x), P(offs),
                                              P(acc), B, 3, C, L, Sc, H, P(gg), n, P(ws), nbytes, st),
    }
    res = {'points': B, 'workspace_MB': nbytes / 1e6, 'pts': os.environ.get('PTS', 'rays')}
    only = os.environ.get('ONLY')
    for name, fn in calls.items():
        if only and name not in only.split(','):
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + '_ms'] = e0.elapsed_time(e1) / 20
    print(json.dumps(res))


if __name__ == '__main__':
    main()
