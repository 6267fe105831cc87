"""Equal-iteration training comparison of the two matrix cores (north-star criterion "PSNR within 0.1 dB of the
reference after equal iterations", here between the fp32 core -- bit-level fp32 arithmetic -- and the bf16x3 core).

A teacher network (random seed 1, sharpened so the scene has structure) renders colour / depth / normal targets
for a fixed set of rays; a student (seed 2) is trained on them with the fused MonoSDFLoss + Adam, once per core,
with identical initial weights, batches and sampling noise.  Training trajectories are chaotic, so a third
run -- the fp32 core again, initial weights perturbed by one part in 1e6 -- gives the spread that two
equally exact implementations would show.  Prints the PSNR of all runs on held-out rays.

    python scripts/train_parity.py [iterations=400]
"""
import json
import math
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402

import bench  # noqa: E402
from monosdf_amd.model.loss import MonoSDFLoss  # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork  # noqa: E402

N = 1024
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N_BATCH = 8


def rays(seed):
    return bench.make_rays(N, seed, 'cuda')


def render(model, r):
    model.eval()
    with torch.no_grad():
        out = model(r, torch.arange(N, device='cuda'), if_pixel_input=True)
    return out


def psnr(a, b):
    return -10.0 * math.log10(torch.mean((a - b) ** 2).item())


def main():
    torch.manual_seed(1)
    teacher = MonoSDFNetwork(bench.model_conf()).cuda()
    with torch.no_grad():                      # give the scene some colour structure
        for n, p in teacher.rendering_network.named_parameters():
            if n.endswith('weight_g'):
                p.mul_(6.0)
    train_rays = [rays(10 + i) for i in range(N_BATCH)]
    test_rays = [rays(100 + i) for i in range(2)]
    targets = []
    for r in train_rays + test_rays:
        o = render(teacher, r)
        targets.append({'rgb': o['rgb_values'][None].clone(), 'depth': (o['depth_values'] / 50.0)[None].clone(),
                        'normal': o['normal_map'][None].clone(), 'mask': torch.ones(1, N, 1, device='cuda')})
    result = {}
    for run, precision in (('fp32', 'fp32'), ('bf16x3', 'bf16x3'), ('fp32_perturbed_1e-6', 'fp32')):
        torch.manual_seed(2)
        student = MonoSDFNetwork(bench.model_conf()).cuda().set_precision(precision)
        if 'perturbed' in run:
            g = torch.Generator(device='cuda').manual_seed(9)
            with torch.no_grad():
                for p in student.parameters():
                    p.mul_(1.0 + 1e-6 * torch.randn(p.shape, device='cuda', generator=g))
        loss_fn = MonoSDFLoss(rgb_loss='torch.nn.L1Loss', eikonal_weight=0.05, smooth_weight=0.005, depth_weight=0.1,
                              normal_l1_weight=0.05, normal_cos_weight=0.05)
        opt = torch.optim.Adam(student.parameters(), lr=5e-4)
        torch.manual_seed(3)
        curve = []
        for it in range(ITERS):
            student.train()
            b = it % N_BATCH
            out = student(train_rays[b], torch.arange(N, device='cuda'), if_pixel_input=True)
            res = loss_fn(out, targets[b], if_pixel_input=True)
            opt.zero_grad(set_to_none=True)
            res['loss'].backward()
            opt.step()
            if it % 50 == 0 or it == ITERS - 1:
                curve.append((it, res['loss'].item()))
        ps = [psnr(render(student, r)['rgb_values'], targets[N_BATCH + i]['rgb'][0]) for i, r in enumerate(test_rays)]
        result[run] = {'psnr_db': sum(ps) / len(ps), 'loss_curve': curve}
        print(run, 'held-out PSNR %.3f dB' % result[run]['psnr_db'], 'loss', curve[0][1], '->', curve[-1][1])
    d = abs(result['fp32']['psnr_db'] - result['bf16x3']['psnr_db'])
    d0 = abs(result['fp32']['psnr_db'] - result['fp32_perturbed_1e-6']['psnr_db'])
    result['abs_psnr_difference_db'] = d
    result['abs_psnr_difference_fp32_control_db'] = d0
    result['iterations'] = ITERS
    print('|PSNR(fp32) - PSNR(bf16x3)| = %.4f dB, |PSNR(fp32) - PSNR(fp32, init perturbed 1e-6)| = %.4f dB after %d '
          'iterations' % (d, d0, ITERS))
    print(json.dumps(result))


if __name__ == '__main__':
    main()
