"""bench.py -- rays/sec of the SDF volume-rendering training step (fwd + loss + bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: 1024 rays x 98 samples per GPU, ImplicitNetwork 8x256
(multires 6, skip [4], weight-norm, geometric init) + RenderingNetwork 289-256-256-3, error-bounded
sampler 64/128/32; synthetic rays (origins U(-0.2,0.2)^3, unit directions) and random-init weights.
Weak scaling: every rank renders its own 1024-ray batch, gradients are averaged with one RCCL
all-reduce of the flat 2.7 MB gradient, as the reference's DDP does.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N=1 only).  `value` is measured
on the fp32 MFMA core (--precision fp32, the default); the same K steps are then repeated on the bf16x3
core and reported under `alt_matrix_core` (never mixed into `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS = 1024
# entry points timed with HIP events inside the timed region (the kernels that make up >95 % of a step)
TIMED = {'msdf_sdf_forward', 'msdf_sdf_fwd_grad', 'msdf_sdf_backward', 'msdf_wgrad', 'msdf_reduce',
         'msdf_color_forward', 'msdf_color_backward', 'msdf_hash_encode_forward', 'msdf_hash_encode_backward',
         'msdf_hash_encode_second_backward'}
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def model_conf(width=256, depth=8, grid=False):
    from monosdf_amd.conf import ConfigTree
    skip = [4] if depth > 4 else []
    implicit = dict(d_in=3, d_out=1, dims=[width] * depth, geometric_init=True, bias=0.9, skip_in=skip,
                    weight_norm=True, multires=6, inside_outside=True)
    if grid:     # configs[2]: scannetGrids.conf:83-128 -- 16-level x 2-feature hash grid + 2x256 MLP
        implicit.update(dims=[width, width], skip_in=[4], use_grid_feature=True, divide_factor=1.1)
    return ConfigTree.from_dict(dict(
        feature_vector_size=width, scene_bounding_sphere=1.1, Grid_MLP=grid,
        implicit_network=implicit,
        rendering_network=dict(mode='idr', d_in=9, d_out=3, dims=[width, width], weight_norm=True,
                               multires_view=4, per_image_code=False),
        density=dict(params_init=dict(beta=0.1), beta_min=0.0001),
        ray_sampler=dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                         max_total_iters=5)))


def make_rays(n, seed, device):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.2, 0.2, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    pose = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    t = lambda a: torch.from_numpy(a).to(device)
    return {'ray_dirs': t(d), 'ray_cam_loc': t(o), 'ray_dirs_tmp': t(d.copy()), 'ray_pose': t(pose)}


def probe_loss(out):
    """mean|rgb| + 0.05 eikonal + 0.05 mean|normal| + 0.1 mean depth + 0.005 smooth."""
    loss = out['rgb_values'].abs().mean() + 0.05 * out['normal_map'].abs().mean() + 0.1 * out['depth_values'].mean()
    g1, g2 = out['grad_theta'], out['grad_theta_nei']
    loss = loss + 0.05 * ((g1.norm(2, dim=1) - 1) ** 2).mean()
    n1 = g1 / (g1.norm(2, dim=1).unsqueeze(-1) + 1e-5)
    n2 = g2 / (g2.norm(2, dim=1).unsqueeze(-1) + 1e-5)
    return loss + 0.005 * torch.norm(n1 - n2, dim=-1).mean()


def sdf_macs_per_point():
    """F_sdf = 39*256 + 2*256^2 + 256*217 + 4*256^2 + 256*257 = 524,544 MAC."""
    return 39 * 256 + 2 * 256 * 256 + 256 * 217 + 4 * 256 * 256 + 256 * 257


def cpu_baseline(n_rays=256, iters=2):
    """The CPU oracle (a port of the reference's PyTorch path, pinned by tests/golden) on a bounded sample."""
    from oracle import config, monosdf_oracle as mo, synth
    conf = config.mlp_config()
    state = synth.make_state(conf, seed=0)
    rays = synth.make_rays(n_rays, seed=1)
    noise = synth.make_noise(conf, n_rays, 128, seed=2)
    idx = torch.arange(n_rays)
    times = []
    for it in range(iters + 1):
        st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
        t0 = time.time()
        out = mo.render(st, conf, rays, idx, True, True, noise)
        mo.probe_loss(out).backward()
        times.append(time.time() - t0)
    dt = float(np.mean(times[1:]))
    return {'value': n_rays / dt, 'unit': 'rays/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d rays x 98 samples, fwd+bwd, %d timed iterations after 1 warm-up, fp32 PyTorch CPU oracle'
                      % (n_rays, iters)}


def grid_report(args, kern, dt, world, rounds, loss):
    """configs[2]: roofline of the hash-grid entry points against HBM (SURVEY.md 8(d) bytes per point)."""
    P_main, P_smp = N_RAYS * 98 + 4 * N_RAYS, N_RAYS * 128
    # bytes per call of each entry point, summed over its launches in one step
    per_step_bytes = {
        'msdf_hash_encode_forward': 1164.0 * P_smp * rounds + 1548.0 * P_main,     # sampler (no dy_dx) + main (dy_dx)
        'msdf_hash_encode_backward': 524.0 * P_main + 1164.0 * P_main,            # input-bwd (d/dx) + grid-bwd
        'msdf_hash_encode_second_backward': (524.0 + 1176.0) * P_main,
    }
    rows = {}
    for n, b in per_step_bytes.items():
        if n in kern:
            rows[n] = {'ms_per_step': kern[n]['ms_per_step'], 'algorithmic_GBps': b / (kern[n]['ms_per_step'] * 1e-3) / 1e9}
    dom = max(rows, key=lambda n: rows[n]['ms_per_step'])
    return {
        'metric': 'rays/sec fwd+bwd, 1024 rays x 98 samples, 16x2 hash grid + 2x256 SDF MLP',
        'value': world * N_RAYS * args.steps / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[2]: multi-res hash grid 16 levels x 2 feats (2^19 entries/level), '
                               '1024 rays x 98 samples, training step', 'sampler_rounds': rounds},
        'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': rows[dom]['algorithmic_GBps'], 'peak': 8000.0,
                     'unit': 'GB/s', 'frac': rows[dom]['algorithmic_GBps'] / 8000.0, 'traffic': None},
        'hash_entry_points': rows,
        'kernels_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in sorted(kern.items())},
        'loss': loss,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--config', choices=['mlp', 'grid'], default='mlp',
                    help="mlp = BASELINE.json configs[1] (the headline metric); grid = configs[2] (hash-grid path)")
    ap.add_argument('--precision', choices=['fp32', 'bf16x3'], default=os.environ.get('MONOSDF_PRECISION', 'fp32'),
                    help='matrix core of the fused MLP kernels that `value` is measured on (default fp32 MFMA)')
    ap.add_argument('--no-alt-precision', dest='alt_precision', action='store_false',
                    help='skip the second measurement on the other matrix core (reported under alt_matrix_core)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('MSDF_FORCE_DIST') == '1'    # the env knob exercises RCCL on one GPU
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', init_method='env://', device_id=device)

    from monosdf_amd import _lib, ops, parallel
    from monosdf_amd.model.network import MonoSDFNetwork

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def measure(precision):
        """W warm-up + K timed steps of the training step on the given matrix core; returns the max over ranks."""
        torch.manual_seed(0)                      # same initial weights on every rank (as DDP would broadcast)
        model = MonoSDFNetwork(model_conf(grid=(args.config == 'grid'))).to(device).train()
        model.set_precision(precision)
        params = [p for p in model.parameters() if p.requires_grad]
        try:
            opt = torch.optim.Adam(params, lr=5e-4, fused=True)      # one multi-tensor launch for the whole update
        except (RuntimeError, TypeError):
            opt = torch.optim.Adam(params, lr=5e-4)
        torch.manual_seed(1234 + rank)            # per-rank sampling noise
        rays = make_rays(N_RAYS, 1 + rank, device)
        indices = torch.arange(N_RAYS, device=device)

        def step():
            opt.zero_grad(set_to_none=True)
            out = model(rays, indices, if_pixel_input=True)
            loss = ops.probe_loss(out)        # the BASELINE.md probe loss, value + gradients in one HIP launch
            loss.backward()
            parallel.average_gradients(params)        # one flat RCCL all-reduce (no-op on one GPU)
            opt.step()
            return loss

        for _ in range(args.warmup):
            step()
        rounds = model.ray_sampler.last_rounds
        _lib.PROFILE = {}
        _lib.PROFILE_NAMES = TIMED
        barrier()
        t0 = time.time()
        for _ in range(args.steps):
            loss = step()
        barrier()
        dt = time.time() - t0
        prof, _lib.PROFILE = _lib.PROFILE, None
        if use_dist:
            t = torch.tensor([dt], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-entry-point device time (HIP events on the launch stream, inside the timed region)
        kern = {}
        for name, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            kern[name] = {'launches_per_step': len(ms) / args.steps, 'avg_ms': float(np.mean(ms)),
                          'ms_per_step': float(np.sum(ms)) / args.steps}
        plan = model.implicit_network._fused(device).mp.plan
        return dict(dt=dt, kern=kern, rounds=rounds, loss=float(loss.item()), precision=precision,
                    slots=(plan.hsum, plan.qsum, plan.absum))

    def mlp_rooflines(m):
        """Roofline of the dominant SDF kernel.  fp32 core: MFMA-bound (algorithmic FLOPs);
        bf16x3 core: the same kernels are HBM-bound on their saved-activation traffic."""
        kern = m['kern']
        P_main, P_eik, P_smp = N_RAYS * 98, 4 * N_RAYS, N_RAYS * 128
        F = sdf_macs_per_point()
        flops = {   # algorithmic FLOPs per launch (2 FLOP / MAC), SURVEY.md 8(d) multipliers
            'msdf_sdf_forward': 2.0 * F * P_smp,                 # no-grad forward, 1 x F_sdf
            'msdf_sdf_fwd_grad': 2.0 * 2 * F * (P_main + P_eik),   # forward + d/dx sweep
            'msdf_sdf_backward': 2.0 * 2 * F * (P_main + P_eik),   # p-bar = W q-bar and h-bar = W^T a-bar sweeps
        }
        hs, qs, ab = m['slots']
        P = P_main + P_eik
        hbm = {     # algorithmic bytes per launch: 4 B x slots read or written per point
            'msdf_sdf_fwd_grad': 4.0 * P * (3 * hs),               # H written, H re-read, PM written
            'msdf_sdf_backward': 4.0 * P * (5 * hs + qs + ab),     # H x2, PM, T read; T, QB, AB written
        }
        dom = max((n for n in kern if n in flops), key=lambda n: kern[n]['ms_per_step'])
        t = kern[dom]['avg_ms'] * 1e-3
        mf = {'bound': 'mfma', 'kernel': dom, 'achieved': flops[dom] / t / 1e12, 'peak': F32_MFMA_PEAK_TFLOPS,
              'unit': 'TFLOP/s', 'frac': flops[dom] / t / 1e12 / F32_MFMA_PEAK_TFLOPS, 'traffic': None,
              'avg_kernel_ms': kern[dom]['avg_ms']}
        # rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch of the same command,
        # profiles/r01_v8_pmc_{fp32,bf16x3}.json (separate passes; the kernels' data movement is fixed by P)
        measured = {'fp32': {'msdf_sdf_backward': 6.650e9, 'msdf_sdf_fwd_grad': 2.937e9, 'msdf_sdf_forward': 2.3e7},
                    'bf16x3': {'msdf_sdf_backward': 6.503e9, 'msdf_sdf_fwd_grad': 3.061e9, 'msdf_sdf_forward': 1.05e8}}
        mf['traffic'] = measured[m['precision']].get(dom)
        hb = None
        if dom in hbm:
            hb = {'bound': 'hbm', 'kernel': dom, 'achieved': hbm[dom] / t / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                  'frac': hbm[dom] / t / 1e9 / 8000.0, 'traffic': mf['traffic'], 'avg_kernel_ms': kern[dom]['avg_ms']}
        return mf, hb

    primary = measure(args.precision)
    alt = None
    if args.alt_precision and args.config == 'mlp':
        alt = measure('bf16x3' if args.precision == 'fp32' else 'fp32')

    if rank == 0:
        kern, dt, rounds = primary['kern'], primary['dt'], primary['rounds']
        dtype = 'f32' if args.precision == 'fp32' else 'bf16x3 (fp32 split into 2 bf16, fp32 accumulate)'
        if args.config == 'grid':
            res = grid_report(args, kern, dt, world, rounds, primary['loss'])
            res['dtype'] = dtype
            print(json.dumps(res))
            if use_dist:
                dist.destroy_process_group()
            return
        mf, hb = mlp_rooflines(primary)
        res = {
            'metric': 'rays/sec fwd+bwd, 1024 rays x 98 samples, 8x256 SDF MLP',
            'value': world * N_RAYS * args.steps / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: 1024 rays x 98 samples per GPU, ImplicitNetwork 8x256 + '
                                   'RenderingNetwork 289-256-256-3, error-bounded sampler (k=%d round), '
                                   'training step = fwd + loss + bwd + Adam' % rounds,
                       'rays_per_gpu': N_RAYS, 'samples_per_ray': 98, 'sampler_rounds': rounds,
                       'matrix_core': args.precision},
            'roofline': mf if args.precision == 'fp32' else (hb or mf),
            'kernels_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in sorted(kern.items())},
            'loss': primary['loss'],
        }
        if alt is not None:
            amf, ahb = mlp_rooflines(alt)
            other = 'bf16x3' if args.precision == 'fp32' else 'fp32'
            res['alt_matrix_core'] = {
                'matrix_core': other, 'value': world * N_RAYS * args.steps / alt['dt'], 'unit': 'rays/s',
                'ms_per_step': 1e3 * alt['dt'] / args.steps,
                'roofline': (ahb or amf) if other == 'bf16x3' else amf,
                'kernels_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in sorted(alt['kern'].items())},
                'note': 'same workload, steps and warm-up on the other matrix core of the fused MLP kernels; both '
                        'pass the same parity tests (tests/test_gpu_parity.py); `value` above is the %s core' % args.precision,
            }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
