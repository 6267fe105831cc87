"""bench.py -- rays/sec of the SDF volume-rendering training step (fwd + loss + bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: 1024 rays x 98 samples per GPU, ImplicitNetwork 8x256
(multires 6, skip [4], weight-norm, geometric init) + RenderingNetwork 289-256-256-3, error-bounded
sampler 64/128/32; synthetic rays (origins U(-0.2,0.2)^3, unit directions) and random-init weights.
Weak scaling: every rank renders its own 1024-ray batch, gradients are averaged with one RCCL
all-reduce of the flat 2.7 MB gradient, as the reference's DDP does.

Every step renders a different one of 8 pre-generated ray batches.  Prints ONE JSON line (rank 0):
`value` = the configs[1] training step at the random-init state (density beta 0.1, the sampler converges in one
round) on the fp32 MFMA core, with `roofline` (dominant kernel, HIP-event timed inside the timed region;
`traffic` read from the rocprofv3 PMC summary named beside it) and `cpu_baseline` (the CPU oracle per
BASELINE.md section 3, rank 0, N=1 only).  Beside it, never mixed into `value` (N=1 only):
`sharp_state` = the same step at density beta 0.01, where the sampler needs 2+ rounds, with the
rounds per step and what the speculation of the round count cost; `hash_grid` = configs[2] with its HBM roofline;
`alt_matrix_core` = the bf16x3 core.
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
N_BATCHES = 8          # ray batches cycled through by the timed steps
# entry points timed with HIP events inside the timed region (the kernels that make up >95 % of a step)
TIMED = {'msdf_sdf_forward_if', 'msdf_sdf_fwd_grad', 'msdf_sdf_backward', 'msdf_wgrad', 'msdf_reduce',
         'msdf_color_forward', 'msdf_color_backward', 'msdf_hash_encode_forward', 'msdf_hash_encode_backward',
         'msdf_hash_encode_second_backward', 'msdf_hash_encode_backward_ws', 'msdf_hash_encode_second_backward_ws',
         'msdf_hash_encode_backward_fused'}
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


def _cpu_time(n_rays, iters, threads):
    from oracle import config, monosdf_oracle as mo, synth
    conf = config.mlp_config()
    state = synth.make_state(conf, seed=0)
    rays = synth.make_rays(n_rays, seed=1)
    noise = synth.make_noise(conf, n_rays, 128, seed=2)
    idx = torch.arange(n_rays)
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        times = []
        for it in range(iters + 1):
            st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
            t0 = time.time()
            out = mo.render(st, conf, rays, idx, True, True, noise)
            mo.probe_loss(out).backward()
            times.append(time.time() - t0)
    finally:
        torch.set_num_threads(before)
    return n_rays / float(np.mean(times[1:]))


def cpu_baseline():
    """the CPU oracle (a port of the reference's PyTorch path, pinned by tests/golden) on
    the configs[1] workload -- 1024 rays, 3 timed steady-state iterations on all host cores; plus the reference
    runner's own setting of ONE thread (monosdf_train.py:37) on a bounded sample."""
    # "all cores": PyTorch's default thread count can exceed the CPUs this process may use (a GPU box hands 16 of
    # its 128 to one GPU's job) and then runs slower than fewer threads: take the fastest of a few counts, measured
    # on a quarter batch, and say which
    most = torch.get_num_threads()
    tried = {}
    for t in sorted({most, min(most, 64), min(most, 32), min(most, 16)}):
        tried[t] = _cpu_time(N_RAYS // 4, 1, t)
    cores = max(tried, key=tried.get)
    return {'value': _cpu_time(N_RAYS, 3, cores), 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': '%d rays x 98 samples (the whole configs[1] batch), fwd+bwd, 3 timed iterations after 1 warm-up, '
                      'fp32 PyTorch CPU oracle, sampler k=1; thread count = the fastest of %s on a quarter batch '
                      '(rays/s: %s)' % (N_RAYS, sorted(tried), {k: round(v, 1) for k, v in sorted(tried.items())}),
            'one_thread': {'value': _cpu_time(192, 1, 1), 'unit': 'rays/s', 'cores': 1,
                           'sample': '192 rays x 98 samples, fwd+bwd, 1 timed iteration after 1 warm-up, '
                                     'torch.set_num_threads(1) as the reference runner sets it'}}


def pmc_traffic(entry, precision):
    """HBM bytes per launch of `entry` from the committed rocprofv3 --pmc summary (FETCH_SIZE x2 + WRITE_SIZE per the
    gfx950 note of MI355X_MICROARCH.md; scripts/pmc_sum.py) -> (bytes or None, file name or None)."""
    kernel = entry.replace('_if', '') + ('_k' if precision == 'fp32' else '_b16_k')
    for name in ('r02_pmc_%s.json' % precision, 'r01_v8_pmc_%s.json' % precision):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            row = json.load(open(path)).get(kernel)
            if row and 'hbm_bytes_per_launch_corrected' in row:
                return row['hbm_bytes_per_launch_corrected'], 'profiles/' + name
    return None, None


def pmc_traffic_grid(entry):
    """HBM bytes per call of a hash entry point = the sum over its kernels (profiles/r02_pmc_grid.json; FETCH_SIZE x2 +
    WRITE_SIZE; the x2 of MI355X_MICROARCH.md is calibrated for wide streaming reads, not for 8-byte gathers: the
    forward kernel's figure is an upper bound)."""
    kernels = {'msdf_hash_encode_forward': ['void hg_forward_kernel'],
               'msdf_hash_encode_backward': ['void hg_backward_input_kernel'],
               'msdf_hash_encode_second_backward_ws': ['void hg_second_backward_grad_kernel'],
               'msdf_hash_encode_backward_fused': ['hb_setup_k', 'void hb_count_k', 'hb_scan_k', 'void hb_place_k',
                                                   'void hb_accumulate_k']}
    path = os.path.join(ROOT, 'profiles', 'r02_pmc_grid.json')
    if not os.path.exists(path) or entry not in kernels:
        return None, None
    table = json.load(open(path))
    tot = 0.0
    for k in kernels[entry]:
        row = next((v for name, v in table.items() if name.startswith(k)), None)
        if row is None or 'hbm_bytes_per_launch_corrected' not in row:
            return None, None
        tot += row['hbm_bytes_per_launch_corrected']
    return tot, 'profiles/r02_pmc_grid.json'


def grid_report(args, kern, dt, world, rounds, loss, sampler):
    """configs[2]: roofline of the hash-grid entry points against HBM (SURVEY.md 8(d) bytes per point)."""
    P_main, P_smp = N_RAYS * 98 + 4 * N_RAYS, N_RAYS * 128
    # bytes per call of each entry point, summed over its launches in one step
    per_step_bytes = {
        # main pass (with dy_dx) + one sampler evaluation (no dy_dx) per further launch of the step
        'msdf_hash_encode_forward': 1548.0 * P_main + 1164.0 * P_smp * max(
            0.0, kern.get('msdf_hash_encode_forward', {}).get('launches_per_step', 1.0 + rounds) - 1.0),
        'msdf_hash_encode_backward': 524.0 * P_main + 1164.0 * P_main,            # input-bwd (d/dx) + grid-bwd
        'msdf_hash_encode_second_backward': (524.0 + 1176.0) * P_main,
        # the fused node (ops.GridSdfFunction): d/dx only; grad_grad only; both embedding scatters in one pass
        'msdf_hash_encode_backward_ws': 524.0 * P_main,
        'msdf_hash_encode_second_backward_ws': 524.0 * P_main,
        'msdf_hash_encode_backward_fused': (1164.0 + 1176.0) * P_main,
    }
    if 'msdf_hash_encode_backward_fused' in kern:      # there the plain entry point computes d/dx only
        per_step_bytes['msdf_hash_encode_backward'] = 524.0 * P_main
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
                               '1024 rays x 98 samples, training step', 'sampler_rounds': sampler['rounds_per_step'],
                   'ray_batches': N_BATCHES},
        'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': rows[dom]['algorithmic_GBps'], 'peak': 8000.0,
                     'unit': 'GB/s', 'frac': rows[dom]['algorithmic_GBps'] / 8000.0,
                     'traffic': pmc_traffic_grid(dom)[0], 'traffic_source': pmc_traffic_grid(dom)[1],
                     'note': 'parity of the hash-grid arithmetic is unpinned by reference outputs (CUDA-only in the '
                             'reference, no vectors): checked against the restated oracle only'},
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
                    help="mlp = BASELINE.json configs[1] (the headline metric); grid = configs[2] alone")
    ap.add_argument('--precision', choices=['fp32', 'bf16x3'], default=os.environ.get('MONOSDF_PRECISION', 'fp32'),
                    help='matrix core of the fused MLP kernels that `value` is measured on (default fp32 MFMA)')
    ap.add_argument('--no-alt-precision', dest='alt_precision', action='store_false',
                    help='skip the measurement on the other matrix core (reported under alt_matrix_core)')
    ap.add_argument('--no-extras', dest='extras', action='store_false',
                    help='skip sharp_state and hash_grid (profiling runs want the headline workload only)')
    ap.add_argument('--beta', type=float, default=0.1,
                    help='density beta of the state `value` is measured at (0.1 = random init; profiling runs of the '
                         'sharp state pass 0.01)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('MSDF_FORCE_DIST') == '1'    # the env knob exercises RCCL on one GPU
    if use_dist:
        import torch.distributed as dist
        # RCCL may print a version banner on stdout when the first communicator is built: keep stdout for the one
        # JSON line (the banner goes to stderr)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend='nccl', init_method='env://', device_id=device)
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from monosdf_amd import _lib, ops, parallel
    from monosdf_amd.model.network import MonoSDFNetwork

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def measure(precision, grid=False, beta=0.1):
        """W warm-up + K timed steps of the training step on the given matrix core; returns the max over ranks."""
        torch.manual_seed(0)                      # same initial weights on every rank (as DDP would broadcast)
        model = MonoSDFNetwork(model_conf(grid=grid)).to(device).train()
        model.set_precision(precision)
        with torch.no_grad():
            model.density.beta.fill_(beta)
        params = [p for p in model.parameters() if p.requires_grad]
        try:
            opt = torch.optim.Adam(params, lr=5e-4, fused=True)      # one multi-tensor launch for the whole update
        except (RuntimeError, TypeError):
            opt = torch.optim.Adam(params, lr=5e-4)
        averager = parallel.GradientAverager(params) if use_dist else None
        torch.manual_seed(1234 + rank)            # per-rank sampling noise
        # every rank cycles through its own 8 batches (weak scaling: no DistributedSampler in the reference)
        batches = [make_rays(N_RAYS, 1 + 1000 * rank + b, device) for b in range(N_BATCHES)]
        indices = torch.arange(N_RAYS, device=device)
        smp = model.ray_sampler
        rounds_seen = []

        def step(i):
            opt.zero_grad(set_to_none=True)
            out = model(batches[i % N_BATCHES], indices, if_pixel_input=True)
            loss = ops.probe_loss(out)        # the BASELINE.md probe loss, value + gradients in one HIP launch
            loss.backward()
            if averager is not None:
                averager.average()            # one flat RCCL all-reduce of a persistent buffer
            opt.step()
            rounds_seen.append(smp.last_rounds)
            return loss

        for i in range(args.warmup):
            step(i)
        del rounds_seen[:]
        stats0 = dict(smp.stats)
        _lib.PROFILE = {}
        _lib.PROFILE_NAMES = TIMED
        barrier()
        t0 = time.time()
        for i in range(args.steps):
            loss = step(args.warmup + i)
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
        hist = {}
        for r in rounds_seen:
            hist[str(r)] = hist.get(str(r), 0) + 1
        d = {k: smp.stats[k] - stats0[k] for k in stats0}
        sampler = {'rounds_per_step': hist, 'mean_rounds': float(np.mean(rounds_seen)),
                   # the round count of a step is guessed from the last steps so that the host never waits for the
                   # device inside a step: `repeated_passes` = steps whose guess was too small (the forward pass ran
                   # twice), `idle_rounds` = rounds enqueued beyond the ones that ran (one SDF evaluation of 131,072
                   # points each, results unused)
                   'repeated_passes': d['repeats'], 'idle_rounds': d['idle_rounds'], 'beta0': beta + 1e-4}
        return dict(dt=dt, kern=kern, rounds=hist, sampler=sampler, loss=float(loss.item()), precision=precision,
                    slots=(plan.hsum, plan.qsum, plan.absum))

    def mlp_rooflines(m):
        """Roofline of the dominant SDF kernel.  fp32 core: MFMA-bound (algorithmic FLOPs);
        bf16x3 core: the same kernels are HBM-bound on their saved-activation traffic."""
        kern = m['kern']
        P_main, P_eik, P_smp = N_RAYS * 98, 4 * N_RAYS, N_RAYS * 128
        F = sdf_macs_per_point()
        flops = {   # algorithmic FLOPs per launch (2 FLOP / MAC), SURVEY.md 8(d) multipliers
            'msdf_sdf_forward_if': 2.0 * F * P_smp,              # no-grad forward, 1 x F_sdf (skipped launches count too)
            'msdf_sdf_fwd_grad': 2.0 * 2 * F * (P_main + P_eik),   # forward + d/dx sweep
            'msdf_sdf_backward': 2.0 * 2 * F * (P_main + P_eik),   # p-bar = W q-bar and h-bar = W^T a-bar sweeps
        }
        hs, qs, ab = m['slots']
        P = P_main + P_eik
        hbm = {     # algorithmic bytes per launch: 4 B x slots read or written per point
            'msdf_sdf_fwd_grad': 4.0 * P * (3 * hs),               # H written, H re-read, PM written
            'msdf_sdf_backward': 4.0 * P * (5 * hs + qs + ab),     # H x2, PM, T read; T, QB, AB written
        }
        # dominant = the largest per-LAUNCH duration among the kernels with a FLOP model (the sampler's forward
        # kernel runs once per round; per launch it is the smallest of the three)
        dom = max((n for n in kern if n in flops), key=lambda n: kern[n]['avg_ms'])
        t = kern[dom]['avg_ms'] * 1e-3
        traffic, source = pmc_traffic(dom, m['precision'])
        mf = {'bound': 'mfma', 'kernel': dom, 'achieved': flops[dom] / t / 1e12, 'peak': F32_MFMA_PEAK_TFLOPS,
              'unit': 'TFLOP/s', 'frac': flops[dom] / t / 1e12 / F32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
              'traffic_source': source, 'avg_kernel_ms': kern[dom]['avg_ms']}
        hb = None
        if dom in hbm:
            hb = {'bound': 'hbm', 'kernel': dom, 'achieved': hbm[dom] / t / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                  'frac': hbm[dom] / t / 1e9 / 8000.0, 'traffic': traffic, 'traffic_source': source,
                  'avg_kernel_ms': kern[dom]['avg_ms']}
        return mf, hb

    def ms(m):
        return {k: round(v['ms_per_step'], 4) for k, v in sorted(m['kern'].items())}

    def rate(m):
        return world * N_RAYS * args.steps / m['dt']

    grid_only = args.config == 'grid'
    primary = measure(args.precision, grid=grid_only, beta=args.beta)
    single = world == 1 and not use_dist
    extras = args.extras and single and not grid_only
    sharp = measure(args.precision, beta=0.01) if extras else None
    alt = measure('bf16x3' if args.precision == 'fp32' else 'fp32') if (extras and args.alt_precision) else None
    grid = measure('fp32', grid=True) if extras else None

    if rank == 0:
        dtype = 'f32' if args.precision == 'fp32' else 'bf16x3 (fp32 split into 2 bf16, fp32 accumulate)'
        if grid_only:
            res = grid_report(args, primary['kern'], primary['dt'], world, primary['sampler']['mean_rounds'],
                              primary['loss'], primary['sampler'])
            res['dtype'] = dtype
            print(json.dumps(res))
            if use_dist:
                dist.destroy_process_group()
            return
        mf, hb = mlp_rooflines(primary)
        res = {
            'metric': 'rays/sec fwd+bwd, 1024 rays x 98 samples, 8x256 SDF MLP',
            'value': rate(primary), 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * primary['dt'] / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: 1024 rays x 98 samples per GPU, ImplicitNetwork 8x256 + '
                                   'RenderingNetwork 289-256-256-3, error-bounded sampler, random-init weights '
                                   '(density beta %g), training step = fwd + loss + bwd + Adam, a different one of '
                                   '%d ray batches every step' % (args.beta, N_BATCHES),
                       'rays_per_gpu': N_RAYS, 'samples_per_ray': 98, 'sampler_rounds': primary['rounds'],
                       'ray_batches': N_BATCHES, 'matrix_core': args.precision},
            'roofline': mf if args.precision == 'fp32' else (hb or mf),
            'sampler': primary['sampler'],
            'kernels_ms_per_step': ms(primary),
            'loss': primary['loss'],
        }
        if sharp is not None:
            smf, shb = mlp_rooflines(sharp)
            res['sharp_state'] = {
                'what': 'the same training step with density beta = 0.01 (SURVEY 8(d) "sharpened" state): the sampler '
                        'needs 2+ rounds, each one more SDF evaluation of 131,072 points',
                'value': rate(sharp), 'unit': 'rays/s', 'ms_per_step': 1e3 * sharp['dt'] / args.steps,
                'sampler': sharp['sampler'], 'roofline': smf if args.precision == 'fp32' else (shb or smf),
                'kernels_ms_per_step': ms(sharp)}
        if grid is not None:
            g = grid_report(args, grid['kern'], grid['dt'], world, grid['sampler']['mean_rounds'], grid['loss'],
                            grid['sampler'])
            res['hash_grid'] = {k: g[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'config', 'roofline',
                                                  'hash_entry_points', 'kernels_ms_per_step')}
        if alt is not None:
            amf, ahb = mlp_rooflines(alt)
            other = 'bf16x3' if args.precision == 'fp32' else 'fp32'
            res['alt_matrix_core'] = {
                'matrix_core': other, 'value': rate(alt), 'unit': 'rays/s',
                'ms_per_step': 1e3 * alt['dt'] / args.steps,
                'roofline': (ahb or amf) if other == 'bf16x3' else amf,
                'kernels_ms_per_step': ms(alt),
                'note': 'same workload, steps and warm-up on the other matrix core of the fused MLP kernels (opt-in: '
                        'narrower arithmetic than the reference); `value` above is the %s core' % args.precision,
            }
        if single and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
