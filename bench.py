"""bench.py -- rays/sec of the SDF volume-rendering training step (fwd + loss + bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) BEFORE any GPU call and exits
    with their code; under a launcher (WORLD_SIZE set) it is one of the ranks and WORLD_SIZE must equal --gpus.
    Fewer than N visible devices is an error, never a silent one-GPU run.

Workload = BASELINE.json configs[1]: 1024 rays x 98 samples per GPU, ImplicitNetwork 8x256
(multires 6, skip [4], weight-norm, geometric init) + RenderingNetwork 289-256-256-3, error-bounded
sampler 64/128/32; synthetic rays (origins U(-0.2,0.2)^3, unit directions) and random-init weights.
Weak scaling: every rank renders its own 1024-ray batch, gradients are averaged over RCCL as the reference's
DDP does (one flat 2.7 MB all-reduce; with the hash grid the 48.8 MB table gradient travels as a second message
that starts behind the scatter kernel, parallel.GradientAverager).  For N > 1 the line also carries `multi_gpu`:
the number of ranks that answered an all-reduce, per-rank ms/step (min / max) and the all-reduce time per step
(HIP events) for configs[1] and configs[2].

Every step renders a different one of 8 pre-generated ray batches.  Prints ONE JSON line (rank 0):
`value` = the configs[1] training step at the random-init state (density beta 0.1, the sampler converges in one
round) on the fp32 MFMA core, with `roofline` (dominant kernel, HIP-event timed inside the timed region;
`traffic` read from the rocprofv3 PMC summary named beside it) and `cpu_baseline` (the CPU oracle per
BASELINE.md section 3, rank 0, N=1 only).  Beside it, never mixed into `value` (N=1 only):
`sharp_state` = the same step at density beta 0.01, where the sampler needs 2+ rounds, with the
rounds per step and what the speculation of the round count cost; `hash_grid` = configs[2] with its HBM roofline;
`alt_matrix_core` = the bf16x3 core, `alt_matrix_core_x6` = the bf16x6 core; `sustained` = a 400-step TRAINING run (closed-form scene, fused
MonoSDFLoss, Adam, fresh rays every step, density beta 0.02 at the start: beta falls and the sampler goes to 2+ rounds).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS = 1024
N_BATCHES = 8          # ray batches cycled through by the timed steps
# entry points timed with HIP events inside the timed region (the kernels that make up >95 % of a step)
TIMED = {'msdf_sdf_forward_if', 'msdf_sdf_forward_lm', 'msdf_sdf_fwd_grad', 'msdf_sdf_backward', 'msdf_wgrad', 'msdf_reduce',
         'msdf_color_forward', 'msdf_color_backward', 'msdf_hash_encode_forward', 'msdf_hash_encode_backward',
         'msdf_hash_encode_second_backward', 'msdf_hash_encode_backward_ws', 'msdf_hash_encode_second_backward_ws',
         'msdf_hash_encode_backward_fused', 'msdf_hash_encode_backward_fused_out', 'msdf_hash_node_forward',
         'msdf_hash_node_input_gradient', 'msdf_hash_node_second_grad', 'msdf_hash_node_scatter'}
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


def grid_sdf_macs_per_point():
    """configs[2] (scannetGrids.conf:83-128): (39 PE + 32 grid features) x 256 + 256 x 256 + 256 x 257 = 149,504 MAC."""
    return 71 * 256 + 256 * 256 + 256 * 257


def scene_targets(rays, radius=0.6):
    """Supervision of the `sustained` training run: a closed-form scene (the one the 200-step reference trajectories of
    tests/golden/traj_*.npz were recorded on) -- rays from inside hit a sphere of `radius`; colour a smooth function of
    the hit point, the depth cue the hit distance up to the monocular scale, the normal cue the inward normal in the
    camera frame.  Shapes as the reference's data loader delivers them ([1, N, C]); computed on the device before
    the timed region."""
    o, d = rays['ray_cam_loc'].double(), rays['ray_dirs'].double()
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - radius * radius
    t = -b + torch.sqrt(b * b - c)
    p = o + t.unsqueeze(-1) * d
    rgb = 0.5 + 0.4 * torch.sin(3.0 * p + torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64, device=o.device))
    rot = rays['ray_pose'][:, :3, :3].double().transpose(1, 2)
    n_cam = (rot @ (-p / radius).unsqueeze(-1)).squeeze(-1)
    depth = t * rays['ray_dirs_tmp'][:, 2].double().abs()
    f = lambda a: a.float()[None].contiguous()
    return {'rgb': f(rgb), 'depth': f((depth / 50.0).unsqueeze(-1)), 'normal': f(n_cam),
            'mask': torch.ones(1, o.shape[0], 1, device=o.device)}


def gather_ceiling(with_dy):
    """Measured ceiling of the hash forward kernel's gathers (profiles/r04_gather_ceiling.json: scripts/dbg/
    gather_ceiling.hip, mode "replay" -- the library's own index arithmetic on a training-like point set, the real table,
    the real loads and stores, no interpolation arithmetic) in GB/s of SURVEY 8(d)'s algorithmic bytes, or None."""
    path = os.path.join(ROOT, 'profiles', 'r04_gather_ceiling.json')
    if not os.path.exists(path):
        return None
    for row in json.load(open(path)).get('rows', []):
        if row.get('mode') == 'replay' and bool(row.get('dy_dx')) == bool(with_dy):
            return row['GBps']
    return None


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
    the configs[1] workload.  Three separately labelled figures: `value` = the WHOLE 1024-ray batch, 3 timed
    steady-state iterations, on the thread count picked by `thread_scan`; `thread_scan` = a QUARTER batch (256 rays,
    1 timed iteration) on a few thread counts -- smaller tensors fit the caches better, so its rays/s are higher than
    `value` at the same thread count and are only used to choose; `one_thread` = the reference runner's own setting
    (monosdf_train.py:37), 192 rays, 2 timed iterations."""
    # "all cores": PyTorch's default thread count can exceed the CPUs this process may use (a GPU box hands 16 of
    # its 128 to one GPU's job) and then runs slower than fewer threads
    most = torch.get_num_threads()
    tried = {}
    for t in sorted({most, min(most, 64), min(most, 32), min(most, 16)}):
        tried[t] = _cpu_time(N_RAYS // 4, 1, t)
    cores = max(tried, key=tried.get)
    return {'value': _cpu_time(N_RAYS, 3, cores), 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': 'FULL batch: %d rays x 98 samples (the whole configs[1] batch), fwd+bwd, 3 timed iterations after '
                      '1 warm-up, fp32 PyTorch CPU oracle, sampler k=1, %d threads' % (N_RAYS, cores),
            'thread_scan': {'sample': 'QUARTER batch: %d rays x 98 samples, fwd+bwd, 1 timed iteration after 1 warm-up; '
                                      'used only to pick the thread count of `value` (a quarter batch runs at more '
                                      'rays/s than the full one on the same threads)' % (N_RAYS // 4),
                            'unit': 'rays/s', 'rays_per_s_by_threads': {str(k): round(v, 1) for k, v in sorted(tried.items())}},
            'one_thread': {'value': _cpu_time(192, 2, 1), 'unit': 'rays/s', 'cores': 1,
                           'sample': '192 rays x 98 samples, fwd+bwd, 2 timed iterations after 1 warm-up, '
                                     'torch.set_num_threads(1) as the reference runner sets it'}}


def pmc_traffic(entry, precision):
    """HBM bytes per launch of `entry` from the committed rocprofv3 --pmc summary (FETCH_SIZE x2 + WRITE_SIZE per the
    gfx950 note of MI355X_MICROARCH.md; scripts/pmc_sum.py) -> (bytes or None, file name or None)."""
    base = entry.replace('_if', '')
    # the bf16 kernels are templates on the number of planes: "void msdf_..._b16_k<2>" (bf16x3) / "<3>" (bf16x6)
    names = {'fp32': [base + '_k'], 'bf16x3': ['void %s_b16_k<2>' % base, base + '_b16_k'],
             'bf16x6': ['void %s_b16_k<3>' % base]}[precision]
    for name in ('r04_pmc_%s.json' % precision, 'r03_pmc_%s.json' % precision, 'r02_pmc_%s.json' % precision,
                 'r01_v8_pmc_%s.json' % precision):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            table = json.load(open(path))
            for kernel in names:
                row = table.get(kernel)
                if row and 'hbm_bytes_per_launch_corrected' in row:
                    return row['hbm_bytes_per_launch_corrected'], 'profiles/' + name
    return None, None


def pmc_traffic_grid(entry):
    """HBM bytes per call of a hash entry point = the sum over its kernels (profiles/r03_pmc_grid.json; FETCH_SIZE x2 +
    WRITE_SIZE; the x2 of MI355X_MICROARCH.md is calibrated for wide streaming reads, not for 8-byte gathers: the
    forward kernel's figure is an upper bound)."""
    kernels = {'msdf_hash_encode_forward': ['void hg_forward_kernel'],
               'msdf_hash_encode_backward': ['void hg_backward_input_kernel'],
               'msdf_hash_encode_second_backward_ws': ['void hg_second_backward_grad_kernel'],
               'msdf_hash_encode_backward_fused_out': ['void hb2_place_k', 'void hb2_accumulate_k'],
               # node forms of the same kernels (ops.GridSdfFunction, the sampler's evaluations)
               'msdf_hash_node_forward': ['void hg_node_forward_kernel'],
               'msdf_hash_node_input_gradient': ['void hg_node_input_gradient_kernel'],
               'msdf_hash_node_second_grad': ['void hg_node_second_grad_kernel'],
               'msdf_hash_node_scatter': ['void hb2_place_k', 'void hb2_accumulate_k']}
    path = next((os.path.join(ROOT, 'profiles', n) for n in ('r04_pmc_grid.json', 'r03_pmc_grid.json')
                 if os.path.exists(os.path.join(ROOT, 'profiles', n))), None)
    if path is None or entry not in kernels:
        return None, None
    table = json.load(open(path))
    tot = 0.0
    for k in kernels[entry]:
        row = next((v for name, v in table.items() if name.startswith(k)), None)
        if row is None or 'hbm_bytes_per_launch_corrected' not in row:
            return None, None
        tot += row['hbm_bytes_per_launch_corrected']
    return tot, 'profiles/' + os.path.basename(path)


def grid_report(args, kern, dt, world, rounds, loss, sampler, backward_alone_ms=None):
    """configs[2]: roofline of the hash-grid entry points against HBM (SURVEY.md 8(d) bytes per point)."""
    P_main, P_smp = N_RAYS * 98 + 4 * N_RAYS, N_RAYS * 128
    # bytes per call of each entry point, summed over its launches in one step
    n_fwd = lambda name: kern.get(name, {}).get('launches_per_step', 1.0 + rounds)
    per_step_bytes = {
        # main pass (with dy_dx) + one sampler evaluation (no dy_dx) per further launch of the step
        'msdf_hash_encode_forward': 1548.0 * P_main + 1164.0 * P_smp * max(0.0, n_fwd('msdf_hash_encode_forward') - 1.0),
        'msdf_hash_encode_backward': 524.0 * P_main + 1164.0 * P_main,            # input-bwd (d/dx) + grid-bwd
        'msdf_hash_encode_second_backward': (524.0 + 1176.0) * P_main,
        # the fused node (ops.GridSdfFunction): d/dx only; grad_grad only; both embedding scatters in one pass
        'msdf_hash_encode_backward_ws': 524.0 * P_main,
        'msdf_hash_encode_second_backward_ws': 524.0 * P_main,
        'msdf_hash_encode_backward_fused': (1164.0 + 1176.0) * P_main,
        'msdf_hash_encode_backward_fused_out': (1164.0 + 1176.0) * P_main,
        # node forms: the same kernels' bytes; x01 / layout / scaling inside add nothing algorithmic
        'msdf_hash_node_forward': 1548.0 * P_main + 1164.0 * P_smp * max(0.0, n_fwd('msdf_hash_node_forward') - 1.0),
        'msdf_hash_node_input_gradient': 524.0 * P_main,
        'msdf_hash_node_second_grad': 524.0 * P_main,
        'msdf_hash_node_scatter': (1164.0 + 1176.0) * P_main,
    }
    if 'msdf_hash_encode_backward_fused' in kern or 'msdf_hash_encode_backward_fused_out' in kern:
        # there the plain entry point computes d/dx only
        per_step_bytes['msdf_hash_encode_backward'] = 524.0 * P_main
    alone = backward_alone_ms if isinstance(backward_alone_ms, dict) else {}
    rows = {}
    for n, b in per_step_bytes.items():
        if n in kern:
            rows[n] = {'ms_per_step': kern[n]['ms_per_step'], 'algorithmic_GBps': b / (kern[n]['ms_per_step'] * 1e-3) / 1e9}
            if n in alone:
                # inside the step this entry point's kernels run beside the colour network's weight-gradient launch
                # (second stream): the HIP events around it then span that launch's workgroups too
                rows[n]['ms_alone'] = alone[n] * kern[n]['launches_per_step']
                rows[n]['algorithmic_GBps_alone'] = b / (rows[n]['ms_alone'] * 1e-3) / 1e9
    own = lambda n: rows[n].get('ms_alone', rows[n]['ms_per_step'])
    dom = max(rows, key=own)
    # the hash forward kernel against its measured gather ceiling (per launch: the main pass has dy_dx, the sampler's not)
    fwd = kern.get('msdf_hash_node_forward') or kern.get('msdf_hash_encode_forward')
    ceiling = None
    if fwd is not None and gather_ceiling(True) is not None:
        n_l = fwd['launches_per_step']
        ceil_ms = 1e-6 * (1548.0 * P_main / gather_ceiling(True) + 1164.0 * P_smp * max(0.0, n_l - 1.0) / gather_ceiling(False))
        ceiling = {'kernel': 'hg_node_forward_kernel', 'ms_per_step': fwd['ms_per_step'], 'ceiling_ms_per_step': ceil_ms,
                   'frac_of_gather_ceiling': ceil_ms / fwd['ms_per_step'],
                   'ceiling_GBps': {'with_dy_dx': gather_ceiling(True), 'without': gather_ceiling(False)},
                   'source': 'profiles/r04_gather_ceiling.json (scripts/dbg/gather_ceiling.hip, mode replay: the same index '
                             'stream, loads and stores without the interpolation arithmetic)'}
    # the MLP kernels own most of the configs[2] step: fraction of the fp32 MFMA peak of the dominant one and of the SDF
    # weight-gradient launch (algorithmic FLOPs: 2 FLOP/MAC x the multipliers of SURVEY 8(d) on the 71-256-256-257 network)
    Fg = grid_sdf_macs_per_point()
    # (the sampler's no-grad forward needs the sdf row of the output layer only: 256 of its 256 x 257 MACs)
    mlp_flops = {'msdf_sdf_forward_if': 2.0 * (Fg - 256 * 256) * P_smp, 'msdf_sdf_fwd_grad': 4.0 * Fg * P_main,
                 'msdf_sdf_backward': 4.0 * Fg * P_main}
    mlp = {}
    for n, fl in mlp_flops.items():
        if n in kern:
            t = kern[n]['avg_ms'] * 1e-3
            mlp[n] = {'avg_kernel_ms': kern[n]['avg_ms'], 'achieved': fl / t / 1e12, 'frac': fl / t / 1e12 / F32_MFMA_PEAK_TFLOPS}
    mlp_dom = max(mlp, key=lambda n: mlp[n]['avg_kernel_ms']) if mlp else None
    mlp_roofline = None
    if mlp_dom is not None:
        mlp_roofline = dict(mlp[mlp_dom], bound='mfma', kernel=mlp_dom, peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                            all_kernels={n: round(v['frac'], 3) for n, v in mlp.items()})
        if mlp_dom == 'msdf_sdf_backward' and alone.get('msdf_sdf_backward'):
            t_alone = alone['msdf_sdf_backward']
            mlp_roofline['note'] = ('inside the step this kernel may share the chip with the colour network\'s weight-gradient '
                                    'launch (second stream); `alone` = its own duration, 5 extra steps without the side stream')
            mlp_roofline['alone'] = {'avg_kernel_ms': t_alone,
                                     'frac': mlp_flops[mlp_dom] / (t_alone * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
    return {
        'metric': 'rays/sec fwd+bwd, 1024 rays x 98 samples, 16x2 hash grid + 2x256 SDF MLP',
        'value': world * N_RAYS * args.steps / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[2]: multi-res hash grid 16 levels x 2 feats (2^19 entries/level), '
                               '1024 rays x 98 samples, training step', 'sampler_rounds': sampler['rounds_per_step'],
                   'ray_batches': N_BATCHES},
        'roofline': {'bound': 'hbm', 'kernel': dom,
                     'achieved': rows[dom].get('algorithmic_GBps_alone', rows[dom]['algorithmic_GBps']), 'peak': 8000.0,
                     'unit': 'GB/s',
                     'frac': rows[dom].get('algorithmic_GBps_alone', rows[dom]['algorithmic_GBps']) / 8000.0,
                     'timing': 'alone (5 extra steps without the side stream)' if 'ms_alone' in rows[dom] else 'inside the step',
                     'traffic': pmc_traffic_grid(dom)[0], 'traffic_source': pmc_traffic_grid(dom)[1],
                     'note': 'parity of the hash-grid arithmetic is unpinned by reference outputs (CUDA-only in the '
                             'reference, no vectors): checked against the restated oracle only'},
        'hash_entry_points': rows,
        'hash_forward_vs_gather_ceiling': ceiling,
        'mlp_roofline': mlp_roofline,
        'kernels_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in sorted(kern.items())},
        'loss': loss,
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--config', choices=['mlp', 'grid'], default='mlp',
                    help="mlp = BASELINE.json configs[1] (the headline metric); grid = configs[2] alone")
    ap.add_argument('--precision', choices=['fp32', 'bf16x3', 'bf16x6'], default=os.environ.get('MONOSDF_PRECISION', 'fp32'),
                    help='matrix core of the fused MLP kernels that `value` is measured on (default fp32 MFMA)')
    ap.add_argument('--no-alt-precision', dest='alt_precision', action='store_false',
                    help='skip the measurement on the other matrix core (reported under alt_matrix_core)')
    ap.add_argument('--no-extras', dest='extras', action='store_false',
                    help='skip sharp_state, hash_grid and sustained (profiling runs want the headline workload only)')
    ap.add_argument('--sustained-steps', type=int, default=400,
                    help='length of the `sustained` run (consecutive Adam steps from random init, fresh rays per step)')
    ap.add_argument('--beta', type=float, default=0.1,
                    help='density beta of the state `value` is measured at (0.1 = random init; profiling runs of the '
                         'sharp state pass 0.01)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous check without a GPU: the ranks meet over gloo on the CPU, the kernels '
                         'are skipped (a step is the flat gradient all-reduce alone), `value` is null')
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N > 1 outside a launcher: start the N ranks as a CHILD process tree (one process per GPU, the
    reference's launch: training/exp_runner.py:73-77 under torch.distributed.launch) and return its exit code.
    Nothing here touches the GPU (counting devices does not initialise it), and nothing is exec'ed over this process."""
    if not args.dry_run:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) visible -- refusing to run fewer ranks\n'
                             % (args.gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes needs it on this host)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


RDZV_TIMEOUT_S = float(os.environ.get('MSDF_RDZV_TIMEOUT', '120'))


class _RankPrefix:
    """stderr of a rank with '[rank k] ' in front of every line (N ranks share the launcher's stderr)."""

    def __init__(self, stream, rank):
        self.stream, self.prefix, self.bol = stream, '[rank %d] ' % rank, True

    def write(self, text):
        for piece in text.splitlines(True):
            if self.bol:
                self.stream.write(self.prefix)
            self.stream.write(piece)
            self.bol = piece.endswith('\n')
        return len(text)

    def flush(self):
        self.stream.flush()

    def __getattr__(self, name):
        return getattr(self.stream, name)


def join_ranks(backend, rank, world, **pg_kwargs):
    """Rendezvous that fails FAST and says who is missing (the reference's launch, training/exp_runner.py:73-77, waits
    the default 10 minutes -- the driver's whole limit -- when a rank does not come up): every rank announces itself in
    the store, then waits at most MSDF_RDZV_TIMEOUT (120 s) for the others; a rank that times out names the missing
    ranks on stderr and the process exits non-zero.  Returns None on success, an exit code otherwise."""
    import datetime
    import torch.distributed as dist
    timeout = datetime.timedelta(seconds=RDZV_TIMEOUT_S)
    try:
        if os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True':
            # under torch.distributed.run the launcher's agent hosts the store: every rank is a client of it
            store, _, _ = next(iter(dist.rendezvous('env://', rank, world, timeout=timeout)))
        else:
            # ranks started by hand: rank 0 hosts the store and does NOT wait for the others inside the constructor
            # (that wait cannot say who is missing)
            store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), world, is_master=(rank == 0),
                                  timeout=timeout, wait_for_workers=False)
    except Exception as e:       # the store itself (hosted by rank 0 / the launcher) is not there
        sys.stderr.write('bench.py: rank %d could not reach the rendezvous store at %s:%s within %g s (%s: %s) -- rank 0 / '
                         'the launcher did not arrive\n' % (rank, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'),
                                                            RDZV_TIMEOUT_S, type(e).__name__, e))
        return 3
    store.set_timeout(timeout)
    store.set('bench/arrived/%d' % rank, '1')
    t_end, missing = time.time() + RDZV_TIMEOUT_S, []
    for r in range(world):
        try:
            store.wait(['bench/arrived/%d' % r], datetime.timedelta(seconds=max(0.05, t_end - time.time())))
        except Exception:
            missing.append(r)
    if missing:
        sys.stderr.write('bench.py: rank(s) %s of %d did not arrive within %g s -- giving up (exit 3)\n'
                         % (', '.join(str(r) for r in missing), world, RDZV_TIMEOUT_S))
        return 3
    dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world, timeout=timeout, **pg_kwargs)
    return None


class _StdoutToStderr:
    """RCCL / gloo may print a banner on stdout when the first communicator is built: stdout is kept for the one
    JSON line, so file descriptor 1 points at stderr while a process group comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def dry_run(args, rank, world):
    """The rendezvous and the collectives of a step without kernels: gloo on the CPU."""
    import torch.distributed as dist
    from monosdf_amd import parallel
    with _StdoutToStderr():
        rc = join_ranks('gloo', rank, world)
        if rc is not None:
            return rc
        count = torch.ones(1)
        dist.all_reduce(count)
    try:
        params = [torch.nn.Parameter(torch.zeros(670395))]           # the 8x256 network's gradient, one message
        averager = parallel.GradientAverager(params)
        dist.barrier()
        t0 = time.time()
        for i in range(args.steps):
            params[0].grad = torch.full_like(params[0], float(rank + 1))
            averager.average()
        dist.barrier()
        dt = torch.tensor([time.time() - t0], dtype=torch.float64)
        every = [torch.zeros_like(dt) for _ in range(world)]
        dist.all_gather(every, dt)
        ok = bool(torch.allclose(params[0].grad, torch.full_like(params[0], (world + 1) / 2.0)))
        if rank == 0:
            per_rank = [1e3 * float(t) / max(args.steps, 1) for t in every]
            print(json.dumps({
                'metric': 'rays/sec fwd+bwd, 1024 rays x 98 samples, 8x256 SDF MLP', 'value': None, 'unit': 'rays/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': max(per_rank),
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'dry_run': True,
                'config': {'workload': 'dry run: rendezvous + the flat gradient all-reduce of configs[1] over gloo on the '
                                       'CPU, no kernels'},
                'multi_gpu': {'backend': 'gloo', 'ranks_answering': int(count.item()), 'gradient_mean_correct': ok,
                              'per_rank_ms_per_step': {'min': min(per_rank), 'max': max(per_rank)}}}))
    finally:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        sys.stderr.write('bench.py: --gpus must be >= 1\n')
        return 2
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        sys.stderr = _RankPrefix(sys.stderr, rank)
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s)\n' % (args.gpus, world))
        return 2
    if args.dry_run:
        return dry_run(args, rank, world)
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write('bench.py: rank %d has no GPU (%d visible)\n' % (rank, torch.cuda.device_count()))
        return 2
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('MSDF_FORCE_DIST') == '1'    # the env knob exercises RCCL on one GPU
    if use_dist:
        import torch.distributed as dist
        with _StdoutToStderr():
            rc = join_ranks('nccl', rank, world, device_id=device)
            if rc is not None:
                return rc
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()

    from monosdf_amd import _lib, ops, parallel
    from monosdf_amd.model.network import MonoSDFNetwork

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def device_rays(n, gen):
        """A fresh ray batch drawn on the device (the `sustained` run: one per step, made before the timed region)."""
        o = (torch.rand(n, 3, device=device, generator=gen) - 0.5) * 0.4
        d = torch.nn.functional.normalize(torch.randn(n, 3, device=device, generator=gen), dim=1)
        pose = torch.eye(4, device=device).expand(n, 4, 4).contiguous()
        return {'ray_dirs': d, 'ray_cam_loc': o, 'ray_dirs_tmp': d.clone(), 'ray_pose': pose}

    def measure(precision, grid=False, beta=0.1, steps=None, fresh=False, train=False):
        """W warm-up + K timed steps of the training step on the given matrix core; returns the max over ranks.
        fresh: a different ray batch EVERY step (pre-generated on the device) instead of cycling through 8.
        train: a training run instead of the probe loss -- targets of a closed-form scene (scene_targets), the fused
        MonoSDFLoss with the weights of the reference's confs, NO warm-up steps (the run starts at the initial state);
        the timed region is cut into windows of 100 steps (one host sync each)."""
        steps = args.steps if steps is None else steps
        warmup = 0 if train else args.warmup
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
        averager = parallel.GradientAverager(params, timing=True) if use_dist else None
        torch.manual_seed(1234 + rank)            # per-rank sampling noise
        # every rank cycles through its own 8 batches (weak scaling: no DistributedSampler in the reference)
        loss_fn, targets = None, None
        if fresh:
            gen = torch.Generator(device=device)
            gen.manual_seed(99 + rank)
            batches = [device_rays(N_RAYS, gen) for _ in range(warmup + steps)]
            if train:
                from monosdf_amd.model.loss import MonoSDFLoss
                # scannet_mlp.conf loss block: eikonal 0.05, smooth 0.005, depth 0.1, normal l1 / cos 0.05
                loss_fn = MonoSDFLoss('torch.nn.L1Loss', eikonal_weight=0.05, smooth_weight=0.005, depth_weight=0.1,
                                      normal_l1_weight=0.05, normal_cos_weight=0.05)
                targets = [scene_targets(b) for b in batches]
        else:
            batches = [make_rays(N_RAYS, 1 + 1000 * rank + b, device) for b in range(N_BATCHES)]
        indices = torch.arange(N_RAYS, device=device)
        smp = model.ray_sampler
        rounds_seen = []

        def step(i):
            opt.zero_grad(set_to_none=True)
            out = model(batches[i % len(batches)], indices, if_pixel_input=True)
            if loss_fn is not None:           # MonoSDFLoss (model/loss.py:252-311), value + gradients in one HIP launch
                loss = loss_fn(out, targets[i % len(batches)], if_pixel_input=True)['loss']
            else:
                loss = ops.probe_loss(out)    # the BASELINE.md probe loss, value + gradients in one HIP launch
            loss.backward()
            if averager is not None:
                averager.average()            # RCCL: the flat MLP block here, the table gradient already in flight
            opt.step()
            rounds_seen.append(smp.last_rounds)
            return loss

        first_loss = None
        for i in range(warmup):
            l0 = step(i)
            first_loss = first_loss if first_loss is not None else float(l0.item())
        del rounds_seen[:]
        stats0 = dict(smp.stats)
        if averager is not None:
            averager.timings = {'flat': [], 'overlapped': []}
        _lib.PROFILE = {}
        _lib.PROFILE_NAMES = TIMED
        barrier()
        t0 = time.time()
        windows = []
        for i in range(steps):
            loss = step(warmup + i)
            if train:
                if first_loss is None:
                    first_loss = loss           # read after the run (no sync inside the first step)
                if (i + 1) % 100 == 0 or i + 1 == steps:
                    torch.cuda.synchronize()
                    windows.append((i + 1, time.time() - t0, float(model.density.get_beta().item())))
        barrier()
        dt_own = dt = time.time() - t0
        prof, _lib.PROFILE = _lib.PROFILE, None
        if torch.is_tensor(first_loss):
            first_loss = float(first_loss.item())
        alone = None
        if grid and world == 1:
            # inside the step the colour network's weight-gradient launch (side stream) shares the chip with the SDF backward
            # kernel and / or the embedding scatter: their own durations, measured on 5 extra steps without the side stream
            side, ops.USE_SIDE_STREAM = ops.USE_SIDE_STREAM, False
            _lib.PROFILE, _lib.PROFILE_NAMES = {}, {'msdf_sdf_backward', 'msdf_hash_node_scatter'}
            for i in range(5):
                step(warmup + steps + i)
            torch.cuda.synchronize()
            alone = {n: float(np.mean([a.elapsed_time(b) for a, b in ev])) for n, ev in _lib.PROFILE.items()}
            _lib.PROFILE, _lib.PROFILE_NAMES, ops.USE_SIDE_STREAM = None, TIMED, side
        multi = None
        if use_dist:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank = [1e3 * float(e.item()) / steps for e in every]
            dt = max(float(e.item()) for e in every)
            count = torch.ones(1, device=device)
            dist.all_reduce(count)
            ar = {k: (float(np.sum([a.elapsed_time(b) for a, b in v])) / steps if v else 0.0)
                  for k, v in averager.timings.items()}
            multi = {'rccl_ranks': int(count.item()),
                     'per_rank_ms_per_step': {'min': min(per_rank), 'max': max(per_rank)},
                     # HIP events on the stream each message runs on (rank 0): `flat` = the MLP block, after the backward
                     # pass on the main stream (exposed); `overlapped` = the hash-grid table gradient on the comm stream,
                     # started behind the scatter kernel, under the weight-gradient kernels
                     'allreduce_ms_per_step': ar,
                     'gradient_bytes': {'flat': 4 * (averager.flat.numel() if averager.flat is not None else 0),
                                        'overlapped': 4 * sum(p.numel() for p in averager.big)}}
            averager.close()
        # per-entry-point device time (HIP events on the launch stream, inside the timed region)
        kern = {}
        for name, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            # the sampler's SDF evaluations go through msdf_sdf_forward_lm (msdf_sdf_forward_if = the same without a
            # level-major feature tensor): reported under the name the earlier rounds' lines use
            name = 'msdf_sdf_forward_if' if name == 'msdf_sdf_forward_lm' else name
            kern[name] = {'launches_per_step': len(ms) / steps, 'avg_ms': float(np.mean(ms)),
                          'ms_per_step': float(np.sum(ms)) / steps}
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
        if averager is not None and not use_dist:
            averager.close()
        return dict(dt=dt, dt_own=dt_own, steps=steps, kern=kern, rounds=hist, sampler=sampler, loss=float(loss.item()),
                    first_loss=first_loss, precision=precision, slots=(plan.hsum, plan.qsum, plan.absum), multi=multi,
                    beta_end=float(model.density.get_beta().item()), windows=windows, rounds_seq=list(rounds_seen),
                    backward_alone_ms=alone)

    def mlp_rooflines(m):
        """Roofline of the dominant SDF kernel.  fp32 core: MFMA-bound (algorithmic FLOPs);
        bf16x3 core: the same kernels are HBM-bound on their saved-activation traffic."""
        kern = m['kern']
        P_main, P_eik, P_smp = N_RAYS * 98, 4 * N_RAYS, N_RAYS * 128
        F = sdf_macs_per_point()
        flops = {   # algorithmic FLOPs per launch (2 FLOP / MAC), SURVEY.md 8(d) multipliers
            # no-grad forward of the sampler: get_sdf_vals needs the sdf ROW of the output layer only (256 of its
            # 256 x 257 MACs) -- rounds 1-3 counted the whole layer here, as SURVEY's 1 x F_sdf does: 12.5 % too much
            'msdf_sdf_forward_if': 2.0 * (F - 256 * 256) * P_smp,
            'msdf_sdf_fwd_grad': 2.0 * 2 * F * (P_main + P_eik),   # forward + d/dx sweep
            'msdf_sdf_backward': 2.0 * 2 * F * (P_main + P_eik),   # p-bar = W q-bar and h-bar = W^T a-bar sweeps
        }
        hs, qs, ab = m['slots']
        P = P_main + P_eik
        hbm = {     # algorithmic bytes per launch: 4 B x slots read or written per point
            'msdf_sdf_fwd_grad': 4.0 * P * (3 * hs),               # H written, H re-read, PM written
            'msdf_sdf_backward': 4.0 * P * (4 * hs + qs + ab),     # H x2, PM, q-bar rows read; QB, AB written
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
        return world * N_RAYS * m['steps'] / m['dt']

    grid_only = args.config == 'grid'
    primary = measure(args.precision, grid=grid_only, beta=args.beta)
    single = world == 1 and not use_dist
    extras = args.extras and single and not grid_only
    sharp = measure(args.precision, beta=0.01) if extras else None
    alt = measure('bf16x3' if args.precision == 'fp32' else 'fp32') if (extras and args.alt_precision) else None
    alt6 = measure('bf16x6') if (extras and args.alt_precision and args.precision == 'fp32') else None
    # the hash-grid configuration: beside the headline at N = 1, and at N > 1 for its gradient exchange
    grid = measure('fp32', grid=True) if (extras or (use_dist and args.extras and not grid_only)) else None
    # density beta at the start of the `sustained` training run: just above the value at which this random-init network's
    # sampler starts to need a third round (all batches take 2 rounds at 0.014, 2-5 at 0.010), so that the falling beta
    # of the run crosses it (from the confs' 0.1 a 400-step window would stay at one round)
    SUSTAINED_BETA0 = 0.0125
    sustained = measure(args.precision, beta=SUSTAINED_BETA0, steps=args.sustained_steps, fresh=True, train=True) \
        if (extras and args.sustained_steps > 0) else None

    if rank == 0:
        dtype = {'fp32': 'f32', 'bf16x3': 'bf16x3 (fp32 split into 2 bf16, 3 products, fp32 accumulate)',
                 'bf16x6': 'bf16x6 (fp32 split into 3 bf16, 6 products, fp32 accumulate)'}[args.precision]
        if grid_only:
            res = grid_report(args, primary['kern'], primary['dt'], world, primary['sampler']['mean_rounds'],
                              primary['loss'], primary['sampler'], primary['backward_alone_ms'])
            res['dtype'] = dtype
            if primary['multi'] is not None:
                res['multi_gpu'] = primary['multi']
            print(json.dumps(res))
            if use_dist:
                dist.destroy_process_group()
            return 0
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
        if primary['multi'] is not None:
            res['multi_gpu'] = dict(primary['multi'], backend='nccl (RCCL)',
                                    note='the overlap of the hash-grid table all-reduce with the weight-gradient kernels is '
                                         'a design until a SCALE record of this line exists: `allreduce_ms_per_step` is '
                                         'the time between HIP events around each message, not its exposed part')
            if grid is not None:
                res['multi_gpu']['hash_grid'] = dict(grid['multi'], value=rate(grid), unit='rays/s',
                                                     ms_per_step=1e3 * grid['dt'] / grid['steps'])
        if sharp is not None:
            smf, shb = mlp_rooflines(sharp)
            res['sharp_state'] = {
                'what': 'the same training step with density beta = 0.01 (SURVEY 8(d) "sharpened" state): the sampler '
                        'needs 2+ rounds, each one more SDF evaluation of 131,072 points',
                'value': rate(sharp), 'unit': 'rays/s', 'ms_per_step': 1e3 * sharp['dt'] / args.steps,
                'sampler': sharp['sampler'], 'roofline': smf if args.precision == 'fp32' else (shb or smf),
                'kernels_ms_per_step': ms(sharp)}
        if sustained is not None:
            wins, prev_n, prev_t = [], 0, 0.0
            seq = sustained['rounds_seq']
            for n, t, b in sustained['windows']:
                wins.append({'steps': '%d-%d' % (prev_n + 1, n), 'ms_per_step': 1e3 * (t - prev_t) / (n - prev_n),
                             'mean_rounds': float(np.mean(seq[prev_n:n])), 'beta_at_end': b})
                prev_n, prev_t = n, t
            res['sustained'] = {
                'what': 'a TRAINING run of %d consecutive steps on the configs[1] network: forward + fused MonoSDFLoss '
                        '(weights of scannet_mlp.conf) against the targets of a closed-form scene (a sphere seen from '
                        'inside, the scene of tests/golden/traj_*.npz) + backward + Adam (lr 5e-4), a FRESH ray batch every '
                        'step, from the random-init weights with density beta = %g (just above where this network\'s sampler starts '
                        'to need a third round; the reference\'s confs start at 0.1 and reach such values after tens of '
                        'thousands of steps): beta falls, the rounds per step go up, the round-count guess misses now and '
                        'then.  No warm-up: the first window contains the first step' % (
                            sustained['steps'], SUSTAINED_BETA0),
                'value': rate(sustained), 'unit': 'rays/s', 'steps': sustained['steps'],
                'ms_per_step': 1e3 * sustained['dt'] / sustained['steps'],
                'ms_per_step_first_100': wins[0]['ms_per_step'] if wins else None,
                'ms_per_step_last_100': wins[-1]['ms_per_step'] if wins else None,
                'windows': wins, 'seconds': sustained['dt'], 'sampler': sustained['sampler'],
                'beta_start': SUSTAINED_BETA0 + 1e-4, 'beta_end': sustained['beta_end'],
                'loss_first_step': sustained['first_loss'], 'loss_last_step': sustained['loss'],
                'kernels_ms_per_step': ms(sustained)}
        if grid is not None and single:
            g = grid_report(args, grid['kern'], grid['dt'], world, grid['sampler']['mean_rounds'], grid['loss'],
                            grid['sampler'], grid['backward_alone_ms'])
            res['hash_grid'] = {k: g[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'config', 'roofline',
                                                  'mlp_roofline', 'hash_forward_vs_gather_ceiling', 'hash_entry_points',
                                                  'kernels_ms_per_step')}
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
        if alt6 is not None:
            a6mf, a6hb = mlp_rooflines(alt6)
            res['alt_matrix_core_x6'] = {
                'matrix_core': 'bf16x6', 'value': rate(alt6), 'unit': 'rays/s',
                'ms_per_step': 1e3 * alt6['dt'] / args.steps, 'roofline': a6hb or a6mf,
                'kernels_ms_per_step': ms(alt6),
                'note': 'same workload, steps and warm-up with every operand of the fused MLP kernels split into THREE bf16 '
                        'planes and six products per term (fp32 accumulation): fp32-grade results -- its parity tests are '
                        'held to the fp32 core\'s rows of the frozen tolerance table -- on the bf16 matrix cores; weight '
                        'gradients on the fp32 kernel.  Opt-in; `value` above is the fp32 MFMA core',
            }
        if single and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
