"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: ray sharding, the single flat gradient
all-reduce, and the all-gather of rendered rows.  The renderer used here is the CPU oracle -- the
point is the distributed logic, which is device independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from monosdf_amd import parallel
        from oracle import config, monosdf_oracle as mo, synth
        torch.set_num_threads(2)
        conf = config.mlp_config(64)
        state = synth.make_state(conf, seed=0, jitter=0.2)
        n = 13                                             # ragged on purpose
        rays = synth.make_rays(n, seed=1, random_pose=True)
        lo, hi = parallel.shard_slice(n)
        mine = {k: v[lo:hi] for k, v in rays.items()}
        # eval-mode render of this rank's rays, gathered
        out = mo.render(state, conf, mine, torch.arange(lo, hi), True, False, None)
        rows = torch.cat([out['rgb_values'], out['depth_values'], out['normal_map']], 1)
        full = parallel.all_gather_rows(rows.detach())
        # gradient averaging: loss over the local rays, mean over ranks == weighted single-process result
        st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
        o2 = mo.render(st, conf, mine, torch.arange(lo, hi), True, False, None)
        o2['rgb_values'].sum().backward()
        params = [torch.nn.Parameter(v.detach().clone()) for v in st.values() if v.grad is not None]
        for p, v in zip(params, [v for v in st.values() if v.grad is not None]):
            p.grad = v.grad.clone()
        parallel.average_gradients(params)
        if rank == 0:
            ref = mo.render(state, conf, rays, torch.arange(n), True, False, None)
            ref_rows = torch.cat([ref['rgb_values'], ref['depth_values'], ref['normal_map']], 1)
            st1 = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
            r1 = mo.render(st1, conf, rays, torch.arange(n), True, False, None)
            r1['rgb_values'].sum().backward()
            g_single = [v.grad / world for v in st1.values() if v.grad is not None]
            err_rows = (full - ref_rows).abs().max().item()
            err_grad = max(((a.grad - b).abs().max() / (b.abs().max() + 1e-12)).item()
                           for a, b in zip(params, g_single))
            q.put((full.shape[0], err_rows, err_grad))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gradient_average():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_rows, err_rows, err_grad = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n_rows == 13
    # each ray is independent given the same sampler round count (both shards converge in one round here)
    assert err_rows < 1e-4
    assert err_grad < 1e-4


def test_shard_slice_is_a_partition():
    from monosdf_amd import parallel
    for n in (0, 1, 7, 144, 1024):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_slice(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _run_workers(target, world=2, timeout=300, args=()):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    t0, res = time.time(), None
    while res is None:
        try:
            res = q.get(timeout=1.0)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > timeout:
                for p in procs:
                    p.kill()
                raise AssertionError('a rank failed (exit codes %r)' % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _setup(rank, world, port):
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.set_num_threads(2)


class _OracleImageModel:
    """Stands in for MonoSDFNetwork in image mode (the distributed drivers only call it and toggle .training)."""

    def __init__(self, state, conf):
        self.state, self.conf, self.training, self.calls = state, conf, True, []

    def eval(self):
        self.training = False

    def train(self, mode=True):
        self.training = mode

    def __call__(self, inputs, indices):
        from oracle import monosdf_oracle as mo
        assert not self.training
        self.calls.append(inputs['uv'].shape[1])
        with torch.enable_grad():        # the oracle takes d sdf / dx by autograd; the HIP model has it in-kernel
            out = mo.render(self.state, self.conf, inputs, indices, False, False, None)
        return {k: v.detach() for k, v in out.items()}


def _render_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from monosdf_amd.utils import render
        from oracle import config, monosdf_oracle as mo, synth
        conf = config.mlp_config(64)
        state = synth.make_state(conf, seed=0, jitter=0.2)
        total, split = 73, 10                       # 8 chunks, the last one ragged (3 pixels) and on rank 1
        g = torch.Generator().manual_seed(5)
        uv = torch.rand(1, total, 2, generator=g) * 384
        intr = torch.eye(4)[None].clone()
        intr[0, 0, 0], intr[0, 1, 1], intr[0, 0, 2], intr[0, 1, 2] = 300., 310., 192., 190.
        pose = torch.eye(4)[None].clone()
        pose[0, :3, 3] = torch.tensor([0.1, -0.15, 0.05])
        inputs = {'uv': uv, 'pose': pose, 'intrinsics': intr}
        model = _OracleImageModel(state, conf)
        out = render.render_image(model, inputs, torch.tensor([3]), total, split_n_pixels=split)
        assert model.training                       # mode restored
        # chunk c -> rank c mod world: this rank rendered exactly its chunks
        sizes = [min(split, total - c * split) for c in range((total + split - 1) // split)]
        assert model.calls == [sizes[c] for c in range(rank, len(sizes), world)]
        ref = mo.render_image(state, conf, inputs, torch.tensor([3]), total, split_n_pixels=split)
        err = max((out[k] - ref[k]).abs().max().item() for k in ref)
        shapes_ok = all(out[k].shape == ref[k].shape for k in ref)
        gathered = [None] * world
        dist.all_gather_object(gathered, (err, shapes_ok))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_render_image_two_ranks_equals_single_process():
    """configs[3] plumbing: chunks dealt round-robin over 2 ranks (ragged last chunk), one all-gather, rows put
    back in pixel order -- every rank ends with the image a single process renders."""
    for err, shapes_ok in _run_workers(_render_worker):
        assert shapes_ok and err == 0.0


def _volume_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from monosdf_amd.utils import render
        from oracle import config, monosdf_oracle as mo, synth
        conf = config.mlp_config(64)
        state = synth.make_state(conf, seed=0, jitter=0.3)
        calls = []

        def sdf_fn(p):
            calls.append(p.shape[0])
            with torch.no_grad():
                return mo.sdf_network_raw(state, conf, p)[:, 0]
        blocks = list(render.sdf_volume(sdf_fn, resolution=128, grid_boundary=(-1.1, 1.1), device='cpu', shard=True))
        assert len(blocks) == 1
        vol = torch.from_numpy(blocks[0][2])
        n_eval = sum(calls)
        if rank == 0:
            ref_calls = []

            def ref_fn(p):
                ref_calls.append(p.shape[0])
                with torch.no_grad():
                    return mo.sdf_network_raw(state, conf, p)[:, 0]
            ref = mo.sdf_volume_block(ref_fn, (-1.1,) * 3, (1.1,) * 3, 128)
            q.put((bool(torch.equal(vol, ref)), n_eval, sum(ref_calls)))
    finally:
        dist.destroy_process_group()


def test_sdf_volume_sharded_equals_single_process():
    """configs[4] plumbing: every pyramid level's (masked) point list is cut evenly over the ranks and the values
    are all-gathered; the volume equals the single-process one bit for bit, at half the evaluations per rank."""
    same, n_rank0, n_single = _run_workers(_volume_worker)
    assert same
    assert abs(n_rank0 - n_single / 2) <= 4         # one point per level of imbalance at most


def _rounds_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)))
        import sys
        sys.path.insert(0, sys_path)
        from helpers import Case
        from monosdf_amd import parallel
        from oracle import monosdf_oracle as mo
        c = Case('mlp_w64_eval_k5')                  # the whole batch needs 5 rounds, its second half alone 3
        rays = c.inputs
        n = rays['ray_dirs'].shape[0]
        lo, hi = parallel.shard_slice(n)
        d, o = rays['ray_dirs'][lo:hi], rays['ray_cam_loc'][lo:hi]

        def all_max(t):
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t
        alone, together = {}, {}
        z_alone, _ = mo.error_bound_sampler(c.state, c.conf, d, o, False, None, trace=alone)
        z_glob, _ = mo.error_bound_sampler(c.state, c.conf, d, o, False, None, trace=together, max_reduce=all_max)
        full = parallel.all_gather_rows(z_glob)
        info = [None] * world
        dist.all_gather_object(info, (alone['rounds'], together['rounds']))
        if rank == 0:
            q.put((info, bool(torch.equal(full, c.out['z_vals'])), c.rounds))
    finally:
        dist.destroy_process_group()


def test_global_round_decision_reproduces_single_process_rounds():
    """SURVEY 8(e), strong scaling: with the max beta all-reduced (MAX) before each round's convergence test, every
    shard runs the rounds the whole batch runs on one process and the gathered z equal the reference's z for the
    whole batch; left alone, a shard stops as soon as ITS rays have converged."""
    info, same, rounds = _run_workers(_rounds_worker)
    assert all(t == rounds for _, t in info)
    assert same
    assert any(a < rounds for a, _ in info)         # the protocol matters: some shard would have stopped early


def _averager_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from monosdf_amd import parallel
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)),
                  torch.nn.Parameter(torch.randn(2, 2), requires_grad=False), torch.nn.Parameter(torch.randn(4))]
        avg = parallel.GradientAverager(params)
        ok = True
        for step in range(3):
            for p in params:
                p.grad = None
            params[0].grad = torch.full((5, 3), float(rank + 1 + step))
            if rank == 0:                            # this parameter has a gradient on one rank only
                params[1].grad = torch.full((7,), 2.0)
            flat = avg.average()
            ok &= bool(torch.all(params[0].grad == (1 + 2) / 2 + step))
            ok &= bool(torch.all(params[1].grad == 1.0))
            ok &= params[2].grad is None and bool(torch.all(params[3].grad == 0))
            ok &= params[0].grad.data_ptr() == flat.data_ptr()          # views of the persistent buffer
        res = [None] * world
        dist.all_gather_object(res, ok)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


def test_gradient_averager_fixed_layout_and_persistent_buffer():
    """same layout on every rank even when a gradient is missing on one of them; no per-step buffer."""
    assert all(_run_workers(_averager_worker))


class _TableGrad(torch.autograd.Function):
    """Stands in for ops.GridSdfFunction: hands autograd a freshly allocated table gradient."""

    @staticmethod
    def forward(ctx, table, scale):
        ctx.scale, ctx.shape = scale, table.shape
        return table.sum().reshape(1) * scale

    @staticmethod
    def backward(ctx, g):
        out = torch.full(ctx.shape, float(ctx.scale)) * g
        _TableGrad.last_ptr = out.data_ptr()
        return out, None


def _overlap_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from monosdf_amd import parallel
        torch.manual_seed(0)
        table = torch.nn.Parameter(torch.zeros(4096, 2))           # the "large" parameter: its own, early message
        w = torch.nn.Parameter(torch.randn(6, 3))
        b = torch.nn.Parameter(torch.randn(6))
        avg = parallel.GradientAverager([table, w, b], overlap_min_numel=4096)
        ok = len(avg.big) == 1 and avg.flat.numel() == 24 and parallel._LISTENERS == 1
        for step in range(3):
            for p in (table, w, b):
                p.grad = None
            fired = len(avg._inflight)
            loss = _TableGrad.apply(table, float(rank + 1 + step)).sum() + (w.sum() + b.sum()) * (rank + 1)
            if not (rank == 1 and step == 2):          # one rank skips the table in one step: reduced as zeros
                loss.backward()
                ok &= len(avg._inflight) == fired + 1              # the hook started the exchange during backward
                ok &= table.grad.data_ptr() == _TableGrad.last_ptr  # autograd adopted the tensor, no 48.8 MB copy
            else:
                ((w.sum() + b.sum()) * (rank + 1)).backward()
            avg.average()
            want = ((1 + step) + (2 + step)) / 2.0 if step < 2 else (1 + step) / 2.0
            ok &= bool(torch.all(table.grad == want))
            ok &= bool(torch.all(w.grad == 1.5)) and bool(torch.all(b.grad == 1.5))
            ok &= w.grad.data_ptr() == avg.flat.data_ptr() and not avg._inflight
        avg.close()
        ok &= parallel._LISTENERS == 0
        res = [None] * world
        dist.all_gather_object(res, ok)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


def test_gradient_averager_two_blocks_table_gradient_leaves_during_backward():
    """the table gradient is its own message, started from a post-accumulate hook (before
    average() is called), the MLP block stays one flat message; a rank without a table gradient sends zeros."""
    assert all(_run_workers(_overlap_worker))


def _gather_sizes_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from monosdf_amd import parallel
        rows = torch.arange(3 * (rank + 2), dtype=torch.float32).reshape(rank + 2, 3) + 100 * rank
        known = parallel.all_gather_rows(rows, sizes=[2, 3])
        asked = parallel.all_gather_rows(rows)
        try:
            parallel.all_gather_rows(rows, sizes=[5, 5])
            raised = False
        except ValueError:
            raised = True
        if rank == 0:
            q.put((known.shape[0], bool(torch.equal(known, asked)), float(known[2, 0]), raised))
    finally:
        dist.destroy_process_group()


def test_all_gather_rows_with_known_sizes_needs_no_size_exchange():
    n, same, first_of_rank1, raised = _run_workers(_gather_sizes_worker)
    assert n == 5 and same and first_of_rank1 == 100.0 and raised


# ---------------------------------------------------------------------------------------------------------------
# bench.py --gpus N: the parent starts N ranks (before any GPU call) or fails -- never a silent one-GPU run
# ---------------------------------------------------------------------------------------------------------------
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(argv), env=e, capture_output=True,
                          text=True, timeout=600)


@pytest.mark.parametrize('n', [2, 3, 8])
def test_bench_launcher_starts_n_ranks(n):
    import json
    r = _bench('--gpus', str(n), '--steps', '2', '--dry-run')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, nothing else on stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == n and line['dry_run'] is True and line['value'] is None
    assert line['multi_gpu']['ranks_answering'] == n and line['multi_gpu']['gradient_mean_correct']


def test_bench_launcher_refuses_fewer_devices_than_ranks():
    """No GPU in this container: --gpus 2 without --dry-run must fail, not run one rank and print n_gpus 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip('two GPUs visible')
    r = _bench('--gpus', '2', '--steps', '1')
    assert r.returncode != 0 and 'GPU(s) visible' in r.stderr and r.stdout.strip() == ''


def test_bench_refuses_world_size_mismatch():
    r = _bench('--gpus', '4', '--steps', '1', '--dry-run', env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'launcher started 1 rank' in r.stderr and r.stdout.strip() == ''


def test_bench_rank_that_does_not_arrive_is_named_quickly():
    """A rank that never comes up must not hold the others for torch.distributed's default 10 minutes (the driver's
    whole limit): with a 4-second rendezvous limit, rank 0 of a two-rank job whose rank 1 is never started gives up,
    names rank 1 on its (rank-prefixed) stderr and exits non-zero without printing a line."""
    import socket
    import time
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    t0 = time.time()
    r = _bench('--gpus', '2', '--steps', '1', '--dry-run',
               env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1',
                    'MASTER_PORT': str(port), 'MSDF_RDZV_TIMEOUT': '4'})
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert 'rank(s) 1 of 2 did not arrive' in r.stderr and '[rank 0] ' in r.stderr
    assert r.stdout.strip() == ''
    assert time.time() - t0 < 120
