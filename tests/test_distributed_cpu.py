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
