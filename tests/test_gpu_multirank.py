"""Two ranks on the one GPU of the test box (gloo rendezvous; the data stay on the GPU): the parts of the
multi-GPU path that need real kernels -- the batch-global sampler decision by all-reduce(MAX) (SURVEY 8(e), strong
scaling), the sharded image render and the sharded SDF volume -- against the single-process results."""
import os
import queue
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        from helpers import Case
        from monosdf_amd import parallel
        from monosdf_amd.conf import ConfigTree
        from monosdf_amd.model.network import MonoSDFNetwork
        from monosdf_amd.utils import render
        res = {}

        def build(case):
            m = MonoSDFNetwork(ConfigTree.from_dict(case.conf))
            m.load_state_dict({k: v.clone() for k, v in case.state.items()}, strict=True)
            return m.cuda().eval()

        # 1. strong scaling of one batch: the whole batch needs 5 rounds, its second half alone 3
        c = Case('mlp_w64_eval_k5')
        m = build(c)
        n = c.inputs['ray_dirs'].shape[0]
        lo, hi = parallel.shard_slice(n)
        mine = {k: v[lo:hi].cuda() for k, v in c.inputs.items()}
        idx = c.indices[lo:hi].cuda()
        rounds = {}
        for mode, spec in (('sync', False), ('speculative', True)):
            m.speculate_rounds = spec
            m.ray_sampler.global_rounds = None
            out = m(mine, idx, if_pixel_input=True)
            rounds[mode + '_alone'] = m.ray_sampler.last_rounds
            m.ray_sampler.global_rounds = True
            out = m(mine, idx, if_pixel_input=True)
            rounds[mode + '_global'] = m.ray_sampler.last_rounds
            rows = torch.cat([out['z_vals'], out['rgb_values'], out['depth_values'], out['normal_map']], 1)
            res[mode] = parallel.all_gather_rows(rows.detach()).cpu()
        # the number of rounds (= collectives) a rank enqueues for the next group call comes from the all-reduced
        # decisions alone: the same on every rank, whatever each rank's own calls needed in between
        m.ray_sampler.global_rounds = True
        res['global_guess'] = (m.ray_sampler.guess_rounds(), list(m.ray_sampler._hist[1]))
        res['own_history'] = list(m.ray_sampler._hist[0])
        m.ray_sampler.global_rounds = None
        m.speculate_rounds = False
        full = m({k: v.cuda() for k, v in c.inputs.items()}, c.indices.cuda(), if_pixel_input=True)
        res['single'] = torch.cat([full['z_vals'], full['rgb_values'], full['depth_values'], full['normal_map']], 1).cpu()
        res['single_rounds'] = m.ray_sampler.last_rounds
        res['rounds'] = rounds

        # 2. sharded image render (configs[3] plumbing on the HIP model): 8 chunks, ragged last one
        ci = Case('mlp_w64_image_eval')
        mi = build(ci)
        total, split = 73, 10
        g = torch.Generator().manual_seed(5)
        inputs = {'uv': (torch.rand(1, total, 2, generator=g) * 384).cuda(), 'pose': ci.inputs['pose'].cuda(),
                  'intrinsics': ci.inputs['intrinsics'].cuda()}
        img = render.render_image(mi, inputs, ci.indices.cuda(), total, split_n_pixels=split)
        res['image'] = {k: v.cpu() for k, v in img.items()}
        # single-process reference: the same loop without the process group's help
        chunks = render.split_input(inputs, total, split)
        with torch.no_grad():
            outs = [mi(ch, ci.indices.cuda()) for ch in chunks]
        res['image_single'] = {k: torch.cat([o[k].reshape(o[k].shape[0], -1) for o in outs], 0).cpu() for k in img}

        # 3. sharded SDF volume (configs[4] plumbing on the HIP model)
        with torch.no_grad():
            fn = lambda p: m.implicit_network(p)[:, 0]
            vol = list(render.sdf_volume(fn, resolution=128, grid_boundary=(-1.1, 1.1), shard=True))[0][2]
            vol1 = list(render.sdf_volume(fn, resolution=128, grid_boundary=(-1.1, 1.1), shard=False))[0][2]
        res['volume_equal'] = bool((vol == vol1).all())
        # 4. training step of the hash-grid model, every rank on its own half of the rays: the table gradient leaves as
        # its own message from the post-accumulate hook (behind the scatter, before the weight-gradient kernels), the
        # MLP block as one flat message; the result is the mean of the two ranks' single-process gradients
        cg = Case('grid_small_train')
        mg = build(cg).train()
        ng = cg.inputs['ray_dirs'].shape[0]
        half = ng // 2

        def grads_of(lo_, hi_):
            for p_ in mg.parameters():
                p_.grad = None
            mg._noise = None
            torch.manual_seed(7)
            o = mg({k: v[lo_:hi_].cuda() for k, v in cg.inputs.items()}, cg.indices[lo_:hi_].cuda(), if_pixel_input=True)
            loss_ = o['rgb_values'].sum() + o['normal_map'].sum() + 0.1 * (o['grad_theta'].norm(dim=1) - 1).pow(2).sum()
            loss_.backward()
        named = [(n_, p_) for n_, p_ in mg.named_parameters() if p_.requires_grad]
        # without a listening averager: this rank's own gradients of both halves -> the expected mean
        want = {}
        for part, (lo_, hi_) in enumerate(((0, half), (half, 2 * half))):
            grads_of(lo_, hi_)
            for n_, p_ in named:
                want[n_] = want.get(n_, 0) + p_.grad.detach().clone() / 2
        avg = parallel.GradientAverager([p_ for _, p_ in named], overlap_min_numel=8000, timing=True)
        res['avg_big'] = [n_ for n_, p_ in named if any(p_ is b_ for b_ in avg.big)]
        grads_of(rank * half, (rank + 1) * half)
        res['avg_started_in_backward'] = len(avg._inflight)
        avg.average()
        torch.cuda.synchronize()
        res['avg_err'] = max(float((p_.grad - want[n_]).abs().max() / (want[n_].abs().max() + 1e-20)) for n_, p_ in named)
        res['avg_overlapped_ms'] = [a_.elapsed_time(b_) for a_, b_ in avg.timings['overlapped']]
        avg.close()
        # 5. the table feeds TWO nodes of one backward pass (the network evaluated on two point sets): autograd adds the
        # second scatter's result into the first one's tensor in place -- same address, which carries the first scatter's
        # "complete" mark.  The mark must not be trusted then (parallel.grad_ready_event: tensor identity + version).
        net = mg.implicit_network
        gen = torch.Generator().manual_seed(3)
        xa, xb = (torch.rand(2, 2, 150, 3, generator=gen) * 1.6 - 0.8).cuda().unbind(0)

        def twice(r_):
            for p_ in mg.parameters():
                p_.grad = None
            sa, fa, ga = net.get_outputs(xa[r_])
            sb, fb, gb = net.get_outputs(xb[r_])
            (sa.sum() + 0.1 * fa.sum() + ga.sum() + 2.0 * sb.sum() + 0.3 * gb.pow(2).sum()).backward()
        want2 = {}
        for r_ in range(2):
            twice(r_)
            for n_, p_ in named:
                if p_.grad is not None:
                    want2[n_] = want2.get(n_, 0) + p_.grad.detach().clone() / 2
        seen = []
        orig_ready = parallel.grad_ready_event
        parallel.grad_ready_event = lambda g_: seen.append(orig_ready(g_)) or seen[-1]
        avg2 = parallel.GradientAverager([p_ for n_, p_ in named if n_ in want2], overlap_min_numel=8000)
        twice(rank)
        avg2.average()
        torch.cuda.synchronize()
        parallel.grad_ready_event = orig_ready
        res['twice_err'] = max(float((p_.grad - want2[n_]).abs().max() / (want2[n_].abs().max() + 1e-20))
                               for n_, p_ in named if n_ in want2)
        res['twice_marks_trusted'] = [e_ is not None for e_ in seen]
        res['twice_marks_left'] = len(parallel._GRAD_READY)
        avg2.close()
        # plain numpy in the queue: a tensor would travel as a shared-memory handle that dies with this process
        def plain(v):
            if torch.is_tensor(v):
                return v.detach().numpy()
            return {k: plain(x) for k, x in v.items()} if isinstance(v, dict) else v
        res = plain(res)
        allres = [None] * world
        dist.all_gather_object(allres, res)
        if rank == 0:
            q.put(allres)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    t0, allres = time.time(), None
    while allres is None:
        try:
            allres = q.get(timeout=1.0)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 420:
                for p in procs:
                    p.kill()
                raise AssertionError('a rank failed (exit codes %r)' % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for res in allres:
        assert res['single_rounds'] == 5
        # with the all-reduced decision every shard runs the batch's 5 rounds, and the gathered rows ARE the
        # single-process rows (bit for bit), whether the rounds were speculated or read back
        for mode in ('sync', 'speculative'):
            assert res['rounds'][mode + '_global'] == 5, res['rounds']
            assert np.array_equal(res[mode], res['single']), mode
        for k in res['image']:
            assert np.array_equal(res['image'][k], res['image_single'][k]), k
        assert res['volume_equal']
        assert res['avg_big'] == ['implicit_network.encoding.embeddings']
        assert res['avg_started_in_backward'] == 1 and len(res['avg_overlapped_ms']) == 1
        assert res['avg_err'] < 2e-6, res['avg_err']
        # two nodes, one table: one hook call, its tensor was written after the first mark -> not trusted; nothing left over
        assert res['twice_marks_trusted'] == [False], res['twice_marks_trusted']
        assert res['twice_marks_left'] == 0
        assert res['twice_err'] < 2e-6, res['twice_err']
    assert allres[0]['global_guess'] == allres[1]['global_guess'] and allres[0]['global_guess'][0] == 5
    assert allres[0]['own_history'] != allres[1]['own_history']          # each rank's own calls differ (3 vs 5 rounds)
    # left alone, the second shard stops after its own 3 rounds (what makes the all-reduce necessary)
    assert sorted(r['rounds']['sync_alone'] for r in allres) == [3, 5]
