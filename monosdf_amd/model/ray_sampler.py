"""Ray samplers (reference: code/model/ray_sampler.py).  ``UniformSampler.get_z_vals`` and
``ErrorBoundSampler.get_z_vals`` keep the reference's signatures; the algorithms run in csrc/sampler.hip
(one wave per ray) with the SDF evaluations done by the fused forward kernel.

The batch-global convergence test (``beta.max() > beta0``, reference ray_sampler.py:179) is evaluated on the
device: the kernels of a round return at once when the previous round did not ask for them, so the host may
enqueue rounds without reading the flags back (``speculate``), and only has to check afterwards that it
enqueued enough of them."""
import ctypes as C

import torch

from .. import _lib


class RaySampler:
    def __init__(self, near, far):
        self.near, self.far = near, far


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('monosdf_amd: the samplers run on the GPU only (no CPU fallback)')
    return t.detach().float().contiguous()


class UniformSampler(RaySampler):
    """reference ray_sampler.py:16-83: N_samples equidistant depths between near and far (far = the exit of the
    cube [-R, R]^3 clipped to 2 R 1.75 with take_sphere_intersection, else that constant), stratified jitter in
    training mode.  Returns (z_vals [N, N_samples], near [N,1], far [N,1])."""

    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1):
        super().__init__(near, 2.0 * scene_bounding_sphere * 1.75 if far == -1 else far)
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection

    def get_z_vals(self, ray_dirs, cam_loc, model, jitter=None):
        ray_dirs, cam_loc = _need_gpu(ray_dirs), _need_gpu(cam_loc)
        dev = ray_dirs.device
        N, n = ray_dirs.shape[0], self.N_samples
        f32 = dict(device=dev, dtype=torch.float32)
        if model.training and jitter is None:
            jitter = torch.rand(N, n, **f32)
        z = torch.empty(N, n, **f32)
        far = torch.empty(N, 1, **f32)
        scratch = torch.empty(N * n * 5 + N, **f32)        # new_z, pts, beta of the shared kernel: not returned
        a = _lib.SamplerArgs()
        a.ray_o, a.ray_d, a.N = cam_loc.data_ptr(), ray_dirs.data_ptr(), N
        a.M, a.m_max, a.n_eval, a.n_final, a.n_extra = n, n, n, 0, 0
        a.max_rounds = 1
        a.near, a.far = float(self.near), float(self.far)
        a.bound = float(self.scene_bounding_sphere) if self.take_sphere_intersection else 0.0
        a.lemma = 1.0
        a.z, a.new_z = z.data_ptr(), scratch.data_ptr()
        a.new_pos = scratch.data_ptr() + 4 * N * n
        a.pts, a.beta = scratch.data_ptr() + 8 * N * n, scratch.data_ptr() + 20 * N * n
        a.jitter = _need_gpu(jitter).data_ptr() if jitter is not None else None
        a.far_out = far.data_ptr()
        _lib.call('msdf_sampler_init', C.byref(a), _lib.stream_ptr())
        self._keep = (jitter, scratch)
        return z, torch.full((N, 1), float(self.near), **f32), far


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=1.0e-6):
        super().__init__(near, 2.0 * scene_bounding_sphere * 1.75)
        if inverse_sphere_bg:
            raise NotImplementedError('monosdf_amd: inverse_sphere_bg is not used by any conf of the reference fork')
        self.N_samples = N_samples
        self.N_samples_eval = N_samples_eval
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval,
                                              take_sphere_intersection=True)
        self.N_samples_extra = N_samples_extra
        self.eps = eps
        self.beta_iters = beta_iters
        self.max_total_iters = max_total_iters
        self.scene_bounding_sphere = scene_bounding_sphere
        self.add_tiny = add_tiny
        self._last_rounds = 0
        # rounds of the last HISTORY calls and the calls left in which all max_total_iters rounds are enqueued (see
        # guess_rounds) -- kept twice: [0] for this rank's own calls, [1] for the calls whose round decision was
        # all-reduced over a rank group (global_rounds).  The second one is a function of the all-reduced flags alone,
        # so it is the same on every rank of the group, and so is the number of rounds (= collectives) guessed from it
        self._hist = [[], []]
        self._miss = [0, 0]
        self._pending = None
        self._eval_columns = {}
        # speculation bookkeeping (bench.py reports it): calls, passes repeated because too few rounds were
        # enqueued, rounds enqueued beyond the ones that ran
        self.stats = {'calls': 0, 'repeats': 0, 'idle_rounds': 0}
        # a torch.distributed group (or True = the default group): the convergence test takes the maximum beta over
        # all ranks, so that a batch split over ranks runs the rounds the whole batch would run on one GPU
        # (SURVEY 8(e)); None = every rank decides for its own rays, as under the reference's DDP
        self.global_rounds = None
        # Lemma-2 constant, formed in fp32 like the reference does (ray_sampler.py:119)
        self._lemma = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))

    def get_z_vals(self, ray_dirs, cam_loc, model):
        z, z_eik, _ = self.sample(ray_dirs, cam_loc, model, want_points=False)
        return z, z_eik

    def get_error_bound(self, beta, model, sdf, z_vals, dists, d_star):
        """reference ray_sampler.py:264-272 (dists is z_vals' difference, recomputed in the kernel)."""
        z = _need_gpu(z_vals)
        N, M = z.shape
        sdf = _need_gpu(sdf).reshape(N, M)
        d_star = _need_gpu(d_star).reshape(N, M - 1)
        beta = _need_gpu(beta if torch.is_tensor(beta) else torch.tensor(beta, device=z.device)).reshape(-1)
        if beta.numel() not in (1, N):
            raise RuntimeError('monosdf_amd: beta must hold one value or one per ray')
        out = torch.empty(N, device=z.device, dtype=torch.float32)
        _lib.call('msdf_sampler_error_bound', _lib.ptr(z), _lib.ptr(sdf), _lib.ptr(d_star), _lib.ptr(beta),
                  int(beta.numel() == N and N > 1), N, M, _lib.ptr(out), _lib.stream_ptr())
        return out

    # -- the columns of the dense sample set that join the final set (ray_sampler.py:242-247) ------------------
    def _extra_columns(self, training, noise, dev):
        """[max_total_iters, N_samples_extra] int64 on the device: row k-1 is used when k rounds ran.  Training:
        a random subset per size, drawn on the CPU generator as the reference does; eval: its linspace."""
        n_extra, n_eval, K = self.N_samples_extra, self.N_samples_eval, self.max_total_iters
        if n_extra <= 0:
            return torch.zeros(K, 1, device=dev, dtype=torch.int64)
        if training:
            given = noise.get('extra_idx')
            if given is not None:
                given = given.to(device=dev, dtype=torch.int64)
                if given.dim() == 2:   # a test's table, one row per possible size
                    return given.contiguous()
                # a test's draw for the size the loop will end with: the same row for all sizes
                return given.reshape(1, n_extra).expand(K, n_extra).contiguous()
            host = torch.empty(K, n_extra, dtype=torch.int64, pin_memory=True)
            for k in range(K):
                host[k] = torch.randperm(n_eval * (k + 1))[:n_extra]
            return host.to(dev, non_blocking=True)
        key = (str(dev), n_eval, K, n_extra)
        if key not in self._eval_columns:
            rows = [torch.linspace(0, n_eval * (k + 1) - 1, n_extra).long() for k in range(K)]
            self._eval_columns[key] = torch.stack(rows).to(dev)
        return self._eval_columns[key]

    @property
    def last_rounds(self):
        """Rounds the last call ran (waits for its flags if they have not been read back yet)."""
        self._resolve()
        return self._last_rounds

    @last_rounds.setter
    def last_rounds(self, value):
        self._resolve()
        self._last_rounds = int(value)

    def _resolve(self):
        """Read the flags of the last speculative call (copied to pinned memory behind the sampler kernels).
        Returns False if its last enqueued round asked for one more."""
        if self._pending is None:
            return True
        ev, host, k, which = self._pending
        self._pending = None
        ev.synchronize()
        ran = 1
        while ran <= k and int(host[2 * (ran - 1) + 1]):
            ran += 1
        if ran > k:
            self.stats['repeats'] += 1
            self._note_rounds(k + 1, which)
            self._miss[which] = self.HISTORY
            return False
        self.stats['idle_rounds'] += k - ran
        self._note_rounds(ran, which)
        return True

    def confirm(self):
        """After a speculative sample(): False if the last round that was enqueued asked for another one (the
        pass has to be repeated with more rounds); otherwise the result is exact, however many were enqueued.
        With max_total_iters rounds enqueued that cannot happen, and nothing is read back here."""
        if self._pending is None or self._pending[2] >= self.max_total_iters:
            return True
        return self._resolve()

    HISTORY = 32

    def _note_rounds(self, rounds, which=0):
        self._last_rounds = rounds
        self._hist[which] = (self._hist[which] + [rounds])[-self.HISTORY:]

    def guess_rounds(self):
        """Rounds to enqueue for the next call without reading a flag back.  An idle round costs three empty
        launches (~15 us), a round too few costs the whole forward pass again (~10 ms at 1024 rays), so the guess
        errs upwards: the most demanding of the last HISTORY calls, and all max_total_iters rounds for HISTORY calls
        after every miss (small beta: most steps need 2 rounds, one in seven needs 5 -- bench.py, sharp_state)."""
        which = 1 if self.global_rounds is not None else 0
        if self._miss[which] > 0:
            self._miss[which] -= 1
            return self.max_total_iters
        return max(self._hist[which]) if self._hist[which] else 1

    def sample(self, ray_dirs, cam_loc, model, want_points=True, speculate=0, beta0=None):
        """get_z_vals plus (optionally) the 3-D points of the ray samples and, in training, the eikonal
        points appended behind them -- written by the finish kernel instead of ~15 small tensor ops.

        speculate = k > 0: enqueue exactly k rounds WITHOUT reading the batch-global convergence flag back (the
        one host sync per round, during which the GPU would drain); rounds beyond the ones the flags ask for do
        nothing.  The caller enqueues the rest of its work and then calls confirm(), which tells whether k was
        enough."""
        dev = ray_dirs.device
        ray_dirs, cam_loc = _need_gpu(ray_dirs), _need_gpu(cam_loc)
        self._resolve()               # bookkeeping of the previous call (its flags arrived long ago)
        N = ray_dirs.shape[0]
        n_eval, n_final, n_extra = self.N_samples_eval, self.N_samples, self.N_samples_extra
        K = self.max_total_iters
        m_max = n_eval * K
        S = n_final + n_extra + 2
        noise = getattr(model, '_noise', None) or {}
        training = bool(model.training)
        net = model.implicit_network
        beta0 = (model.density.get_beta() if beta0 is None else beta0).detach().float().reshape(1).contiguous()
        f32 = dict(device=dev, dtype=torch.float32)
        z = torch.empty(N, m_max, **f32)
        sdf = torch.empty(N, m_max, **f32)
        new_z = torch.empty(N, n_eval, **f32)
        new_pos = torch.empty(N, n_eval, device=dev, dtype=torch.int32)
        pts = torch.empty(N * n_eval, 3, **f32)
        beta = torch.empty(N, **f32)
        flags = torch.empty(2 * K, device=dev, dtype=torch.int32)        # zeroed by msdf_sampler_init
        final_z = torch.empty(N, n_final, **f32)
        jitter = u_final = nei_drawn = eik_u_drawn = eik_unit_drawn = None
        if training:
            # ONE launch for every random draw of the call: stratified jitter, inverse-CDF u, neighbour jitter, the
            # eikonal sample's column (floor(u S): the reference's randint, ray_sampler.py:254) and the uniform
            # eikonal points ((2u - 1) R inside the finish kernel: the reference's uniform_(-R, R), network.py:587)
            need = [k for k in ('jitter', 'final_u', 'nei_rand', 'eik_idx', 'eik_uniform') if noise.get(k) is None]
            sizes = {'jitter': N * n_eval, 'final_u': N * n_final, 'nei_rand': 6 * N if want_points else 0,
                     'eik_idx': N, 'eik_uniform': 3 * N if want_points else 0}
            pool = torch.rand(sum(sizes[k] for k in need), **f32) if need else None
            drawn, off = {}, 0
            for k in need:
                drawn[k] = pool[off:off + sizes[k]]
                off += sizes[k]
            jitter = noise.get('jitter')
            jitter = drawn['jitter'].view(N, n_eval) if jitter is None else jitter.to(**f32).contiguous()
            u_final = noise.get('final_u')
            u_final = drawn['final_u'].view(N, n_final) if u_final is None else u_final.to(**f32).contiguous()
            nei_drawn = drawn.get('nei_rand')
            eik_u_drawn, eik_unit_drawn = drawn.get('eik_idx'), drawn.get('eik_uniform')
        a = _lib.SamplerArgs()
        a.ray_o, a.ray_d, a.N = cam_loc.data_ptr(), ray_dirs.data_ptr(), N
        a.m_max, a.n_eval, a.n_final, a.n_extra = m_max, n_eval, n_final, n_extra
        a.max_rounds, a.training, a.beta_iters = K, int(training), self.beta_iters
        a.near, a.far, a.bound = float(self.near), float(self.far), float(self.scene_bounding_sphere)
        a.eps, a.add_tiny, a.lemma = float(self.eps), float(self.add_tiny), self._lemma
        a.beta0, a.z, a.sdf = beta0.data_ptr(), z.data_ptr(), sdf.data_ptr()
        a.new_z, a.new_pos, a.pts, a.beta = new_z.data_ptr(), new_pos.data_ptr(), pts.data_ptr(), beta.data_ptr()
        a.flags = flags.data_ptr()
        a.jitter = jitter.data_ptr() if jitter is not None else None
        a.u_final = u_final.data_ptr() if u_final is not None else None
        a.final_z = final_z.data_ptr()
        st = _lib.stream_ptr()
        a.M = n_eval
        _lib.call('msdf_sampler_init', C.byref(a), st)
        z_out = z_eik = x_all = extra_idx = eik_idx = None

        def prepare_finish():
            """Everything the finish kernel needs -- none of it depends on the number of rounds -- issued while the
            GPU is still busy with the first round, so that after a host sync only the launch is left."""
            nonlocal z_out, z_eik, x_all, extra_idx, eik_idx
            eik_idx = noise.get('eik_idx')
            a.eik_u, a.eik_unit = None, 0
            if eik_idx is None and eik_u_drawn is not None:
                a.eik_idx, a.eik_u = None, eik_u_drawn.data_ptr()        # the column is floor(u S), formed in the kernel
            else:
                if eik_idx is None:
                    eik_idx = torch.randint(S, (N,), device=dev)
                eik_idx = eik_idx.to(device=dev, dtype=torch.int64).contiguous()
                a.eik_idx = eik_idx.data_ptr()
            z_out = torch.empty(N, S, **f32)
            z_eik = torch.empty(N, 1, **f32)
            a.z_out, a.z_eik, a.pts_out = z_out.data_ptr(), z_eik.data_ptr(), None
            if want_points:
                n_eik = 4 * N if training else 0
                x_all = torch.empty(N * S + n_eik, 3, **f32)
                a.pts_out = x_all.data_ptr()
                if training:
                    R = self.scene_bounding_sphere
                    eik_uniform = noise.get('eik_uniform')
                    if eik_uniform is None and eik_unit_drawn is not None and eik_unit_drawn.numel() == 3 * N:
                        eik_uniform, a.eik_unit = eik_unit_drawn.view(N, 3), 1       # U[0,1): scaled to the cube in the kernel
                    else:
                        eik_uniform = (torch.empty(N, 3, **f32).uniform_(-R, R) if eik_uniform is None
                                       else eik_uniform.to(**f32).contiguous())
                    nei = noise.get('nei_rand')
                    nei = nei_drawn.view(2 * N, 3) if nei is None else nei.to(**f32).contiguous()
                    a.eik_uniform, a.nei_rand = eik_uniform.data_ptr(), nei.data_ptr()
                    self._keep = (eik_uniform, nei)
            extra_idx = self._extra_columns(training, noise, dev)
            a.extra_idx = extra_idx.data_ptr()

        group = self.global_rounds
        # one all-reduce is enqueued per enqueued round, so the count must be the same on every rank of the group: the
        # caller's guess comes from guess_rounds(), which in this mode looks only at the history of all-reduced
        # decisions -- identical on every rank as long as the ranks make their group calls in lockstep (they must:
        # each one is a collective).  Round 3 enqueued all max_total_iters rounds here: K - 1 idle rounds and
        # collectives per call in the usual one-round state.
        which = 1 if group is not None else 0
        rounds = 0
        with torch.no_grad():
            while True:
                # fused forward kernel on the new points, [N*n_eval, 1]; in a speculated round it returns at once
                # when the previous round did not ask for this one
                run_flag = flags.data_ptr() + 4 * (2 * rounds - 1) if (speculate > 0 and rounds > 0) else None
                new_sdf = net._sdf_only(pts, run_flag=run_flag)
                a.new_sdf = new_sdf.data_ptr()
                a.M, a.round_idx = n_eval * (rounds + 1), rounds
                _lib.call('msdf_sampler_beta', C.byref(a), st)
                if group is not None:
                    # bits of a positive float order like the float: an integer MAX over ranks is the batch's max beta
                    import torch.distributed as dist
                    dist.all_reduce(flags[2 * rounds:2 * rounds + 1], op=dist.ReduceOp.MAX,
                                    group=None if group is True else group)
                _lib.call('msdf_sampler_resample', C.byref(a), st)
                if rounds == 0:
                    prepare_finish()
                rounds += 1
                if speculate > 0:
                    more = rounds < min(speculate, K)
                else:
                    more = bool(flags[2 * rounds - 1].item())         # the one host sync of the round
                if not more:
                    break
        self._pending = None
        self.stats['calls'] += 1
        if speculate > 0:
            host = torch.empty(flags.shape, dtype=flags.dtype, pin_memory=True)
            host.copy_(flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, host, rounds, which)
        else:
            self._note_rounds(rounds, which)
        # final set: 64 importance samples + near + far + 32 columns of the dense set
        _lib.call('msdf_sampler_finish', C.byref(a), st)
        return z_out, z_eik, x_all
