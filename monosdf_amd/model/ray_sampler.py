"""Ray samplers (reference: code/model/ray_sampler.py).  ``ErrorBoundSampler.get_z_vals`` keeps the
reference's signature ``(ray_dirs, cam_loc, model) -> (z_vals, z_samples_eik)``; the algorithm runs in
csrc/sampler.hip (one wave per ray) with the SDF evaluations done by the fused forward kernel.
One 4-byte device->host read per round carries the batch-global convergence test
(``beta.max() > beta0``, reference ray_sampler.py:179)."""
import ctypes as C
import math

import torch

from .. import _lib, ops


class RaySampler:
    def __init__(self, near, far):
        self.near, self.far = near, far


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=1.0e-6):
        super().__init__(near, 2.0 * scene_bounding_sphere * 1.75)
        if inverse_sphere_bg:
            raise NotImplementedError('monosdf_amd: inverse_sphere_bg is not used by any conf of the reference fork')
        self.N_samples = N_samples
        self.N_samples_eval = N_samples_eval
        self.N_samples_extra = N_samples_extra
        self.eps = eps
        self.beta_iters = beta_iters
        self.max_total_iters = max_total_iters
        self.scene_bounding_sphere = scene_bounding_sphere
        self.add_tiny = add_tiny
        self.last_rounds = 0
        self._pending = None
        # Lemma-2 constant, formed in fp32 like the reference does (ray_sampler.py:119)
        self._lemma = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))

    def get_z_vals(self, ray_dirs, cam_loc, model):
        z, z_eik, _ = self.sample(ray_dirs, cam_loc, model, want_points=False)
        return z, z_eik

    @staticmethod
    def _draw_columns(M, n, dev):
        """n distinct columns of the dense sample set, the same for all rays: drawn on the CPU generator as the
        reference does (ray_sampler.py:244) -- 256 bytes uploaded instead of a device sort."""
        return torch.randperm(M)[:n].pin_memory().to(dev, non_blocking=True)

    def confirm(self):
        """After a speculative sample(): True if the convergence flags ask for exactly the rounds that were run."""
        if self._pending is None:
            return True
        ev, host, k = self._pending
        self._pending = None
        ev.synchronize()
        more = [int(host[2 * r + 1]) for r in range(k)]
        return all(more[:k - 1]) and not more[k - 1]

    def sample(self, ray_dirs, cam_loc, model, want_points=True, speculate=0):
        """get_z_vals plus (optionally) the 3-D points of the ray samples and, in training, the eikonal
        points appended behind them -- written by the finish kernel instead of ~15 small tensor ops.

        speculate = k > 0: run exactly k rounds WITHOUT reading the batch-global convergence flag back (the one
        host sync per round, during which the GPU would drain); the caller enqueues the rest of its work and
        then calls confirm(), which tells whether k was the number of rounds the flags ask for."""
        dev = ray_dirs.device
        if not ray_dirs.is_cuda:
            raise RuntimeError('monosdf_amd: the sampler runs on the GPU only (no CPU fallback)')
        ray_dirs = ray_dirs.detach().float().contiguous()
        cam_loc = cam_loc.detach().float().contiguous()
        N = ray_dirs.shape[0]
        n_eval, n_final, n_extra = self.N_samples_eval, self.N_samples, self.N_samples_extra
        m_max = n_eval * self.max_total_iters
        S = n_final + n_extra + 2
        noise = getattr(model, '_noise', None) or {}
        training = bool(model.training)
        net = model.implicit_network
        beta0 = model.density.get_beta().detach().float().reshape(1).contiguous()
        f32 = dict(device=dev, dtype=torch.float32)
        z = torch.empty(N, m_max, **f32)
        sdf = torch.empty(N, m_max, **f32)
        new_z = torch.empty(N, n_eval, **f32)
        new_pos = torch.empty(N, n_eval, device=dev, dtype=torch.int32)
        pts = torch.empty(N * n_eval, 3, **f32)
        beta = torch.empty(N, **f32)
        flags = torch.zeros(2 * self.max_total_iters, device=dev, dtype=torch.int32)
        final_z = torch.empty(N, n_final, **f32)
        jitter = u_final = nei_drawn = None
        if training:
            # one launch for every U[0,1) draw of the call (stratified jitter, inverse-CDF u, neighbour jitter)
            need = [k for k in ('jitter', 'final_u', 'nei_rand') if noise.get(k) is None]
            sizes = {'jitter': N * n_eval, 'final_u': N * n_final, 'nei_rand': 6 * N if want_points else 0}
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
        lemma = self._lemma
        a = _lib.SamplerArgs()
        a.ray_o, a.ray_d, a.N = cam_loc.data_ptr(), ray_dirs.data_ptr(), N
        a.m_max, a.n_eval, a.n_final, a.n_extra = m_max, n_eval, n_final, n_extra
        a.max_rounds, a.training, a.beta_iters = self.max_total_iters, int(training), self.beta_iters
        a.near, a.far, a.bound = float(self.near), float(self.far), float(self.scene_bounding_sphere)
        a.eps, a.add_tiny, a.lemma = float(self.eps), float(self.add_tiny), lemma
        a.beta0, a.z, a.sdf = beta0.data_ptr(), z.data_ptr(), sdf.data_ptr()
        a.new_z, a.new_pos, a.pts, a.beta = new_z.data_ptr(), new_pos.data_ptr(), pts.data_ptr(), beta.data_ptr()
        a.jitter = jitter.data_ptr() if jitter is not None else None
        a.u_final = u_final.data_ptr() if u_final is not None else None
        a.final_z = final_z.data_ptr()
        st = _lib.stream_ptr()
        a.M = n_eval
        _lib.call('msdf_sampler_init', C.byref(a), st)
        rounds, M = 0, n_eval
        z_out = z_eik = x_all = extra_idx = eik_idx = None

        def prepare_finish():
            """Everything the finish kernel needs that does not depend on the number of rounds -- issued while the
            GPU is still busy with the first round, so that after the host sync below only the launch is left."""
            nonlocal z_out, z_eik, x_all, extra_idx, eik_idx
            eik_idx = noise.get('eik_idx')
            if eik_idx is None:
                eik_idx = torch.randint(S, (N,), device=dev)
            eik_idx = eik_idx.to(device=dev, dtype=torch.int64).contiguous()
            z_out = torch.empty(N, S, **f32)
            z_eik = torch.empty(N, 1, **f32)
            a.eik_idx, a.z_out, a.z_eik, a.pts_out = eik_idx.data_ptr(), z_out.data_ptr(), z_eik.data_ptr(), None
            if want_points:
                n_eik = 4 * N if training else 0
                x_all = torch.empty(N * S + n_eik, 3, **f32)
                a.pts_out = x_all.data_ptr()
                if training:
                    R = self.scene_bounding_sphere
                    eik_uniform = noise.get('eik_uniform')
                    eik_uniform = (torch.empty(N, 3, **f32).uniform_(-R, R) if eik_uniform is None
                                   else eik_uniform.to(**f32).contiguous())
                    nei = noise.get('nei_rand')
                    nei = nei_drawn.view(2 * N, 3) if nei is None else nei.to(**f32).contiguous()
                    a.eik_uniform, a.nei_rand = eik_uniform.data_ptr(), nei.data_ptr()
                    self._keep = (eik_uniform, nei)
            # the 32 extra columns for the one-round case (the common one); redrawn below if more rounds ran
            if n_extra > 0 and training and noise.get('extra_idx') is None:
                extra_idx = self._draw_columns(n_eval, n_extra, dev)

        with torch.no_grad():
            while True:
                new_sdf = net.get_sdf_vals(pts)                       # fused forward kernel, [N*n_eval, 1]
                a.new_sdf = new_sdf.data_ptr()
                a.M, a.round_idx = M, rounds
                a.flag = flags.data_ptr() + 8 * rounds
                _lib.call('msdf_sampler_beta', C.byref(a), st)
                _lib.call('msdf_sampler_resample', C.byref(a), st)
                if rounds == 0:
                    prepare_finish()
                if speculate > 0:
                    more = int(rounds + 1 < speculate)
                else:
                    more = int(flags[2 * rounds + 1].item())          # the one host sync of the round
                rounds += 1
                if not more:
                    break
                M += n_eval
        self._pending = None
        if speculate > 0:
            host = torch.empty(flags.shape, dtype=flags.dtype, pin_memory=True)
            host.copy_(flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, host, rounds)
        self.last_rounds = rounds
        # final set: 64 importance samples + near + far + 32 columns of the dense set
        if n_extra > 0:
            if not training or noise.get('extra_idx') is not None or M != n_eval:
                extra_idx = noise.get('extra_idx') if training else None
                if extra_idx is None:
                    extra_idx = (self._draw_columns(M, n_extra, dev) if training
                                 else torch.linspace(0, M - 1, n_extra, device=dev).long())
            extra_idx = extra_idx.to(device=dev, dtype=torch.int64).contiguous()
        else:
            extra_idx = torch.zeros(1, device=dev, dtype=torch.int64)
        a.M = M
        a.extra_idx = extra_idx.data_ptr()
        _lib.call('msdf_sampler_finish', C.byref(a), st)
        return z_out, z_eik, x_all
