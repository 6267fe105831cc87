"""Where the device part of sdf_volume's 512^3 block goes (diagnostic): times of the tensor operations around the network."""
import time
import torch
n = 256
dev = 'cuda'
vals = torch.randn(n ** 3, device=dev)
up = torch.nn.Upsample(scale_factor=2, mode='nearest')


def t(fn, name, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print('%-40s %.2f ms' % (name, 1e3 * (time.time() - t0) / reps))
    return r


m256 = t(lambda: (vals.abs() < 0.05).reshape(n, n, n)[None, None], 'threshold mask 256^3')
mf = t(lambda: up(m256.float()), 'upsample mask (float) -> 512^3')
mb = t(lambda: mf.bool(), 'float -> bool 512^3')
v512 = t(lambda: up(vals.reshape(n, n, n)[None, None]).reshape(-1), 'upsample vals -> 512^3')
idx = t(lambda: mb.reshape(-1).nonzero().reshape(-1), 'nonzero over 512^3 (%.1f %% set)' % (100 * float(mb.float().mean())))
t(lambda: (torch.div(idx, 512 * 512, rounding_mode='floor'), idx % (512 * 512)), 'index arithmetic')
new = torch.randn(idx.numel(), device=dev)
t(lambda: v512.index_put_((idx,), new), 'vals[idx] = ...')
t(lambda: m256.repeat_interleave(2, 2).repeat_interleave(2, 3).repeat_interleave(2, 4), 'mask by repeat_interleave (bool)')
t(lambda: vals.reshape(n, n, n).repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2), 'vals by repeat_interleave')
