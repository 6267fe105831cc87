"""How does `(x / divide_factor + 1.0) / 2.0` round on the device?  (which of the candidate sequences is bit-identical)"""
import numpy as np
import torch
g = torch.Generator().manual_seed(0)
x = ((torch.rand(1 << 16, 3, generator=g) * 2 - 1) * 1.15).cuda()
df = 1.1
ref = (x / df + 1.0) / 2.0
inv32 = torch.tensor(np.float32(1.0) / np.float32(df), device='cuda')
inv64 = torch.tensor(np.float32(1.0 / df), device='cuda')
cands = {
    'x * (1f/1.1f) + 1 then * 0.5': (x * inv32 + 1.0) * 0.5,
    'x * float(1/1.1 in double) + 1 then * 0.5': (x * inv64 + 1.0) * 0.5,
    'true division by 1.1f': (x / torch.tensor(np.float32(df), device='cuda') + 1.0) * 0.5,
    'fma(x, inv32, 1) * 0.5': (torch.addcmul(torch.ones_like(x), x, inv32.expand_as(x))) * 0.5,
}
for k, v in cands.items():
    print('%-45s mismatches %d of %d' % (k, int((v != ref).sum()), ref.numel()))
print('1f/1.1f = %.10g, float(1/1.1) = %.10g' % (inv32.item(), inv64.item()))
