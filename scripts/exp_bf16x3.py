"""CPU experiment: how much accuracy does a 3-term bf16 split (hi*hi + hi*lo + lo*hi, fp32 accumulate)
lose on the 8x256 SDF network?  (fp32 MFMA vs bf16 MFMA.)  Uses the oracle only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import monosdf_oracle as mo, config, synth


def split(t):
    hi = t.bfloat16().float()
    return hi, (t - hi).bfloat16().float()


def lin3(x, w, b=None):
    xh, xl = split(x)
    wh, wl = split(w)
    y = xh @ wh.t() + xh @ wl.t() + xl @ wh.t()
    return y if b is None else y + b


def lin1(x, w, b=None):
    y = x.bfloat16().float() @ w.bfloat16().float().t()
    return y if b is None else y + b


orig = F.linear
conf = config.mlp_config(256, 8)
st = synth.make_state(conf, seed=0, jitter=0.05)
x = (torch.rand(2000, 3) * 2 - 1) * 0.9
ref = mo.get_outputs(st, conf, x, create_graph=False)
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for name, fn in [('bf16 x3 split', lin3), ('plain bf16', lin1)]:
    torch.nn.functional.linear = fn
    out = mo.get_outputs(st, conf, x, create_graph=False)
    torch.nn.functional.linear = orig
    print('%-14s max rel err: sdf %.2e  feat %.2e  grad_x sdf %.2e' % (name, rel(out[0], ref[0]), rel(out[1], ref[1]), rel(out[2], ref[2])))
