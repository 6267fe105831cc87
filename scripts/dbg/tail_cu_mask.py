"""Timing experiment (no product code involved): does keeping the partly filled last round of the forward + gradient
launch and the colour network's launch on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) shorten the pair?

  baseline : fwd+grad(104,448 points) ; colour(100,352)                        one stream
  split    : fwd+grad(98,304 = 3 full rounds) ; then  fwd+grad(6,144) on a stream masked to 3 of every 8 CUs (96)
             beside colour(98,304) on a stream masked to the other 5 of 8 (160) ; then colour(2,048)
  unmasked : the same split on two ordinary streams (round 2's experiment)

Separate dummy inputs per launch (only the time matters).  Prints ms per sequence."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                     # noqa: E402
from monosdf_amd.model.network import MonoSDFNetwork   # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(keep):
    """keep(cu) -> bool over 256 CUs"""
    words = (ctypes.c_uint32 * 8)()
    for cu in range(256):
        if keep(cu):
            words[cu // 32] |= (1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


torch.manual_seed(0)
model = MonoSDFNetwork(bench.model_conf()).cuda().train()
net, rn = model.implicit_network, model.rendering_network
net.share('cuda')
rn.share('cuda')
S = 98
g = torch.Generator(device='cuda').manual_seed(0)


def sdf_inputs(P):
    return (torch.rand(P, 3, device='cuda', generator=g) - 0.5)


def col_inputs(P):
    N = P // S if P % S == 0 else None
    pts = torch.randn(P, 3, device='cuda', generator=g)
    nrm = torch.randn(P, 3, device='cuda', generator=g)
    feat = torch.randn(P, 256, device='cuda', generator=g) * 0.3
    n_rays = (P + S - 1) // S
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device='cuda', generator=g), dim=-1)
    return pts, nrm, dirs, feat, torch.arange(n_rays, device='cuda')


def run_sdf(x):
    P = x.shape[0]
    net.evaluate(x, P, P, save=True)


def run_col(inp, P):
    pts, nrm, dirs, feat, idx = inp
    spr = S if P % S == 0 else 64
    n_rays = P // spr
    rn(pts, nrm, dirs[:n_rays].contiguous(), feat, idx[:n_rays], if_pixel_input=True, samples_per_ray=spr)


PA, PB, PC = 98304, 6144, 100352
xa, xb, xall = sdf_inputs(PA), sdf_inputs(PB), sdf_inputs(PA + PB)
ca = col_inputs(1536 * 64)       # 98,304 points as 1536 "rays" of 64
cb = col_inputs(2048)
call = col_inputs(PC)
main = torch.cuda.current_stream()
tail_m = masked_stream(lambda cu: cu % 8 < 3)
rest_m = masked_stream(lambda cu: cu % 8 >= 3)
tail_u, rest_u = torch.cuda.Stream(), torch.cuda.Stream()


def baseline():
    run_sdf(xall)
    with torch.no_grad():
        run_col(call, PC)


def split(s_tail, s_rest):
    run_sdf(xa)
    ev = torch.cuda.Event()
    ev.record(main)
    s_tail.wait_event(ev)
    s_rest.wait_event(ev)
    with torch.cuda.stream(s_tail):
        run_sdf(xb)
    with torch.cuda.stream(s_rest), torch.no_grad():
        run_col(ca, 1536 * 64)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s_tail)
    e2.record(s_rest)
    main.wait_event(e1)
    main.wait_event(e2)
    with torch.no_grad():
        run_col(cb, 2048)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.enable_grad():
    print('baseline (one stream)            %.3f ms' % timeit(baseline))
    print('split, two ordinary streams      %.3f ms' % timeit(lambda: split(tail_u, rest_u)))
    print('split, CU-masked streams 96/160  %.3f ms' % timeit(lambda: split(tail_m, rest_m)))
    print('baseline again                   %.3f ms' % timeit(baseline))
