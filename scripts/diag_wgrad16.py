"""Compares the bf16x3 weight-gradient kernel with the fp32 one on the same workspace (debug aid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import Case, rel_err
from monosdf_amd.conf import ConfigTree
from monosdf_amd.model.network import MonoSDFNetwork

for name in ['mlp_w64_eval', 'mlp_w256_eval']:
    grads = {}
    for prec in ['fp32', 'bf16x3']:
        c = Case(name)
        m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state); m = m.cuda().train()
        m.set_precision('fp32')
        net = m.implicit_network
        g = torch.Generator().manual_seed(5)
        P = 777
        x = ((torch.rand(P, 3, generator=g) * 2 - 1) * 1.2).cuda()
        F = c.conf['feature_vector_size']
        ca, cb, cc = torch.randn(P, 1, generator=g).cuda(), (torch.randn(P, F, generator=g) * 0.1).cuda(), torch.randn(P, 3, generator=g).cuda()
        fused = net._fused(x.device)
        fused.precision_wgrad = prec
        if prec == 'bf16x3':
            # same fp32 forward/backward kernels, only the weight-gradient GEMM on the bf16 core
            orig = fused.run_wgrad
            def run(P_pad, base, fused=fused, orig=orig):
                fused.precision = 'bf16x3'
                try:
                    return orig(P_pad, base)
                finally:
                    fused.precision = 'fp32'
            fused.run_wgrad = run
        sdf, feat, grad = net.get_outputs(x)
        loss = (ca * sdf).sum() + (cb * feat).sum() + (cc * grad).sum()
        loss.backward()
        grads[prec] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for n in grads['fp32']:
        print(name, n, 'rel err %.2e' % rel_err(grads['bf16x3'][n], grads['fp32'][n]))
