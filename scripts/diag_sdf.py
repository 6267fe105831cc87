import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import Case, rel_err
from oracle import monosdf_oracle as mo
from monosdf_amd.conf import ConfigTree
from monosdf_amd.model.network import MonoSDFNetwork

def bad_idx(a, b, tol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().double()
    e = (a - b).abs().reshape(a.shape[0], -1).max(1)[0] / (b.abs().max() + 1e-12)
    idx = torch.nonzero(e > tol).flatten()
    return idx

for name in sys.argv[1:] or ['mlp_w64_eval', 'mlp_w256_eval']:
    c = Case(name)
    m = MonoSDFNetwork(ConfigTree.from_dict(c.conf)); m.load_state_dict(c.state); m = m.cuda().eval()
    for P in (64, 1037, 4096):
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.3
        sdf_o, feat_o, grad_o = mo.get_outputs(c.state, c.conf, x, create_graph=False)
        raw_o = mo.sdf_network_raw(c.state, c.conf, x)
        sdf, feat, grad = m.implicit_network.get_outputs(x.cuda())
        with torch.no_grad():
            vals = m.implicit_network.get_sdf_vals(x.cuda())
        for nm, a, b in [('sdf', sdf, sdf_o), ('feat', feat, feat_o), ('grad', grad, grad_o), ('vals', vals, sdf_o)]:
            bi = bad_idx(a, b)
            print(name, 'P', P, nm, 'rel_err %.3e' % rel_err(a, b), 'n_bad', len(bi), bi[:12].tolist(), bi[-4:].tolist())
        nclamp = int((sdf_o < raw_o[:, :1] - 1e-9).sum())
        print('   clamped points in oracle:', nclamp)
