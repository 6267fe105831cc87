"""torch.autograd bindings of the HIP kernels (device memory + streams come from PyTorch,
all arithmetic on the hot path happens in libmonosdf_hip.so).

Each Function owns one stage of the reference's autograd graph (reference:
code/model/network.py) so DDP / Adam / checkpoints keep working on ordinary
nn.Parameters: gradients of the *effective* (weight-normalised) matrices come out
of the kernels and flow back to weight_g / weight_v through PyTorch's own autograd.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, parallel, plan as planlib


def _need_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError('monosdf_amd: %s must be a GPU tensor (the HIP path has no CPU fallback)' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('monosdf_amd: %s must be float32' % name)
    return t.contiguous()


def _pad64(n):
    return (n + 63) // 64 * 64


# matrix core of the fused MLP kernels (include/monosdf_plan.h MSDF_PRECISION_*)
PRECISIONS = ('fp32', 'bf16x3', 'bf16x6')      # index = MSDF_PRECISION_*
_PLANES = {'bf16x3': 2, 'bf16x6': 3}


# ---------------------------------------------------------------------------
# side stream: the colour network's weight-gradient GEMM feeds nothing but the optimiser, and the SDF network's
# weight-gradient launch ends with a partly filled chip (1344 workgroups on 256 slots), so the two run side
# by side.  ColorMlpFunction.backward only queues its launch; SdfMlpFunction.backward starts it on the side
# stream right after the SDF backward kernel, next to its own weight-gradient launch; the first consumer of
# the gradients (FusedWeightNormFunction.backward) joins.  Without an SDF backward in the graph the queued
# launch simply runs at the join, on the main stream.
# ---------------------------------------------------------------------------
import os as _os

_SIDE_STREAMS = {}
_DEFERRED = []         # launches queued for the side stream: callables
_PENDING = []          # events of side-stream work whose results the main stream has not waited for yet
USE_SIDE_STREAM = _os.environ.get('MSDF_SIDE_STREAM', '1') != '0'
COLOR_WGRAD_EARLY = _os.environ.get('MSDF_COLOR_WGRAD_EARLY', '1') != '0'
# hash-grid node: the encoder's Jacobian applied inside the SDF kernels (0: msdf_hash_node_input_gradient /
# msdf_hash_node_second_grad as launches of their own -- comparison runs)
FUSE_JACOBIAN = _os.environ.get('MSDF_FUSE_JACOBIAN', '1') != '0'


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


def start_side_work(device):
    """Launch everything queued on the side stream, ordered after the work already on the current stream."""
    if not _DEFERRED:
        return
    main, side = torch.cuda.current_stream(), _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for launch in _DEFERRED:
            launch(side)
        done = torch.cuda.Event()
        done.record(side)
    _DEFERRED.clear()
    _PENDING.append(done)


def join_side_work():
    """Make the current stream wait for the side stream (cheap when there is nothing to wait for)."""
    if _DEFERRED:                      # nobody started it: run it here, in order
        cur = torch.cuda.current_stream()
        for launch in _DEFERRED:
            launch(cur)
        _DEFERRED.clear()
    if _PENDING:
        cur = torch.cuda.current_stream()
        for ev in _PENDING:
            cur.wait_event(ev)
        _PENDING.clear()


class FusedMlp:
    """Device-side state of one fused network (plan, pack rules, maps, caches)."""

    def __init__(self, mlp_plan, device, precision='fp32'):
        if precision not in PRECISIONS:
            raise ValueError('monosdf_amd: precision must be one of %s, got %r' % (PRECISIONS, precision))
        self.mp = mlp_plan
        self.device = device
        self.precision = precision
        # the plan the kernels get: same geometry, K counts / pack offsets of the chosen matrix core
        self.plan = mlp_plan.plan if precision == 'fp32' else mlp_plan.build_b16(_PLANES[precision])
        self.rules_dev = torch.from_numpy(mlp_plan.rules_np).to(device)
        self.maps_dev = torch.from_numpy(mlp_plan.maps_np).to(device)
        self._wgrad_cache = {}

    # -- weights -----------------------------------------------------------------
    def pack(self, flat_w, flat_b):
        """Fragment-ordered weight packs of this network's matrix core (+ one LDS chunk of slack: the
        kernels stage whole chunks)."""
        mp = self.mp
        flat_w = _need_cuda(flat_w.detach(), 'weights')
        flat_b = _need_cuda(flat_b.detach(), 'biases')
        assert flat_w.numel() == mp.n_w and flat_b.numel() == mp.n_b
        units16 = mp.wpack_f4 if self.precision == 'fp32' else mp.wpack16_units(_PLANES[self.precision])
        wpack = torch.empty(units16 * 4 + 4 * 17 * 64 * 4, device=self.device, dtype=torch.float32)
        bpack = torch.empty(mp.bpack_f + 64, device=self.device, dtype=torch.float32)
        _lib.call('msdf_pack_weights', C.byref(self.plan), _lib.ptr(self.rules_dev), _lib.ptr(self.maps_dev),
                  _lib.ptr(flat_w), _lib.ptr(flat_b), _lib.ptr(wpack), _lib.ptr(bpack), _lib.stream_ptr())
        return wpack, bpack

    # -- weight gradients ----------------------------------------------------------
    def wgrad_program(self, P_pad):
        key = P_pad
        if key not in self._wgrad_cache:
            build = planlib.build_sdf_wgrad if self.mp.kind == 'sdf' else planlib.build_color_wgrad
            # the split model is fitted to the fp32 kernel (also what the bf16x6 core runs); the bf16x3 kernel has its own
            # stage times and wave grids and keeps round 1's plan: the same ~54-stage split for every item
            uniform = None
            if self.precision == 'bf16x3':
                S = max(1, (P_pad + 1727) // 1728)
                uniform = {w: S for w in planlib.WgradProgram.CLASS_COST_US}
            prog = planlib.balanced_program(build, self.mp, P_pad, splits=uniform)
            rules_dev = torch.from_numpy(prog.rules_bytes()).to(self.device)
            wg_map = torch.from_numpy(prog.wg_map()).to(self.device)
            # the item table holds offsets, not addresses: one host-to-device copy per (network, point count), ever
            items_dev = torch.from_numpy(prog.items_bytes()).to(self.device)
            # when the reduce rules store to every gradient element the flat buffer needs no zero fill
            full = prog.writes_every_element(self.mp.n_w + self.mp.n_b, self.mp.maps_np)
            self._wgrad_cache[key] = dict(prog=prog, rules=rules_dev, wg_map=wg_map, items=items_dev, full=full)
        return self._wgrad_cache[key]

    def run_wgrad(self, P_pad, base_addr, defer=False):
        """base_addr: buffer name -> tensor.  Returns the flat gradient [n_w + n_b].
        defer: only queue the two launches (start_side_work / join_side_work run them); the returned tensor
        is filled by then."""
        ent = self.wgrad_program(P_pad)
        prog = ent['prog']
        items_dev = ent['items']
        bases = [base_addr.get(name) for name in prog.BUFFERS]
        part = torch.empty(prog.part_f + 64, device=self.device, dtype=torch.float32)
        alloc = torch.empty if ent['full'] else torch.zeros
        grad = alloc(self.mp.n_w + self.mp.n_b, device=self.device, dtype=torch.float32)
        wg_map = ent['wg_map']
        held = list(base_addr.values())

        rules_dev = ent['rules']

        def launch(stream=None):
            if stream is not None:
                # allocated on the main stream, used on this one
                for t in held + [part, grad, items_dev, wg_map, rules_dev, self.maps_dev]:
                    t.record_stream(stream)
            st = _lib.stream_ptr()
            _lib.call('msdf_wgrad', _lib.ptr(items_dev), _lib.ptr(wg_map), wg_map.numel() // 2,
                      _lib.ptr(part), P_pad, PRECISIONS.index(self.precision), _lib.ptr(bases[0]), _lib.ptr(bases[1]), st)
            _lib.call('msdf_reduce', _lib.ptr(rules_dev), len(prog.rules), _lib.ptr(self.maps_dev),
                      _lib.ptr(part), _lib.ptr(grad), st)

        if defer:
            _DEFERRED.append(launch)
        else:
            launch()
        return grad


# ---------------------------------------------------------------------------
# weight normalisation of a whole network (one launch forward, one backward)
# ---------------------------------------------------------------------------
class WeightNormTables:
    """One immutable set of device tables for msdf_weightnorm_*: pointers of (v, g, b) per layer."""

    def __init__(self, shapes, tensors):
        """shapes: [(rows, cols, has_g)] per layer; tensors: the parameters in (v, [g], b) order."""
        dev = tensors[0].device
        recs, row_layer, w_off, b_off, it = [], [], 0, 0, iter(tensors)
        for i, (rows, cols, has_g) in enumerate(shapes):
            r = _lib.WnLayer()
            v = next(it)
            g = next(it) if has_g else None
            b = next(it)
            r.v, r.g, r.b = v.data_ptr(), (g.data_ptr() if has_g else None), b.data_ptr()
            r.rows, r.cols = rows, cols
            r.w_off, r.b_off, r.row_off, r.has_g = w_off, b_off, b_off, int(has_g)
            recs.append(bytes(r))
            row_layer += [i] * rows
            w_off += rows * cols
            b_off += rows
        self.layers_dev = torch.from_numpy(np.frombuffer(b''.join(recs), dtype=np.uint8).copy()).to(dev)
        self.row_layer_dev = torch.tensor(row_layer, dtype=torch.int32, device=dev)
        self.n_w, self.total_rows = w_off, b_off


class WeightNormState:
    """Cache of WeightNormTables keyed by the parameters' storage addresses (rebuilt when a parameter moves).
    A table object is never modified once built: a pending backward keeps the one it was given."""

    def __init__(self):
        self.key, self.cur = None, None

    def tables(self, shapes, tensors):
        key = tuple(t.data_ptr() for t in tensors)
        if key != self.key:
            self.cur = WeightNormTables(shapes, tensors)
            self.key = key
        return self.cur


class FusedWeightNormFunction(torch.autograd.Function):
    """(v_0, g_0, b_0, v_1, g_1, b_1, ...) -> (flat effective weights, flat biases)."""

    @staticmethod
    def forward(ctx, state, shapes, *params):
        st = state.tables(shapes, params)
        dev = params[0].device
        flat_w = torch.empty(st.n_w, device=dev, dtype=torch.float32)
        flat_b = torch.empty(st.total_rows, device=dev, dtype=torch.float32)
        norms = torch.empty(st.total_rows, device=dev, dtype=torch.float32)
        _lib.call('msdf_weightnorm_forward', _lib.ptr(st.layers_dev), _lib.ptr(st.row_layer_dev), st.total_rows,
                  _lib.ptr(flat_w), _lib.ptr(flat_b), _lib.ptr(norms), _lib.stream_ptr())
        # the backward kernel re-reads v and g: saving them puts them under autograd's version check (an in-place
        # update between forward and backward raises instead of silently giving the gradient of other weights)
        ctx.state, ctx.shapes = state, shapes
        ctx.save_for_backward(norms, *params)
        return flat_w, flat_b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, g_b):
        norms, *params = ctx.saved_tensors
        shapes = ctx.shapes
        st = ctx.state.tables(shapes, params)       # the saved tensors' addresses (same table unless storage moved)
        dev = norms.device
        join_side_work()            # g_w may have been produced on the side stream
        g_w = g_w.contiguous()
        dv = torch.empty(st.n_w, device=dev, dtype=torch.float32)
        dg = torch.empty(st.total_rows, device=dev, dtype=torch.float32)
        _lib.call('msdf_weightnorm_backward', _lib.ptr(st.layers_dev), _lib.ptr(st.row_layer_dev), st.total_rows,
                  _lib.ptr(g_w), _lib.ptr(norms), _lib.ptr(dv), _lib.ptr(dg), _lib.stream_ptr())
        grads, w_off, b_off = [], 0, 0
        for rows, cols, has_g in shapes:
            n = rows * cols
            grads.append(dv[w_off:w_off + n].view(rows, cols))
            if has_g:
                grads.append(dg[b_off:b_off + rows].view(rows, 1))
            grads.append(g_b[b_off:b_off + rows])
            w_off += n
            b_off += rows
        return (None, None) + tuple(grads)


def fused_weight_norm(state, layers):
    params, shapes = [], []
    for l in layers:
        params.append(l.weight_v if l.has_weight_norm else l.weight)
        if l.has_weight_norm:
            params.append(l.weight_g)
        params.append(l.bias)
        shapes.append((l.out_features, l.in_features, bool(l.has_weight_norm)))
    return FusedWeightNormFunction.apply(state, tuple(shapes), *params)


# ---------------------------------------------------------------------------
# SDF network
# ---------------------------------------------------------------------------
def sdf_forward_nograd(mlp, wpack, bpack, x, aux, clamp_radius, sphere_scale, run_flag=None, aux_lm=None):
    """get_sdf_vals: forward only (sampler).  run_flag: device address of a uint32; the launch does nothing when
    it holds 0 (a sampler round that the previous round did not ask for).  aux_lm = (C, L C): `aux` is the hash
    encoder's level-major tensor [L, P, C] instead of rows [P, 16 * aux_tiles]."""
    x = _need_cuda(x, 'points')
    P = x.shape[0]
    out = torch.empty(P, 1, device=x.device, dtype=torch.float32)
    aC, aLC = aux_lm if aux_lm is not None else (0, 0)
    _lib.call('msdf_sdf_forward_lm', C.byref(mlp.plan), _lib.ptr(wpack), _lib.ptr(bpack),
              _lib.ptr(x), _lib.ptr(aux), int(aC), int(aLC), P, float(clamp_radius), float(sphere_scale), _lib.ptr(out),
              C.c_void_p(run_flag) if run_flag else None, _lib.stream_ptr())
    return out


class SdfMlpFunction(torch.autograd.Function):
    """(x, aux, W, b) -> (sdf [:n_split], sdf [n_split:], feat [n_feat,F], d sdf/dx [:n_split], d sdf/dx [n_split:],
    d sdf/d aux [P,A]).  The two point groups (ray samples | eikonal points) come out as separate tensors --
    views of one buffer -- because their gradients come from different consumers: joining them by slicing one
    output would cost a zero-fill, a copy and an add per group in the backward pass."""

    @staticmethod
    def forward(ctx, x, aux, flat_w, flat_b, wpack, bpack, mlp, n_clamp, n_feat, clamp_radius,
                sphere_scale, save, n_split=None, aux_lm=None, aux_jac=None):
        """aux_lm = (C, L C): aux, d sdf / d aux (the last output) and -- in backward -- their gradients are the hash
        encoder's level-major tensors [L, P, C] instead of rows [P, 16 * aux_tiles].
        aux_jac = (dy_dx, k) (with aux_lm, C = 2): the encoder's Jacobian [L, P, 3, 2] and the chain-rule factor of
        x -> x01; the kernel then adds the grid part of d sdf / d x to the returned gradient itself, and the backward
        kernel forms the gradient arriving at d sdf / d aux from it (ctx.gg_out, set by the caller, receives k * g_nrm)."""
        ctx.set_materialize_grads(False)
        mp = mlp.mp
        plan = mlp.plan
        x = _need_cuda(x.detach(), 'points')
        P = x.shape[0]
        P_pad = _pad64(max(P, 1))
        dev = x.device
        has_aux = plan.aux_tiles > 0
        if has_aux:
            aux = _need_cuda(aux.detach(), 'aux features')
        woff, total = planlib.sdf_workspace(mp, P_pad)
        if not save:
            total = woff['PM']           # only H is touched in inference
        ws = torch.empty(max(total, 64), device=dev, dtype=torch.float32)
        F = 16 * plan.feat_tiles
        sdf = torch.empty(P, 1, device=dev, dtype=torch.float32)
        # rows up to the next multiple of 64 exist (zero) so the colour network's weight-gradient GEMM
        # can stream whole 64-point tiles of this buffer
        feat_full = torch.empty(_pad64(max(n_feat, 1)), F, device=dev, dtype=torch.float32)
        feat_full[n_feat:].zero_()
        feat = feat_full[:n_feat]
        nrm = torch.empty(P, 3, device=dev, dtype=torch.float32)
        aC, aLC = aux_lm if (aux_lm is not None and has_aux) else (0, 0)
        aux_shape = (aLC // aC, P, aC) if aC else (P, 16 * plan.aux_tiles)
        r_aux = torch.empty(*aux_shape, device=dev, dtype=torch.float32) if has_aux else None
        clamped = torch.empty(max(P, 1), device=dev, dtype=torch.uint8)
        a = _lib.FgArgs()
        a.wpack, a.bpack, a.x, a.aux = wpack.data_ptr(), bpack.data_ptr(), x.data_ptr(), \
            (aux.data_ptr() if has_aux else None)
        a.P, a.P_pad, a.n_clamp, a.n_feat = P, P_pad, n_clamp, n_feat
        a.clamp_radius, a.sphere_scale = float(clamp_radius), float(sphere_scale)
        a.sdf, a.feat, a.nrm = sdf.data_ptr(), feat.data_ptr(), nrm.data_ptr()
        a.r_aux = r_aux.data_ptr() if has_aux else None
        a.clamped = clamped.data_ptr()
        base = ws.data_ptr()
        a.H = base + 4 * woff['H']
        a.PM = base + 4 * woff['PM'] if save else None
        a.IN0 = base + 4 * woff['IN0'] if save else None
        a.save = 1 if save else 0
        a.aux_C, a.aux_LC = int(aC), int(aLC)
        ctx.aux_jac = None
        if aux_jac is not None and aC == 2:
            a.dy_dx, a.aux_dx_scale = aux_jac[0].data_ptr(), float(aux_jac[1])
            ctx.aux_jac = (aux_jac[0], float(aux_jac[1]))
        if P > 0:
            _lib.call('msdf_sdf_fwd_grad', C.byref(plan), C.byref(a), _lib.stream_ptr())
        ctx.mlp, ctx.P, ctx.P_pad, ctx.n_feat, ctx.saved = mlp, P, P_pad, n_feat, save
        ctx.has_aux, ctx.aux_lm, ctx.aux_shape = has_aux, (int(aC), int(aLC)), aux_shape
        ctx.n_split = ns = P if n_split is None else int(n_split)
        ctx.save_for_backward(x, ws, clamped, wpack, bpack)
        return sdf[:ns], sdf[ns:], feat, nrm[:ns], nrm[ns:], (r_aux if has_aux else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_sdf_b, g_feat, g_nrm, g_nrm_b, g_raux):
        if not ctx.saved:
            raise RuntimeError('monosdf_amd: backward through an inference-mode SDF evaluation')
        x, ws, clamped, wpack, bpack = ctx.saved_tensors
        mlp, P, P_pad = ctx.mlp, ctx.P, ctx.P_pad
        mp = mlp.mp
        plan = mlp.plan
        dev = x.device
        woff, _ = planlib.sdf_workspace(mp, P_pad)
        cont = lambda t: None if t is None else t.contiguous()
        g_sdf, g_feat, g_nrm, g_raux = cont(g_sdf), cont(g_feat), cont(g_nrm), cont(g_raux)
        g_sdf_b, g_nrm_b = cont(g_sdf_b), cont(g_nrm_b)
        g_aux = torch.empty(*ctx.aux_shape, device=dev, dtype=torch.float32) if ctx.has_aux else None
        b = _lib.BwArgs()
        b.aux_C, b.aux_LC = ctx.aux_lm
        jac = getattr(ctx, 'aux_jac', None)
        if jac is not None:
            gg_out = getattr(ctx, 'gg_out', None)
            b.dy_dx, b.aux_dx_scale = jac[0].data_ptr(), jac[1]
            b.gg_out = gg_out.data_ptr() if gg_out is not None else None
        b.wpack, b.bpack, b.x = wpack.data_ptr(), bpack.data_ptr(), x.data_ptr()
        b.P, b.P_pad, b.n_feat, b.n_split = P, P_pad, ctx.n_feat, ctx.n_split
        b.g_sdf_b = g_sdf_b.data_ptr() if g_sdf_b is not None else None
        b.g_nrm_b = g_nrm_b.data_ptr() if g_nrm_b is not None else None
        b.g_sdf = g_sdf.data_ptr() if g_sdf is not None else None
        b.g_feat = g_feat.data_ptr() if (g_feat is not None and ctx.n_feat > 0) else None
        b.g_nrm = g_nrm.data_ptr() if g_nrm is not None else None
        b.g_raux = g_raux.data_ptr() if (g_raux is not None and ctx.has_aux) else None
        b.clamped = clamped.data_ptr()
        base = ws.data_ptr()
        for k in ('H', 'PM', 'QB', 'AB', 'GSDF', 'QLAST'):
            setattr(b, k, base + 4 * woff[k])
        b.T = None           # the second-order term is formed again in the sweep down: no buffer
        b.g_aux = g_aux.data_ptr() if g_aux is not None else None
        if P > 0:
            _lib.call('msdf_sdf_backward', C.byref(plan), C.byref(b), _lib.stream_ptr())
            start_side_work(dev)                 # the colour network's queued weight-gradient launch, if any
            between = getattr(ctx, 'after_sweeps', None)
            if between is not None:
                between(g_aux)                   # consumers of d loss / d aux that should precede the weight gradients
            grad = mlp.run_wgrad(P_pad, {'ws': ws})
        else:
            between = getattr(ctx, 'after_sweeps', None)
            if between is not None:
                between(g_aux)
            grad = torch.zeros(mp.n_w + mp.n_b, device=dev)
        return (None, g_aux, grad[:mp.n_w], grad[mp.n_w:]) + (None,) * 11


class _InnerCtx:
    """Stands in for autograd's ctx when one Function runs another Function's forward / backward inside its own.
    The tensors it is given to save are handed to the OUTER ctx.save_for_backward by the caller (autograd's
    in-place version check then covers them) and put back before the inner backward runs."""

    def __init__(self):
        self.saved_tensors = ()
        self.after_sweeps = None

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, flag):
        pass


class GridSdfFunction(torch.autograd.Function):
    """Hash-grid encoding + SDF network + d sdf / d x as ONE autograd node (ImplicitNetworkGrid.get_outputs /
    gradient_sdf, reference network.py:247-309 with hashgrid.py:14-101).

    As three nodes (encode -> MLP -> grid part of d sdf/dx) the backward pass scatters into the embedding table
    twice: the second-order term when the gradient of d sdf/dx arrives, the first-order term after the MLP's
    backward kernel.  Both visit the same corners of the same points, so here they are ONE binned scatter
    (msdf_hash_encode_backward_fused) at the end of the node's backward."""

    @staticmethod
    def forward(ctx, x, embeddings, flat_w, flat_b, wpack, bpack, mlp, enc, n_clamp, n_feat, sphere_scale, save,
                n_split, divide_factor):
        ctx.set_materialize_grads(False)
        x = _need_cuda(x.detach(), 'points')
        emb = _need_cuda(embeddings.detach(), 'embeddings')
        B, D = x.shape
        L, Cdim, S, H = enc
        offsets = mlp.grid_offsets
        A = 16 * mlp.plan.aux_tiles                 # row pitch of the grid features as the SDF kernels read them
        # the node forms of the hash kernels (csrc/hashgrid.hip): x01 formed inside the encoder kernel, which writes its
        # own level-major [L, B, C] output and the Jacobian dy_dx [L, B, 3, C]
        x01 = torch.empty(B, D, device=x.device, dtype=torch.float32)
        outputs = torch.empty(L, B, Cdim, device=x.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * Cdim, device=x.device, dtype=torch.float32)
        st = _lib.stream_ptr()
        _lib.call('msdf_hash_node_forward', _lib.ptr(x), float(divide_factor), _lib.ptr(x01), _lib.ptr(emb),
                  _lib.ptr(offsets), _lib.ptr(outputs), 0, B, Cdim, L, S, H, _lib.ptr(dy_dx), st)
        # two features per level (every configuration of the reference): the SDF kernels read and write the encoder's
        # level-major tensors themselves; other channel counts go through rows and the LDS-tiled transpose
        lm = (Cdim, L * Cdim) if (Cdim == 2 and mlp.precision == 'fp32') else None      # the bf16 cores take rows
        if lm is not None:
            aux = outputs
        else:
            aux = torch.empty(B, A, device=x.device, dtype=torch.float32)
            _lib.call('msdf_hash_transpose', _lib.ptr(outputs), _lib.ptr(aux), None, None, L, B, Cdim, A, 1, st)
        inner = _InnerCtx()
        # d sdf / d x through the grid: sum_{l,c} (d sdf / d feature) * d feature / d x01, chain rule to x, added to the
        # MLP's own d sdf / d x -- inside the SDF kernel with the level-major tensors (FUSE_JACOBIAN), otherwise by
        # msdf_hash_node_input_gradient in place (nrm_a / nrm_b are the two halves of ONE [B,3] buffer starting at nrm_a)
        k = 0.5 / divide_factor
        jac = (dy_dx, k) if (lm is not None and FUSE_JACOBIAN) else None
        # the grid class never clamps (network.py:290-309): clamp radius 0
        sdf_a, sdf_b, feat, nrm_a, nrm_b, r_aux = SdfMlpFunction.forward(
            inner, x, aux, flat_w, flat_b, wpack, bpack, mlp, n_clamp, n_feat, 0.0, sphere_scale, save, n_split, lm, jac)
        if jac is None:
            assert nrm_a.data_ptr() + 12 * nrm_a.shape[0] == nrm_b.data_ptr() or nrm_b.shape[0] == 0
            _lib.call('msdf_hash_node_input_gradient', _lib.ptr(r_aux), 0 if lm is not None else A, _lib.ptr(dy_dx), B,
                      Cdim, L, float(k), _lib.ptr(nrm_a), st)
        ctx.inner, ctx.enc, ctx.k, ctx.n_entries, ctx.lm = inner, enc, k, emb.shape[0], lm
        ctx.offsets = offsets
        ctx.save_for_backward(x01, dy_dx, r_aux, *inner.saved_tensors)
        inner.saved_tensors = ()
        return sdf_a, sdf_b, feat, nrm_a, nrm_b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_sdf_b, g_feat, g_nrm, g_nrm_b):
        x01, dy_dx, r_aux, *inner_saved = ctx.saved_tensors
        inner, (L, Cdim, S, H), k = ctx.inner, ctx.enc, ctx.k
        inner.saved_tensors = tuple(inner_saved)
        B, D = x01.shape
        dev = x01.device
        ns = inner.n_split
        st = _lib.stream_ptr()
        A = 16 * inner.mlp.plan.aux_tiles
        # gradient arriving at the grid part of d sdf/dx (the reference's grad_grad_inputs, hashgrid.py:71-84), scaled
        # by the chain-rule factor, and its term for d sdf / d feature: grad_grad[b, l C + c] = sum_d gg[b,d] dy_dx --
        # one launch, written as the rows the SDF backward kernel reads
        lm = ctx.lm
        gg = torch.empty(B, D, device=dev, dtype=torch.float32)
        if getattr(inner, 'aux_jac', None) is not None:
            # the SDF backward kernel forms this gradient from dy_dx itself and writes gg (k * g_nrm) for the scatter
            g_raux = None
            inner.gg_out = gg
        else:
            g_raux = torch.empty(*((L, B, Cdim) if lm is not None else (B, A)), device=dev, dtype=torch.float32)
            cont = lambda t: None if t is None else t.contiguous()
            g_nrm_c, g_nrm_b_c = cont(g_nrm), cont(g_nrm_b)
            _lib.call('msdf_hash_node_second_grad', _lib.ptr(g_nrm_c) if g_nrm_c is not None else None,
                      _lib.ptr(g_nrm_b_c) if g_nrm_b_c is not None else None, ns, float(k), _lib.ptr(gg), _lib.ptr(dy_dx),
                      _lib.ptr(g_raux), 0 if lm is not None else A, B, Cdim, L, st)
        done = []

        def scatter(g_aux):
            """Both embedding gradients in ONE binned scatter, launched between the MLP's sweeps and its weight-gradient
            kernels: the 48.8 MB result is then complete ~2 ms before the node's backward ends, and a multi-GPU run
            exchanges it under the weight-gradient kernels (parallel.GradientAverager)."""
            # the "=" form: the table gradient is written, not added to -- no 48.8 MB zero fill, no read of the table
            g_emb = torch.empty(ctx.n_entries, Cdim, device=dev, dtype=torch.float32)
            nbytes = _lib.load().msdf_hash_scatter_workspace_bytes(B, Cdim, L, ctx.n_entries)
            ws = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
            # the scatter reads level-major gradients (coalesced): what the SDF kernels wrote (two features per level),
            # otherwise both operands transposed in ONE launch
            if lm is not None:
                g_lm = (g_aux, r_aux)
            else:
                g_lm = torch.empty(2, L, B, Cdim, device=dev, dtype=torch.float32)
                _lib.call('msdf_hash_transpose', _lib.ptr(g_aux), _lib.ptr(g_lm[0]), _lib.ptr(r_aux), _lib.ptr(g_lm[1]),
                          L, B, Cdim, A, 0, st)
            _lib.call('msdf_hash_node_scatter', _lib.ptr(g_lm[0]), _lib.ptr(g_lm[1]), 0, _lib.ptr(x01),
                      _lib.ptr(ctx.offsets), _lib.ptr(g_emb), B, Cdim, L, S, H, _lib.ptr(gg), ctx.n_entries,
                      _lib.ptr(ws), int(nbytes), st)
            parallel.mark_grad_ready(g_emb)
            done.append(g_emb)

        inner.after_sweeps = scatter
        try:
            res = SdfMlpFunction.backward(inner, g_sdf, g_sdf_b, g_feat, g_nrm, g_nrm_b, g_raux)
        finally:
            inner.after_sweeps = None
            inner.saved_tensors = ()
            inner.gg_out = None
        g_w, g_b = res[2], res[3]
        # the only reference to the table gradient leaves with the return value: autograd then adopts the tensor as
        # embeddings.grad instead of copying 48.8 MB
        return (None, done.pop(), g_w, g_b) + (None,) * 10


def hash_node_features(x, divide_factor, embeddings, offsets, enc, pitch, level_major=True):
    """Grid features of the points x (world coordinates), no gradient: what the sampler's SDF evaluations feed the fused
    forward kernel -- x01 inside the encoder kernel.  Returns (tensor, aux_lm): the encoder's level-major [L, B, C]
    tensor with aux_lm = (C, L C) for two features per level, else rows of `pitch` floats (one LDS-tiled transpose)
    with aux_lm = None."""
    x = _need_cuda(x.detach(), 'points')
    emb = _need_cuda(embeddings.detach(), 'embeddings')
    L, Cdim, S, H = enc
    B = x.shape[0]
    st = _lib.stream_ptr()
    outputs = torch.empty(L, B, Cdim, device=x.device, dtype=torch.float32)
    _lib.call('msdf_hash_node_forward', _lib.ptr(x), float(divide_factor), None, _lib.ptr(emb), _lib.ptr(offsets),
              _lib.ptr(outputs), 0, B, Cdim, L, S, H, None, st)
    if Cdim == 2 and level_major:
        return outputs, (Cdim, L * Cdim)          # the SDF kernel reads the level-major tensor itself
    aux = torch.empty(B, pitch, device=x.device, dtype=torch.float32)
    _lib.call('msdf_hash_transpose', _lib.ptr(outputs), _lib.ptr(aux), None, None, L, B, Cdim, pitch, 1, st)
    return aux, None


# ---------------------------------------------------------------------------
# colour network
# ---------------------------------------------------------------------------
class ColorMlpFunction(torch.autograd.Function):
    """(points, per-ray dirs, normals, feat, per-ray code, W, b) -> rgb [P,3]."""

    @staticmethod
    def forward(ctx, x, dirs, nrm, feat, code, flat_w, flat_b, wpack, bpack, cmlp, spr, save):
        mp = cmlp.mp
        plan = cmlp.plan
        x = _need_cuda(x.detach(), 'points')
        dirs = _need_cuda(dirs.detach(), 'view dirs')
        nrm = _need_cuda(nrm.detach(), 'normals')
        feat = feat.detach()
        P = x.shape[0]
        P_pad = _pad64(max(P, 1))
        dev = x.device
        # the kernels read whole 16-slot tiles of the feature vector: a narrower tensor is taken as the view of the
        # SDF kernel's padded rows it normally is (ImplicitNetwork.evaluate slices them to feature_vector_size; the
        # pad columns are zero there), otherwise padded with zeros
        Fk = 16 * plan.layer[0].kt
        ctx.feat_cols = feat.shape[1]
        if feat.shape[1] > Fk or Fk - feat.shape[1] >= 16:
            raise RuntimeError('monosdf_amd: feature width %d does not fit the colour network plan (%d slots)'
                               % (feat.shape[1], Fk))
        if feat.shape[1] < Fk:
            if feat.is_cuda and feat.dim() == 2 and feat.stride() == (Fk, 1) and \
                    feat.untyped_storage().nbytes() // 4 - feat.storage_offset() >= P * Fk:
                feat = feat.as_strided((P, Fk), (Fk, 1))
            else:
                feat = torch.nn.functional.pad(feat, (0, Fk - feat.shape[1]))
        feat = _need_cuda(feat, 'features')
        room = feat.untyped_storage().nbytes() // 4 - feat.storage_offset()
        if room < P_pad * feat.shape[1]:
            # the weight-gradient GEMM streams whole 64-point tiles: give the tail rows finite values
            padded = torch.zeros(P_pad, feat.shape[1], device=dev, dtype=torch.float32)
            padded[:P] = feat
            feat = padded[:P]
        has_code = plan.aux_tiles > 0
        if has_code:
            code = _need_cuda(code.detach(), 'image code')
        woff, total = planlib.color_workspace(mp, P_pad)
        ws = torch.empty(max(total if save else 64, 64), device=dev, dtype=torch.float32)
        rgb = torch.empty(P, 3, device=dev, dtype=torch.float32)
        a = _lib.ColorFwdArgs()
        a.wpack, a.bpack = wpack.data_ptr(), bpack.data_ptr()
        a.x, a.dirs, a.nrm, a.feat = x.data_ptr(), dirs.data_ptr(), nrm.data_ptr(), feat.data_ptr()
        a.code = code.data_ptr() if has_code else None
        a.P, a.P_pad, a.spr, a.save = P, P_pad, int(spr), 1 if save else 0
        a.rgb = rgb.data_ptr()
        base = ws.data_ptr()
        a.H = base + 4 * woff['H'] if save else None
        a.MISC = base + 4 * woff['MISC'] if save else None
        if P > 0:
            _lib.call('msdf_color_forward', C.byref(plan), C.byref(a), _lib.stream_ptr())
        ctx.cmlp, ctx.P, ctx.P_pad, ctx.saved, ctx.has_code, ctx.spr = cmlp, P, P_pad, save, has_code, spr
        ctx.save_for_backward(rgb, feat, ws, wpack, bpack)
        return rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb):
        if not ctx.saved:
            raise RuntimeError('monosdf_amd: backward through an inference-mode colour evaluation')
        rgb, feat, ws, wpack, bpack = ctx.saved_tensors
        cmlp, P, P_pad = ctx.cmlp, ctx.P, ctx.P_pad
        mp = cmlp.mp
        plan = cmlp.plan
        dev = rgb.device
        woff, _ = planlib.color_workspace(mp, P_pad)
        g_rgb = g_rgb.contiguous()
        g_feat = torch.empty_like(feat)
        g_misc = torch.empty(P, 16 * mp.misc_tiles, device=dev, dtype=torch.float32)
        b = _lib.ColorBwdArgs()
        b.wpack, b.bpack, b.rgb, b.g_rgb = wpack.data_ptr(), bpack.data_ptr(), rgb.data_ptr(), g_rgb.data_ptr()
        b.P, b.P_pad = P, P_pad
        base = ws.data_ptr()
        b.H, b.AB = base + 4 * woff['H'], base + 4 * woff['AB']
        b.g_feat, b.g_misc = g_feat.data_ptr(), g_misc.data_ptr()
        g_nrm = torch.empty(P, 3, device=dev, dtype=torch.float32) if plan.mode == 1 else None
        b.g_nrm = g_nrm.data_ptr() if g_nrm is not None else None
        if P > 0:
            _lib.call('msdf_color_backward', C.byref(plan), C.byref(b), _lib.stream_ptr())
            grad = cmlp.run_wgrad(P_pad, {'ws': ws, 'feat': feat}, defer=USE_SIDE_STREAM)
            if USE_SIDE_STREAM and COLOR_WGRAD_EARLY:
                # start it now: it then runs beside the SDF backward kernel and fills the partly filled last round
                # of that kernel's workgroups, instead of starting after it
                start_side_work(dev)
        else:
            grad = torch.zeros(mp.n_w + mp.n_b, device=dev)
        g_code = None
        if ctx.has_code:
            g_code = g_misc[:, 48:].reshape(P // ctx.spr, ctx.spr, -1).sum(1)
        if ctx.feat_cols < g_feat.shape[1]:
            g_feat = g_feat[:, :ctx.feat_cols]
        return (None, None, g_nrm, g_feat, g_code, grad[:mp.n_w], grad[mp.n_w:], None, None, None, None, None)


# ---------------------------------------------------------------------------
# compositor
# ---------------------------------------------------------------------------
class CompositeFunction(torch.autograd.Function):
    """(z, sdf, rgb, normals, beta, depth_scale) -> weights, rgb_values, depth_values, normal_map [, depth_vals].

    beta_raw / beta_min (optional): `beta` is then |beta_raw| + beta_min formed by effective_beta() and the gradient
    goes to `beta_raw` directly -- sign(beta_raw) times the sum of the kernel's per-ray partials in ONE launch -- instead
    of through abs / add / sum / sgn / mul launches of autograd.  want_depth_vals: also z * depth_scale per sample
    (the reference's `depth_vals`), written by the same kernel."""

    @staticmethod
    def forward(ctx, z, sdf, rgb, nrm, beta, depth_scale, white_bkgd, bg, pose=None, beta_raw=None,
                want_depth_vals=False):
        ctx.set_materialize_grads(False)      # unused outputs (e.g. `weights`) arrive as None, not as zero fills
        z = _need_cuda(z.detach(), 'z_vals')
        N, S = z.shape
        ctx.in_shapes = (sdf.shape, rgb.shape, nrm.shape)
        sdf = _need_cuda(sdf.detach(), 'sdf').reshape(N, S)
        rgb = _need_cuda(rgb.detach(), 'rgb').reshape(N, S, 3)
        nrm = _need_cuda(nrm.detach(), 'normals').reshape(N, S, 3)
        beta = _need_cuda(beta.detach(), 'beta').reshape(1)
        ctx.beta_shape = None if beta_raw is None else beta_raw.shape
        raw = beta if beta_raw is None else _need_cuda(beta_raw.detach(), 'beta').reshape(1)
        # one scale per ray: a dense [N] vector, or a column of a [N, k] table read in place (pitch k floats)
        depth_scale = depth_scale.detach()
        if depth_scale.dim() == 2 and depth_scale.shape[0] == N and depth_scale.shape[1] == 1 and N > 0 \
                and depth_scale.is_cuda and depth_scale.dtype == torch.float32 and depth_scale.stride(1) == 1:
            ds_stride = int(depth_scale.stride(0))
        else:
            depth_scale = _need_cuda(depth_scale, 'depth_scale').reshape(N)
            ds_stride = 1
        dev = z.device
        weights = torch.empty(N, S, device=dev)
        rgb_values = torch.empty(N, 3, device=dev)
        depth_values = torch.empty(N, 1, device=dev)
        normal_map = torch.empty(N, 3, device=dev)
        wsum = torch.empty(max(N, 1), device=dev)
        depth_vals = torch.empty(N, S, device=dev) if want_depth_vals else None
        a = _lib.CompositeArgs()
        a.z, a.sdf, a.rgb, a.nrm = z.data_ptr(), sdf.data_ptr(), rgb.data_ptr(), nrm.data_ptr()
        a.beta, a.depth_scale = beta.data_ptr(), depth_scale.data_ptr()
        a.depth_scale_stride = ds_stride
        a.depth_vals = depth_vals.data_ptr() if depth_vals is not None else None
        a.N, a.S, a.white_bkgd = N, S, 1 if white_bkgd else 0
        a.bg0, a.bg1, a.bg2 = [float(v) for v in bg]
        a.weights, a.rgb_values, a.depth_values = weights.data_ptr(), rgb_values.data_ptr(), depth_values.data_ptr()
        a.normal_map, a.wsum = normal_map.data_ptr(), wsum.data_ptr()
        if pose is not None:
            pose = _need_cuda(pose.detach(), 'pose').reshape(-1, 4, 4)
            if pose.shape[0] not in (1, N):
                raise RuntimeError('monosdf_amd: pose must be [1,4,4] or [N,4,4]')
            a.pose, a.pose_stride = pose.data_ptr(), (16 if pose.shape[0] == N and N > 1 else 0)
        else:
            pose = z.new_zeros(1)
            a.pose, a.pose_stride = None, 0
        ctx.has_pose, ctx.pose_stride = a.pose is not None, a.pose_stride
        _lib.call('msdf_composite_forward', C.byref(a), _lib.stream_ptr())
        ctx.save_for_backward(z, sdf, rgb, nrm, beta, depth_scale, weights, wsum, depth_values, pose, raw)
        ctx.white_bkgd, ctx.bg, ctx.ds_stride = white_bkgd, [float(v) for v in bg], ds_stride
        if want_depth_vals:
            ctx.mark_non_differentiable(depth_vals)       # z_vals carries no gradient (the sampler runs under no_grad)
            return weights, rgb_values, depth_values, normal_map, depth_vals
        return weights, rgb_values, depth_values, normal_map

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, g_rgbv, g_depth, g_nmap, g_dvals=None):
        z, sdf, rgb, nrm, beta, depth_scale, weights, wsum, depth_values, pose, raw = ctx.saved_tensors
        N, S = z.shape
        dev = z.device
        cont = lambda t: None if t is None else t.contiguous()
        g_w, g_rgbv, g_depth, g_nmap = cont(g_w), cont(g_rgbv), cont(g_depth), cont(g_nmap)
        g_sdf = torch.empty(N, S, device=dev)
        g_rgb = torch.empty(N, S, 3, device=dev)
        g_nrm = torch.empty(N, S, 3, device=dev)
        g_beta_part = torch.empty(max(N, 1), device=dev)
        b = _lib.CompositeBwdArgs()
        b.z, b.sdf, b.rgb, b.nrm = z.data_ptr(), sdf.data_ptr(), rgb.data_ptr(), nrm.data_ptr()
        b.beta, b.depth_scale = beta.data_ptr(), depth_scale.data_ptr()
        b.weights, b.wsum, b.depth_values = weights.data_ptr(), wsum.data_ptr(), depth_values.data_ptr()
        b.g_rgb_values = g_rgbv.data_ptr() if g_rgbv is not None else None
        b.g_depth = g_depth.data_ptr() if g_depth is not None else None
        b.g_normal = g_nmap.data_ptr() if g_nmap is not None else None
        b.g_weights = g_w.data_ptr() if g_w is not None else None
        b.N, b.S, b.white_bkgd = N, S, 1 if ctx.white_bkgd else 0
        b.bg0, b.bg1, b.bg2 = ctx.bg
        b.g_sdf, b.g_rgb, b.g_nrm, b.g_beta_part = g_sdf.data_ptr(), g_rgb.data_ptr(), g_nrm.data_ptr(), \
            g_beta_part.data_ptr()
        b.pose, b.pose_stride = (pose.data_ptr() if ctx.has_pose else None), ctx.pose_stride
        b.depth_scale_stride = ctx.ds_stride
        _lib.call('msdf_composite_backward', C.byref(b), _lib.stream_ptr())
        sh = ctx.in_shapes
        if ctx.beta_shape is not None:
            # d loss / d beta_raw = sign(beta_raw) * sum over rays, one launch (fixed summation order)
            g_raw = torch.empty(1, device=dev)
            _lib.call('msdf_beta_grad', _lib.ptr(raw), _lib.ptr(g_beta_part), N, _lib.ptr(g_raw), _lib.stream_ptr())
            return (None, g_sdf.reshape(sh[0]), g_rgb.reshape(sh[1]), g_nrm.reshape(sh[2]), None, None, None, None, None,
                    g_raw.reshape(ctx.beta_shape), None)
        g_beta = g_beta_part[:N].sum().reshape(beta.shape)
        return (None, g_sdf.reshape(sh[0]), g_rgb.reshape(sh[1]), g_nrm.reshape(sh[2]), g_beta, None, None, None, None,
                None, None)


def effective_beta(beta_raw, beta_min):
    """|beta_raw| + beta_min as a detached device tensor [1] (LaplaceDensity.get_beta, reference density.py:28-30) in one
    launch; the gradient path is CompositeFunction's beta_raw argument."""
    raw = _need_cuda(beta_raw.detach(), 'beta').reshape(1)
    out = torch.empty(1, device=raw.device, dtype=torch.float32)
    _lib.call('msdf_beta_eff', _lib.ptr(raw), float(beta_min), _lib.ptr(out), _lib.stream_ptr())
    return out


# ---------------------------------------------------------------------------
# Laplace density as an operator of its own (the fused path evaluates it inside the compositor / sampler)
# ---------------------------------------------------------------------------
class LaplaceDensityFunction(torch.autograd.Function):
    """(sdf [..., cols], beta [1] or one per row) -> sigma (reference: model/density.py:21-26)."""

    @staticmethod
    def forward(ctx, sdf, beta):
        s = _need_cuda(sdf.detach(), 'sdf')
        b = _need_cuda(beta.detach().to(s.device), 'beta').reshape(-1)
        cols = s.shape[-1] if s.dim() > 0 else 1
        rows = s.numel() // max(cols, 1)
        if b.numel() not in (1, rows):
            raise RuntimeError('monosdf_amd: beta must hold one value or one per row of sdf')
        stride = int(b.numel() == rows and rows > 1)
        out = torch.empty_like(s)
        _lib.call('msdf_laplace_density', _lib.ptr(s), _lib.ptr(b), stride, s.numel(), max(cols, 1), _lib.ptr(out),
                  _lib.stream_ptr())
        ctx.save_for_backward(s, b)
        ctx.meta = (stride, cols, beta.shape)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        s, b = ctx.saved_tensors
        stride, cols, beta_shape = ctx.meta
        g = g.contiguous()
        g_sdf, g_be = torch.empty_like(s), torch.empty_like(s)
        _lib.call('msdf_laplace_density_backward', _lib.ptr(s), _lib.ptr(b), stride, s.numel(), max(cols, 1),
                  _lib.ptr(g), _lib.ptr(g_sdf), _lib.ptr(g_be), _lib.stream_ptr())
        g_beta = (g_be.reshape(-1, max(cols, 1)).sum(1) if stride else g_be.sum()).reshape(beta_shape)
        return g_sdf, g_beta


# ---------------------------------------------------------------------------
# hash grid (mirrors the reference's two autograd Functions, hashencoder/hashgrid.py:14-101)
# ---------------------------------------------------------------------------
# 'binned' (default): the embedding gradients are summed per table slice in LDS (msdf_hash_encode_*_ws, one record per
# corner through a caller-owned workspace); 'atomic': one float atomic per corner like the reference's kernels
HASH_SCATTER = _os.environ.get('MSDF_HASH_SCATTER', 'binned')


def _hash_workspace(B, Cdim, L, embeddings):
    n = _lib.load().msdf_hash_scatter_workspace_bytes(B, Cdim, L, embeddings.shape[0])
    return torch.empty(int(n), device=embeddings.device, dtype=torch.uint8), int(n)


class HashEncodeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, S, H, calc_grad_inputs):
        inputs = _need_cuda(inputs, 'inputs')
        embeddings = _need_cuda(embeddings, 'embeddings')
        offsets = offsets.contiguous()
        if offsets.dtype != torch.int32 or not offsets.is_cuda:
            raise RuntimeError('monosdf_amd: offsets must be a GPU int32 tensor')
        B, D = inputs.shape
        L, Cdim = offsets.shape[0] - 1, embeddings.shape[1]
        outputs = torch.empty(L, B, Cdim, device=inputs.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * Cdim if calc_grad_inputs else 1, device=inputs.device, dtype=torch.float32)
        _lib.call('msdf_hash_encode_forward', _lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(offsets),
                  _lib.ptr(outputs), B, D, Cdim, L, float(S), int(H), int(bool(calc_grad_inputs)),
                  _lib.ptr(dy_dx), _lib.stream_ptr())
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, Cdim, L, float(S), int(H))
        ctx.calc_grad_inputs = bool(calc_grad_inputs)
        return outputs.permute(1, 0, 2).reshape(B, L * Cdim)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cdim, L, S, H = ctx.dims
        grad = grad.view(B, L, Cdim).permute(1, 0, 2).contiguous()
        g_in, g_emb = HashEncodeBackwardFunction.apply(grad, inputs, embeddings, offsets, dy_dx, ctx.dims,
                                                       ctx.calc_grad_inputs, True)
        return (g_in if ctx.calc_grad_inputs else None), g_emb, None, None, None, None


class HashEncodeBackwardFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad, inputs, embeddings, offsets, dy_dx, dims, calc_grad_inputs, want_emb=True):
        B, D, Cdim, L, S, H = dims
        grad = _need_cuda(grad, 'grad')
        g_in = torch.zeros_like(inputs)
        # the reference always scatters into grad_embeddings, even when autograd discards it
        # (SURVEY 8a11); want_emb=False skips that wasted pass, the result the caller sees is identical
        g_emb = torch.zeros_like(embeddings) if want_emb else None
        if want_emb and HASH_SCATTER == 'binned':
            ws, nbytes = _hash_workspace(B, Cdim, L, embeddings)
            _lib.call('msdf_hash_encode_backward_ws', _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings),
                      _lib.ptr(offsets), _lib.ptr(g_emb), B, D, Cdim, L, S, H, int(calc_grad_inputs),
                      _lib.ptr(dy_dx), _lib.ptr(g_in), embeddings.shape[0], _lib.ptr(ws), nbytes, _lib.stream_ptr())
        else:
            _lib.call('msdf_hash_encode_backward', _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings),
                      _lib.ptr(offsets), _lib.ptr(g_emb), B, D, Cdim, L, S, H, int(calc_grad_inputs),
                      _lib.ptr(dy_dx), _lib.ptr(g_in), _lib.stream_ptr())
        if g_emb is None:
            g_emb = embeddings.new_zeros(1)
        ctx.save_for_backward(grad, inputs, embeddings, offsets, dy_dx)
        ctx.dims, ctx.calc_grad_inputs = dims, calc_grad_inputs
        return g_in, g_emb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg_in, gg_emb):
        # like the reference: gg_emb is ignored and nothing flows to the inputs (hashgrid.py:87,101)
        grad, inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cdim, L, S, H = ctx.dims
        gg_in = _need_cuda(gg_in, 'grad_grad_inputs')
        grad_grad = torch.zeros_like(grad)
        grad2_emb = torch.zeros_like(embeddings)
        if HASH_SCATTER == 'binned' and Cdim > 1:
            ws, nbytes = _hash_workspace(B, Cdim, L, embeddings)
            _lib.call('msdf_hash_encode_second_backward_ws', _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings),
                      _lib.ptr(offsets), B, D, Cdim, L, S, H, int(ctx.calc_grad_inputs), _lib.ptr(dy_dx),
                      _lib.ptr(gg_in), _lib.ptr(grad_grad), _lib.ptr(grad2_emb), embeddings.shape[0], _lib.ptr(ws),
                      nbytes, _lib.stream_ptr())
        else:
            _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings),
                      _lib.ptr(offsets), B, D, Cdim, L, S, H, int(ctx.calc_grad_inputs), _lib.ptr(dy_dx),
                      _lib.ptr(gg_in), _lib.ptr(grad_grad), _lib.ptr(grad2_emb), _lib.stream_ptr())
        return grad_grad, None, grad2_emb, None, None, None, None, None


class HashEncodeWithJacobian(torch.autograd.Function):
    """Forward that also returns dy_dx (for the fused-MLP path, where d/dx is applied explicitly)."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, S, H):
        inputs = _need_cuda(inputs, 'inputs')
        embeddings_c = _need_cuda(embeddings, 'embeddings')
        B, D = inputs.shape
        L, Cdim = offsets.shape[0] - 1, embeddings.shape[1]
        outputs = torch.empty(L, B, Cdim, device=inputs.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * Cdim, device=inputs.device, dtype=torch.float32)
        _lib.call('msdf_hash_encode_forward', _lib.ptr(inputs), _lib.ptr(embeddings_c), _lib.ptr(offsets),
                  _lib.ptr(outputs), B, D, Cdim, L, float(S), int(H), 1, _lib.ptr(dy_dx), _lib.stream_ptr())
        ctx.save_for_backward(inputs, embeddings_c, offsets, dy_dx)
        ctx.dims = (B, D, Cdim, L, float(S), int(H))
        ctx.mark_non_differentiable(dy_dx)
        return outputs.permute(1, 0, 2).reshape(B, L * Cdim), dy_dx

    @staticmethod
    def backward(ctx, grad, _g_dy):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cdim, L, S, H = ctx.dims
        grad = grad.view(B, L, Cdim).permute(1, 0, 2).contiguous()
        _, g_emb = HashEncodeBackwardFunction.apply(grad, inputs, embeddings, offsets, dy_dx, ctx.dims, False, True)
        return None, g_emb, None, None, None


# ---------------------------------------------------------------------------
# fused benchmark loss (value + gradients in one launch)
# ---------------------------------------------------------------------------
class ProbeLossFunction(torch.autograd.Function):
    """mean|rgb| + w_n mean|normal| + w_d mean depth + w_e mean (|g1|-1)^2 + w_s mean |n1 - n2|  (BASELINE.md 2)."""

    @staticmethod
    def forward(ctx, rgb, nrm, depth, g1, g2, w_normal, w_depth, w_eik, w_smooth):
        rgb, nrm = _need_cuda(rgb.detach(), 'rgb_values'), _need_cuda(nrm.detach(), 'normal_map')
        depth = _need_cuda(depth.detach(), 'depth_values').reshape(-1)
        g1, g2 = _need_cuda(g1.detach(), 'grad_theta'), _need_cuda(g2.detach(), 'grad_theta_nei')
        N, M = rgb.shape[0], g1.shape[0]
        dev = rgb.device
        # the five gradient tensors are views of one buffer: the backward scales them with ONE multiply
        sizes = [t.numel() for t in (rgb, nrm, depth, g1, g2)]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        outs, off = [], 0
        for t, n in zip((rgb, nrm, depth, g1, g2), sizes):
            outs.append(flat[off:off + n].view(t.shape))
            off += n
        partial = torch.empty(1, device=dev, dtype=torch.float32)        # the complete loss value (one workgroup)
        a = _lib.ProbeLossArgs()
        a.rgb, a.nrm, a.depth, a.g1, a.g2 = [t.data_ptr() for t in (rgb, nrm, depth, g1, g2)]
        a.N, a.M = N, M
        a.w_normal, a.w_depth, a.w_eik, a.w_smooth = float(w_normal), float(w_depth), float(w_eik), float(w_smooth)
        a.g_rgb, a.g_nrm, a.g_depth, a.g_g1, a.g_g2 = [t.data_ptr() for t in outs]
        a.partial = partial.data_ptr()
        _lib.call('msdf_probe_loss', C.byref(a), _lib.stream_ptr())
        ctx.save_for_backward(flat)
        ctx.sizes, ctx.shapes = sizes, [t.shape for t in (rgb, nrm, depth, g1, g2)]
        return partial.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        flat, = ctx.saved_tensors
        scaled = flat * g
        res, off = [], 0
        for n, shape in zip(ctx.sizes, ctx.shapes):
            res.append(scaled[off:off + n].view(shape))
            off += n
        res[2] = res[2].reshape(-1, 1)
        return tuple(res) + (None, None, None, None)


def probe_loss(out, w_normal=0.05, w_depth=0.1, w_eik=0.05, w_smooth=0.005):
    return ProbeLossFunction.apply(out['rgb_values'], out['normal_map'], out['depth_values'], out['grad_theta'],
                                   out['grad_theta_nei'], w_normal, w_depth, w_eik, w_smooth)


class SplitRowsFunction(torch.autograd.Function):
    """t -> (t[:k], t[k:]) whose backward is ONE concatenation (slicing twice costs two zero-fills, two copies
    and an add in autograd)."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.set_materialize_grads(False)
        ctx.k, ctx.shape = int(k), t.shape
        return t[:k], t[k:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        tail = ctx.shape[1:]
        if ga is None:
            ga = gb.new_zeros((ctx.k,) + tuple(tail))
        if gb is None:
            gb = ga.new_zeros((ctx.shape[0] - ctx.k,) + tuple(tail))
        return torch.cat([ga, gb], 0), None


class MonoSdfLossFunction(torch.autograd.Function):
    """(rgb_values, depth_values, normal_map, grad_theta, grad_theta_nei | data) -> [8] scalars of MonoSDFLoss.
    Element 0 is `loss`; gradients flow from it only (the other entries are the reference's logging scalars)."""

    @staticmethod
    def forward(ctx, rgb, depth, normal, g1, g2, sdf, rgb_gt, depth_gt, normal_gt, mask_gt, weights, gamma,
                scale_invariant):
        rgb, normal = _need_cuda(rgb.detach(), 'rgb_values'), _need_cuda(normal.detach(), 'normal_map')
        ctx.depth_shape = depth.shape
        depth = _need_cuda(depth.detach(), 'depth_values').reshape(-1)
        sdf = _need_cuda(sdf.detach(), 'sdf')
        N = rgb.shape[0]
        sdf = sdf.reshape(N, -1)
        has_eik = g1 is not None
        if has_eik:
            g1, g2 = _need_cuda(g1.detach(), 'grad_theta'), _need_cuda(g2.detach(), 'grad_theta_nei')
            if g1.shape != g2.shape:
                raise RuntimeError('monosdf_amd: grad_theta and grad_theta_nei must have the same shape')
        dev = rgb.device
        data = [_need_cuda(t.detach().to(dev).float(), n).reshape(-1)
                for t, n in ((rgb_gt, 'rgb ground truth'), (depth_gt, 'depth cue'), (normal_gt, 'normal cue'),
                             (mask_gt, 'mask'))]
        if data[0].numel() != 3 * N or data[1].numel() != N or data[2].numel() != 3 * N or data[3].numel() != N:
            raise RuntimeError('monosdf_amd: ground-truth tensors do not match the %d rays of the batch' % N)
        out = torch.empty(8, device=dev, dtype=torch.float32)
        mask = torch.empty(N, device=dev, dtype=torch.float32)
        # the gradient tensors are views of one buffer: the backward scales them with ONE multiply
        like = [rgb, depth, normal] + ([g1, g2] if has_eik else [])
        sizes = [t.numel() for t in like]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        grads, off = [], 0
        for t, n in zip(like, sizes):
            grads.append(flat[off:off + n].view(t.shape))
            off += n
        a = _lib.MonoSdfLossArgs()
        a.rgb, a.depth, a.normal, a.sdf = rgb.data_ptr(), depth.data_ptr(), normal.data_ptr(), sdf.data_ptr()
        a.grad_theta = g1.data_ptr() if has_eik else None
        a.grad_nei = g2.data_ptr() if has_eik else None
        a.rgb_gt, a.depth_gt, a.normal_gt, a.mask_gt = [t.data_ptr() for t in data]
        a.N, a.S, a.E = N, sdf.shape[1], (g1.shape[0] if has_eik else 0)
        a.gamma, a.scale_invariant = int(bool(gamma)), int(bool(scale_invariant))
        a.w_eik, a.w_smooth, a.w_depth, a.w_nl1, a.w_ncos = [float(w) for w in weights]
        a.mask, a.out = mask.data_ptr(), out.data_ptr()
        a.g_rgb, a.g_depth, a.g_normal = [t.data_ptr() for t in grads[:3]]
        a.g_theta = grads[3].data_ptr() if has_eik else None
        a.g_nei = grads[4].data_ptr() if has_eik else None
        _lib.call('msdf_monosdf_loss', C.byref(a), _lib.stream_ptr())
        ctx.has_eik = has_eik
        ctx.save_for_backward(flat)
        ctx.sizes, ctx.shapes = sizes, [t.shape for t in like]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        flat, = ctx.saved_tensors
        scaled = flat * g_out[0]
        res, off = [], 0
        for n, shape in zip(ctx.sizes, ctx.shapes):
            res.append(scaled[off:off + n].view(shape))
            off += n
        res[1] = res[1].reshape(ctx.depth_shape)
        if not ctx.has_eik:
            res += [None, None]
        return tuple(res) + (None,) * 8


# ---------------------------------------------------------------------------
# ray generation (image-mode inputs)
# ---------------------------------------------------------------------------
def camera_rays(uv, pose, intrinsics):
    """uv [n,2], pose [4,4], intrinsics [4,4] -> (ray_dirs [n,3], ray_dirs_cam [n,3], cam_loc [n,3]);
    rend_util.get_camera_params with the pose and with the identity in one launch (no gradients: the
    reference's poses / intrinsics are data)."""
    uv = _need_cuda(uv.detach(), 'uv')
    pose = _need_cuda(pose.detach(), 'pose')
    intrinsics = _need_cuda(intrinsics.detach(), 'intrinsics')
    if pose.shape != (4, 4) or intrinsics.shape != (4, 4):
        raise NotImplementedError('monosdf_amd: 4x4 pose / intrinsics matrices only (quaternion poses are not '
                                  'used on this path)')
    n = uv.shape[0]
    out = torch.empty(3, n, 3, device=uv.device, dtype=torch.float32)
    _lib.call('msdf_camera_rays', _lib.ptr(uv), _lib.ptr(pose), _lib.ptr(intrinsics), n, _lib.ptr(out[0]),
              _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.stream_ptr())
    return out[0], out[1], out[2]
