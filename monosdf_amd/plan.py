"""Host-side geometry of the fused MLP kernels: slot layouts, weight-pack rules,
workspace layout, weight-gradient work items and reduction rules.

A "slot" is one lane-register position of the kernels' activation tiles (16 slots
per tile, see csrc/mlp_core.h).  The maps built here say which original row /
column of a reference weight matrix (reference: code/model/network.py:43-75 for
the SDF network's dims/skip logic, 344-384 for the colour network) each slot
carries; padding slots map to -1 and get zero weights.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

MT = _lib.MAX_TILES


def _ceil16(n):
    return (n + 15) // 16


def _even(n):
    return (n + 1) & ~1


def _pad_k(k):
    """K of a packed operand: 12..16 tiles share the unrolled K=16 kernel path."""
    return 16 if 12 <= k <= 16 else k


def _ident_map(n_valid, n_tiles, shift=0):
    m = np.full(16 * n_tiles, -1, dtype=np.int32)
    m[:n_valid] = np.arange(n_valid, dtype=np.int32) + shift
    return m


class MlpPlan:
    """Everything static about one network: ctypes plan, pack rules, maps, sizes."""

    def __init__(self):
        self.plan = _lib.Plan()
        self.rules = []          # PackRule per pack unit
        self.maps = []           # list of int32 arrays, concatenated at finalise
        self._map_off = 0
        self.w_shapes = []       # original (rows, cols) per weight tensor
        self.unit_weight = []    # pack unit -> index of the weight tensor
        self.wpack_f4 = 0
        self.bpack_f = 0
        self.rowmaps = []        # per unit: (offset, array)
        self.colmaps = []

    def add_map(self, arr):
        off = self._map_off
        self.maps.append(np.ascontiguousarray(arr, dtype=np.int32))
        self._map_off += len(arr)
        return off

    def add_unit(self, widx, rowmap, colmap, scale, skip_tile=-1):
        u = len(self.rules)
        ot, kt = len(rowmap) // 16, len(colmap) // 16
        if ot > MT or kt > MT:
            raise RuntimeError('monosdf_amd: layer with %d/%d tiles exceeds the %d-tile register file layout'
                               % (kt, ot, MT))
        L = self.plan.layer[u]
        L.kt, L.ot, L.ktp, L.otp = kt, ot, _pad_k(kt), _pad_k(ot)
        L.wf_off = self.wpack_f4
        self.wpack_f4 += _even(ot) * L.ktp * 64
        L.wb_off = self.wpack_f4
        self.wpack_f4 += _even(kt) * L.otp * 64
        L.bias_off = self.bpack_f
        self.bpack_f += 16 * ot
        L.skip_tile = skip_tile
        r = _lib.PackRule()
        rows, cols = self.w_shapes[widx]
        r.rows, r.cols = rows, cols
        r.rowmap_off = self.add_map(rowmap)
        r.colmap_off = self.add_map(colmap)
        r.scale = scale
        self.rules.append(r)
        self.unit_weight.append(widx)
        self.rowmaps.append((r.rowmap_off, rowmap))
        self.colmaps.append((r.colmap_off, colmap))
        return u

    def build_b16(self, planes=2):
        """Second plan for the bf16-split kernels (planes = 2: bf16x3, 3: bf16x6): ktp / otp = K-block counts (32 slots),
        wf_off / wb_off = 16-byte offsets of the plane packs; everything else identical."""
        cache = self.__dict__.setdefault('_plans16', {})
        if planes in cache:
            return cache[planes][0]
        p16 = _lib.Plan()
        C.memmove(C.byref(p16), C.byref(self.plan), C.sizeof(_lib.Plan))
        p16.precision = {2: 1, 3: 2}[planes]            # MSDF_PRECISION_BF16X3 / MSDF_PRECISION_BF16X6
        off = 0
        for u in range(self.plan.n_layers):
            L = p16.layer[u]
            # K blocks of 32 slots; 6..8 blocks share the unrolled 8-block path (zero weights in the padding)
            kb = lambda t: 8 if 6 <= (t + 1) // 2 <= 8 else (t + 1) // 2
            L.ktp, L.otp = kb(L.kt), kb(L.ot)
            L.wf_off = off
            off += (L.ot + 3) // 4 * 4 * L.ktp * planes * 64      # out tiles padded (the kernels stage whole chunks)
            L.wb_off = off
            off += (L.kt + 3) // 4 * 4 * L.otp * planes * 64
        cache[planes] = (p16, off)
        return p16

    def wpack16_units(self, planes=2):
        """16-byte units of the plane packs of build_b16(planes)."""
        self.build_b16(planes)
        return self._plans16[planes][1]

    def finalise(self):
        # flat weight / bias buffers are the concatenation of the original tensors
        w_off = np.cumsum([0] + [r * c for r, c in self.w_shapes])
        b_off = np.cumsum([0] + [r for r, _ in self.w_shapes])
        self.w_offsets, self.b_offsets = w_off, b_off
        for u, r in enumerate(self.rules):
            r.w_off = int(w_off[self.unit_weight[u]])
            r.b_off = int(b_off[self.unit_weight[u]])
        self.n_w = int(w_off[-1])
        self.n_b = int(b_off[-1])
        self.plan.n_layers = len(self.rules)
        self.maps_np = np.concatenate(self.maps) if self.maps else np.zeros(1, np.int32)
        self.rules_np = np.frombuffer(b''.join(bytes(r) for r in self.rules), dtype=np.uint8).copy()


def build_sdf_plan(w_shapes, skip_in, n_freqs, aux_cols, aux_active, feature_size, d_in=3):
    """Plan of ImplicitNetwork / ImplicitNetworkGrid.

    w_shapes: [(out_l, in_l)] of the reference's nn.Linear layers; aux_cols: width of the
    hash-feature block in the network input (0 for ImplicitNetwork, 32 for the grid class);
    aux_active: whether those features are non-zero (use_grid_feature)."""
    mp = MlpPlan()
    mp.w_shapes = list(w_shapes)
    n = len(w_shapes)
    pe_dim = d_in + 2 * d_in * n_freqs if n_freqs > 0 else d_in
    d0 = pe_dim + aux_cols
    assert w_shapes[0][1] == d0, (w_shapes[0], d0)
    e_tiles = 3
    if pe_dim > 48:
        raise RuntimeError('monosdf_amd: positional encoding wider than 48 is not supported')
    aux_tiles = _ceil16(aux_cols) if aux_active else 0
    if aux_tiles > 2:
        raise RuntimeError('monosdf_amd: more than 32 grid features are not supported')
    in0_tiles = e_tiles + aux_tiles
    in0_map = np.full(16 * in0_tiles, -1, dtype=np.int32)
    in0_map[:pe_dim] = np.arange(pe_dim)
    if aux_tiles:
        in0_map[48:48 + aux_cols] = pe_dim + np.arange(aux_cols)
    P = mp.plan
    P.e_tiles, P.aux_tiles, P.n_freqs = e_tiles, aux_tiles, n_freqs
    feat_tiles = _ceil16(feature_size)
    P.feat_tiles = feat_tiles
    P.sdf_slot = 16 * feat_tiles
    hpre = qpre = abpre = 0
    prev_ot, prev_out = None, None
    for l, (out, inn) in enumerate(w_shapes):
        last = (l == n - 1)
        if last:
            assert out == 1 + feature_size
            rowmap = np.full(16 * (feat_tiles + 1), -1, dtype=np.int32)
            rowmap[:feature_size] = 1 + np.arange(feature_size)
            rowmap[16 * feat_tiles] = 0
        else:
            rowmap = _ident_map(out, _ceil16(out))
        skip_tile = -1
        scale = 1.0
        if l == 0:
            colmap = in0_map.copy()
        elif l in skip_in:
            assert inn == prev_out + d0
            skip_tile = prev_ot
            colmap = np.concatenate([_ident_map(prev_out, prev_ot),
                                     np.where(in0_map >= 0, in0_map + prev_out, -1).astype(np.int32)])
            scale = 1.0 / math.sqrt(2.0)
        else:
            assert inn == prev_out, (l, inn, prev_out)
            colmap = _ident_map(prev_out, prev_ot)
        u = mp.add_unit(l, rowmap, colmap, scale, skip_tile)
        L = P.layer[u]
        L.hpre, L.qpre, L.abpre = hpre, qpre, abpre
        if not last:
            hpre += 16 * L.ot
            qpre += 16 * L.kt
        abpre += 16 * L.ot
        prev_ot, prev_out = L.ot, out
    P.hsum, P.qsum, P.absum = hpre, qpre, abpre
    P.wsdf_off = mp.bpack_f
    P.out_rows = 1
    mp.bpack_f += 16 * P.layer[n - 1].kt
    P.mode, P.out_act = 0, 0
    mp.in0_tiles = in0_tiles
    mp.kind = 'sdf'
    mp.finalise()
    return mp


def build_color_plan(w_shapes, mode, n_freqs_view, feature_size, code_cols=0, out_relu=False):
    """Plan of RenderingNetwork (mode 'idr' or 'nerf'); pack units 0/1 split the first layer."""
    mp = MlpPlan()
    mp.w_shapes = list(w_shapes)
    n = len(w_shapes)
    pev = 3 + 6 * n_freqs_view if n_freqs_view > 0 else 3
    lead = (3 + pev + 3) if mode == 'idr' else pev            # columns before the feature block
    if lead > 48:
        raise RuntimeError('monosdf_amd: view encoding too wide')
    assert w_shapes[0][1] == lead + feature_size + code_cols, (w_shapes[0], lead, feature_size, code_cols)
    feat_tiles = _ceil16(feature_size)
    aux_tiles = _ceil16(code_cols)
    if aux_tiles > 2:
        raise RuntimeError('monosdf_amd: per-image code wider than 32 is not supported')
    P = mp.plan
    P.e_tiles, P.aux_tiles, P.n_freqs, P.feat_tiles = 3, aux_tiles, n_freqs_view, feat_tiles
    P.mode = 1 if mode == 'idr' else 0
    P.out_act = 1 if out_relu else 0
    P.sdf_slot = 0
    out0 = w_shapes[0][0]
    row0 = _ident_map(out0, _ceil16(out0))
    feat_map = _ident_map(feature_size, feat_tiles, shift=lead)
    misc_map = np.full(16 * (3 + aux_tiles), -1, dtype=np.int32)
    misc_map[:lead] = np.arange(lead)
    if aux_tiles:
        misc_map[48:48 + code_cols] = lead + feature_size + np.arange(code_cols)
    u0 = mp.add_unit(0, row0, feat_map, 1.0)
    u1 = mp.add_unit(0, row0, misc_map, 1.0)
    P.layer[u0].abpre = P.layer[u1].abpre = 0
    P.layer[u0].hpre = P.layer[u1].hpre = 0
    abpre, hpre = 16 * P.layer[u0].ot, 0
    prev_out, prev_ot = out0, P.layer[u0].ot
    for l in range(1, n):
        out, inn = w_shapes[l]
        assert inn == prev_out
        rowmap = _ident_map(out, _ceil16(out))
        u = mp.add_unit(l, rowmap, _ident_map(prev_out, prev_ot), 1.0)
        L = P.layer[u]
        L.abpre, L.hpre = abpre, hpre
        abpre += 16 * L.ot
        hpre += 16 * L.kt
        prev_out, prev_ot = out, L.ot
    P.hsum, P.absum, P.qsum = hpre, abpre, 0
    # the 3 colour rows of the output layer in input-slot order: the kernels form rgb as dot products over the last
    # hidden activation (and its adjoint as an outer product) instead of a 16-row matrix product with 13 rows of zeros
    assert w_shapes[n - 1][0] == 3
    P.wsdf_off = mp.bpack_f
    P.out_rows = 3
    mp.bpack_f += 3 * 16 * P.layer[u].kt
    mp.misc_tiles = 3 + aux_tiles
    mp.lead = lead
    mp.kind = 'color'
    mp.finalise()
    return mp


# ---------------------------------------------------------------------------
# workspace layouts (float offsets for a given P_pad)
# ---------------------------------------------------------------------------
def sdf_workspace(mp, P_pad):
    P = mp.plan
    n = P.n_layers
    sizes = [('H', P.hsum * P_pad), ('PM', P.hsum * P_pad), ('IN0', 16 * mp.in0_tiles * P_pad),
             ('QB', P.qsum * P_pad), ('AB', P.absum * P_pad),
             ('GSDF', P_pad), ('QLAST', 16 * P.layer[n - 1].kt * P_pad)]
    off, total = {}, 0
    for k, s in sizes:
        off[k] = total
        total += (s + 63) & ~63
    return off, total


def color_workspace(mp, P_pad):
    P = mp.plan
    sizes = [('H', P.hsum * P_pad), ('MISC', 16 * mp.misc_tiles * P_pad), ('AB', P.absum * P_pad)]
    off, total = {}, 0
    for k, s in sizes:
        off[k] = total
        total += (s + 63) & ~63
    return off, total


# ---------------------------------------------------------------------------
# weight-gradient work items + reduction rules
# ---------------------------------------------------------------------------
class WgradProgram:
    """Work items (absolute device addresses are filled in per call), reduce rules and the
    workgroup -> (item, split) map.  Every item cuts its point range into its own number of splits,
    chosen so that all workgroups of a launch run about equally long and fill the 256 CUs."""

    def __init__(self, split_fn):
        self.split_fn = split_fn
        self.items = []      # dicts: x, y, v = (buffer name, float offset); part/colsum/vrow offsets; n_splits
        self.rules = []      # ReduceRule
        self.part_f = 0

    def alloc(self, n):
        off = self.part_f
        self.part_f += n
        return off

    @staticmethod
    def weight(wx, wy):
        """Class of an item = the wave grid msdf_wgrad_k runs it on (csrc/wgrad.hip), named by the relative duration of one
        32-point stage: 1.0 wide (2 x 4 waves of 128 x 64), 0.6 / 0.5 / 0.3 the 8 x 1 grids of <= 128 / 96 / 64 columns,
        0.2 thin (<= 32 rows: 1 x 8 waves of 32 x 32), 0.0 column sums only."""
        if wy == 0:
            return 0.0
        if wx <= 32 and wy > 32:
            return 0.2
        if wy <= 128:
            return {2: 0.3, 3: 0.5, 4: 0.6}[max(2, (wy + 31) // 32)]
        return 1.0

    # class -> (microseconds per 32-point stage of a workgroup, fixed cost of a workgroup), MI355X: one item alone over
    # 1 / 2 / 4 rounds of workgroups (scripts/bench_wgrad.py calib), then fitted to 80 timed split plans of the three
    # networks (`target` sweeps; profiles/r04_wgrad_splits.md: rms error 3 % for workgroups of <= 500 us)
    CLASS_COST_US = {1.0: (8.4, 6.0), 0.6: (5.2, 6.0), 0.5: (4.1, 5.0), 0.3: (3.0, 8.0), 0.2: (2.2, 6.0),
                     0.0: (1.45, 2.0)}
    CORUN_SLOWDOWN = {0.5: 1.45, 0.6: 1.45}
    N_CU = 256
    REDUCE_US_PER_MB = 0.54          # msdf_reduce_k reads the partial blocks at ~1.85 TB/s

    def splits(self, wx, wy):
        return self.split_fn(self.weight(wx, wy))

    def add_item(self, x, x_ld, wx, y, y_ld, wy, part_off, n_splits, colsum_off=-1, v=None, vrow_off=-1):
        assert wx % 16 == 0 and wy % 16 == 0 and wx <= 256 and wy <= 256, (wx, wy)
        self.items.append(dict(x=x, y=y, v=v, x_ld=x_ld, y_ld=y_ld, wx=wx, wy=wy, part_off=part_off,
                               colsum_off=colsum_off, vrow_off=vrow_off, n_splits=n_splits,
                               weight=self.weight(wx, wy)))

    def add_rule(self, part_off, n_blocks, wx, wy, rowmap_off, colmap_off, dst_off, dst_ld, scale,
                 fixed_row=0):
        r = _lib.ReduceRule()
        r.part_off, r.dst_off, r.n_blocks, r.wx, r.wy = part_off, dst_off, n_blocks, wx, wy
        r.rowmap_off, r.colmap_off, r.dst_ld, r.fixed_row, r.scale = rowmap_off, colmap_off, dst_ld, fixed_row, scale
        self.rules.append(r)

    def writes_every_element(self, n_total, maps_np):
        """True when the reduce rules together store to every element of the flat gradient [n_total] (then the caller
        need not zero-fill it first).  Mirrors msdf_reduce_k's addressing: dst_off + row * dst_ld + col, rows / columns
        through the slot maps, padding slots (-1) skipped."""
        hit = np.zeros(n_total, dtype=bool)
        for r in self.rules:
            rows = maps_np[r.rowmap_off:r.rowmap_off + r.wx] if r.rowmap_off >= 0 else np.full(r.wx, r.fixed_row)
            cols = maps_np[r.colmap_off:r.colmap_off + r.wy] if r.colmap_off >= 0 else np.zeros(r.wy, dtype=np.int64)
            rows, cols = rows[rows >= 0].astype(np.int64), cols[cols >= 0].astype(np.int64)
            if rows.size == 0 or cols.size == 0:
                continue
            idx = (r.dst_off + rows[:, None] * r.dst_ld + cols[None, :]).reshape(-1)
            if idx.min() < 0 or idx.max() >= n_total:
                return False
            hit[idx] = True
        return bool(hit.all())

    def rules_bytes(self):
        return np.frombuffer(b''.join(bytes(r) for r in self.rules), dtype=np.uint8).copy()

    def wg_duration_us(self, it):
        """Modelled duration of one workgroup of item `it` (all its splits are equally long up to one stage)."""
        tau, t0 = self.CLASS_COST_US[it['weight']]
        return t0 + tau * it.get('stages_per_wg', 1)

    def wg_map(self):
        """int32 (item, split) pairs; long workgroups first so the tail of the launch is made of short ones."""
        order = sorted(range(len(self.items)), key=lambda i: -self.wg_duration_us(self.items[i]))
        pairs = [(i, s) for i in order for s in range(self.items[i]['n_splits'])]
        return np.asarray(pairs, dtype=np.int32).reshape(-1)

    BUFFERS = ('ws', 'feat')      # operand buffers of a launch, in the order msdf_wgrad takes their base pointers

    def items_bytes(self):
        """The work-item table (no device addresses in it: built once, msdf_wgrad takes the buffers' base pointers)."""
        out = []
        buf = lambda ref: self.BUFFERS.index(ref[0])
        for it in self.items:
            w = _lib.WgradItem()
            bx = buf(it['x'])
            w.x = it['x'][1]
            by, w.y = (buf(it['y']), it['y'][1]) if it['y'] is not None else (bx, it['x'][1])
            bv, w.v = (buf(it['v']), it['v'][1]) if it['v'] is not None else (0xff, 0)
            w.bufs = bx | (by << 8) | (bv << 16)
            w.part_off, w.colsum_off, w.vrow_off = it['part_off'], it['colsum_off'], it['vrow_off']
            w.x_ld, w.y_ld, w.wx, w.wy = it['x_ld'], it['y_ld'], it['wx'], it['wy']
            w.n_splits = it['n_splits']
            out.append(bytes(w))
        return np.frombuffer(b''.join(out), dtype=np.uint8).copy()


def _makespan_us(durations, n_cu):
    """In-order dispatch of workgroups (longest first) onto n_cu CUs that hold one workgroup each."""
    import heapq
    free = [0.0] * n_cu
    for d in durations:
        heapq.heapreplace(free, free[0] + d)
    return max(free)


_SPLIT_CACHE = {}
# the second evaluation of a candidate plan slows EVERY class but the widest by 30 % (0 switches that off: comparison
# runs).  Stand-alone the colour network's launch is faster without it (0.27 against 0.33 ms: 96 instead of 77 splits of
# its wide items, 314 workgroups), but inside the training step -- where that launch runs beside the scatter and the
# SDF launch -- the plan that fits ONE round of the 256 CUs measured better: configs[2] 3.37 against 3.39 ms per step
# (three runs each on one box), configs[1] the same
import os as _os
_ALL_NARROW_SLOW = _os.environ.get('MSDF_WGRAD_ALL_NARROW_SLOW', '1') != '0'


def model_launch_us(prog):
    """Modelled duration of a weight-gradient launch + its reduction (the objective of choose_splits)."""
    durs = sorted((prog.wg_duration_us(it) for it in prog.items for _ in range(it['n_splits'])), reverse=True)
    part = sum(it['n_splits'] * it['wx'] * max(it['wy'], 1) for it in prog.items)
    return _makespan_us(durs, WgradProgram.N_CU), WgradProgram.REDUCE_US_PER_MB * 4e-6 * part


def choose_splits(classes, n_stages, part_floats, target_us=None):
    """Split count per item class for a launch of the given items.

    classes: {class: number of items}, part_floats: {class: floats of partial sums one split of ALL items of the class
    writes}.  The wide items are cut into workgroups of modelled duration D, every other class into workgroups of at
    most D / 2 (short workgroups are what fills the end of the launch; they are dispatched last); D is the candidate
    that minimises the modelled launch -- in-order dispatch, longest first, one workgroup per CU -- plus the time
    msdf_reduce_k needs for the partial sums.  Round 3 used one split count for every item, sized for the 21 items of
    the 8 x 256 network (61 splits: 5 rounds of workgroups); the 7 items of the network behind the hash grid then made
    1.67 rounds."""
    key = (tuple(sorted(classes.items())), n_stages, tuple(sorted(part_floats.items())), target_us)
    if key in _SPLIT_CACHE:
        return _SPLIT_CACHE[key]
    cost = WgradProgram.CLASS_COST_US
    top = max(classes)
    best = None
    # workgroups longer than ~500 us are outside what the model was fitted on (and measured 10-20 % slower than it says:
    # the narrow classes are latency-bound when they run for that long beside wide workgroups).  Inside that range the
    # model ranks plans as the GPU does -- a same-process sweep of the wide split count (scripts/bench_wgrad.py fine,
    # profiles/r04_wgrad_splits.md): the cliffs where the wide workgroups spill into one more round of the 256 CUs and
    # the best plan of each of the three networks are where it puts them
    tau_top, t0_top = cost[top]
    if target_us is None:
        # every split count of the widest class whose workgroups last 100 ... 500 us, each with its own duration as D
        s_lo = max(1, math.ceil(n_stages * tau_top / (500.0 - t0_top)))
        s_hi = min(n_stages, max(s_lo, math.ceil(n_stages * tau_top / (100.0 - t0_top))))
        cands = sorted({t0_top + tau_top * -(-n_stages // sw) for sw in range(s_lo, s_hi + 1)})
    else:
        cands = [float(target_us)]
    for D in cands:
        S, part = {}, 0.0
        for c, n_items in classes.items():
            tau, t0 = cost[c]
            d = D if c == top else 0.5 * D
            S[c] = int(min(n_stages, max(1, math.ceil(n_stages * tau / max(d - t0, tau)))))
            part += part_floats.get(c, 0) * S[c]
        t = 0.0
        for corun in (False, True):
            # as calibrated, and with the 96- / 128-column classes at the per-stage time they measure while they run
            # BESIDE wide workgroups (6 us instead of 4.1: a plan of the network behind the hash grid whose wide
            # workgroups fill 0.7 of a round, with the 80-column workgroups on the other CUs from the start, measured
            # 0.59-0.63 ms where the model said 0.50; the 48-column and thin classes show no such effect): a plan has
            # to be good either way
            durs = []
            for c, n_items in classes.items():
                tau, t0 = cost[c]
                if corun and c in WgradProgram.CORUN_SLOWDOWN:
                    tau *= WgradProgram.CORUN_SLOWDOWN[c]
                elif corun and c != top and _ALL_NARROW_SLOW:
                    tau *= 1.3
                durs += [t0 + tau * -(-n_stages // S[c])] * (n_items * S[c])
            durs.sort(reverse=True)
            t = max(t, _makespan_us(durs, WgradProgram.N_CU))
        t += WgradProgram.REDUCE_US_PER_MB * 4e-6 * part
        if best is None or t < best[0] - 1e-9:
            best = (t, S)
    _SPLIT_CACHE[key] = best[1]
    return best[1]


def balanced_program(build, mp, P_pad, splits=None, target_us=None):
    """The weight-gradient program of a network with its split counts chosen by choose_splits (or given per class;
    target_us: the workgroup duration instead of the modelled optimum -- tuning runs)."""
    if splits is None:
        probe = build(mp, P_pad, lambda w: 1)
        classes, part = {}, {}
        for it in probe.items:
            classes[it['weight']] = classes.get(it['weight'], 0) + 1
            part[it['weight']] = part.get(it['weight'], 0) + it['wx'] * max(it['wy'], 1)
        splits = choose_splits(classes, max(1, P_pad // 32), part, target_us)
    prog = build(mp, P_pad, lambda w: splits[w])
    n_stages = max(1, P_pad // 32)
    for it in prog.items:
        it['stages_per_wg'] = -(-n_stages // it['n_splits'])
    return prog


def _col_parts(L, in0_tiles):
    """Column ranges (slot start, width) of a layer's input, each <= 256 wide."""
    if L.skip_tile >= 0:
        return [(0, 16 * L.skip_tile), (16 * L.skip_tile, 16 * in0_tiles)]
    return [(0, 16 * L.kt)]


def build_sdf_wgrad(mp, P_pad, split_fn):
    """Items/rules of d W_l, d b_l for the SDF network.  Buffer names refer to the workspace
    ('ws', float offsets from sdf_workspace)."""
    P = mp.plan
    n = P.n_layers
    woff, _ = sdf_workspace(mp, P_pad)
    prog = WgradProgram(split_fn)
    dW_off, dB_off = mp.w_offsets, mp.b_offsets + mp.n_w     # gradient buffer: all dW then all db
    for l in range(n):
        L = P.layer[l]
        rows, cols = mp.w_shapes[l]
        rm_off = mp.rowmaps[l][0]
        cm_off = mp.colmaps[l][0]
        scale = float(mp.rules[l].scale)
        last = (l == n - 1)
        wx = 16 * (P.feat_tiles if last else L.ot)
        x_ld = 16 * L.ot
        ab = ('ws', woff['AB'] + L.abpre * P_pad)
        parts = _col_parts(L, mp.in0_tiles)
        S0 = prog.splits(wx, parts[0][1])               # splits of the item that also carries the bias sums
        colsum_off = prog.alloc(S0 * wx)
        for pi, (c0, w) in enumerate(parts):
            S = prog.splits(wx, w)
            # input activation of this layer for the a-bar term
            if l == 0 or (L.skip_tile >= 0 and pi == 1):
                y2, y2_ld = ('ws', woff['IN0']), 16 * mp.in0_tiles
            else:
                Lp = P.layer[l - 1]
                y2, y2_ld = ('ws', woff['H'] + Lp.hpre * P_pad), 16 * Lp.ot
            nterms = 1 if last else 2
            part = prog.alloc(nterms * S * wx * w)
            t2 = part
            if not last:
                pm = ('ws', woff['PM'] + L.hpre * P_pad)
                qb = ('ws', woff['QB'] + L.qpre * P_pad + c0)
                prog.add_item(pm, x_ld, wx, qb, 16 * L.kt, w, part, S)
                t2 = part + S * wx * w
            vrow_off, v = -1, None
            if last:
                # sdf row of the output layer: sum_p gsdf[p] h[p][:]  (+ colsum of QLAST)
                Sq = prog.splits(w, 0)
                vrow_off = prog.alloc((S + Sq) * w)
                v = ('ws', woff['GSDF'])
                prog.add_item(('ws', woff['QLAST']), 16 * L.kt, w, None, 0, 0, 0, Sq, colsum_off=vrow_off + S * w)
                prog.add_rule(vrow_off, S + Sq, 1, w, -1, cm_off, int(dW_off[l]), cols, scale,
                              fixed_row=int(mp.rowmaps[l][1][P.sdf_slot]))
            prog.add_item(ab, x_ld, wx, y2, y2_ld, w, t2, S, colsum_off=(colsum_off if pi == 0 else -1),
                          v=v, vrow_off=vrow_off)
            prog.add_rule(part, nterms * S, wx, w, rm_off, cm_off + c0, int(dW_off[l]), cols, scale)
        prog.add_rule(colsum_off, S0, wx, 1, rm_off, -1, int(dB_off[l]), 1, 1.0)
        if last:
            # bias of the sdf row: column sums of the sdf tile of a-bar
            Sc = prog.splits(16, 0)
            cs = prog.alloc(Sc * 16)
            prog.add_item(('ws', woff['AB'] + L.abpre * P_pad + 16 * P.feat_tiles), x_ld, 16, None, 0, 0, 0, Sc,
                          colsum_off=cs)
            prog.add_rule(cs, Sc, 16, 1, rm_off + 16 * P.feat_tiles, -1, int(dB_off[l]), 1, 1.0)
    return prog


def build_color_wgrad(mp, P_pad, split_fn):
    """Items/rules for the colour network.  Buffers: 'ws' (colour workspace) and 'feat' (SDF features)."""
    P = mp.plan
    nu = P.n_layers
    woff, _ = color_workspace(mp, P_pad)
    prog = WgradProgram(split_fn)
    dW_off, dB_off = mp.w_offsets, mp.b_offsets + mp.n_w
    # first layer: units 0 (feature columns) and 1 (misc columns) share a-bar_0
    for u in range(nu):
        L = P.layer[u]
        widx = mp.unit_weight[u]
        rows, cols = mp.w_shapes[widx]
        wx = 16 * L.ot
        ab = ('ws', woff['AB'] + L.abpre * P_pad)
        if u == 0:
            y, y_ld = ('feat', 0), 16 * L.kt
        elif u == 1:
            y, y_ld = ('ws', woff['MISC']), 16 * mp.misc_tiles
        else:
            y, y_ld = ('ws', woff['H'] + L.hpre * P_pad), 16 * L.kt
        w = 16 * L.kt
        S = prog.splits(wx, w)
        part = prog.alloc(S * wx * w)
        cs = -1
        if u != 1:
            cs = prog.alloc(S * wx)
        prog.add_item(ab, wx, wx, y, y_ld, w, part, S, colsum_off=cs)
        prog.add_rule(part, S, wx, w, mp.rowmaps[u][0], mp.colmaps[u][0], int(dW_off[widx]), cols, 1.0)
        if cs >= 0:
            prog.add_rule(cs, S, wx, 1, mp.rowmaps[u][0], -1, int(dB_off[widx]), 1, 1.0)
    return prog
