"""Micro-benchmark of the weight-gradient launch on a synthetic workspace (tuning aid).

    python scripts/bench_wgrad.py [mlp|grid|color] [sweep]

mlp = the 8x256 SDF network of configs[1], grid = the 2x256 SDF network behind the hash grid of configs[2] (input
39 PE + 32 grid features), color = the colour network.  Prints the launch time for the split plan the library picks
(plan.balanced_program) and, with `sweep`, for a table of split counts per item class."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from monosdf_amd import plan as planlib, ops, _lib

which = sys.argv[1] if len(sys.argv) > 1 else 'mlp'
sweep = 'sweep' in sys.argv[2:]
dev = torch.device('cuda')
P_pad = 104448
if which == 'mlp':
    shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
    mp = planlib.build_sdf_plan(shapes, [4], 6, 0, False, 256)
    build, wsfn = planlib.build_sdf_wgrad, planlib.sdf_workspace
elif which == 'grid':
    shapes = [(256, 71), (256, 256), (257, 256)]
    mp = planlib.build_sdf_plan(shapes, [4], 6, 32, True, 256)
    build, wsfn = planlib.build_sdf_wgrad, planlib.sdf_workspace
else:
    P_pad = 100352
    shapes = [(256, 289), (256, 256), (3, 256)]
    mp = planlib.build_color_plan(shapes, 'idr', 4, 256)
    build, wsfn = planlib.build_color_wgrad, planlib.color_workspace
woff, total = wsfn(mp, P_pad)
ws = torch.randn(total, device=dev) * 0.01
feat = torch.randn(P_pad * 256, device=dev) * 0.01
PREC = 0


def run(prog, label, iters=10):
    items = torch.from_numpy(prog.items_bytes()).to(dev)
    wg_map = torch.from_numpy(prog.wg_map()).to(dev)
    part = torch.empty(prog.part_f + 64, device=dev)
    st = _lib.stream_ptr()
    macs = sum(it['wx'] * it['wy'] for it in prog.items) * P_pad

    def once():
        _lib.call('msdf_wgrad', _lib.ptr(items), _lib.ptr(wg_map), wg_map.numel() // 2, _lib.ptr(part), P_pad, PREC, _lib.ptr(ws), _lib.ptr(feat), st)
    once(); once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    for it in prog.items:
        it.setdefault('stages_per_wg', -(-(P_pad // 32) // it['n_splits']))
    model = planlib.model_launch_us(prog)
    print('%-60s WGs %4d  part %6.1f MB  %.3f ms  %.1f TFLOP/s (padded tiles)  model %.3f + reduce %.3f' % (
        label, wg_map.numel() // 2, 4e-6 * prog.part_f, ms, 2 * macs / ms / 1e9, 1e-3 * model[0], 1e-3 * model[1]), flush=True)
    return ms


prog = planlib.balanced_program(build, mp, P_pad)
cls = sorted(set((it['weight'], it['n_splits']) for it in prog.items))
print(which, 'items (wx, wy, weight, splits):', [(it['wx'], it['wy'], it['weight'], it['n_splits']) for it in prog.items])
run(prog, 'library plan %s' % cls)
if sweep:
    weights = sorted(set(it['weight'] for it in prog.items), reverse=True)
    if which == 'mlp':
        for S in (48, 56, 59, 61, 64, 68, 72, 85):
            run(build(mp, P_pad, lambda w, S=S: S), 'all items S=%d' % S)
    for Sw in (40, 51, 60, 64, 68, 76, 85, 102, 128, 153, 170):
        for f in (1.0, 0.6, 0.4):
            for Sc in (4, 16):
                def fn(w, Sw=Sw, f=f, Sc=Sc):
                    if w >= 1.0:
                        return Sw
                    if w == 0.0:
                        return Sc
                    return max(1, int(round(Sw * f)))
                run(build(mp, P_pad, fn), 'wide %d, narrow x%.1f, colsum %d' % (Sw, f, Sc), iters=5)


def calibrate():
    """Per item class: time of a launch of that item alone with 256 / 512 / 1024 splits (1 / 2 / 4 rounds of the 256 CUs)
    -> per-stage time and fixed cost of a workgroup (the constants of plan.WgradProgram.stage_us / WG_FIXED_US)."""
    import numpy as np
    seen = set()
    n_stages = P_pad // 32
    for S in (256, 512, 1024):
        prog = build(mp, P_pad, lambda w, S=S: S)
        items = torch.from_numpy(prog.items_bytes()).to(dev)
        part = torch.empty(prog.part_f + 64, device=dev)
        for i, it in enumerate(prog.items):
            key = (it['wx'], it['wy'], it['colsum_off'] >= 0, it['vrow_off'] >= 0)
            if (S, key) in seen:
                continue
            seen.add((S, key))
            wg = torch.tensor([[i, s] for s in range(S)], dtype=torch.int32, device=dev).reshape(-1)
            st = _lib.stream_ptr()

            def once():
                _lib.call('msdf_wgrad', _lib.ptr(items), _lib.ptr(wg), S, _lib.ptr(part), P_pad, PREC, _lib.ptr(ws), _lib.ptr(feat), st)
            once(); once(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                once()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print('calib S=%4d item wx=%3d wy=%3d colsum=%d vrow=%d: %.1f us  (%.2f us per stage of the %d per CU)' % (
                S, key[0], key[1], key[2], key[3], 1e3 * ms, 1e3 * ms / (n_stages / 256.0), n_stages // 256), flush=True)


if 'target' in sys.argv[2:]:
    for D in (80, 100, 115, 130, 150, 165, 180, 200, 215, 230, 245, 260, 280, 300, 325, 350, 375, 400, 435, 470, 510, 550, 600, 650, 720, 800):
        pr = planlib.balanced_program(build, mp, P_pad, target_us=D)
        run(pr, 'target %4d us %s' % (D, sorted(set((it['weight'], it['n_splits']) for it in pr.items))), iters=8)
if 'calib' in sys.argv[2:]:
    calibrate()


def fine():
    """Same process, interleaved repetitions: the chooser's plan against plans with the wide class pinned to a range of
    split counts (the other classes by the chooser's rule for that workgroup duration)."""
    import numpy as np
    cost = planlib.WgradProgram.CLASS_COST_US
    n_stages = P_pad // 32
    rng = {'mlp': range(54, 76), 'grid': list(range(76, 92)) + list(range(106, 118)), 'color': range(70, 100)}[which]
    progs = [('library', planlib.balanced_program(build, mp, P_pad))]
    for sw in rng:
        D = cost[1.0][1] + cost[1.0][0] * -(-n_stages // sw)
        progs.append(('wide %d' % sw, planlib.balanced_program(build, mp, P_pad, target_us=D)))
    res = {n: [] for n, _ in progs}
    for rep in range(3):
        for n, pr in progs:
            res[n].append(run(pr, '%s %s' % (n, sorted(set((it['weight'], it['n_splits']) for it in pr.items))), iters=6))
    print('--- median of 3 ---')
    for n, _ in progs:
        print('%-12s %.3f ms' % (n, float(np.median(res[n]))))


if 'fine' in sys.argv[2:]:
    fine()
