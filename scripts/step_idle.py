"""GPU idle time inside ONE steady-state training step, from a rocprofv3 kernel trace of bench.py
(`*_kernel_trace.csv`): step length, union of the kernel intervals (all streams), idle time and the largest gaps.

    python scripts/step_idle.py gpurun_out/r04prof/kernel_trace_grid.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'weightnorm_forward' in r['Kernel_Name']]      # two per step
seg = rows[starts[-4]:starts[-2]]
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[starts[-2]]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in seg)
busy, (cs, ce, last), gaps = 0, iv[0], []
for s, e, n in iv[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ce - t0, last, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
    last = n
busy += ce - cs
gaps.append((t1 - ce, ce - t0, last, '(next step)'))
print('step %.1f us, kernels running %.1f us, idle %.1f us (%.1f %%)' % ((t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3,
                                                                          100.0 * (t1 - t0 - busy) / (t1 - t0)))
for g, at, a, b in sorted(gaps, reverse=True)[:8]:
    print('  %6.1f us idle at %7.1f us  between %s and %s' % (g / 1e3, at / 1e3, a, b))
