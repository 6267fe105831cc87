"""Launch census of ONE steady-state training step from a rocprofv3 kernel trace of bench.py
(`*_kernel_trace.csv`): every launch in order with its start offset and duration, PyTorch's own marked.

    python scripts/step_launches.py profiles/r03_kernel_trace_fp32.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'weightnorm_forward' in n]      # two per step (SDF, colour network)
seg = rows[starts[-4]:starts[-2]]
t0 = int(seg[0]['Start_Timestamp'])
ours = lambda n: ('msdf' in n or 'smp_' in n or 'hg_' in n or 'hb2_' in n or 'hb_' in n)
n_torch = 0
for r in seg:
    n = r['Kernel_Name']
    mine = ours(n)
    n_torch += 0 if mine else 1
    print('%9.1f us +%8.1f  %s%s' % ((int(r['Start_Timestamp']) - t0) / 1e3,
                                      (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                      n.split('(')[0][-64:], '' if mine else '    <-- PyTorch'))
print('%d launches in the step, %d of them PyTorch\'s' % (len(seg), n_torch))
