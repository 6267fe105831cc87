"""Per-kernel averages of rocprofv3 --pmc passes (one directory per pass) as JSON.

    python scripts/pmc_sum.py <dir with pass sub-directories> [kernel-name substring] > summary.json

FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); FETCH/WRITE_SIZE are in KiB.
"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if sub in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k][r['Counter_Name']].add((f, r['Dispatch_Id']))
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if sub in k:
            dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k in sorted(acc):
    c = {n: v / len(disp[k][n]) for n, v in acc[k].items()}
    row = {'counters_avg_per_launch': c, 'avg_ns_under_pmc': sum(dur[k]) / max(1, len(dur[k])), 'launches': len(dur[k])}
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        row['hbm_bytes_per_launch_corrected'] = (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        row['mfma_pipe_util'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0)
    out[k] = row
print(json.dumps(out, indent=1))
