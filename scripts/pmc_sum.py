"""Sums rocprofv3 --pmc counter CSVs per kernel (tuning aid): python scripts/pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if sub in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in acc:
    n = len(cnt[k])
    print(k, 'dispatches', n)
    for c, v in sorted(acc[k].items()):
        print('   %-32s %.4g per launch' % (c, v / n))
