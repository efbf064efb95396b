"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel family (tuning aid).
usage: python tools/pmc_summary.py gpurun_out/pmc1/run_counter_collection.csv"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(k_\w+)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:36]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
names = sorted({c for v in agg.values() for c in v})
print('%-38s %8s ' % ('kernel', 'launches') + ' '.join('%22s' % n for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0))[:14]:
    print('%-38s %8d ' % (k, len(disp[k])) + ' '.join('%22.4g' % v.get(n, 0) for n in names))
