"""Summarise a rocprofv3 kernel trace CSV by kernel and grid shape (tuning aid).
usage: python tools/trace_summary.py gpurun_out/prof/run_kernel_trace.csv [steps]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    m = re.search(r'(k_\w+)', r['Kernel_Name'])
    short = m.group(1) if m else r['Kernel_Name'][:40]
    key = (short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = collections.defaultdict(float)
cnt = collections.defaultdict(int)
for k, v in agg.items():
    tot[k[0]] += sum(v)
    cnt[k[0]] += len(v)
print('kernel time per step (ms), launches per step')
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:16]:
    print('  %-44s %8.2f %8.1f' % (k, v / 1e3 / steps, cnt[k] / steps))
print('  %-44s %8.2f' % ('TOTAL', sum(tot.values()) / 1e3 / steps))
# NT kernels use a 1-D grid of tiles (padded to a multiple of 8): group launches by tile count
for kn in ('k_nt64', 'k_ntF', 'k_nt32', 'k_ntFs'):
    items = collections.defaultdict(list)
    for k, v in agg.items():
        if k[0] == kn:
            items[k[1]].extend(v)
    if not items:
        continue
    n = sum(len(v) for v in items.values())
    t = sum(sum(v) for v in items.values())
    print('%s: %d launches, %.2f ms/step' % (kn, n, t / 1e3 / steps))
    sizes = sorted(items)
    pick = sorted(set(sizes[i * (len(sizes) - 1) // 15] for i in range(16))) if len(sizes) > 16 else sizes
    for g in pick:
        v = items[g]
        print('     tiles %5d  n %4d  avg %7.1f us  min %7.1f' % (g, len(v), sum(v) / len(v), min(v)))
