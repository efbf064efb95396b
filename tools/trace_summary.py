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
for kn in ('k_nt64', 'k_ntF', 'k_nt32', 'k_ntFs'):
    shapes = sorted({(k[1], k[3]) for k in agg if k[0] == kn})
    for gx, gz in shapes:
        items = {k[2]: v for k, v in agg.items() if k[0] == kn and k[1] == gx and k[3] == gz}
        n = sum(len(v) for v in items.values())
        t = sum(sum(v) for v in items.values())
        print('%s grid.x %d grid.z %d: %d launches, %.2f ms/step' % (kn, gx, gz, n, t / 1e3 / steps))
        gys = sorted(items)
        pick = sorted(set([gys[0], gys[len(gys) // 4], gys[len(gys) // 2], gys[3 * len(gys) // 4], gys[-1]]))
        for gy in pick:
            v = items[gy]
            print('     grid.y %4d  n %4d  avg %7.1f us  min %7.1f' % (gy, len(v), sum(v) / len(v), min(v)))
