"""List the launches of kernels whose name contains a pattern in a rocprofv3 kernel trace (grid, duration), grouped.
    python tools/trace_list.py <kernel_trace.csv> <pattern>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    key = (name, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-42s grid %6d x %4d x %3d  n %4d  avg %8.1f us  total %8.2f ms' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), sum(v) / 1e3))
