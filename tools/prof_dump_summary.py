"""Summarise a PPK_PROF_DUMP file (one "kind work ms" line per launch of one profiled step):
time and achieved rate per kernel kind, binned by the launch's work.
usage: PPK_PROF_DUMP=gpurun_out/dump.txt python bench.py ... ; python tools/prof_dump_summary.py gpurun_out/dump.txt"""
import collections
import sys

rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
by = collections.defaultdict(list)
for k, w, ms in rows:
    by[k].append((float(w), float(ms)))
for k, v in sorted(by.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    tot = sum(x[1] for x in v)
    work = sum(x[0] for x in v)
    print('%-22s launches %5d  %8.3f ms  %8.1f Gwork/s' % (k, len(v), tot, work / tot / 1e6 if tot else 0))
    if not k.startswith('k_nt'):
        continue
    wmax = max(x[0] for x in v)
    bins = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for w, ms in v:
        b = min(7, int(8 * w / wmax))
        bins[b][0] += 1
        bins[b][1] += ms
        bins[b][2] += w
    for b in sorted(bins):
        n, ms, w = bins[b]
        print('      work %4.1f-%4.1f GF  n %4d  %7.3f ms  avg %6.1f us  %6.1f TFLOP/s  (at 100 TFLOP/s: %6.3f ms)'
              % (b * wmax / 8e9, (b + 1) * wmax / 8e9, n, ms, ms / n * 1e3, w / ms / 1e9, w / 1e11))
