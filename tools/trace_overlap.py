"""How much do kernels of different queues overlap in a rocprofv3 --kernel-trace CSV?
    python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per_q = collections.Counter()
dur = collections.defaultdict(list)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', '?')
    per_q[q] += 1
    ev.append((s, 1)); ev.append((e, -1))
    name = r['Kernel_Name'].split('(')[0].split('::')[-1]
    dur[name].append(e - s)
ev.sort()
busy = over = 0
active = 0
last = ev[0][0]
for t, d in ev:
    if active >= 1: busy += t - last
    if active >= 2: over += t - last
    active += d
    last = t
total = sum(sum(v) for v in dur.values())
print(f'launches {len(rows)}  queues {dict(per_q)}')
print(f'span {(ev[-1][0] - ev[0][0]) / 1e6:.1f} ms  busy (>=1 kernel) {busy / 1e6:.1f} ms  overlapped (>=2) {over / 1e6:.1f} ms  sum of durations {total / 1e6:.1f} ms')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f'{k:28s} n={len(v):6d} avg {sum(v) / len(v) / 1e3:8.1f} us  total {sum(v) / 1e6:8.1f} ms')
