"""Merges rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of tools/bench_graph.py --only <case> into one text summary
per case: per kernel launches, HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE: MI355X_MICROARCH.md, HBM
section) and, for K7, the ratio to the algorithmic bytes bench_graph.py prices the launch at.
usage: python tools/graph_traffic.py <case> <fetch csv> <write csv> <bench_graph jsonl of the same case>"""
import collections
import csv
import json
import re
import sys

case, fcsv, wcsv, jl = sys.argv[1:5]


def per_kernel(path):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        m = re.search(r'(k_\w+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        tot[k] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
    return tot, {k: len(v) for k, v in disp.items()}


f, nf = per_kernel(fcsv)
w, nw = per_kernel(wcsv)
alg = {}
for line in open(jl):
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        if 'roofline' in d:
            alg[d['kernel']] = d['roofline']['algorithmic_bytes']
print(f'# case {case}: HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / launches')
print('%-28s %9s %16s %16s %16s' % ('kernel', 'launches', 'fetch B/launch', 'write B/launch', 'traffic B/launch'))
rows = []
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
    if not k.startswith('k_'):
        continue
    n = nf[k]
    fb, wb = 2 * f[k] * 1024 / n, w.get(k, 0) * 1024 / max(nw.get(k, n), 1)
    rows.append((k, n, fb, wb))
    print('%-28s %9d %16.4g %16.4g %16.4g' % (k, n, fb, wb, fb + wb))
spmm = [r for r in rows if r[0] in ('k_spmm', 'k_spmm_chunks', 'k_spmm_long_sum')]
if spmm and any('k_spmm' in k for k in alg):
    a = next(v for k, v in alg.items() if 'k_spmm forward' in k)
    t = sum(r[2] + r[3] for r in spmm)
    print(f'K7 (k_spmm + long-row kernels) per product: traffic {t:.4g} B vs algorithmic {a:.4g} B = {t / a:.3f}')
sp = {r[0]: r[2] + r[3] for r in rows if r[0].startswith('k_sp_')}
if sp:
    # the all-anchors case (one GPU holds the graph): forward = k_sp_dist_once + terms + sums + loss, backward = coef + dz + dz_fix
    fwd = sum(sp.get(k, 0.0) for k in ('k_sp_dist_once', 'k_sp_terms', 'k_sp_sum', 'k_sp_sum2', 'k_sp_loss'))
    bwd = sum(sp.get(k, 0.0) for k in ('k_sp_coef', 'k_sp_dz', 'k_sp_dz_fix'))
    af = next((v for k, v in alg.items() if k.startswith('K5 sparse graph loss forward (') and 'rank' not in k), None)
    afb = next((v for k, v in alg.items() if k.startswith('K5 sparse graph loss forward + backward') and 'rank' not in k), None)
    if af and afb:
        print(f'K5 all anchors, forward (dist_once + terms + sums): traffic {fwd:.4g} B vs algorithmic {af:.4g} B = {fwd / af:.3f} '
              f'(latent rows of neighbouring anchors are re-read from L2 / Infinity Cache, not HBM)')
        print(f'K5 all anchors, forward + backward (+ coef + dz + dz_fix; k_sp_terms / k_sp_dz averaged over all-anchor and '
              f'sharded launches of the run): traffic {fwd + bwd:.4g} B vs algorithmic {afb:.4g} B = {(fwd + bwd) / afb:.3f}')
