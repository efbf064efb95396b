import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
# take the last 60% of the trace (steady state)
rows=rows[int(n*0.4):]
span=(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)/1e3
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(rows,rows[1:])]
gaps.sort()
print(f'kernels {len(rows)}, span {span:.0f} us, busy {busy:.0f} us ({100*busy/span:.1f} %), per launch: wall {span/len(rows):.2f} us, kernel {busy/len(rows):.2f} us, median gap {gaps[len(gaps)//2]:.2f} us, mean gap {sum(gaps)/len(gaps):.2f} us')
d=collections.defaultdict(list)
for r in rows:
    import re
    m=re.search(r'(k_\w+)', r['Kernel_Name']); k=m.group(1) if m else r['Kernel_Name'][:30]
    d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print(f'  {k:24s} n {len(v):6d} mean {sum(v)/len(v):6.2f} us total {sum(v)/1e3:8.2f} ms')
