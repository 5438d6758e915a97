import re,sys
rows=[]
for l in open(sys.argv[1]):
    m=re.match(r'\s*([\d.]+) us\s+\+\s*([\d.]+) us\s+(.*)',l)
    if m: rows.append((float(m.group(1)),float(m.group(2)),m.group(3).strip()))
ras=[i for i,r in enumerate(rows) if 'raster_kernel' in r[2]]
i0=ras[-4]; i1=ras[-2]
t0=rows[i0][0]
prev_end=None
for s,d,n in rows[i0-3:i1+1]:
    gap = (s-prev_end) if prev_end is not None else 0
    if 'icp_pass' not in n or gap>5: print(f"{s-t0:9.1f} +{d:7.1f} gap {gap:6.1f} {n[:60]}")
    prev_end=max(prev_end or 0, s+d)
print('2-step span', rows[i1][0]-t0)
