import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps: delimited by adam kernel
ends=[i for i,r in enumerate(rows) if "adam" in r[2].lower()]
print("adam launches", len(ends))
for k in range(max(0,len(ends)-6), len(ends)-1):
    seg=rows[ends[k]+1:ends[k+1]+1]
    span=seg[-1][1]-seg[0][0]
    busy=0; cur=seg[0][0]; gaps=[]
    for s,e,n in seg:
        if s>cur: gaps.append((s-cur,n))
        busy+=max(0,e-max(s,cur)); cur=max(cur,e)
    big=sorted(gaps,reverse=True)[:6]
    print("step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps; largest: %s" % (len(seg), span/1e6, busy/1e6, (span-busy)/1e6, len(gaps), [(round(g/1e3,1), n[:40]) for g,n in big]))
