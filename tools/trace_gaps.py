"""Idle time between consecutive kernels of the bench steps: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps start at stem_pack_kernel
starts = [i for i, r in enumerate(rows) if "stem_pack_kernel" in r[2]]
for si in range(len(starts) - 1):
    seg = rows[starts[si]:starts[si + 1]]
    span = seg[-1][1] - seg[0][0]
    busy = 0; cur_end = seg[0][0]; gaps = []
    for s, e, n in seg:
        if s > cur_end: gaps.append((s - cur_end, n))
        busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    nxt = rows[starts[si + 1]][0] - seg[-1][1]
    print("step %d: %d kernels, span %.3f ms, union-busy %.3f ms, idle inside %.3f ms (%d gaps, median %.1f us), gap to next step %.1f us"
          % (si, len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3 if gaps else 0, nxt / 1e3))
