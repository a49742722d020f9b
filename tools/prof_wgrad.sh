#!/bin/bash
# Kernel trace of one weight-gradient shape under several knob sets: per-kernel average durations (main kernel vs finish).
#   bash tools/prof_wgrad.sh tower "RTN_WGRAD_WIN=1 RTN_WGRAD_WIN=0"
shape=${1:-tower}; variants=${2:-"RTN_WGRAD_WIN=1 RTN_WGRAD_WIN=0"}
O=$PWD/gpurun_out/prof_wgrad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in $variants; do
  d=$O/$(echo $v | tr '=+' '__')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/ab_wgrad.py $shape $v > $d.log 2>&1 < /dev/null
  f=$(find $d -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "== $shape $v"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print("  %-90s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  else echo "  no stats file"; tail -3 $d.log; fi
done
