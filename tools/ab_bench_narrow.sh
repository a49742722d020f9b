cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
for rep in 1 2; do
 for v in 1 0; do
  RTN_CONV_G8_NARROW=$v python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('infer narrow=$v rep$rep', round(j['value'],1), round(j['ms_per_step'],4))"
 done
done
for rep in 1 2; do
 for v in 1 0; do
  RTN_CONV_G8_NARROW=$v python bench.py --mode train 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('train narrow=$v rep$rep', round(j['value'],1), round(j['ms_per_step'],3))"
 done
done
