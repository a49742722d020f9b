set -e
timeout -k 10 400 python -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/chain_test.log 2>&1 || { tail -30 gpurun_out/chain_test.log; exit 1; }
tail -2 gpurun_out/chain_test.log
for v in "RTN_FUSE_CHAIN=0" "RTN_FUSE_CHAIN=1 RTN_CHAIN_SPREAD=1" "RTN_FUSE_CHAIN=1 RTN_CHAIN_SPREAD=0" "RTN_FUSE_CHAIN=0" "RTN_FUSE_CHAIN=1 RTN_CHAIN_SPREAD=1" "RTN_FUSE_CHAIN=1 RTN_CHAIN_SPREAD=0"; do
  echo "== $v" >> gpurun_out/chain_bench.log
  env $v timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_one_batch_at_a_time'))" >> gpurun_out/chain_bench.log
done
cat gpurun_out/chain_bench.log
