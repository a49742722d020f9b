cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4_v1; mkdir -p $O
RTN_WGRAD_LANE=0 RTN_TWO_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt1 -- python3 bench.py --mode train --steps 5 --warmup 2 > $O/ktt1.log 2>&1
cp "$(find $O/ktt1 -name '*kernel_stats.csv' | head -1)" $O/train_kernel_stats_one_stream.csv; rm -rf $O/ktt1
timeout -k 10 200 python tools/hbm_table.py $O/train_kernel_stats_one_stream.csv profiles/r4_v1_bench_kernel_stats_one_stream.csv 2>&1 | grep -v amdgpu > $O/hbm_bound_kernels.txt
cat $O/hbm_bound_kernels.txt; tail -2 $O/ktt1.log | cut -c1-300
