#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (every rocprofv3 command has the program itself behind `--`):
#   bash tools/collect_profiles.sh <tag, e.g. r3_v1> <git commit>
# Output under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=$1; COMMIT=$2; O=gpurun_out/$TAG
mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --in-flight 1"     # (tools/pmc_traffic.py counts the batch-8 passes of the process itself)
T="python3 bench.py --mode train --steps 3 --warmup 1"
echo "== kernel trace / stats (inference, the bench default: two batches in flight)"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/kt2.log 2>&1
echo "== kernel trace / stats (inference, one batch at a time, lanes on)"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --in-flight 1 > $O/kt.log 2>&1
echo "== kernel trace / stats (inference, one stream)"; RTN_TWO_STREAMS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --in-flight 1 > $O/kt1.log 2>&1
echo "== kernel trace / stats (training)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python3 bench.py --mode train --steps 5 --warmup 2 > $O/ktt.log 2>&1
echo "== kernel trace / stats (training, every launch alone on the device: no lanes)"; RTN_WGRAD_LANE=0 RTN_TWO_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt1 -- python3 bench.py --mode train --steps 5 --warmup 2 > $O/ktt1.log 2>&1
for P in "f:FETCH_SIZE" "w:WRITE_SIZE" "m:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "s:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  k=${P%%:*}; c=${P#*:}
  echo "== pmc $c (inference)"; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$k -- $B > $O/pmc_$k.log 2>&1
done
for P in "f:FETCH_SIZE" "w:WRITE_SIZE" "m:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  k=${P%%:*}; c=${P#*:}
  echo "== pmc $c (training)"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmct_$k -- $T > $O/pmct_$k.log 2>&1
done
csvof() { find $O/$1 -name "*counter_collection.csv" | head -1; }
statof() { find $O/$1 -name "*kernel_stats.csv" | head -1; }
cp "$(statof kt2)" $O/bench_kernel_stats_two_in_flight.csv 2>/dev/null
cp "$(statof kt)" $O/bench_kernel_stats.csv 2>/dev/null
cp "$(statof kt1)" $O/bench_kernel_stats_one_stream.csv 2>/dev/null
cp "$(statof ktt)" $O/train_kernel_stats.csv 2>/dev/null
cp "$(statof ktt1)" $O/train_kernel_stats_one_stream.csv 2>/dev/null
RTN_GIT_COMMIT=$COMMIT python3 tools/pmc_traffic.py "$(csvof pmc_f)" "$(csvof pmc_w)" auto $O/pmc_traffic.json $O/pmc_f.log > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma_util.py "$(csvof pmc_m)" "$(csvof pmc_s)" $O/pmc_mfma_util.json > $O/pmc_mfma_util.txt 2>&1
python3 tools/pmc_train.py "$(csvof pmct_f)" "$(csvof pmct_w)" "$(csvof pmct_m)" 4 16 $COMMIT $O/pmc_train_traffic.json $O/pmc_train_mfma_util.json > $O/pmc_train.txt 2>&1
echo "== layer times"; timeout -k 10 200 python3 tools/profile_layers.py > $O/layer_times.txt 2>&1
echo "== train breakdown"; timeout -k 10 300 python3 tools/profile_train.py 8 > $O/train_step_breakdown.txt 2>&1
rm -rf $O/kt2 $O/kt $O/kt1 $O/ktt $O/ktt1 $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_s $O/pmct_f $O/pmct_w $O/pmct_m
ls -la $O; tail -3 $O/pmc_traffic.txt; tail -5 $O/pmc_train.txt; head -4 $O/train_step_breakdown.txt
