#!/bin/bash
# Everything profiles/ quotes for a round, in one GPU-box session:  bash tools/collect_profiles.sh OUTDIR
#   kernel_stats.csv              rocprofv3 --kernel-trace --stats of the default bench (stream lanes on: the two head towers overlap)
#   kernel_stats_one_stream.csv   the same with RTN_TWO_STREAMS=0: every launch alone on the device, so a kernel's AverageNs is the
#                                 solo duration bench.py reports as roofline.dominant.avg_launch_ms_solo
#   pmc_traffic.json              HBM bytes per step (separate FETCH_SIZE / WRITE_SIZE passes; tools/pmc_traffic.py)
#   pmc_mfma_util.json            MFMA-pipe busy share and wave-cycle shares per kernel (tools/pmc_mfma_util.py)
#   bench.json, layer_times.txt, train_step_breakdown.txt
set -o pipefail
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
run() { timeout -k 10 400 "$@"; }
run python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
run python3 tools/profile_layers.py > "$OUT/layer_times.txt" 2>&1 || exit 1
run python3 tools/profile_train.py > "$OUT/train_step_breakdown.txt" 2>&1 || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > "$OUT/bench_stats.out" 2> "$OUT/bench_stats.err" || exit 1
cp $(find "$OUT/stats" -name "*kernel_stats.csv") "$OUT/kernel_stats.csv"; rm -rf "$OUT/stats"
export RTN_TWO_STREAMS=0
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > "$OUT/bench_stats1.out" 2> "$OUT/bench_stats1.err" || exit 1
unset RTN_TWO_STREAMS
cp $(find "$OUT/stats1" -name "*kernel_stats.csv") "$OUT/kernel_stats_one_stream.csv"; rm -rf "$OUT/stats1"
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -- $B > "$OUT/bench_f.out" 2> "$OUT/bench_f.err" || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -- $B > "$OUT/bench_w.out" 2> "$OUT/bench_w.err" || exit 1
run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_m" -- $B > "$OUT/bench_m.out" 2> "$OUT/bench_m.err" || exit 1
run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_s" -- $B > "$OUT/bench_s.out" 2> "$OUT/bench_s.err" || exit 1
python3 tools/pmc_traffic.py $(find "$OUT/pmc_f" -name "*counter_collection.csv") $(find "$OUT/pmc_w" -name "*counter_collection.csv") 7.125 "$OUT/pmc_traffic.json" "$OUT/bench_f.out" || exit 1
python3 tools/pmc_mfma_util.py $(find "$OUT/pmc_m" -name "*counter_collection.csv") $(find "$OUT/pmc_s" -name "*counter_collection.csv") "$OUT/pmc_mfma_util.json" > "$OUT/mfma_top.txt" || exit 1
rm -rf "$OUT/pmc_f" "$OUT/pmc_w" "$OUT/pmc_m" "$OUT/pmc_s" "$OUT"/bench_*.err
ls "$OUT"
