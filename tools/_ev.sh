set -u
mkdir -p gpurun_out/r4_v3
bash tools/collect_profiles.sh r4_v3 f28f187 > gpurun_out/r4_v3/collect.log 2>&1
timeout -k 10 300 python3 tools/launch_plan.py --markdown > gpurun_out/r4_v3/launch_plan.txt 2>gpurun_out/r4_v3/launch_plan.err
tail -3 gpurun_out/r4_v3/collect.log
