set -u
mkdir -p gpurun_out/r4_v5
bash tools/collect_profiles.sh r4_v5 9daaa8b > gpurun_out/r4_v5/collect.log 2>&1
timeout -k 10 300 python3 tools/launch_plan.py --markdown > gpurun_out/r4_v5/launch_plan.txt 2>gpurun_out/r4_v5/launch_plan.err
tail -3 gpurun_out/r4_v5/collect.log; cat gpurun_out/r4_v5/pmc_traffic.txt
