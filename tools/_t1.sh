mkdir -p gpurun_out/t2
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/t2/tests.log 2>&1; tail -22 gpurun_out/t2/tests.log
timeout -k 10 400 python tools/fp8_plans.py 2>&1 | grep -v amdgpu > gpurun_out/t2/fp8_plans.txt; cat gpurun_out/t2/fp8_plans.txt
