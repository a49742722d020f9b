mkdir -p gpurun_out/t4
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/t4/tests.log 2>&1; tail -18 gpurun_out/t4/tests.log
( timeout -k 10 120 python tools/exp_first_repack.py cold; timeout -k 10 120 python tools/exp_first_repack.py warm ) 2>&1 | grep -v amdgpu > gpurun_out/t4/first_repack.txt; cat gpurun_out/t4/first_repack.txt
