mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; tail -5 gpurun_out/r3/gpu_tests.log
