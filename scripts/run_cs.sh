mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; tail -5 gpurun_out/r3/gpu_tests.log
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; tail -c 3000 gpurun_out/r3/bench_a.json
