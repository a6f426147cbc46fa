mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r3/t_conv.log 2>&1; tail -3 gpurun_out/r3/t_conv.log
GP_SKIP_V1=1 timeout -k 10 300 python scripts/bench_conv.py > gpurun_out/r3/bench_conv.log 2>&1; grep -v amdgpu gpurun_out/r3/bench_conv.log | tail -10
GP_SKIP_V1=1 GP_CHUNK=16384 timeout -k 10 300 python scripts/bench_conv.py > gpurun_out/r3/bench_conv16k.log 2>&1; grep -v amdgpu gpurun_out/r3/bench_conv16k.log | tail -8
