set -e
mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pool_cs" > gpurun_out/r3/t_cs.log 2>&1 || { tail -20 gpurun_out/r3/t_cs.log; exit 1; }
tail -2 gpurun_out/r3/t_cs.log
timeout -k 10 300 python scripts/bench_pool.py cs > gpurun_out/r3/bench_pool_cs.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/bench_pool_cs.log | tail -4
timeout -k 10 300 python scripts/bench_pool.py cs 150000 32 > gpurun_out/r3/bench_pool_cs_nolate.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/bench_pool_cs_nolate.log | tail -4
timeout -k 10 120 python scripts/stamp_pool.py 0 0 > gpurun_out/r3/stamp_cs.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/stamp_cs.log | tail -24
timeout -k 10 120 python scripts/stamp_pool.py 0 64 > gpurun_out/r3/stamp_cs_issue.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/stamp_cs_issue.log | sed -n 4,10p
