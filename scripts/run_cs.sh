mkdir -p gpurun_out/r3
timeout -k 10 200 python scripts/bench_pool.py "engine" > gpurun_out/r3/eng.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/eng.log | tail -6
for a in 0 1 4 5; do timeout -k 10 100 python scripts/stamp_engine.py $a 2>&1 | grep -v amdgpu.ids; done
for a in 1 4 5 2 3; do
  timeout -k 10 200 python scripts/bench_pool.py "engine cs128 x 128c (split" 150000 $a > gpurun_out/r3/eabl_$a.log 2>&1
  echo "ablate=$a: $(grep 'engine cs128 x 128c (split' gpurun_out/r3/eabl_$a.log | tail -1)"
done
