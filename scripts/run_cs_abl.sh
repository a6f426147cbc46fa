mkdir -p gpurun_out/r3
for a in 0 1 2 3 8 10 16 4 5; do
  timeout -k 10 200 python scripts/bench_pool.py "cs128 column-sliced (split" 150000 $a > gpurun_out/r3/abl_$a.log 2>&1
  echo "ablate=$a: $(grep 'cs128 column-sliced (split' gpurun_out/r3/abl_$a.log | tail -1)"
done
