mkdir -p gpurun_out/r3
for o in rcb rcb8192 rcb4096 rcb2048; do
GP_ORDER=$o timeout -k 10 300 python scripts/bench_pool.py "cs128 column" > gpurun_out/r3/order_$o.log 2>&1; echo "== $o"; grep -v amdgpu gpurun_out/r3/order_$o.log | grep "cs BR\|cs128 column\|vs ELL" 
done
