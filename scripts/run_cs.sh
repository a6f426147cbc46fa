mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_reference_golden.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu -k "lift_masks or tuple or as_tuple or drive_evaluate or boundary" > gpurun_out/r3/t_lift.log 2>&1; tail -6 gpurun_out/r3/t_lift.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-train > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3/bench_b.json') if l.startswith('{')][-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])
print(json.dumps(r.get('api_tuple'), indent=1))
PY
tail -3 gpurun_out/r3/bench_b.err
