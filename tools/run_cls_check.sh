timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "row_class or pcg_on_row or lagged or single_sync" > gpurun_out/t_cls.log 2>&1 && timeout -k 10 500 python bench.py --no-pmc --no-csr-section --no-cpu-baseline > gpurun_out/bench_cls.json 2> gpurun_out/bench_cls.err; tail -3 gpurun_out/t_cls.log; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_cls.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["us_per_pcg_iteration"], d["config"]["pcg_iteration_breakdown_us"], d["roofline"]["achieved"])
PY
