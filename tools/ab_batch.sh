set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "multidot or vector_ops" > gpurun_out/ab_md_test.log 2>&1
for CFG in cfg2 cfg3 cfg5small; do :; done
for B in 1 256 1 256; do
  for CFG in cfg2 cfg3; do
    PGD_BATCH_FUNCTIONALS=$B timeout -k 10 300 python tools/run_config.py $CFG 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('BATCH=$B', d['config'], 'solve_s %.4f' % d['solve_s'], d['fp_passes'], d['pcg_iterations'], 'passes/s %.2f' % d['fp_it_per_s'], d['num_fp_it'])" >> gpurun_out/ab_md.log
  done
done
cat gpurun_out/ab_md_test.log | tail -3; cat gpurun_out/ab_md.log
