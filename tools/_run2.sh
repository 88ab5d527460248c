timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "gpu tests rc=$?"; tail -8 gpurun_out/t_all.log
bash tools/ab_env.sh "PGD_TUNE=28=1" "PGD_TUNE=28=0"
for CFG in cfg2 cfg3; do timeout -k 10 300 python tools/run_config.py $CFG 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['config'], 'solve_s %.4f' % d['solve_s'], d['fp_passes'], d['pcg_iterations'], 'passes/s %.2f' % d['fp_it_per_s'], d['num_fp_it'])"; done
