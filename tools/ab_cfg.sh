# A/B of a BASELINE config under different PGD_TUNE settings in ONE box.  usage: bash tools/ab_cfg.sh cfg3 "k=v" "k=v" ...
CFG=$1; shift
for T in "$@"; do
  PGD_TUNE="$T" timeout -k 10 500 python tools/run_config.py $CFG 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('PGD_TUNE=%-12s' % '$T', d['config'], 'solve_s %.4f' % d['solve_s'], d['fp_passes'], d['pcg_iterations'], 'passes/s %.2f' % d['fp_it_per_s'])"
done
