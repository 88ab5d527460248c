# A/B of whole bench runs under different environment settings in ONE box.  usage: bash tools/ab_env.sh "A=1 B=2" "A=0" ...
for E in "$@"; do
  env $E timeout -k 10 400 python bench.py --no-pmc --no-csr-section --no-cpu-baseline > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || exit 1
  python - "$E" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
print("%-28s value %.3f us/it %.1f %s" % (sys.argv[1], d["value"], d["config"]["us_per_pcg_iteration"], d["config"]["pcg_iteration_breakdown_us"]))
PY
done
