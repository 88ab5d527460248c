# A/B of bench runs at a given grid size under different environment settings in ONE box.
# usage: N=128 bash tools/ab_env_n.sh "A=1 B=2" "A=0" ...   (EXTRA="--dist-driver" for the sharded driver)
for E in "$@"; do
  env $E timeout -k 10 400 python bench.py --n ${N:-128} --no-pmc --no-csr-section --no-cpu-baseline --no-general-paths ${EXTRA:-} > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || exit 1
  python - "$E" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
print("%-44s value %.3f us/it %.1f %s" % (sys.argv[1], d["value"], d["config"]["us_per_pcg_iteration"], d["config"]["pcg_iteration_breakdown_us"]))
PY
done
