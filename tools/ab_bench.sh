# A/B of whole bench runs under different PGD_TUNE settings in ONE box.  usage: bash tools/ab_bench.sh TAG "k=v,..." "k=v,..." ...
TAG=$1; shift
n=0
for T in "$@"; do
  n=$((n+1))
  PGD_TUNE="$T" timeout -k 10 400 python bench.py --no-pmc --no-csr-section --no-cpu-baseline > gpurun_out/ab_${TAG}_$n.json 2> gpurun_out/ab_${TAG}_$n.err || exit 1
  python - "$T" gpurun_out/ab_${TAG}_$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("PGD_TUNE=%-16s value %.3f  it/step %.1f  us/it %.1f  %s" % (sys.argv[1], d["value"], d["config"]["pcg_iterations_per_step"], d["config"]["us_per_pcg_iteration"], d["config"]["pcg_iteration_breakdown_us"]))
PY
done
