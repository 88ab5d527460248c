# kernel durations of the PCG product at 256^3 for several march lengths.  usage: bash tools/prof_zchunk.sh TAG zc1 zc2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for ZC in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pz_${TAG}_$ZC -- python3 tools/pmc_spmv_sym.py 256 ${MODE:-coded} $ZC > gpurun_out/pz_${TAG}_$ZC.log 2>&1 || exit 1
done
python3 - "$TAG" "$@" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for zc in sys.argv[2:]:
    for f in glob.glob("gpurun_out/pz_%s_%s/**/*kernel_stats.csv" % (tag, zc), recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmv_dia" in r["Name"] and "<true, true" in r["Name"]:
                print("zchunk %4s %-60s calls %4s avg %8.1f us min %8.1f" % (zc, r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
