# kernel timeline of a few bench passes (rocprofv3 --kernel-trace, no stats): where the GPU idles between the solves
# usage: bash tools/timeline_pass.sh TAG ; then python tools/timeline_pass.py gpurun_out/TAG_trace.csv
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_${TAG} -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-csr-section ${BENCH_ARGS:-} > gpurun_out/${TAG}_tl_bench.json 2> gpurun_out/${TAG}_tl.err || exit 1
F=$(find gpurun_out/tl_${TAG} -name '*kernel_trace.csv' | head -1)
python3 - "$F" gpurun_out/${TAG}_trace.csv <<'PY'
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w") as g:
    r = csv.DictReader(f)
    g.write("name,start,end\n")
    for row in r:
        n = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("pgd::", "")
        g.write("%s,%s,%s\n" % (n.replace(",", ";"), row["Start_Timestamp"], row["End_Timestamp"]))
PY
rm -rf gpurun_out/tl_${TAG}
ls -la gpurun_out/${TAG}_trace.csv
