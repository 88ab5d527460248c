# k_assemble_p1<3> at 256^3: HIP-event times, then rocprofv3 --pmc passes (one counter set per pass).  usage: bash tools/prof_assembly.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}
python3 tools/bench_assembly.py 128 256 > gpurun_out/${TAG}_assembly.jsonl 2> gpurun_out/${TAG}_assembly.err || exit 1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pa_${TAG}_$name -- python3 tools/bench_assembly.py 256 > gpurun_out/pa_${TAG}_$name.log 2>&1 || echo "pass $name failed"
done
python3 - <<PY
import csv, glob, collections
tag="$TAG"
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pa_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_assemble_p1" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open("gpurun_out/%s_assembly.jsonl" % tag).read())
for k, v in sorted(acc.items()):
    print("%-32s n=%3d mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
