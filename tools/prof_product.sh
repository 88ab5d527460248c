# kernel durations + a few PMC passes of the PCG product at 256^3 (12 eager PCG iterations).  usage: bash tools/prof_product.sh TAG [zchunk]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}; ZC=${2:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_${TAG}_trace -- python3 tools/pmc_spmv_sym.py 256 grid $ZC > gpurun_out/pp_${TAG}_trace.log 2>&1 || exit 1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pp_${TAG}_$name -- python3 tools/pmc_spmv_sym.py 256 grid $ZC > gpurun_out/pp_${TAG}_$name.log 2>&1 || echo "pass $name failed"
done
python3 - <<PY
import csv, glob, collections
tag="$TAG"
for f in glob.glob("gpurun_out/pp_%s_trace/**/*kernel_stats.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Name"] or "pcg1" in r["Name"]:
            print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pp_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if ("k_spmv_dia" in r["Kernel_Name"] or "k_spmv_stencil" in r["Kernel_Name"]) and "<true, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s n=%3d mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
