# what lies between the loops of two sharded solves (one rank as its own neighbour): kernels with start offsets, from a rocprofv3 trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sst
rocprofv3 --kernel-trace -d /tmp/sst -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_self_periodic.py --json --only ${1:-stream_ordered_one_march} > /dev/null 2>&1
f=$(find /tmp/sst -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
upd = [i for i, r in enumerate(rows) if "k_pcg1_update" in r["Kernel_Name"]]
init = [i for i, r in enumerate(rows) if "k_cg_init_s" in r["Kernel_Name"] or "k_pcg_init_s" in r["Kernel_Name"]]
at = init[-1]                                   # the last solve's initial residual
prev_upd = max(u for u in upd if u < at)        # ... and the last update of the solve before it
last = min(u for u in upd if u > at)
t0 = int(rows[prev_upd]["End_Timestamp"])
print("between the last update of one solve and the first update of the next: %.3f ms" % ((int(rows[last]["Start_Timestamp"]) - t0) / 1e6))
for r in rows[prev_upd + 1:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pgd::", "")[:70]))
PY
