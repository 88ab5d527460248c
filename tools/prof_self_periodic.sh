# per-kernel durations of the self-periodic sharded iteration (one rank as its own neighbour), RCCL exchange vs direct halo:
#   bash tools/prof_self_periodic.sh [variant ...]      (run on the GPU box; writes the tables to stdout)
cd /tmp && export TMPDIR=/tmp
for v in ${@:-stream_ordered_one_march direct_halo_one_march}; do
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_self_periodic.py --json --only $v $PROF_SP_ARGS > /tmp/prof_$v.json 2>/dev/null
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" /tmp/prof_$v.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-60s calls %6s  avg %9.1f ns  min %8s  total %6.2f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
try:
    d = json.load(open(sys.argv[2]))
    for k, v in d.items():
        if isinstance(v, dict) and "us_per_iteration" in v:
            print("   under the profiler: %s %.1f us per iteration, %d iterations" % (k, v["us_per_iteration"], v["iterations"]))
except Exception as e:
    print("   (no result line: %r)" % e)
PY
done
