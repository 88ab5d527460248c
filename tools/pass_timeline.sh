# Where a fixed-point pass of the bench spends its time OUTSIDE the two kernels of the PCG iteration: a rocprofv3 kernel trace of
# a short bench run, kernels summed per name over the passes of the timed window, gaps (stream idle) reported too.
#   bash tools/pass_timeline.sh            (on the GPU box; prints the table)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-general-paths --no-pmc --no-cpu-baseline --no-csr-section > /tmp/pt.json 2>/tmp/pt.err
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python3 - "$f" /tmp/pt.json <<'PY'
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = json.load(open(sys.argv[2]))
K, W = d["steps"], d["warmup"]
# a pass = one spatial solve + one parameter solve; the spatial solve starts with k_combine_dia on the big operator: use the LAST
# K + 1 occurrences of a combine of >= 1e6 rows ... simpler: the spatial solves are the launches of k_pcg_init_s
idx = [i for i, r in enumerate(rows) if "k_pcg_init_s" in r["Kernel_Name"]]
print("solves seen:", len(idx), " passes/s reported:", d["value"], " its/pass:", d["config"]["pcg_iterations_per_step"])
use = idx[-(K + 1):]            # K complete solve-to-solve windows at the end of the run = the timed passes
t0, t1 = int(rows[use[0]]["Start_Timestamp"]), int(rows[use[-1]]["Start_Timestamp"])
sel = rows[use[0]:use[-1]]
by = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
last_end = None
gaps = 0.0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:64]
    by[name][0] += 1
    by[name][1] += (e - s)
    busy += e - s
    if last_end is not None and s > last_end:
        gaps += s - last_end
    last_end = max(last_end or 0, e)
print("window %.2f ms per pass: kernels %.2f ms, stream idle %.2f ms" % ((t1 - t0) / 1e6 / K, busy / 1e6 / K, gaps / 1e6 / K))
gap_by = collections.defaultdict(lambda: [0, 0.0])
last_end, last_name = None, None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pgd::", "")[:34]
    if last_end is not None and s - last_end > 15000:          # idle longer than 15 us: a host round trip, not a launch gap
        key = last_name + " -> " + name
        gap_by[key][0] += 1
        gap_by[key][1] += s - last_end
    last_end, last_name = max(last_end or 0, e), name
print("idle stretches > 15 us (a host round trip), by the kernels on either side:")
for key, (n, t) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:22]:
    print("  %-72s %5.1f per pass  %7.3f ms/pass  avg %7.1f us" % (key, n / K, t / 1e6 / K, t / 1e3 / n))
for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-66s %7.1f calls/pass  %8.3f ms/pass  avg %8.1f us" % (name, n / K, t / 1e6 / K, t / 1e3 / n))
PY
