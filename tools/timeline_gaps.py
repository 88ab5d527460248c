"""Kernel timeline summary: per kernel name the number of dispatches, the median duration and the median idle gap in
FRONT of it (ns timeline written by tools/timeline_pass.sh: name,start,end).  Shows what a launch-bound loop pays between
its kernels.  usage: python tools/timeline_gaps.py gpurun_out/TAG_trace.csv [min_count]"""
import sys, csv, collections, statistics
rows = [(r["name"], int(r["start"]), int(r["end"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dur, gap, prev = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(collections.Counter)
for i, (n, s, e) in enumerate(rows):
    dur[n].append(e - s)
    if i:
        gap[n].append(s - rows[i - 1][2])
        prev[n][rows[i - 1][0]] += 1
print("%-58s %7s %9s %9s  %s" % ("kernel", "count", "med us", "gap us", "mostly after"))
tot = 0.0
for n in sorted(dur, key=lambda n: -sum(dur[n])):
    if len(dur[n]) < minc: continue
    g = statistics.median(gap[n]) if gap[n] else 0
    print("%-58s %7d %9.2f %9.2f  %s" % (n[:58], len(dur[n]), statistics.median(dur[n]) / 1e3, g / 1e3, prev[n].most_common(1)[0][0][:40] if prev[n] else ""))
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
print("span %.1f ms, busy %.1f ms (%.0f %%), %d dispatches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows)))
