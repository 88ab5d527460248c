"""Per-kernel statistics of a `rocprofv3 --kernel-trace` run WITHOUT the no-op dispatches.

    python tools/kernel_stats_filtered.py <dir with *_kernel_trace.csv> [--min-us 10] > profiles/<tag>_kernel_stats_filtered.csv

The PCG loops queue whole chunks of 16 iterations; what is queued behind the iteration that converged returns on the done flag
(3-4 us per dispatch).  rocprofv3's own --stats averages those in, so its AverageNs of k_pcg1_update / k_spmv_diac_march2 sits
3-6 % below the HIP-event timing of bench.py, which drops them.  This filter drops, per kernel, the dispatches shorter than
--min-us WHEN the kernel's median is at least four times that (a kernel that is short by nature keeps all its dispatches), and
prints raw and filtered figures side by side."""
import csv
import glob
import statistics
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_us = 10.0
    if "--min-us" in sys.argv:
        min_us = float(sys.argv[sys.argv.index("--min-us") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--min-us") + 1]]
    files = glob.glob(args[0] + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        raise SystemExit("no *_kernel_trace.csv under %s" % args[0])
    dur = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                dur.setdefault(name, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "AverageUs", "CallsKept", "AverageUsKept", "DroppedBelowUs", "MedianUs", "MinUs", "MaxUs", "TotalMs"])
    for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        med = statistics.median(d)
        cut = min_us if med >= 4.0 * min_us else 0.0
        kept = [t for t in d if t >= cut]
        w.writerow([name, len(d), "%.3f" % (sum(d) / len(d)), len(kept), "%.3f" % (sum(kept) / len(kept)), cut, "%.3f" % med,
                    "%.3f" % min(d), "%.3f" % max(d), "%.3f" % (sum(d) / 1e3)])


if __name__ == "__main__":
    main()
