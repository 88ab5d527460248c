"""Where a bench pass spends its time, from the kernel timeline tools/timeline_pass.sh wrote (name,start,end in ns).

A pass = from the end of one spatial solve (k_scale_out) to the end of the next.  Prints, per pass: wall time,
time inside the PCG loop (first product after k_pcg_init_s .. k_scale_out), GPU busy / idle time outside the loop,
and the kernels that ran outside the loop."""
import sys, csv, collections
rows = [(r["name"], int(r["start"]), int(r["end"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
ends = [i for i, r in enumerate(rows) if r[0].startswith("k_scale_out")]
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], rows[b][2]
    init = next(i for i, r in enumerate(seg) if r[0].startswith("k_pcg_init_s"))
    loop0 = seg[init][2]
    pre = seg[:init + 1]
    busy = sum(e - s for _, s, e in pre)
    by = collections.Counter()
    cnt = collections.Counter()
    for n, s, e in pre:
        by[n] += e - s
        cnt[n] += 1
    loop = seg[init + 1:]
    lbusy = sum(e - s for _, s, e in loop)
    print("pass %.2f ms | PCG loop %.2f ms (busy %.2f, %d kernels) | outside %.2f ms: busy %.2f idle %.2f" % (
        (t1 - t0) / 1e6, (t1 - loop0) / 1e6, lbusy / 1e6, len(loop), (loop0 - t0) / 1e6, busy / 1e6, (loop0 - t0 - busy) / 1e6))
    for n, t in by.most_common(14):
        print("      %-60s %3d x  %8.1f us" % (n[:60], cnt[n], t / 1e3))
    # the largest idle gaps outside the loop
    gaps = sorted(((pre[i + 1][1] - pre[i][2], pre[i][0], pre[i + 1][0]) for i in range(len(pre) - 1)), reverse=True)[:6]
    g0 = pre[0][1] - t0
    print("      gap after k_scale_out -> %s: %.1f us" % (pre[0][0][:40], g0 / 1e3))
    for g, x, y in gaps:
        print("      gap %.1f us between %s -> %s" % (g / 1e3, x[:40], y[:40]))
