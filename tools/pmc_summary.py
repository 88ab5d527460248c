"""Reduce the counter CSVs of tools/pmc_round.sh to one JSON: calibration factors (known bytes / counter) for 8- and
16-byte accesses and the product's HBM-side bytes per launch with those factors applied."""
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def per_kernel(dirname, counter):
    out = {}
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % dirname, recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out


def unit(vals, known):
    """FETCH_SIZE / WRITE_SIZE are reported in KiB on some ROCm releases and in bytes on others: pick the unit that puts
    the known stream within a factor 4."""
    v = sum(vals) / len(vals)
    for u in (1.0, 1024.0):
        if known / 4 <= v * u <= known * 4:
            return u
    return 1.0


res = {"tag": tag, "calibration": {}}
GIB = float(1 << 30)
factors = {}
for kind, counter in (("load", "FETCH_SIZE"), ("store", "WRITE_SIZE")):
    for w in (8, 16):
        d = per_kernel("pmc_%s_calib_%s%d" % (tag, kind, w), counter)
        vals = [v for k, vs in d.items() if "k_calib_stream" in k for v in vs]
        if not vals:
            continue
        u = unit(vals, GIB)
        mean = sum(vals) / len(vals) * u
        factors[(kind, w)] = GIB / mean
        res["calibration"]["%s_%dB_per_lane" % (kind, w)] = {"counter": counter, "counter_unit_bytes": u, "known_bytes": GIB,
                                                             "counter_bytes_mean": mean, "factor_known_over_counter": GIB / mean,
                                                             "dispatches": len(vals)}
prod = {}
for counter, kind in (("FETCH_SIZE", "load"), ("WRITE_SIZE", "store")):
    d = per_kernel("pmc_%s_march_%s" % (tag, counter), counter)
    for k, vs in d.items():
        if "k_spmv_dia_march" in k and "<true, true" in k:        # the PCG instance (fused dot, y stored)
            u = res["calibration"].get("%s_8B_per_lane" % kind, {}).get("counter_unit_bytes", 1.0)
            mean = sum(vs) / len(vs) * u
            prod[counter] = {"kernel": k[:80], "counter_bytes_per_launch": mean, "factor_8B": factors.get((kind, 8)),
                             "bytes_per_launch": mean * factors.get((kind, 8), 1.0), "dispatches": len(vs)}
res["product"] = prod
if "FETCH_SIZE" in prod and "WRITE_SIZE" in prod:
    res["hbm_bytes_per_launch"] = prod["FETCH_SIZE"]["bytes_per_launch"] + prod["WRITE_SIZE"]["bytes_per_launch"]
print(json.dumps(res, indent=1))
