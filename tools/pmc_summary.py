"""Summarise rocprofv3 outputs under a directory into small tracked files for profiles/:
  * kernel stats (name, calls, total/avg ns) from *kernel_stats.csv or *kernel_trace.csv
  * per-kernel average FETCH_SIZE / WRITE_SIZE (KB per dispatch) from *counter_collection.csv
Usage: python tools/pmc_summary.py <rocprof_dir> <out_prefix>"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    for k in ("k_import", "k_resize", "k_fast_cells", "k_quadtree", "k_blur", "k_layout", "k_orient_brief", "k_stereo_match",
              "k_stereo_median", "k_knn2", "k_hamming_matrix"):
        if k in name:
            return k
    return name[:60]


def main(d, out):
    res = {"kernels": {}, "counters": {}}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            try:
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except Exception:
                continue
            a = agg[short(r.get("Kernel_Name", "?"))]; a[0] += 1; a[1] += dur
        for k, (n, t) in agg.items():
            res["kernels"][k] = {"calls": n, "total_ms": round(t / 1e6, 4), "avg_us": round(t / n / 1e3, 3)}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            try:
                v = float(r["Counter_Value"])
            except Exception:
                continue
            a = agg[short(r.get("Kernel_Name", "?"))][r.get("Counter_Name", "?")]; a[0] += 1; a[1] += v
        for k, cs in agg.items():
            res["counters"].setdefault(k, {})
            for c, (n, t) in cs.items():
                res["counters"][k][c] = {"dispatches": n, "avg": round(t / n, 3)}
    json.dump(res, open(out + ".json", "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
