"""profiles/pmc_traffic.json from the rocprofv3 PMC summaries (tools/pmc_summary.py output of the FETCH_SIZE and WRITE_SIZE
passes): HBM-side bytes per launch of each pipeline stage = (FETCH_SIZE + WRITE_SIZE) KB * 1024.
gfx950 note (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports 16-B/lane streams by 2x and is uncalibrated for other
widths; these kernels use 4-B/lane loads, calibrated here on k_blur, whose unique read set is known exactly
(images x sum of level pixels): FETCH_SIZE = 142.5 MB vs 143.0 MB -> factor 1.0, so no doubling is applied."""
import json, sys
f = json.load(open(sys.argv[1]))["counters"]; w = json.load(open(sys.argv[2]))["counters"]
stage = {"k_import": "import", "k_resize": "pyramid", "k_fast_cells": "fast_cells", "k_quadtree": "quadtree", "k_blur": "blur",
         "k_layout": "layout", "k_orient_brief": "orient_brief", "k_stereo_match": "match"}
out = {}
for k, s in stage.items():
    fe = f.get(k, {}).get("FETCH_SIZE", {}).get("avg", 0.0); wr = w.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0)
    mult = 7 if k == "k_resize" else 1          # the pyramid stage is 7 launches
    out[s] = int((fe + wr) * 1024 * mult)
out["_note"] = "bytes per stage launch at 128 images/step = (FETCH_SIZE+WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes; 4-B/lane loads, factor 1.0 (calibrated on k_blur)"
import os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
stamp = {"_kernel_sources_sha": bench.kernel_sources_sha(), "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes of %s (%s)" % (time.strftime("%Y-%m-%d"), os.path.dirname(sys.argv[3]))}
out.update(stamp)
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(out)

# optional: VALU wave-instructions per stage launch from the SQ pass (tools/gpu_round2.sh) -> profiles/pmc_valu.json
if len(sys.argv) > 5:
    q = json.load(open(sys.argv[4]))["counters"]
    v = {}
    for k, s_ in stage.items():
        mult = 7 if k == "k_resize" else 1
        v[s_] = int(q.get(k, {}).get("SQ_INSTS_VALU", {}).get("avg", 0.0) * mult)
    v["_note"] = "VALU wave-instructions per stage launch at 128 images/step (rocprofv3 --pmc SQ_INSTS_VALU, own pass)"
    v.update(stamp)
    json.dump(v, open(sys.argv[5], "w"), indent=1, sort_keys=True)
    print(v)
