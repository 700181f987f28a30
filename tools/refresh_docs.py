"""Rewrites the measured numbers of DESIGN.md §6b / §6c, BASELINE.md §4 and INTEGRATION.md from profiles/r02_final/*.json (after a
tools/gpu_round2.sh visit).  The prose around the numbers is edited by hand."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r02_final")
b = json.load(open(os.path.join(P, "bench_n1.json"))); nr = json.load(open(os.path.join(P, "next_rows.json")))
mono, fe, rg = (json.load(open(os.path.join(P, "bench_%s.json" % c))) for c in ("mono", "fisheye", "rgbd"))
cb = b["cpu_baseline"]; r = b["roofline"]
log = open(os.path.join(P, "serial_stage_times_and_parity.log")).read()
single = float(re.search(r"single pair, graph=False: ([0-9.]+) ms", log).group(1))
st2 = eval(re.search(r"^B 2 .*stages (\{.*?\})", log, re.M).group(1)); st128 = eval(re.search(r"^B 128 .*stages (\{.*?\})", log, re.M).group(1))


def g(prefix):
    k = [k for k in nr if k.startswith(prefix)]
    assert k, prefix
    return nr[k[0]]


def sub_line(s, prefix, new):
    lines = s.split("\n"); idx = [i for i, l in enumerate(lines) if l.startswith(prefix)]
    assert len(idx) == 1, (prefix, len(idx))
    lines[idx[0]] = new
    return "\n".join(lines)


p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
s = sub_line(s, "| `bench.py` (defaults:", "| `bench.py` (defaults: 128 pairs per step, 200 steps, four handles) | **%.0f stereo pairs/s**, %.3f ms/step; step completion intervals median %.2f / p10 %.2f / p90 %.2f ms (round 1: 75.7 k; first half of this round: 80.5 k; between boxes of the pool one build varies by ±1 %%) |" % (b["value"], b["ms_per_step"], b["step_ms"]["median"], b["step_ms"]["p10"], b["step_ms"]["p90"]))
s = sub_line(s, "| the same with the inputs uploaded", "| the same with the inputs uploaded inside the timed region (`h2d_inclusive`) | **%.0f pairs/s**, %.3f ms/step, %.1f GB/s over PCIe (round 1: 39.8 k with the copies on the kernel stream) |" % (b["h2d_inclusive"]["value"], b["h2d_inclusive"]["ms_per_step"], b["h2d_inclusive"]["PCIe_GBps"]))
s = sub_line(s, "| single pair per call", "| single pair per call (B = 2 images, one handle, sync after every pair) | **%.3f ms** → %.0f pairs/s (round 1: 0.37 ms; quadtree %.3f of it, pyramid %.3f, stereo %.3f, FAST %.3f, orient+BRIEF %.3f) |" % (single, 1000.0 / single, st2["quadtree"], st2["pyramid"], st2["match"], st2["fast_cells"], st2["orient_brief"]))
tot = sum(st128[k] for k in ("import", "pyramid", "fast_cells", "quadtree", "blur", "layout", "orient_brief", "match"))
s = sub_line(s, "| serial stage times at 128 images", "| serial stage times at 128 images (ms) | import %.3f · pyramid %.3f · FAST %.3f · quadtree %.3f · blur %.3f · layout %.3f · orient+BRIEF %.3f · stereo %.3f = %.2f (round 1: 1.20) |" % (st128["import"], st128["pyramid"], st128["fast_cells"], st128["quadtree"], st128["blur"], st128["layout"], st128["orient_brief"], st128["match"], tot))
s = sub_line(s, "| the reference's two stages", "| the reference's two stages, kernels alone, per 128-pair step | \"ORB Extraction\" %.3f ms · \"Stereo Matching\" %.3f ms (CPU reference on one core: %.1f ms + %.2f ms **per pair**) |" % (b["reference_stage_ms_alone"]["ORB Extraction"], b["reference_stage_ms_alone"]["Stereo Matching"], cb["one_core"]["stage_ms"]["ORB Extraction"], cb["one_core"]["stage_ms"]["Stereo Matching"]))
s = sub_line(s, "| dominant kernel |", "| dominant kernel | `k_fast_cells` (256 images per launch): 326 MB algorithmic / %.3f ms (HIP events, overlapped schedule) = %.0f GB/s = **%.1f %% of 8 TB/s** (alone: %.3f ms = %.0f GB/s); rocprofv3 average in `rocprofv3_kernel_stats.csv`; PMC traffic 155 MB per 128 images = the algorithmic bytes; 341 M VALU wave-instructions per launch → %.0f G/s alone = **%.1f %% of the 1228.8 G/s issue peak** |" % (r["avg_launch_ms"], r["achieved"], 100 * r["frac"], r["alone_launch_ms"], r["alone_GBps"], r["valu_issue"]["alone_wave_instr_per_s"] / 1e9, 100 * r["valu_issue"]["alone_frac"]))
s = sub_line(s, "| other configurations", "| other configurations (`bench.py --config`) | mono 752×480 n1000: %.0f frames/s · fisheye 512×512 n1500 incl. the triangulation gate: %.0f pairs/s · RGB-D 640×480 n1000 + `SearchLocalPoints` against 5000 map points per frame: %.0f frames/s (the per-frame search call from Python dominates) |" % (mono["value"], fe["value"], rg["value"]))
s = sub_line(s, "| CPU baseline, same run", "| CPU baseline, same run (256-core host) | one core %.1f pairs/s · two cores %.1f · all cores (128 concurrent Frame constructors) %.0f pairs/s |" % (cb["one_core"]["value"], cb["two_cores"]["value"], cb["value"]))
slp = g("Tracking::SearchLocalPoints")
s = sub_line(s, "| `Tracking::SearchLocalPoints`: `isInFrustum`", "| `Tracking::SearchLocalPoints`: `isInFrustum` + `SearchByProjection` for 5000 map points (`orbm_search_local_points`, results identical) | %.3f ms (C ABI alone: **%.3f ms**; with the points resident, `orbm_search_local_points_resident`: **%.3f ms**) | %.2f ms (reference `Frame.cc` + `ORBmatcher.cc`) |" % (slp["gpu_ms"], slp["gpu_ms_c_abi"], slp["gpu_ms_c_abi_resident_points"], slp["cpu_reference_ms"]))
for pre, key, fmt in (("| `SearchByProjection(Frame, 5000 MapPoints)`", "SearchByProjection(Frame, 5000", "| `SearchByProjection(Frame, 5000 MapPoints)` (projections from the caller) | %.3f ms | %.2f ms |"),
                      ("| `SearchByProjection(KeyFrame, Sim3, 5000 points)`", "SearchByProjection(KeyFrame, Sim3", "| `SearchByProjection(KeyFrame, Sim3, 5000 points)` | %.3f ms | %.2f ms |"),
                      ("| `Fuse` candidate search", "Fuse", "| `Fuse` candidate search (5000 points, χ² gate) | %.3f ms | %.2f ms |"),
                      ("| `SearchByProjection(Frame, LastFrame)`", "SearchByProjection(Frame, LastFrame)", "| `SearchByProjection(Frame, LastFrame)` (1000 points) | %.3f ms | %.2f ms |"),
                      ("| `ComputeDistinctiveDescriptors`", "ComputeDistinctiveDescriptors", "| `ComputeDistinctiveDescriptors`, 5000 map points / 110 k descriptors | %.3f ms | %.1f ms |")):
    s = sub_line(s, pre, fmt % (g(key)["gpu_ms"], g(key)["cpu_oracle_ms"]))
s = sub_line(s, "| `SearchForTriangulation` (1000 × 1000", "| `SearchForTriangulation` (1000 × 1000 features, 64 nodes) | %.3f ms | %.3f ms |" % (nr["SearchForTriangulation"]["gpu_ms"], nr["SearchForTriangulation"]["cpu_oracle_ms"]))
s = sub_line(s, "| `SearchByBoW(KeyFrame, Frame)` |", "| `SearchByBoW(KeyFrame, Frame)` | %.3f ms | %.3f ms |" % (nr["SearchByBoW(KeyFrame, Frame)"]["gpu_ms"], nr["SearchByBoW(KeyFrame, Frame)"]["cpu_oracle_ms"]))
a, c = g("SearchForTriangulation x 20 neighbours (C ABI"), g("SearchByBoW x 10 relocalisation candidates (C ABI")
s = sub_line(s, "| `SearchForTriangulation` × 20 neighbours (`LocalMapping", "| `SearchForTriangulation` × 20 neighbours (`LocalMapping::CreateNewMapPoints`) | %.3f ms | **%.3f ms** | %.3f ms (20 calls) |" % (a["gpu_ms_host_views"], a["gpu_ms_resident_key_frames"], a["cpu_oracle_ms"]))
s = sub_line(s, "| `SearchByBoW` × 10 relocalisation candidates |", "| `SearchByBoW` × 10 relocalisation candidates | %.3f ms | **%.3f ms** | %.3f ms (10 calls) |" % (c["gpu_ms_host_views"], c["gpu_ms_resident_key_frames"], c["cpu_oracle_ms"]))
open(p, "w").write(s)

p = os.path.join(ROOT, "BASELINE.md"); s = open(p).read()
s = sub_line(s, "| (2) stereo 752×480, N=1200", "| (2) stereo 752×480, N=1200, extract L+R + `ComputeStereoMatches` | CPU, process pinned to 1 core | %.1f pairs/s | \"ORB Extraction\" %.1f ms · \"Stereo Matching\" %.2f ms per pair |" % (cb["one_core"]["value"], cb["one_core"]["stage_ms"]["ORB Extraction"], cb["one_core"]["stage_ms"]["Stereo Matching"]))
s = sub_line(s, "| | CPU, 2 cores", "| | CPU, 2 cores (left ‖ right as `Frame.cc:136-141`) | %.1f pairs/s | %.1f ms · %.2f ms |" % (cb["two_cores"]["value"], cb["two_cores"]["stage_ms"]["ORB Extraction"], cb["two_cores"]["stage_ms"]["Stereo Matching"]))
s = sub_line(s, "| | CPU, all 256 cores", "| | CPU, all 256 cores (128 concurrent constructors) | %.0f pairs/s | (contended: %.0f ms · %.1f ms per pair) |" % (cb["value"], cb["stage_ms"]["ORB Extraction"], cb["stage_ms"]["Stereo Matching"]))
s = sub_line(s, "| | **1× MI355X** |", "| | **1× MI355X** | **%.0f pairs/s** | %.3f ms per 128-pair step (median %.2f, p10 %.2f, p90 %.2f); kernels alone: \"ORB Extraction\" %.3f ms · \"Stereo Matching\" %.3f ms per step; one pair per call: %.3f ms |" % (b["value"], b["ms_per_step"], b["step_ms"]["median"], b["step_ms"]["p10"], b["step_ms"]["p90"], b["reference_stage_ms_alone"]["ORB Extraction"], b["reference_stage_ms_alone"]["Stereo Matching"], single))
s = sub_line(s, "| | 1× MI355X, inputs uploaded", "| | 1× MI355X, inputs uploaded over PCIe inside the timed region | %.0f pairs/s | %.3f ms per step, %.1f GB/s |" % (b["h2d_inclusive"]["value"], b["h2d_inclusive"]["ms_per_step"], b["h2d_inclusive"]["PCIe_GBps"]))
s = sub_line(s, "| (1) mono 752×480", "| (1) mono 752×480, N=1000, lapping {0,1000} | 1× MI355X | %.0f frames/s | %.3f ms per 128-frame step |" % (mono["value"], mono["ms_per_step"]))
s = sub_line(s, "| (3) fisheye stereo", "| (3) fisheye stereo 512×512, N=1500, lapping {0,511}, 2-NN + ratio + triangulation gate | 1× MI355X | %.0f pairs/s | %.3f ms per 128-pair step |" % (fe["value"], fe["ms_per_step"]))
s = sub_line(s, "| (4) RGB 640×480", "| (4) RGB 640×480 → grey on the device, N=1000, + `SearchLocalPoints` (5000 map points) per frame | 1× MI355X | %.0f frames/s | per-frame search call (%.2f ms device path + Python marshalling) dominates; extraction alone ≈ 90 k frames/s |" % (rg["value"], slp["gpu_ms_c_abi"]))
i0 = s.index("Roofline view (dominant kernel `k_fast_cells`"); i1 = s.index("target is exceeded")
s = s[:i0] + "Roofline view (dominant kernel `k_fast_cells`, 256 images per launch): 326 MB algorithmic / %.3f ms = %.0f GB/s = %.1f %% of 8 TB/s (PMC\ntraffic = the algorithmic bytes); the kernel is instruction-issue bound — 341 M VALU wave-instructions per launch = %.1f %% of the 1228.8 G\nwave-instr/s issue peak when it runs alone. End to end: 18.5 MB/pair × %.1f k pairs/s = %.2f TB/s = %.1f %% of 8 TB/s. The streaming kernels\nalone at 128 images per launch: `k_blur` 2.9 TB/s (37 %%), `k_import` 2.8 TB/s (35 %%), `k_resize_rows` ×7 2.3 TB/s (29 %%). The 2000 pairs/s\n" % (r["avg_launch_ms"], r["achieved"], 100 * r["frac"], 100 * r["valu_issue"]["alone_frac"], b["value"] / 1e3, r["end_to_end_GBps"] / 1e3, 100 * r["end_to_end_frac"]) + s[i1:]
open(p, "w").write(s)
p = os.path.join(ROOT, "INTEGRATION.md"); s = open(p).read()
s = re.sub(r"\d+\.\d k pairs/s at \d+ GB/s over PCIe, against \d+\.\d k with resident inputs\)", "%.1f k pairs/s at %.0f GB/s over PCIe, against %.1f k with resident inputs)" % (b["h2d_inclusive"]["value"] / 1e3, b["h2d_inclusive"]["PCIe_GBps"], b["value"] / 1e3), s)
open(p, "w").write(s)
print("docs refreshed: %.0f pairs/s, single pair %.3f ms" % (b["value"], single))
