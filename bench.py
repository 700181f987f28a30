#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): stereo pairs/s of ORB extract (left+right) + ComputeStereoMatches on 752x480
EuRoC-shaped rectified pairs, nFeatures=1200, 8 levels — configs[1] of BASELINE.json — on N GPUs of one node.

One "step" = one pass of the hot path over one batch of `--pairs` synthetic stereo pairs per GPU that are already
resident in HBM: import -> pyramid -> FAST cells -> quadtree -> blur -> IC-angle + rBRIEF -> stereo row search/SAD/
sub-pixel/median, and the results (keypoints, descriptors, uRight, depth) copied back to host memory.
Multi-GPU: independent image streams, one process per GPU, no data-path collective (weak scaling).

`--config` selects another BASELINE.json configuration (mono / fisheye / rgbd: parity cases, not the headline metric).
Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the roofline / cpu_baseline definitions.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NLEVELS, SCALE, INI, MIN = 8, 1.2, 20, 7                                   # Examples/*/*.yaml ORBextractor.*
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375                         # Examples/Stereo/EuRoC.yaml:17-20
BF, BASE = 458.654 * 0.110074, 0.110074                                    # EuRoC.yaml:23,57
HBM_PEAK_GBS = 8000.0                                                      # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2                              # 1024 SIMD-32 x 2.4 GHz, one wave64 instruction per 2 clk
# Examples/Stereo/TUM-VI.yaml:11-32
KB_CAM1 = [190.978477, 190.973307, 254.931706, 256.897442, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736]
KB_CAM2 = [190.442369, 190.434438, 252.598711, 254.917238, 0.003400603976, 0.001766924711, -0.002663898171, 0.000329921072]
KB_RLR = np.array([[0.999999445773493, 0.000791687752817, 0.000694034010224], [-0.000823363992158, 0.998899461915674, 0.046895490788700],
                   [-0.000656143613422, -0.046896036240590, 0.998899559977407]], np.float32)
KB_TLR = np.array([0.100931237881590, 0.000570764538347, 0.001046438762054], np.float32)

CONFIGS = {
    "stereo": dict(W=752, H=480, nf=1200, lap=(0, 0), kind="stereo", unit="stereo pairs/s",
                   metric="frames/sec ORB extract+match, 752x480 stereo @1200 feat (1 frame = 1 stereo pair; BASELINE.json configs[1])",
                   workload="EuRoC-shaped stereo 752x480, nFeatures=1200, 8 levels: extract L+R + ComputeStereoMatches (BASELINE.json configs[1])"),
    "mono": dict(W=752, H=480, nf=1000, lap=(0, 1000), kind="mono", unit="frames/s",
                 metric="frames/sec ORB extraction, 752x480 mono @1000 feat (BASELINE.json configs[0])",
                 workload="EuRoC-shaped mono 752x480, nFeatures=1000, 8 levels, lapping {0,1000}: ORBextractor::operator() (BASELINE.json configs[0])"),
    "fisheye": dict(W=512, H=512, nf=1500, lap=(0, 511), kind="fisheye", unit="stereo pairs/s",
                    metric="frames/sec ORB extract+match, 512x512 fisheye stereo @1500 feat (1 frame = 1 stereo pair; BASELINE.json configs[2])",
                    workload="TUM-VI-shaped fisheye stereo 512x512 (Kannala-Brandt), nFeatures=1500, lapping {0,511}: extract L+R + ComputeStereoFishEyeMatches "
                             "(2-NN + ratio + triangulation gate) (BASELINE.json configs[2])"),
    "rgbd": dict(W=640, H=480, nf=1000, lap=(0, 0), kind="rgbd", unit="frames/s",
                 metric="frames/sec ORB extract + SearchLocalPoints, 640x480 RGB-D @1000 feat, 5000 map points (BASELINE.json configs[3])",
                 workload="TUM-RGB-D-shaped 640x480 RGB frames, nFeatures=1000: cvtColor + extraction, ComputeStereoFromRGBD (uRight from the depth image), then per "
                          "frame isInFrustum + SearchByProjection (right-coordinate gate active, accept loop on the device) against a resident 5000-point local map, "
                          "one pose per frame, all frames of a step in one batch (BASELINE.json configs[3])"),
}


def level_pixels(W, H):
    inv = [1.0]
    s = np.float32(1.0)
    for _ in range(1, NLEVELS):
        s = np.float32(s * np.float64(np.float32(SCALE)))
        inv.append(float(np.float32(1.0) / s))
    return [int(np.rint(np.float32(W) * np.float32(i))) * int(np.rint(np.float32(H) * np.float32(i))) for i in inv]


def algorithmic_bytes(W, H, n_kp, n_cand):
    """Compulsory bytes per IMAGE of each extractor kernel and per PAIR of the stereo matcher."""
    px = level_pixels(W, H)
    P, P0, P7 = sum(px), px[0], px[-1]
    return {
        "import": 2 * P0,
        "pyramid": (P - P7) + (P - P0),
        "fast_cells": P,                       # SURVEY.md section 8(d) prices E3 on the pyramid pixels it reads; its 4-byte candidate records are reported apart
        "fast_cells_with_candidates": P + 4 * n_cand,
        "quadtree": 3 * 4 * n_cand + 4 * n_kp,
        "blur": 2 * P,
        "layout": 8 * n_kp,
        "orient_brief": (749 + 512 + 60) * n_kp,
        "match": 25 * 32 * n_kp + (121 + 11 * 121) * 0.7 * n_kp + 8 * n_kp,   # per pair
    }


def cpu_baseline(seconds_budget=6.0):
    """The reference CPU path on the host cores, same workload, same run: per stereo pair the reference's OWN stereo Frame constructor
    (src/Frame.cc:105-230, compiled unmodified into oracle/_ref/libref_frame.so with REGISTER_TIMES: two ORBextractor calls on two
    threads, Frame::ComputeStereoMatches, the grid assignment), long-lived extractors as Tracking holds them.  Three settings
    (SURVEY.md §8d): the process pinned to one core, to two cores (left || right as the reference runs them), and every core with
    cores / 2 independent pair streams; `value` is the last one.  The two stage times are the reference's own timers
    ("ORB Extraction" src/Frame.cc:132-146, "Stereo Matching" :158-170, printed at src/Tracking.cc:324-334).
    Without oracle/_ref the oracle restatement is timed ("port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import oracle_lib as ol
    from orb_slam3_detailed_comments_amd import synth
    c = CONFIGS["stereo"]
    W, H, NFEAT = c["W"], c["H"], c["nf"]
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = len(avail)
    if ol.reference_frame_lib() is not None:
        def run(streams, secs):
            pairs = [synth.stereo_pair(W, H, seed=1000 + i) for i in range(streams)]
            out = [None] * streams

            def work(t):
                out[t] = ol.reference_frame_repeat(pairs[t][0], pairs[t][1], secs, NFEAT, fx=FX, bf=BF)
            t0 = time.time()
            th = [threading.Thread(target=work, args=(t,)) for t in range(streams)]
            [x.start() for x in th]
            [x.join() for x in th]
            dt = time.time() - t0
            n = sum(o[0] for o in out)
            ext = sum(o[3] for o in out) / max(n, 1); st = sum(o[4] for o in out) / max(n, 1)
            return n, dt, out[0][2], ext, st

        res = {}
        for name, cpus in (("one_core", avail[:1]), ("two_cores", avail[:2])):
            if hasattr(os, "sched_setaffinity"):
                os.sched_setaffinity(0, set(cpus))
            n, dt, matches, ext, st = run(1, seconds_budget)
            res[name] = {"value": round(n / dt, 2), "cores": len(cpus), "pair_streams": 1, "pairs": n, "seconds": round(dt, 2),
                         "stage_ms": {"ORB Extraction": round(ext, 3), "Stereo Matching": round(st, 3)}, "stereo_matches_last_pair": matches}
        if hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, set(avail))
        # every core: one PROCESS per pair stream (a fresh interpreter: no threads, no GPU runtime inherited), each pinned to two cores of its own - the
        # two std::threads of the reference's constructor.  (Rounds 1-4 ran the streams as threads of this process: 128 streams x 2 std::threads created
        # per frame in one address space measured the allocator and thread creation, 11.6x on 256 cores.)
        quota = cpu_quota_cores()
        usable = cores if quota is None else max(1, min(cores, int(quota + 0.5)))      # a container may see every core of the host and be granted a few
        streams = max(1, usable // 2)
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        start_at = time.time() + 2.0 + 0.02 * streams
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%.3f,%.3f,%d" % (avail[2 * t], avail[min(2 * t + 1, cores - 1)], start_at, seconds_budget, 1000 + t)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for t in range(streams)]
        outs = []
        for pr in procs:
            o, _ = pr.communicate()
            lines = [l for l in o.splitlines() if l.startswith("{")]
            if pr.returncode == 0 and lines:
                outs.append(json.loads(lines[-1]))
        if outs:
            n = sum(o["pairs"] for o in outs)
            window = max(o["t_end"] for o in outs) - min(o["t_start"] for o in outs)
            per = [o["pairs"] / o["seconds"] for o in outs]
            ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            busy = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
            res["all_cores"] = {"value": round(n / window, 2), "cores": usable, "visible_cores": cores, "cpu_quota_cores": quota,
                                # CPU time the worker processes were actually given (setup included) per second of the measurement window: far below
                                # 2 x pair_streams means the box throttles (cgroup limit, other tenants), whatever the core count says
                                "cpu_seconds_of_workers": round(busy, 1), "pair_streams": len(outs), "pairs": n, "seconds": round(window, 2),
                                "stage_ms": {"ORB Extraction": round(sum(o["ext_ms"] for o in outs) / max(n, 1), 3), "Stereo Matching": round(sum(o["stereo_ms"] for o in outs) / max(n, 1), 3)},
                                "stereo_matches_last_pair": outs[0]["matches"], "processes": len(outs), "late_starters": sum(1 for o in outs if o["late"]),
                                "per_stream_pairs_per_s": {"min": round(min(per), 2), "max": round(max(per), 2)},
                                "scaling_efficiency_vs_two_cores": round(n / window / (res["two_cores"]["value"] * len(outs)), 3)}
        else:
            res["all_cores"] = dict(res["two_cores"], note="worker processes failed; two_cores repeated")
        a = res["all_cores"]
        sample = ("%d pairs in %.1f s: %d concurrent PROCESSES, each one stream of the reference's own stereo Frame constructor (src/Frame.cc:105-230 = 2 extractor threads + "
                  "ComputeStereoMatches + grid) pinned to 2 cores of its own, on %d cores; one_core / two_cores = one stream with the process pinned to 1 / 2 cores; OpenCV "
                  "primitives are the scalar shim, not SIMD OpenCV, so this under-states a real OpenCV build" % (a["pairs"], a["seconds"], a["pair_streams"], a["cores"]))
        return {"value": a["value"], "unit": "stereo pairs/s", "cores": a["cores"], "kind": "reference", "sample": sample,
                "one_core": res["one_core"], "two_cores": res["two_cores"], "all_cores": a, "stage_ms": a["stage_ms"]}
    pairs = [synth.stereo_pair(W, H, seed=1000 + i) for i in range(cores)]
    done = [0] * cores
    state = [(ol.OracleExtractor(NFEAT), ol.OracleExtractor(NFEAT)) for _ in range(cores)]
    t_end = time.time() + 2.5 * seconds_budget

    def work(t):
        oL, oR = state[t]
        L, R = pairs[t]
        while time.time() < t_end:
            (mL, kL, dL), (mR, kR, dR) = oL.extract(L), oR.extract(R)
            ol.oracle_stereo(oL, oR, kL, dL, kR, dR, BF, BASE)
            done[t] += 1

    t0 = time.time()
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t0
    n = sum(done)
    sample = "%d pairs in %.1f s on %d threads (one pair stream per thread); oracle restatement of extractor + stereo association" % (n, dt, cores)
    return {"value": round(n / dt, 2), "unit": "stereo pairs/s", "cores": cores, "kind": "port", "sample": sample}


def cpu_quota_cores():
    """CPU time this container is granted per second, in cores (cgroup v2 cpu.max / v1 cfs quota); None = no limit found."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_worker(spec):
    """One pair stream of cpu_baseline's all-core setting, in a process of its own: pinned to two cores, waits for the common start time, runs the
    reference's stereo Frame constructor for the given time and prints its counts."""
    c0, c1, start_at, secs, seed = spec.split(",")
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, {int(c0), int(c1)})
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from orb_slam3_detailed_comments_amd import synth
    c = CONFIGS["stereo"]
    L, R = synth.stereo_pair(c["W"], c["H"], seed=int(seed))
    ol.reference_frame_repeat(L, R, 0.0, c["nf"], fx=FX, bf=BF)                   # library loaded, tables built, one frame through
    late = time.time() > float(start_at)
    while time.time() < float(start_at):
        time.sleep(0.001)
    t0 = time.time()
    n, el, m, ext, st = ol.reference_frame_repeat(L, R, float(secs), c["nf"], fx=FX, bf=BF)
    print(json.dumps({"pairs": n, "seconds": el, "t_start": t0, "t_end": time.time(), "matches": m, "ext_ms": ext, "stereo_ms": st, "late": late}))


def kernel_sources_sha():
    """sha256 (16 hex digits) over the device sources of the library: the stamp that ties counter summaries under profiles/ to the kernels they measured."""
    import hashlib
    d = os.path.join(ROOT, "orb_slam3_detailed_comments_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        # device code: the kernel files and every header they include; orbx_internal.h / orbx_rt.h are host-only (the handle, the runtime wrappers) and no
        # .hip file includes them - a change there does not touch a kernel
        if f.endswith((".hip", ".h", ".inc")) and f not in ("orbx_internal.h", "orbx_rt.h"):
            hsh.update(f.encode()); hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def thread_cpu_seconds():
    """CPU seconds (user + system) of every thread of this process, by thread id, with the thread's name: which thread the host cost of a step sits in"""
    out = {}
    try:
        tck = float(os.sysconf("SC_CLK_TCK"))
        for t in os.listdir("/proc/self/task"):
            try:
                st = open("/proc/self/task/%s/stat" % t).read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(t)] = (name, (int(f[11]) + int(f[12])) / tck)
            except Exception:
                continue
    except Exception:
        pass
    return out


def pcie_links():
    """What sysfs says about the PCIe links of the AMD GPUs of this host (speed, width, NUMA node): the host-fed rate follows the link, and boxes differ."""
    import glob
    out = []
    for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002" or not open(os.path.join(d, "class")).read().strip().startswith(("0x0302", "0x0300", "0x1200")):
                continue
            rd = lambda n: open(os.path.join(d, n)).read().strip() if os.path.exists(os.path.join(d, n)) else None
            out.append({"bdf": os.path.basename(d), "current_link_speed": rd("current_link_speed"), "current_link_width": rd("current_link_width"),
                        "max_link_speed": rd("max_link_speed"), "max_link_width": rd("max_link_width"), "numa_node": rd("numa_node"), "local_cpulist": rd("local_cpulist")})
        except Exception:
            continue
    return out


def live_traffic(dom, timeout=60):
    """HBM bytes of the dominant kernel, measured now: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter per pass, nothing else enabled), each over a
    child run of this file with 64 pairs per step on three handles (= 128 images per launch, the shape of profiles/pmc_traffic.json).  Returns None when rocprofv3 is
    missing or a pass fails; the caller then keeps the tracked figure.  Units and the gfx950 caveat as in tools/make_pmc_traffic.py: KB per dispatch, 4-B/lane loads."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    kname = {"fast_cells": "k_fast_cells", "blur": "k_blur", "quadtree": "k_quadtree", "orient_brief": "k_orient_brief", "match": "k_stereo_match", "pyramid": "k_resize"}.get(dom)
    if not exe or not kname:
        return None
    tmp = tempfile.mkdtemp(prefix="orbx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "2", "--pairs", "64", "--handles", "3", "--no-cpu-baseline", "--no-h2d", "--no-other-configs",
             "--no-latency", "--no-parity-check", "--no-live-traffic", "--min-seconds", "0"]
    out = {}
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            pr = subprocess.Popen([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child, cwd="/tmp", env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                pr.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                pr.kill(); pr.wait()                      # this very process, by its handle
                return None
            n = 0; tot = 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kname in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        n += 1; tot += float(r["Counter_Value"])
            if n == 0:
                return None
            out[counter] = {"dispatches": n, "avg_KB": round(tot / n, 1)}
        mult = 7 if kname == "k_resize" else 1
        return {"kernel": kname, "FETCH_SIZE": out["FETCH_SIZE"], "WRITE_SIZE": out["WRITE_SIZE"],
                "bytes_per_launch_128_images": int((out["FETCH_SIZE"]["avg_KB"] + out["WRITE_SIZE"]["avg_KB"]) * 1024 * mult),
                "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --pairs 64 --handles 3 --steps 4` inside this run", "seconds": round(time.time() - t0, 1)}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(seconds=2.5):
    """BASELINE.json configs 0, 2 and 3 beside the headline, in the driver's own line: each is this file run as a child process (`--config X`), i.e. the
    same setup, warm-up, barrier-bracketed timed region (>= `seconds`) and roofline arithmetic as the headline, reduced to the key figures."""
    import subprocess
    out = {}
    for name in ("mono", "fisheye", "rgbd", "stereo_natural"):
        # stereo_natural: the headline's configuration on the second generator (--workload natural: camera-like imagery, 1-5 % FAST corner density
        # instead of the corner field's 23.5 %) - no EuRoC frame can be had offline, this is the closest the run can get to one
        cmd = [sys.executable, os.path.abspath(__file__), "--config", "stereo" if name == "stereo_natural" else name, "--steps", "20", "--warmup", "5", "--min-seconds", str(seconds),
               "--no-cpu-baseline", "--no-h2d", "--no-other-configs", "--no-latency", "--no-live-traffic"] + (["--workload", "natural"] if name == "stereo_natural" else [])
        t0 = time.time()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
            if pr.returncode != 0 or not line:
                out[name] = {"value": None, "error": (pr.stderr or pr.stdout)[-400:]}
                continue
            r = json.loads(line[-1])
            out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "timed_seconds": r["timed_seconds"], "timed_steps": r["timed_steps"],
                         "units_per_step": r["config"]["units_per_step_per_gpu"], "workload": r["config"]["workload"], "input_mode": r["config"]["input_mode"],
                         "roofline": {k: r["roofline"][k] for k in ("kernel", "frac", "achieved", "avg_launch_ms", "alone_launch_ms", "end_to_end_frac")},
                         "avg_keypoints_per_image": r["config"]["avg_keypoints_per_image"], "avg_matches_per_unit": r["config"]["avg_matches_per_unit"],
                         "generator": r["config"].get("generator"), "fast_corner_density": next((v for k, v in r["config"].items() if k.startswith("fast_corner_density")), None),
                         "parity_check": r.get("parity_check"),
                         "wall_seconds_of_child": round(time.time() - t0, 1)}
        except Exception as e:
            out[name] = {"value": None, "error": repr(e)}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one process per GPU) under torch.distributed.run on
    127.0.0.1 and pass its output and exit code through.  (The round driver starts the ranks itself; WORLD_SIZE is then set and this is skipped.)"""
    import socket
    import subprocess
    if os.environ.get("ORBX_BENCH_BACKEND", "nccl") == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py --gpus %d: this node shows %d GPU(s); one process per GPU needs %d (ORBX_BENCH_BACKEND=gloo lets ranks share a GPU for "
                             "protocol tests)\n" % (n, have, n))
            return 2
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="stereo", choices=sorted(CONFIGS), help="BASELINE.json configuration (stereo = configs[1] = the headline metric)")
    ap.add_argument("--pairs", type=int, default=128, help="units (stereo pairs, or frames for mono / rgbd) per step per GPU: one step = one batch through the "
                    "whole path (32 -> 52 k, 64 -> 57 k, 96 -> 59 k, 128 -> 60 k pairs/s measured in round 1)")
    ap.add_argument("--handles", type=int, default=0, help="extractor handles in flight per GPU (each owns its streams).  Default: 4 "
                    "(2 / 3 / 4 / 5 / 6 handles: 68.0 / 75.7 / 80.3 / 70.2 / 69.0 k pairs/s in round 2), 3 with --allgather: a live RCCL communicator "
                    "has streams of its own on the same hardware queues and four chains then fall behind three (86 k against 95 k pairs/s)")
    ap.add_argument("--workload", default="corner_field", choices=["corner_field", "natural"], help="synthetic image generator: corner_field = the headline "
                    "workload (random rectangles, 24-52 %% of the pixels are FAST corners at t = 7); natural = 1/f^2 spectrum + sparse edges (1-5 %% corners, "
                    "camera-like statistics) - a second reported line, never the headline")
    ap.add_argument("--allgather", action="store_true", help="BASELINE.json configs[4]: after every batch all-gather the descriptor blocks [B, cap, 32] + counts "
                    "of all ranks (RCCL over xGMI) inside the timed region, on a side stream beside the next extraction; the line then also carries the "
                    "collective's own time per batch")
    ap.add_argument("--min-seconds", type=float, default=6.0, help="the timed region repeats whole blocks of --steps steps until it lasts at least this long "
                    "(`repeats` in the output; value = all timed units / the whole timed region).  6 s by default: long enough for an outside observer "
                    "sampling GPU activity every 5 s (the round driver's rocm-smi sampler) to see the device busy")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` block (mono / fisheye / rgbd = BASELINE.json configs 0, 2, 3, each "
                    "measured for >= 1 s by this same file in a child process after the headline)")
    ap.add_argument("--import-copy", action="store_true", help="keep the resident inputs in a separate linear device buffer and copy them into the pyramid inside "
                    "every step (the rounds 1-2 measurement) instead of letting the producer write pyramid level 0 directly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the post-run check of the last timed step's outputs against the reference (oracle/_ref)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-pair latency loop (220 tiny launches of every kernel: profiling runs leave it out so that "
                    "per-kernel averages of rocprofv3 are averages over the timed launches)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the second, PCIe-inclusive measurement (never `value`)")
    ap.add_argument("--h2d", action="store_true", help="make the PCIe-inclusive variant the timed loop (NOT the headline value)")
    ap.add_argument("--host-wait", choices=["auto", "spin", "block"], default="auto",
                    help="how the host thread waits for the GPU (orbx_set_host_wait): spin = the HIP default, lowest latency, one core per waiting thread; block = sleep on the "
                         "completion interrupt. auto = spin for one rank, block for --gpus N > 1, where N ranks share the host's cores (8 ranks x 2 spinning threads on a "
                         "16-core grant is the whole grant)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic with two short rocprofv3 --pmc passes inside this run (keep the tracked figure); the passes belong to the full "
                    "default line and are also left out with --no-other-configs and under a profiler")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)        # one pair stream of cpu_baseline's all-core setting (internal)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    cfg = CONFIGS[args.config]
    W, H, NFEAT, LAP, kind = cfg["W"], cfg["H"], cfg["nf"], cfg["lap"], cfg["kind"]
    paired = kind in ("stereo", "fisheye")

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.allgather or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1":    # the env switch lets a 1-GPU box exercise the torch.distributed path
        import torch
        import torch.distributed as dist_
        # ORBX_BENCH_BACKEND=gloo: test switch - several ranks may then share one GPU (RCCL refuses that) or run without one (ORBX_BENCH_LIB), which
        # lets a box without 8 GPUs run the multi-process path end to end; the barrier and the max-reduce go over CPU tensors in that case
        backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")
        coll_dev = "cpu"                                # where the barriers and the max over ranks run
        kw = {}
        if "WORLD_SIZE" not in os.environ:              # --allgather in a plain process: a one-rank group on a free local port
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1])); sk.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = dict(rank=0, world_size=1)
        if backend == "nccl":
            ndev = torch.cuda.device_count()
            if ndev < world:
                raise SystemExit("bench.py: %d ranks but %d visible GPU(s); one process per GPU (ORBX_BENCH_BACKEND=gloo lets ranks share a device in "
                                 "protocol tests)" % (world, ndev))
            torch.cuda.set_device(local)                # rank -> GPU: LOCAL_RANK
            # The hot path has no collective: ranks only meet at the barriers around the timed region and in the max over ranks, and those go
            # over CPU tensors (gloo).  The RCCL communicator is created lazily, by the first collective on device tensors (--allgather), so
            # a plain scaling run has no RCCL streams competing with the extractors' hardware queues: with the communicator alive four
            # handles per GPU fall from 101 k to 86 k pairs/s (three: 95 k), without it a rank runs as a plain process does
            # (profiles/r03_final/dist_handles.txt).
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: gloo need not resolve the container's hostname
            try:
                dist_.init_process_group("cpu:gloo,cuda:nccl", **kw)
            except Exception as e:                                       # no usable gloo transport: everything over RCCL, as in rounds 1-2
                sys.stderr.write("bench.py: gloo + nccl group failed (%r); falling back to an nccl-only group\n" % (e,))
                if dist_.is_initialized():
                    dist_.destroy_process_group()
                dist_.init_process_group("nccl", device_id=torch.device("cuda", local), **kw)
                coll_dev = "cuda"
        else:
            local = 0
            dist_.init_process_group(backend, **kw)
        dist = dist_
        dist_on_gpu = backend == "nccl"

        def dist_barrier():                             # an all-reduce over CPU tensors: every rank has to arrive
            dist_.all_reduce(torch.zeros(1, device=coll_dev))

    from orb_slam3_detailed_comments_amd import ORBextractor, load_hip, synth, _lib, sophus
    from orb_slam3_detailed_comments_amd import matcher as M
    # ORBX_BENCH_LIB: test switch (tests/test_multi_gloo.py runs this file's distributed path on the CPU emulator build of the kernels)
    lib = _lib.OrbxLib(os.environ["ORBX_BENCH_LIB"]) if os.environ.get("ORBX_BENCH_LIB") else load_hip()
    host_wait_mode = args.host_wait if args.host_wait != "auto" else ("block" if world > 1 else "spin")
    if host_wait_mode == "block" and not os.environ.get("ORBX_BENCH_LIB"):
        if lib.L.orbx_set_host_wait(local, 1) != 0:          # never fatal: a rank that cannot block spins, as in rounds 1-5
            host_wait_mode = "spin (blocking waits refused: %s)" % (lib.L.orbx_last_error() or b"").decode()
    P = args.pairs
    # synthetic inputs shaped like the configuration's dataset; every rank (= camera stream shard) gets its own seeds
    nat = args.workload == "natural"
    if paired:
        ls, rs = [], []
        for i in range(P):
            sd = rank * 100003 + i
            if nat:
                l, r = synth.natural_stereo_pair(W, H, seed=sd) if kind == "stereo" else synth.natural_stereo_pair(W, H, seed=sd, max_disp=24, band=64)
            else:
                l, r = synth.stereo_pair(W, H, seed=sd) if kind == "stereo" else synth.stereo_pair(W, H, seed=sd, nrect=2000, max_disp=24, band=64)
            ls.append(l); rs.append(r)
        batch = np.stack(ls + rs)                               # [2P, H, W]: lefts then rights
    elif kind == "mono":
        batch = np.stack([(synth.natural(W, H, seed=rank * 100003 + i) if nat else synth.corner_field(W, H, seed=rank * 100003 + i)) for i in range(P)])
    else:
        nrect = int(3000 * W * H / (752 * 480))
        gen = (lambda sd: synth.natural(W, H, seed=sd)) if nat else (lambda sd: synth.corner_field(W, H, seed=sd, nrect=nrect))
        batch = np.stack([np.stack([gen(rank * 100003 + i + 7 * c) for c in range(3)], axis=2) for i in range(P)])
    # the statistic the FAST kernel's cost depends on: share of level-0 pixels that are FAST-9/16 corners at minThFAST (first images of the batch)
    probe = [batch[i] if batch.ndim == 3 else batch[i][..., 1] for i in range(min(4, len(batch)))]
    fast_density = float(np.mean([synth.fast_corner_density(im, MIN) for im in probe]))
    NIMG = 2 * P if paired else P
    NH = args.handles if args.handles > 0 else (3 if (dist is not None and (args.allgather or coll_dev == "cuda")) else 4)
    handles = [ORBextractor(NFEAT, SCALE, NLEVELS, INI, MIN, device_id=local, lib=lib) for _ in range(NH)]
    if os.environ.get("ORBX_BENCH_QT_LDS_NODES"):            # experiment switch: quadtree levels above this many nodes keep their node lists in the global pool
        for h in handles:
            h.debug_quadtree_lds_nodes(int(os.environ["ORBX_BENCH_QT_LDS_NODES"]))
    if kind == "rgbd":
        for h in handles:
            h.set_input(3, rgb=True)
    # inputs resident in HBM before the timed region.  Grey frames are written where the extractor wants them - its own pyramid level 0
    # (orbx_input_buffer: what a camera DMA or a decoder on the GPU would do) - so no import pass copies them there inside the step;
    # --import-copy keeps them in a separate linear device buffer, as rounds 1-2 measured (one more read + write of every pixel per step)
    shape3 = batch.shape[:3]; stride = lin_stride = batch.strides[1]; istride = None
    zero_copy = kind != "rgbd" and not args.import_copy
    if zero_copy:
        ups = [h.input_upload(batch) for h in handles]
        dptrs = [u[0] for u in ups]; stride = ups[0][2]; istride = ups[0][3]
    else:
        dptrs = [h.device_upload(batch) for h in handles]
    cap = handles[0].max_keypoints()
    for h in handles:
        h.profile(not os.environ.get("ORBX_BENCH_NOPROFILE"), serial=bool(os.environ.get("ORBX_BENCH_SERIAL")))
    out = [dict(k=h.pinned_empty((NIMG, cap, 28), np.uint8), d=h.pinned_empty((NIMG, cap, 32), np.uint8), n=np.zeros(NIMG, np.int32),
                m=np.zeros(NIMG, np.int32), u=h.pinned_empty((P, cap), np.float32), z=h.pinned_empty((P, cap), np.float32), nm=np.zeros(P, np.int32),
                l2r=np.zeros((P, cap), np.int32), r2l=np.zeros((P, cap), np.int32), p3=np.zeros((P, cap, 3), np.float32))
           for h in handles]
    stage_sum = {}
    stage_cnt = [0]
    nkp = [0, 0]
    nmatch = [0, 0]
    step_end = []

    import ctypes as C
    kb = None
    if kind == "fisheye":
        kb = M.KB8Stereo()
        kb.cam1[:] = KB_CAM1; kb.cam2[:] = KB_CAM2; kb.t12[:] = KB_TLR.tolist()
        kb.R12[:] = sophus.SE3f(KB_RLR, KB_TLR).rotationMatrix().ravel().tolist()          # mRlr = mTlr.rotationMatrix() (src/Frame.cc:1498-1501)
    local_map = None
    if kind == "rgbd":
        # BASELINE.json configs[3]: per frame ComputeStereoFromRGBD (uRight from the depth image) + Tracking::SearchLocalPoints against a local map
        # of 5000 points (50 key frames' worth), all of it on the device (orbm_stereo_from_depth + orbm_search_local_points_batch): the frames are
        # read where the extractor left them, the local map is resident, every frame has its own pose; mvuRight / mvDepth, the keypoint -> map
        # point assignments and the match counts come back to the host inside the timed region.
        RGBD_BF = 40.0                                              # Examples/RGB-D/TUM1.yaml Camera.bf
        rng = np.random.default_rng(7 + rank)
        yy, xx = np.mgrid[0:H, 0:W]
        depth = (2.5 + 1.5 * np.sin(xx / 57.0) * np.cos(yy / 43.0)).astype(np.float32)      # the scene's depth (m), the same for every frame
        depth[rng.uniform(size=(H, W)) < 0.1] = 0.0                                          # missing readings
        handles[0].enqueue(None, LAP, device_ptr=dptrs[0], shape=shape3, stride=stride); r_all = handles[0].fetch()
        Mp = 5000
        fsrc = rng.integers(0, P, Mp)                                # the key frame (here: batch frame) a map point was created from
        X = np.zeros((Mp, 3), np.float32); dsc = np.zeros((Mp, 32), np.uint8); octv = np.zeros(Mp, np.int64)
        for i in range(Mp):
            k0, d0 = r_all[fsrc[i]][1], r_all[fsrc[i]][2]
            j = int(rng.integers(0, len(k0)))
            u, v = float(k0["x"][j]), float(k0["y"][j])
            z = float(depth[int(v), int(u)]) or float(rng.uniform(1.0, 4.0))
            z *= float(rng.uniform(0.97, 1.03))
            X[i] = ((u + rng.normal(0, 0.7) - CX) / FX * z, (v + rng.normal(0, 0.7) - CY) / FY * z, z)
            dsc[i] = d0[j]; octv[i] = k0["octave"][j]
            if rng.uniform() < 0.6:
                for bit in rng.choice(256, int(rng.integers(0, 41)), replace=False):
                    dsc[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
            else:
                dsc[i] = rng.integers(0, 256, 32, dtype=np.uint8)
        dn = np.linalg.norm(X, axis=1); nrm = (X / dn[:, None]).astype(np.float32)
        maxd = (dn * 1.2 ** octv).astype(np.float32); mind = (maxd / 1.2 ** 7).astype(np.float32)
        sfs = handles[0].GetScaleFactors()
        poses = []
        for b in range(P):                                           # small per-frame motion around the map's reference pose
            a = rng.normal(0, 0.004, 3); t = rng.normal(0, 0.01, 3).astype(np.float32)
            Rx = np.array([[1, -a[2], a[1]], [a[2], 1, -a[0]], [-a[1], a[0], 1]], np.float64)
            uu, _, vv = np.linalg.svd(Rx)
            poses.append(((uu @ vv).astype(np.float32), t))
        local_map = []
        for h in handles:
            rp = M.ResidentPoints(h, X, nrm, mind, maxd, dsc)
            lp = M.LocalPointsBatch(h, rp, P, (FX, FY, CX, CY), (0.0, float(W), 0.0, float(H)), RGBD_BF, sfs)
            lp.set_poses(poses)
            local_map.append(dict(rp=rp, lp=lp, depth=h.device_upload(np.broadcast_to(depth, (P, H, W)).copy()), bf=RGBD_BF,
                                  depth_host=depth, poses=poses, X=X, nrm=nrm, mind=mind, maxd=maxd, dsc=dsc))

    # PCIe-inclusive variant: page-locked host copies of the inputs and two device buffers per handle (upload of the next batch beside the kernels)
    host_in, dbuf, dsel = None, None, None

    def setup_h2d():
        nonlocal host_in, dbuf, dsel
        if host_in is None:
            host_in, dbuf, dsel = [], [], [0] * NH
            for h in handles:
                b = h.pinned_empty(batch.shape, np.uint8); b[...] = batch; host_in.append(b)
                dbuf.append([h.device_alloc(batch.nbytes), h.device_alloc(batch.nbytes)])

    def enqueue(i, h2d=False):
        h = handles[i]
        if h2d:
            # the upload of this batch was issued when the previous one of this handle was enqueued (or just now, the first time)
            if not getattr(h, "_primed", False):
                h.device_upload_async(dbuf[i][dsel[i]], host_in[i]); h._primed = True
            h.enqueue(None, LAP, device_ptr=dbuf[i][dsel[i]], shape=shape3, stride=lin_stride)
            dsel[i] ^= 1
            h.device_upload_async(dbuf[i][dsel[i]], host_in[i])        # next batch of this handle, into the other buffer
        else:
            h.enqueue(None, LAP, device_ptr=dptrs[i], shape=shape3, stride=stride, image_stride=istride)
        if kind == "stereo":
            lib.check(lib.L.orbm_stereo_match(h._h, 0, h._h, P, P, BF, BASE))
        elif kind == "fisheye":
            lib.check(lib.L.orbm_stereo_fisheye(h._h, 0, h._h, P, P, C.byref(kb)))
        elif kind == "rgbd":
            lm = local_map[i]
            M.ComputeStereoFromRGBD(h, None, lm["bf"], device_ptr=lm["depth"], shape=(P, H, W))
            lm["lp"].enqueue(0, use_u_right=True, viewing_cos_limit=0.5, th=3.0, far_points=False, nnratio=0.8)     # th = 3: the RGB-D setting, src/Tracking.cc:4038-4039

    # --allgather (BASELINE.json configs[4]): the descriptor blocks of every finished batch go to all ranks; the collective of batch i runs on a
    # side stream beside the extraction of the following batches and is waited for when its handle comes round again
    ag_works = [None] * NH
    ag_stream = None
    lib_comm = None
    if args.allgather:
        from orb_slam3_detailed_comments_amd import multi
        if dist_on_gpu:
            # the exchange runs INSIDE the library (include/orbx.h: orbx_comm_*, orbx_allgather_descriptors - the entry a C++ host calls): RCCL on the
            # communicator's own stream behind a device-to-device snapshot; torch.distributed only carries rank 0's ncclUniqueId to the other ranks
            lib_comm = multi.Communicator.from_torch_distributed(lib, local)
        # (gloo test runs keep the torch.distributed path: their ranks are processes without GPUs, which the emulator's in-process stand-in cannot join)

    def ag_wait(i):
        if lib_comm is not None:
            if ag_works[i] is not None:
                lib_comm.wait(); ag_works[i] = None
            return
        if ag_works[i] is not None:
            for w in ag_works[i][2]:
                w.wait()
            ag_works[i] = None

    def fetch(i, record, gather=True):
        h, o = handles[i], out[i]
        lib.check(lib.L.orbx_fetch(h._h, o["k"].ctypes.data, o["d"].ctypes.data, cap, o["n"].ctypes.data, o["m"].ctypes.data))
        if args.allgather and gather:            # (collective: every rank calls it the same number of times)
            ag_wait(i)
            if lib_comm is not None:
                lib_comm.all_gather(h); ag_works[i] = True          # one communicator: it waits for its previous exchange itself
            else:
                ag_works[i] = multi.all_gather_extracted(h, ag_stream)
        if kind == "stereo":
            lib.check(lib.L.orbm_stereo_fetch(h._h, P, o["u"].ctypes.data, o["z"].ctypes.data, cap, o["nm"].ctypes.data))
        elif kind == "fisheye":
            lib.check(lib.L.orbm_stereo_fisheye_fetch(h._h, P, o["l2r"].ctypes.data, o["r2l"].ctypes.data, o["z"].ctypes.data, o["p3"].ctypes.data, o["nm"].ctypes.data, cap))
        elif kind == "rgbd":
            lib.check(lib.L.orbm_stereo_fetch(h._h, P, o["u"].ctypes.data, o["z"].ctypes.data, cap, o["nm"].ctypes.data))
            asg_frames, nm_frames, _ = local_map[i]["lp"].fetch()
            o["nm"][0] = int(nm_frames.sum())
            o["lp_asg"], o["lp_nm"] = asg_frames, nm_frames             # (kept for the post-run parity check)
        if record:
            step_end.append(time.perf_counter())
            for k, v in h.stage_ms().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            stage_cnt[0] += 1
            nkp[0] += int(o["n"].sum()); nkp[1] += NIMG
            nmatch[0] += int(o["nm"].sum()) if kind != "rgbd" else int(o["nm"][0]); nmatch[1] += P

    def run(nsteps, record, h2d=False):
        pending = []
        for s in range(nsteps):
            i = s % NH
            if len(pending) == NH:
                fetch(pending.pop(0), record)
            enqueue(i, h2d)
            pending.append(i)
        while pending:
            fetch(pending.pop(0), record)
        for i in range(NH):
            ag_wait(i)

    def sync_all():
        for h in handles:
            h.sync()
        if dist is not None:
            if dist_on_gpu:
                import torch
                torch.cuda.synchronize()
            dist_barrier()

    timed_cpu = [0.0]

    def timed(nsteps, h2d, min_seconds=0.0):
        """W untimed warm-up steps, then ONE timed region of `repeats` whole blocks of nsteps steps between two barriers (repeats = 1 unless the
        warm-up predicts a region shorter than min_seconds: 20 steps last 30 ms, and four handles' completions bunch differently from run to run).
        Returns (seconds, completion intervals in ms, repeats)."""
        nwarm = args.warmup if not h2d else min(6, nsteps)
        sync_all()
        tw = time.perf_counter()
        run(nwarm, False, h2d)
        sync_all()
        tw = time.perf_counter() - tw
        repeats = 1
        if nwarm > 0 and min_seconds > 0:
            # the warm-up steps are slower than steady state (first launches, the clocks ramping up) and would predict too few blocks: a second untimed
            # run of the same length, now warm, sets the rate the prediction uses, so that the timed region does last min_seconds
            ncal = max(nwarm, 40)                           # (a handful of steps is dominated by filling and draining the handles in flight)
            tc = time.perf_counter()
            run(ncal, False, h2d)
            sync_all()
            tc = time.perf_counter() - tc
            repeats = int(min(max(1, np.ceil(1.03 * min_seconds / max(tc / ncal * nsteps, 1e-6))), 10000))
        if dist is not None:                       # every rank times the same number of steps
            import torch
            t = torch.tensor([repeats], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
            sync_all()
        del step_end[:]
        thr0 = thread_cpu_seconds()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        run(nsteps * repeats, True, h2d)
        sync_all()
        dt = time.perf_counter() - t0
        timed_cpu[0] = time.process_time() - cpu0           # host CPU seconds (user + system, all threads) this rank spent inside the timed region
        thr1 = thread_cpu_seconds()
        timed_cpu[1:] = [sorted(((round(thr1[t][1] - thr0.get(t, (None, 0.0))[1], 3), thr1[t][0]) for t in thr1), reverse=True)[:4]]
        ends = [t0] + list(step_end)
        per = np.diff(np.array(ends)) * 1e3
        return dt, per, repeats

    # setup, not a warm-up step: every handle allocates its device buffers and uploads its tables on first use
    for i in range(NH):
        enqueue(i); fetch(i, False)
    if args.h2d:
        setup_h2d()
    dt, per_step, repeats = timed(args.steps, args.h2d, args.min_seconds)
    own_cpu_s = timed_cpu[0]
    own_threads = timed_cpu[1] if len(timed_cpu) > 1 else None

    def parity_check(per_handle=4):
        """CHECKER, after the timed region: what the LAST timed step of every handle left in its output buffers - keypoints, descriptors and the
        association results of `per_handle` units per handle (other units on every handle) - against the reference's own code compiled into
        oracle/_ref (src/Frame.cc:105-230 stereo constructor; :1432-1528 fisheye rig; :235-345 RGB-D constructor + Frame::isInFrustum +
        ORBmatcher::SearchByProjection; src/ORBextractor.cc for mono), or the oracle restatement where oracle/_ref is absent.  Byte-identical
        or the run fails (exit code 3).  tests/headline_check.py is test infrastructure: nothing here runs inside the timed region."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        import headline_check as hc
        t0 = time.time()
        checked, mism, against = 0, [], None
        for i in range(NH):
            o = out[i]
            for j in range(per_handle):
                p = (i * per_handle + j) % P
                if kind == "stereo":
                    e = hc.StereoExpectation(batch[p], batch[P + p], NFEAT, FX, BF, BASE)
                    bad = e.differences(o["k"][p], o["d"][p], o["n"][p], o["k"][P + p], o["d"][P + p], o["n"][P + p], o["u"][p], o["z"][p], o["nm"][p])
                    against = "src/Frame.cc:105-230 (oracle/_ref/libref_frame.so)" if e.kind == "reference" else "oracle restatement (oracle/_ref absent)"
                elif kind == "mono":
                    bad = hc.mono_differences(batch[p], LAP, NFEAT, o["k"][p], o["d"][p], o["n"][p], o["m"][p])
                    against = "src/ORBextractor.cc (oracle/_ref/libref_orb.so)" if ol.reference() is not None else "oracle restatement (oracle/_ref absent)"
                elif kind == "fisheye":
                    bad = hc.fisheye_differences(batch[p], batch[P + p], LAP, NFEAT, (KB_CAM1, KB_CAM2, KB_RLR, KB_TLR), o["k"][p], o["d"][p], o["n"][p], o["k"][P + p], o["d"][P + p],
                                                 o["n"][P + p], o["l2r"][p], o["r2l"][p], o["z"][p], o["p3"][p], o["nm"][p])
                    against = "src/Frame.cc:1432-1528 + src/CameraModels/KannalaBrandt8.cpp (oracle/_ref/libref_frame.so), mvDepth / mvStereo3Dpoints included"
                else:
                    if ol.reference_frame_lib() is None:
                        bad = ["oracle/_ref/libref_frame.so missing"]
                    else:
                        lm = local_map[i]
                        grey = ol.oracle_gray(batch[p], True, 0)
                        F = ol.ReferenceFrame(grey, None, NFEAT, fx=FX, fy=FY, cx=CX, cy=CY, bf=lm["bf"], depth=lm["depth_host"])
                        bad = []
                        if int(o["n"][p]) != F.N:
                            bad.append("count %d != %d" % (o["n"][p], F.N))
                        else:
                            if o["k"][p][:F.N].tobytes() != F.keys.tobytes(): bad.append("mvKeys")
                            if o["d"][p][:F.N].tobytes() != F.desc.tobytes(): bad.append("mDescriptors")
                            if o["u"][p][:F.N].tobytes() != F.u_right.tobytes(): bad.append("mvuRight")
                            if o["z"][p][:F.N].tobytes() != F.depth.tobytes(): bad.append("mvDepth")
                            Rp, tp = lm["poses"][p]
                            _, ref_as, ref_n = F.search_local_points(Rp, tp, lm["X"], lm["nrm"], lm["mind"], lm["maxd"], np.zeros(len(lm["X"]), np.uint8),
                                                                     np.ones(len(lm["X"]), np.uint8), lm["dsc"], 0.5, True, 3.0, False, 50.0, 0.8)
                            if int(o["lp_nm"][p]) != ref_n or not np.array_equal(o["lp_asg"][p][:F.N], ref_as): bad.append("SearchLocalPoints assignments")
                    against = "src/Frame.cc:235-345 + Frame::isInFrustum + ORBmatcher::SearchByProjection (oracle/_ref/libref_frame.so)"
                checked += 1
                if bad:
                    mism.append({"handle": i, "unit": p, "fields": bad})
        return {"units": checked, "pairs": checked if paired else None, "per_handle": per_handle, "handles": NH, "identical": not mism, "against": against,
                "what": "outputs of the last timed step of every handle, fetched inside the timed region", "mismatches": mism[:8], "seconds": round(time.time() - t0, 2)}

    parity = None
    if rank == 0 and not args.no_parity_check and not args.h2d:
        try:
            parity = parity_check()
        except Exception as e:
            parity = {"units": 0, "identical": False, "error": repr(e)}
    per_rank_dt = None; per_rank_cpu = [own_cpu_s]
    if dist is not None:
        import torch
        own = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(every, own)                 # each rank's own clock over the same barrier-bracketed region: a straggler GPU shows up here
        per_rank_dt = [float(e.item()) for e in every]
        own = torch.tensor([own_cpu_s], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(every, own)
        per_rank_cpu = [float(e.item()) for e in every]
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ag_alone = None
    if args.allgather:
        # the exchange by itself: snapshot + both collectives + wait, nothing else running, barrier-bracketed (max over ranks)
        import torch
        nrep = 20
        sync_all()
        ta = time.perf_counter()
        for _ in range(nrep):
            if lib_comm is not None:
                lib_comm.all_gather(handles[0]); lib_comm.wait()
            else:
                _, _, ws = multi.all_gather_extracted(handles[0], None)
                for w in ws:
                    w.wait()
        sync_all()
        ta = (time.perf_counter() - ta) / nrep
        t = torch.tensor([ta], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blk_bytes = NIMG * cap * 32 + NIMG * 4
        ag_alone = {"ms_per_batch_alone": round(float(t.item()) * 1e3, 4), "bytes_per_rank": blk_bytes, "bytes_gathered_per_rank": blk_bytes * world,
                    "bus_GBps_per_rank": round(blk_bytes * (world - 1) / max(float(t.item()), 1e-9) / 1e9, 2), "backend": backend,
                    "path": "orbx_allgather_descriptors (RCCL inside the library, C ABI)" if lib_comm is not None else "torch.distributed.all_gather_into_tensor",
                    "where": "side stream, overlapping the next extractions; included in the timed region"}
    if rank == 0:
        total_units = P * args.steps * repeats * world
        value = total_units / dt
        # per-block rates of rank 0 (a block = --steps steps): how much `value` depends on where the region starts and ends
        cum = np.concatenate([[0.0], np.cumsum(per_step)]) * 1e-3          # completion time of step k since the start of the region
        block_values = [round(P * args.steps / max(cum[(b + 1) * args.steps] - cum[b * args.steps], 1e-9), 1) for b in range(repeats)] if len(per_step) == repeats * args.steps else []
        avg_kp = nkp[0] / max(nkp[1], 1)
        stage_ms = {k: v / max(stage_cnt[0], 1) for k, v in stage_sum.items()}
        n_timed_records = stage_cnt[0]
        avg_matches = nmatch[0] / max(nmatch[1], 1)
        # average FAST candidates per image (for the algorithmic-byte model): probe the last batch
        ncand = 0
        for l in range(NLEVELS):
            ncand += len(handles[0].debug_candidates(l, 0))
        ab = algorithmic_bytes(W, H, avg_kp, ncand)
        if zero_copy and not args.h2d:
            ab["import"] = 0                                   # level 0 is read where the producer wrote it
        if kind == "rgbd":
            # M4 = 5000 map points x (32 B descriptor + 20 candidates x 32 B) = 3.36 MB per frame.  (Rounds 3-4 priced this
            # kernel on what the batched search actually moves - 5000 x (88 + 64) + 4 x 5000 x (64 + 16) = 2.36 MB - the survey's figure is the contract.)
            ab["match"] = 5000 * (32 + 20 * 32)
        units = {k: NIMG for k in ab}
        units["match"] = P
        # Which kernel dominates is decided on a clean schedule: a few extra steps on ONE handle with every kernel alone on one
        # stream (with several handles in flight an event pair also brackets the time a launch waits for CUs that another
        # handle's kernels occupy, which inflates short latency-bound stages).  The reported launch duration of that kernel is
        # its HIP-event average over the TIMED region (what rocprofv3 --kernel-trace sees under the same command).
        h0 = handles[0]
        h0.profile(True, serial=True)
        serial_sum = {}
        for _ in range(3):
            enqueue(0)
            if kind in ("stereo", "fisheye"):
                fetch(0, False, gather=False)
            h0.sync()
            for k, v in h0.stage_ms().items():
                serial_sum[k] = serial_sum.get(k, 0.0) + v / 3.0
        ext_stages = ("import", "pyramid", "fast_cells", "quadtree", "blur", "layout", "orient_brief")
        dom = max((k for k in serial_sum if serial_sum[k] > 0 and k in ab), key=lambda k: serial_sum[k])
        achieved = ab[dom] * units[dom] / (stage_ms[dom] * 1e-3) / 1e9
        # ... and the other view: the stage with the largest launch duration UNDER THE TIMED SCHEDULE (several handles in flight: an event pair then also brackets
        # the time a launch's workgroups wait for CUs other handles' kernels hold - the quadtree's few long workgroups stretch most)
        dom_wall = max((k for k in stage_ms if stage_ms[k] > 0 and k in ab), key=lambda k: stage_ms[k])
        # HBM traffic of that kernel: NOT measured in this run (counter collection needs rocprofv3 --pmc passes of their own) but read from the
        # tracked summary of the last such passes and scaled to this launch size; `traffic_source` says which file, which collection and
        # whether the kernel sources have changed since (then the figure is stale and says so)
        traffic = None; traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and kind == "stereo":
            try:
                pj = json.load(open(pmc))
                traffic = pj.get(dom)
                if traffic is not None:
                    traffic = int(traffic * NIMG / 128.0)          # the PMC passes ran at 128 images per launch
                now = kernel_sources_sha()
                traffic_source = {"file": "profiles/pmc_traffic.json", "collected": pj.get("_source", "round 3 (profiles/r03_final/), before sources carried a stamp"),
                                  "kernel_sources_sha_then": pj.get("_kernel_sources_sha"), "kernel_sources_sha_now": now,
                                  "stale": pj.get("_kernel_sources_sha") != now, "measured_in_this_run": False}
            except Exception:
                traffic = None
        # ... and, in the default run on one GPU, MEASURED in this run: two rocprofv3 --pmc passes of their own (FETCH_SIZE, WRITE_SIZE; never combined with
        # a trace) over a short child run of this file at 128 images per launch, summed over the dominant kernel's dispatches.  The file above stays the
        # fallback (no rocprofv3, a pass that times out) and the cross-check (`traffic_file`).
        under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
        if (kind == "stereo" and world == 1 and dist is None and not args.no_live_traffic and not args.no_other_configs and not args.h2d and not under_profiler
                and not os.environ.get("ORBX_BENCH_LIB")):
            live = live_traffic(dom)
            if live is not None:
                traffic_source = dict(traffic_source or {}, traffic_file=traffic, measured_in_this_run=True, live=live)
                traffic = int(live["bytes_per_launch_128_images"] * NIMG / 128.0)
        # VALU issue view of the same kernel (the HBM fraction says little for a compute-heavy integer kernel): wave-instructions per launch
        # from the SQ counter pass (profiles/pmc_valu.json, scaled to this launch size) against the SIMD-32 issue peak of one wave64
        # instruction per 2 clk (1024 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instr/s; profiles/r02/valu_survey.txt: only the VOP2 integer forms
        # reach ~2.5 clk, every VOP3 form takes ~4.4 clk, so a kernel built from both classes sits between 45 % and 80 % of that peak at best)
        valu = None
        pv = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if os.path.exists(pv) and kind == "stereo":
            try:
                n_instr = json.load(open(pv)).get(dom, 0) * NIMG / 128.0
                if n_instr > 0:
                    valu = {"wave_instr_per_launch": int(n_instr), "peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR_PER_S,
                            "achieved_wave_instr_per_s": round(n_instr / (stage_ms[dom] * 1e-3), 0), "alone_wave_instr_per_s": round(n_instr / (serial_sum[dom] * 1e-3), 0),
                            "frac": round(n_instr / (stage_ms[dom] * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S, 4),
                            "alone_frac": round(n_instr / (serial_sum[dom] * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S, 4)}
            except Exception:
                valu = None
        # the whole step in the same units: every kernel's wave-instructions per step / ms_per_step
        step_valu = None
        try:
            if os.path.exists(pv) and kind == "stereo":
                pvj = json.load(open(pv))
                tot = sum(float(v) for k, v in pvj.items() if not k.startswith("_")) * NIMG / 128.0
                step_ms_now = dt / (args.steps * repeats) * 1e3
                step_valu = {"wave_instr_per_step": int(tot), "achieved_wave_instr_per_s": round(tot / (step_ms_now * 1e-3), 0),
                             "frac": round(tot / (step_ms_now * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S, 4), "kernel_sources_sha_of_counts": pvj.get("_kernel_sources_sha"),
                             "stale": pvj.get("_kernel_sources_sha") != kernel_sources_sha()}
        except Exception:
            step_valu = None
        per_unit_bytes = (2 if paired else 1) * sum(v for k, v in ab.items() if k not in ("match", "fast_cells_with_candidates")) + (ab["match"] if kind == "stereo" else 0)
        pct = lambda a, q: float(np.percentile(a, q)) if len(a) else None
        res = {
            "metric": cfg["metric"], "value": round(value, 1),
            "unit": cfg["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / (args.steps * repeats) * 1e3, 4), "repeats": repeats, "timed_steps": args.steps * repeats, "timed_seconds": round(dt, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config, "input_mode": "h2d" if args.h2d else ("zero_copy" if zero_copy else "import_copy"),
                       "units_per_step_per_gpu": P, "images_per_step_per_gpu": NIMG, "outputs_copied_to_host": True, "handles_in_flight": NH,
                       "inputs": "uploaded from pinned host memory inside the timed region (PCIe-inclusive variant)" if args.h2d else
                                 ("resident in HBM, written by the producer into the extractor's level-0 layout (orbx_input_buffer): no import pass" if zero_copy else
                                  "resident in HBM in a linear buffer, copied into the pyramid inside the step"),
                       "avg_keypoints_per_image": round(avg_kp, 1), "avg_matches_per_unit": round(avg_matches, 1),
                       "generator": args.workload, "fast_corner_density_t%d" % MIN: round(fast_density, 4),
                       "library": os.path.basename(lib.path) + (" (ORBX_BENCH_LIB override)" if os.environ.get("ORBX_BENCH_LIB") else ""),
                       "parallelism": "independent streams, %d GPU(s), one process per GPU, %s" % (world, "all-gather of the descriptor blocks per batch (RCCL)" if args.allgather
                                                                                                 else "no collective")},
            "block_values": {"per_block": block_values if len(block_values) <= 64 else block_values[:64], "min": min(block_values) if block_values else None,
                             "max": max(block_values) if block_values else None},
            # per-step completion intervals of the timed region (rank 0; with several handles in flight a step completes every ms_per_step on average)
            "step_ms": {"median": round(pct(per_step, 50), 4), "p10": round(pct(per_step, 10), 4), "p90": round(pct(per_step, 90), 4), "n": int(len(per_step))},
            # achieved / peak / frac: the kernel's ALGORITHMIC bytes per launch over its HIP-event launch duration in the timed region, against the 8 TB/s
            # of HBM - the figure the contract asks for.  `bound` says what actually limits that kernel: for k_fast_cells the vector and LDS issue ports (valu_issue:
            # instruction counts from the SQ counter passes; its HBM traffic is 0.95 x algorithmic - nothing is re-read - and a small fraction of the link)
            "roofline": {"bound": "valu_lds_issue" if (valu is not None and valu["frac"] > achieved / HBM_PEAK_GBS) else "hbm",
                         "kernel": dom, "kernel_chosen_by": "largest launch duration alone on one stream (stage_ms_alone)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(ab[dom] * units[dom]), "avg_launch_ms": round(stage_ms[dom], 4),
                         "alone_launch_ms": round(serial_sum[dom], 4), "alone_GBps": round(ab[dom] * units[dom] / (serial_sum[dom] * 1e-3) / 1e9, 2),
                         "with_candidate_output": None if dom != "fast_cells" else {
                             "algorithmic_bytes_per_launch": int(ab["fast_cells_with_candidates"] * units[dom]),
                             "achieved": round(ab["fast_cells_with_candidates"] * units[dom] / (stage_ms[dom] * 1e-3) / 1e9, 2),
                             "frac": round(ab["fast_cells_with_candidates"] * units[dom] / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "what": "pixels read + the 4-byte record of every FAST candidate written (the figure rounds 1-5 printed as `frac`)"},
                         "by_wall": {"kernel": dom_wall, "kernel_chosen_by": "largest launch duration in the timed region, %d handles in flight (stage_ms_per_step)" % NH,
                                     "avg_launch_ms": round(stage_ms[dom_wall], 4), "alone_launch_ms": round(serial_sum.get(dom_wall, 0.0), 4),
                                     "algorithmic_bytes_per_launch": int(ab[dom_wall] * units[dom_wall]),
                                     "achieved": round(ab[dom_wall] * units[dom_wall] / (stage_ms[dom_wall] * 1e-3) / 1e9, 2),
                                     "frac": round(ab[dom_wall] * units[dom_wall] / (stage_ms[dom_wall] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     "bound": "latency (one workgroup per tree; its launch duration beside other handles' kernels is mostly waiting)" if dom_wall == "quadtree" else None},
                         "end_to_end_GBps": round(per_unit_bytes * value / world / 1e9, 2),
                         "end_to_end_frac": round(per_unit_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5), "valu_issue": valu, "step_valu_issue": step_valu},
            "parity_check": parity,
            # each rank's own rate over the same barrier-bracketed region (its units / its own clock): `value` uses the slowest rank's time
            "per_rank": None if per_rank_dt is None else {
                "unit": cfg["unit"], "values": [round(P * args.steps * repeats / t, 1) for t in per_rank_dt],
                "min": round(P * args.steps * repeats / max(per_rank_dt), 1), "max": round(P * args.steps * repeats / min(per_rank_dt), 1)},
            # what the host side costs: CPU seconds (user + system, every thread of the rank's process) per step inside the timed region - enqueueing four handles'
            # launches, waiting for and fetching ~20 MB of pinned results per step; x n ranks must fit the cores the node grants
            "host_cpu": {"cpu_ms_per_step_per_rank": [round(c / (args.steps * repeats) * 1e3, 4) for c in per_rank_cpu],
                         "cpu_cores_busy_per_rank": [round(c / max(dt, 1e-9), 3) for c in per_rank_cpu],
                         "cpu_cores_busy_all_ranks": round(sum(per_rank_cpu) / max(dt, 1e-9), 3),
                         "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cpu_quota_cores": cpu_quota_cores(),
                         "rank0_busiest_threads_cpu_s": own_threads, "timed_seconds": round(dt, 3), "host_wait": host_wait_mode},
            "allgather": ag_alone,
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_alone": {k: round(v, 4) for k, v in serial_sum.items()},
            # the reference's two instrumented regions (src/Frame.cc:132-146 / :158-170, printed by Tracking::PrintTimeStats), per step, each kernel alone
            "reference_stage_ms_alone": {"ORB Extraction": round(sum(serial_sum.get(k, 0.0) for k in ext_stages), 4), "Stereo Matching": round(serial_sum.get("match", 0.0), 4)},
        }
        if kind == "stereo" and not args.h2d and not args.no_h2d and world == 1 and dist is None:
            # second measurement, never `value`: the same steps with every input batch uploaded from page-locked host memory inside the timed
            # region, on each handle's copy stream, double-buffered (upload of batch i + 1 beside the kernels of batch i)
            try:
                setup_h2d()
                for h in handles:
                    h.profile(False)
                n2 = max(min(args.steps, 100), 10)
                dt2, per2, _ = timed(n2, True)
                # what the link gives by itself, same box, same run: the same page-locked batch uploaded back to back on one copy stream, nothing else running
                sync_all()
                h0 = handles[0]
                h0.device_upload_async(dbuf[0][0], host_in[0]); h0.sync()
                tp = time.perf_counter()
                for _ in range(8):
                    h0.device_upload_async(dbuf[0][0], host_in[0])
                h0.sync()
                probe = 8 * batch.nbytes / (time.perf_counter() - tp) / 1e9
                res["h2d_inclusive"] = {"value": round(P * n2 / dt2, 1), "unit": cfg["unit"], "steps": n2, "ms_per_step": round(dt2 / n2 * 1e3, 4),
                                        "input_MB_per_step": round(batch.nbytes / 1e6, 1), "PCIe_GBps": round(batch.nbytes * n2 / dt2 / 1e9, 1),
                                        "PCIe_probe_GBps": round(probe, 1), "PCIe_frac": round(batch.nbytes * n2 / dt2 / 1e9 / probe, 3),
                                        "PCIe_probe": "8 uploads of one %0.1f-MB batch from page-locked memory (hipMemcpyAsync) on one copy stream, alone" % (batch.nbytes / 1e6),
                                        "step_ms": {"median": round(pct(per2, 50), 4), "p10": round(pct(per2, 10), 4), "p90": round(pct(per2, 90), 4)},
                                        # the link and the host side of it, as this box reports them (PCIe_frac moved between 0.77 and 0.85 from box to box in round 5)
                                        "pcie_links": pcie_links(), "host_cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                                        "pinned_memory": "hipHostMalloc default flags: placed by the runtime on the NUMA node of the device"}
                # the second headline, at the top level of the line: the rate a host that FEEDS frames over PCIe sees (the drop-in's operator()(cv::Mat) takes this road)
                res["value_host_fed"] = res["h2d_inclusive"]["value"]
            except Exception as e:
                res["h2d_inclusive"] = {"value": None, "error": repr(e)}
        if kind == "stereo" and world == 1 and not args.no_latency and not os.environ.get("ORBX_BENCH_LIB"):
            # reported beside the throughput, never `value`: one stereo pair per call on one fresh handle, synchronised after every pair
            # (Tracking's rhythm: extract L + R, ComputeStereoMatches, wait), the frames written into pyramid level 0 by the producer
            try:
                for h in handles:
                    h.sync()
                hl = ORBextractor(NFEAT, SCALE, NLEVELS, INI, MIN, device_id=local, lib=lib)
                lp, lshape, lstride, listride = hl.input_upload(np.stack([batch[0], batch[P]]))
                npair = 200
                for it in range(npair + 20):
                    if it == 20:
                        tl = time.perf_counter()
                    hl.enqueue(None, LAP, device_ptr=lp, shape=lshape, stride=lstride, image_stride=listride)
                    lib.check(lib.L.orbm_stereo_match(hl._h, 0, hl._h, 1, 1, BF, BASE))
                    hl.sync()
                res["latency"] = {"single_pair_ms": round((time.perf_counter() - tl) / npair * 1e3, 4), "pairs": npair,
                                  "what": "one pair per call on one handle, host waits after every pair; inputs resident in pyramid level 0"}
                hl.close()
                # the monocular rhythm: one image per call (extract, wait)
                hm = ORBextractor(NFEAT, SCALE, NLEVELS, INI, MIN, device_id=local, lib=lib)
                lp, lshape, lstride, listride = hm.input_upload(batch[:1])
                for it in range(npair + 20):
                    if it == 20:
                        tl = time.perf_counter()
                    hm.enqueue(None, LAP, device_ptr=lp, shape=lshape, stride=lstride, image_stride=listride)
                    hm.sync()
                res["latency"]["single_image_ms"] = round((time.perf_counter() - tl) / npair * 1e3, 4)
                hm.close()
                # The road a drop-in ORBextractor::operator()(cv::Mat) takes (include/orb_slam3_amd/ORBextractor.h -> orbx_extract): the frame in PAGEABLE host
                # memory, uploaded, extracted, keypoints + descriptors copied back into caller arrays, the host waiting - one image per call, and eight per call
                # (orbx_extract_batch: a multi-camera rig's frames together).  The facade adds the cv::KeyPoint conversion and, unless SetExportPyramid(false),
                # the copy of the pyramid back to mvImagePyramid (tools/gpu_facade_latency.sh times those against the reference build).
                hd = ORBextractor(NFEAT, SCALE, NLEVELS, INI, MIN, device_id=local, lib=lib)
                one = np.array(batch[0], copy=True); eight = np.array(batch[:8], copy=True)         # plain numpy = pageable
                for it in range(60 + 10):
                    if it == 10:
                        tl = time.perf_counter()
                    hd(one, None, LAP)
                t1 = (time.perf_counter() - tl) / 60
                for it in range(30 + 5):
                    if it == 5:
                        tl = time.perf_counter()
                    hd.extract_batch(eight, LAP)
                t8 = (time.perf_counter() - tl) / 30
                hd.close()
                res["dropin_call"] = {"B1_ms_per_call": round(t1 * 1e3, 4), "B1_images_per_s": round(1.0 / t1, 1), "B8_ms_per_call": round(t8 * 1e3, 4),
                                      "B8_images_per_s": round(8.0 / t8, 1),
                                      "what": "orbx_extract / orbx_extract_batch from pageable host memory, results copied back to caller arrays, host waits after every call (Python ctypes caller)"}
            except Exception as e:
                res["latency"] = {"single_pair_ms": None, "error": repr(e)}
        if kind == "stereo" and world == 1 and dist is None and not args.no_other_configs and not args.h2d and not os.environ.get("ORBX_BENCH_LIB"):
            for h in handles:
                h.sync()
            res["other_configs"] = other_configs()
        if kind == "stereo" and not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # the baseline is reporting only; never fail the bench on it
                res["cpu_baseline"] = {"value": None, "unit": "stereo pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe and would come out at exit, after the
        # result: push it out first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist_barrier()
        dist.destroy_process_group()
    if parity is not None and not parity.get("identical"):
        sys.stderr.write("bench.py: parity_check FAILED - the timed loop's outputs differ from the reference: %s\n" % json.dumps(parity))
        return 3


if __name__ == "__main__":
    sys.exit(main() or 0)
