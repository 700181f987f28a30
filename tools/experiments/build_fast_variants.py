"""Experimental builds of liborbx_hip.so with parts of k_fast_cells cut out (timing attribution on the GPU; results are wrong by construction).
   python tools/experiments/build_fast_variants.py  ->  build/variants/liborbx_hip_<name>.so"""
import os, shutil, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam3_detailed_comments_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
SRCS = "k_image.hip k_fast.hip k_quadtree.hip k_describe.hip k_match.hip k_search.hip k_vocab.hip k_input.hip orbx_api.cpp orbm_search.cpp orbv_api.cpp orbx_comm.cpp".split()

def patch(src, name):
    if name == "base":
        return src
    if name == "load_only":        # return after the tile load
        return src.replace("    // exact score of two (pixel, polarity) entries", "    if (lane < 64) { if (lane == 0) *count_out = 0; return; }\n    // exact score of two (pixel, polarity) entries", 1)
    if name == "no_score":         # phase A only: pending entries are dropped instead of scored
        return src.replace("        for (int i0 = pbeg; i0 < cnt; i0 += 2 * kFastThreads) {", "        for (int i0 = pbeg; i0 < cnt && iniTh < 0; i0 += 2 * kFastThreads) {", 1)
    if name == "no_cd":            # no NMS / output
        return src.replace("    if (corners_listed) {\n        // ---- C ----", "    if (iniTh < 0) {\n        // ---- C ----", 1).replace("    } else {\n        // ---- C', D' ----", "    } else if (iniTh < 0) {\n        // ---- C', D' ----", 1)
    if name == "no_gather":        # phase B without the LDS ring gather: ring values made up from the entry
        return src.replace("#define ORBX_D(k, dx, dy) d[k] = pk_xor(pk_bytes(qa + ((dy) + 3) * wp + (dx) + 3, qb + ((dy) + 3) * wp + (dx) + 3), X);",
                           "#define ORBX_D(k, dx, dy) d[k] = pk_xor(pk_make((uint32_t)((eA * (k + 3)) & 0xFF) | ((uint32_t)((eB * (k + 5)) & 0xFF) << 16)), X);", 1)
    if name == "no_network":       # phase B with the gather but without the min/max network
        return src.replace("        fast_score_pk(d, vp, t0, sA, sB);\n    };", "        { pk2 acc = d[0]; for (int k = 1; k < 16; k++) acc = pk_xor(acc, __builtin_bit_cast(uint32_t, d[k])); const pk2 df = pk_sub(vp, acc); sA = pk_lo(df) & 31; sB = pk_hi(df) & 31; }\n    };", 1)
    NS = ("        for (int i0 = pbeg; i0 < cnt; i0 += 2 * kFastThreads) {", "        for (int i0 = pbeg; i0 < cnt && iniTh < 0; i0 += 2 * kFastThreads) {")
    if name == "A_nowrite":        # phase A without the list writes (no scoring either)
        t = src.replace(*NS, 1)
        a = "if (ORBX_IN_BALLOT(bal[j])) list[cnt + lanes_below(bal[j])] = "
        assert a in t
        t = t.replace(a, "if (ORBX_IN_BALLOT(bal[j]) && iniTh < 0) list[cnt + lanes_below(bal[j])] = ", 1)
        return t
    if name == "A_noread":         # phase A without its LDS reads
        t = src.replace(*NS, 1)
        a = "const uint32_t C0 = hb[1], L1 = hb[wpd], C1 = hb[wpd + 1], R1 = hb[wpd + 2], L3 = hb[3 * wpd], C3 = hb[3 * wpd + 1], R3 = hb[3 * wpd + 2],"
        assert a in t
        t = t.replace(a, "const uint32_t hv = (uint32_t)(size_t)hb * 2654435761u; const uint32_t C0 = hv, L1 = hv >> 1, C1 = hv * 3, R1 = hv ^ 77, L3 = hv + 5, C3 = hv >> 3, R3 = hv * 7,", 1)
        t = t.replace("L5 = hb[5 * wpd], C5 = hb[5 * wpd + 1], R5 = hb[5 * wpd + 2], C6 = hb[6 * wpd + 1];", "L5 = hv * 11, C5 = hv >> 5, R5 = hv * 13, C6 = hv ^ 0x5555;", 1)
        return t
    if name == "A_noappend":       # phase A: compute only, flags folded into one store per lane
        t = src.replace(*NS, 1)
        a = "            const uint32_t ANY = SD | SB, BOTH = SD & SB;\n"
        assert a in t
        t = t.replace(a, "            const uint32_t ANY = SD | SB, BOTH = SD & SB;\n            if (iniTh >= 0) { ((uint32_t*)list)[lane] ^= ANY + BOTH; continue; }\n", 1)
        return t
    raise SystemExit("unknown variant " + name)

def build(name, extra=()):
    d = tempfile.mkdtemp(prefix="orbx_" + name)
    for f in os.listdir(CSRC):
        shutil.copy(os.path.join(CSRC, f), d)
    p = os.path.join(d, "k_fast.hip")
    s = open(p).read(); t = patch(s, name.split("+")[0])
    assert name.startswith("base") or t != s, name
    if not name.startswith("base"):        # a cut-out build finds no corners: it must not run the cell a second time at minThFAST
        a = "    if (base > 0 || pass == 1 || minTh >= iniTh) break;"
        assert a in t
        t = t.replace(a, "    break;", 1)
    open(p, "w").write(t)
    os.makedirs(os.path.join(d, "..", "..", "include"), exist_ok=True)
    out = os.path.join(OUT, "liborbx_hip_%s.so" % name.replace("+", "_"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w", "-shared", "-x", "hip",
           "-I" + CSRC, "-include", os.path.join(ROOT, "include", "orbx.h")] + list(extra) + [os.path.join(d, f) for f in SRCS] + ["-o", out]
    # the sources include "../../include/orbx.h" relative to csrc: compile inside a mirror of the tree
    mirror = os.path.join(d, "m", "a", "b"); os.makedirs(mirror)
    for f in os.listdir(d):
        if os.path.isfile(os.path.join(d, f)): shutil.copy(os.path.join(d, f), mirror)
    os.makedirs(os.path.join(d, "m", "include")); shutil.copy(os.path.join(ROOT, "include", "orbx.h"), os.path.join(d, "m", "include"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w", "-shared", "-x", "hip"] + list(extra) + [os.path.join(mirror, f) for f in SRCS] + ["-o", out]
    subprocess.run(cmd, check=True)
    shutil.rmtree(d)
    return out

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    jobs = [("base", ()), ("load_only", ()), ("no_score", ()), ("no_cd", ()), ("no_gather", ()), ("no_network", ()),
            ("base+list1k", ("-DORBX_FAST_LIST_BYTES=1024",)), ("base+list3k", ("-DORBX_FAST_LIST_BYTES=3072",)), ("base+xcd1", ("-DORBX_FAST_XCD_RUN=1",)),
            ("A_nowrite", ()), ("A_noread", ()), ("A_noappend", ()),
            ("base+strip4", ("-DORBX_RESIZE_STRIP=4",)), ("base+strip8", ("-DORBX_RESIZE_STRIP=8",)), ("base+strip32", ("-DORBX_RESIZE_STRIP=32",))]
    if len(sys.argv) > 1: jobs = [j for j in jobs if j[0] in sys.argv[1:]]
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda j: build(*j), jobs): print(o)
