#!/bin/bash
# experimental build for a look inside one k_fast_cells workgroup at one pair per call: wall-clock stamps (100 MHz) between the phases of two
# cells of image 0, printed from the device -> build/variants/liborbx_hip_fastspan.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
D=$(mktemp -d)
mkdir -p $D/m/a/b $D/m/include $R/build/variants
cp $R/orb_slam3_detailed_comments_amd/csrc/* $D/m/a/b/ && cp $R/include/orbx.h $D/m/include/
python3 - $D/m/a/b <<'P'
import sys
d = sys.argv[1]
p = d + '/k_fast.hip'; s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (s.count(a), a)
    s = s.replace(a, b)
ST = "    if (probe) st[%d] = wall_clock64();\n"
rep("    const int lane = (int)threadIdx.x & 63;\n    const unsigned long long lt = (1ull << lane) - 1ull;\n",
    "    const int lane = (int)threadIdx.x & 63;\n    const unsigned long long lt = (1ull << lane) - 1ull;\n    const bool probe = blockIdx.y == 0;\n    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n_flush = 0;\n" + ST % 0)
rep("    // score tile: same pitch as the window tile,", ST % 1 + "    // score tile: same pitch as the window tile,")
rep("    // ---- A ----  quick rejection, 4 adjacent pixels per lane", ST % 2 + "    // ---- A ----  quick rejection, 4 adjacent pixels per lane")
rep("                score_pending();\n                if (cnt + trip > list_cap)", "                score_pending(); n_flush++;\n                if (cnt + trip > list_cap)")
rep("    ORBX_WAVE_SYNC();\n    score_pending();\n    if (corners_listed) {", "    ORBX_WAVE_SYNC();\n" + ST % 3 + "    score_pending();\n" + ST % 4 + "    if (corners_listed) {")
rep("        // ---- D ----  output in bitmap order", ST % 5 + "        // ---- D ----  output in bitmap order")
rep("    if (base > 0 || pass == 1 || minTh >= iniTh) break;", ST % 6 + "    if (probe && lane == 0) printf(\"FASTCELL %d %d %d %lld %lld %.2f %.2f %.2f %.2f %.2f %.2f %d\\n\", (int)ci.level, cx0, cy0, st[0], st[6], (st[1]-st[0])/100., (st[2]-st[1])/100., (st[3]-st[2])/100., (st[4]-st[3])/100., (st[5]-st[4])/100., (st[6]-st[5])/100., base);\n    if (base > 0 || pass == 1 || minTh >= iniTh) break;")
open(p, 'w').write(s)
P
cd $D/m/a/b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -w -shared -x hip k_image.hip k_fast.hip k_quadtree.hip k_describe.hip k_match.hip k_search.hip k_vocab.hip k_input.hip orbx_api.cpp orbm_search.cpp orbv_api.cpp orbx_comm.cpp -o $R/build/variants/liborbx_hip_fastspan.so
rm -rf $D
ls -la $R/build/variants/liborbx_hip_fastspan.so
