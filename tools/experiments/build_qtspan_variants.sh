#!/bin/bash
# experimental builds for tools/experiments/qt_spans_probe.py: the product sources + two wall-clock stamps per quadtree workgroup and the phase stamps of one level
# (ORBX_QT_STAMP_LEVEL) -> build/variants/liborbx_hip_qtspan<level>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
D=$(mktemp -d)
mkdir -p $D/m/a/b $D/m/include $R/build/variants
cp $R/orb_slam3_detailed_comments_amd/csrc/* $D/m/a/b/ && cp $R/include/orbx.h $D/m/include/
python3 - $D/m/a/b <<'P'
import sys
d = sys.argv[1]
p = d + '/k_quadtree.hip'; s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b)
rep("    QT_STAMP(0)\n", "    QT_STAMP(0)\n    if (qt_prof && tid == 0 && b < 2) qt_prof[16 + (b * 8 + level) * 2] = wall_clock64();\n    int n_final_rounds = 0;\n")
a = "    if (tid == 0) lvl_count[(size_t)b * nlevels + level] = nnodes;\n    QT_STAMP(9)\n"
rep(a, a + "    if (qt_prof && tid == 0 && b < 2) qt_prof[16 + (b * 8 + level) * 2 + 1] = wall_clock64();\n    if (qt_prof && tid == 0 && level == ORBX_QT_STAMP_LEVEL && b == 0) { qt_prof[14] = n_final_rounds; qt_prof[15] = NT; }\n")
G = "    if (qt_prof && tid == 0 && level == ORBX_QT_STAMP_LEVEL && b == 0) qt_prof[%d] = wall_clock64();\n"
rep("    int lgt = 0;\n", G % 40 + "    int lgt = 0;\n")
rep("    const bool narrow_counters = n <= kPresortU16Max;\n", G % 41 + "    const bool narrow_counters = n <= kPresortU16Max;\n")
rep("    {\n        int run = 0;\n        for (int c0 = 0; c0 < L.cell_count; c0 += cpt) {\n", G % 42 + "    {\n        int run = 0;\n        for (int c0 = 0; c0 < L.cell_count; c0 += cpt) {\n")
rep("qt_prof[10] = n; qt_prof[11] = nnodes; qt_prof[12] = nexp; }", "qt_prof[10] = n; qt_prof[11] = nnodes; qt_prof[12] = nexp; qt_prof[13] = passno; }")
a = "                const int prev2 = nnodes;\n                QT_STAMP(4)\n"
rep(a, a + "                n_final_rounds++;\n")
open(p, 'w').write(s)
p = d + '/orbx_api.cpp'; s = open(p).read()
rep("e |= h->d_qtprof.ensure(32);", "e |= h->d_qtprof.ensure(64);")
rep("if (rt::copy_d2h(out, h->d_qtprof.p, sizeof(long long) * 16, h->s0)", "if (rt::copy_d2h(out, h->d_qtprof.p, sizeof(long long) * 48, h->s0)")
open(p, 'w').write(s)
P
cd $D/m/a/b
for lv in ${@:-0 2 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -w -shared -x hip -DORBX_QT_STAMP_LEVEL=$lv k_image.hip k_fast.hip k_quadtree.hip k_describe.hip k_match.hip k_search.hip k_vocab.hip k_input.hip orbx_api.cpp orbm_search.cpp orbv_api.cpp orbx_comm.cpp -o $R/build/variants/liborbx_hip_qtspan$lv.so 2>/dev/null &
done
wait
rm -rf $D
ls $R/build/variants | grep qtspan
