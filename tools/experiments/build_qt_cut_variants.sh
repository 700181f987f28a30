#!/bin/bash
# experimental builds of liborbx_hip.so with k_quadtree cut short after a phase (timing attribution at large batches; results are wrong by construction):
# gather = return after the gather, sort = after the counting sort, tree = before the selection -> build/variants/liborbx_hip_qtcut_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/variants
for v in gather sort tree; do
  D=$(mktemp -d); mkdir -p $D/m/a/b $D/m/include
  cp $R/orb_slam3_detailed_comments_amd/csrc/* $D/m/a/b/ && cp $R/include/orbx.h $D/m/include/
  python3 - $D/m/a/b/k_quadtree.hip $v <<'P'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
cut = "    if (iniCut) { if (tid == 0) lvl_count[(size_t)b * nlevels + level] = 0; return; }\n"
mark = {"gather": "    QT_STAMP(1)\n", "sort": "    QT_STAMP(2)\n", "tree": "    QT_STAMP(8)\n"}[v]
assert s.count(mark) == 1
s = s.replace(mark, mark + cut.replace("iniCut", "nlevels > 0"))
open(p, "w").write(s)
P
  (cd $D/m/a/b && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -w -shared -x hip k_image.hip k_fast.hip k_quadtree.hip k_describe.hip k_match.hip k_search.hip k_vocab.hip k_input.hip orbx_api.cpp orbm_search.cpp orbv_api.cpp orbx_comm.cpp -o $R/build/variants/liborbx_hip_qtcut_$v.so 2>/dev/null; rm -rf $D) &
done
wait
ls $R/build/variants | grep qtcut
