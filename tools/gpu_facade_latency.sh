#!/bin/bash
# Per-call latency of the drop-in ORBextractor facade (one 752x480 image per call, the way Frame::ExtractORB uses it) next to the reference's own
# source on one host thread.  Run on a GPU box from the repo root.
set -e
O=gpurun_out; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, ".")
from orb_slam3_detailed_comments_amd import synth
synth.stereo_pair(752, 480, seed=100)[0].tofile("/tmp/facade_im.raw")
synth.natural(752, 480, seed=100).tofile("/tmp/facade_nat.raw")
PY
g++ -std=c++14 -O2 -w -DORBX_FACADE -Iinclude/orb_slam3_amd -Ioracle/opencv_shim tests/cpp/facade_latency.cpp -Lorb_slam3_detailed_comments_amd -lorbx_hip -Wl,-rpath,$PWD/orb_slam3_detailed_comments_amd -o /tmp/facade_latency_ours
{
for im in /tmp/facade_im.raw /tmp/facade_nat.raw; do
  echo "== $im"
  /tmp/facade_latency_ours $im 752 480 1200 300 1
  /tmp/facade_latency_ours $im 752 480 1200 300 0
  [ -x oracle/_ref/facade_latency_ref ] && oracle/_ref/facade_latency_ref $im 752 480 1200 20
done
} | tee $O/facade_latency.txt
