#!/bin/bash
# Round 6: variant builds of the library for A/B runs on the GPU box (tools/experiments/gpu_r6_ab.sh).  name:EXTRA pairs; each goes to
# build/variants/liborbx_hip_<name>.so with an object directory of its own.
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/variants
build() {
  name=$1; shift
  make -s -j8 -C orb_slam3_detailed_comments_amd/csrc OUT=../../build/variants/liborbx_hip_$name.so OBJDIR=../../build/var_$name EXTRA="$*" 2>&1 | grep -E "error|warning" || true
  echo "built $name: $*"
}
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}; [ "$extra" = "$spec" ] && extra=""
  build $name $extra
done
