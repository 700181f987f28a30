#!/bin/bash
# developer loop: build the HIP library and the emulator library, dump the device ISA of one kernel file ($1, default k_fast.hip) to /tmp/isa
set -e
R=/root/repo
make -s -C $R/orb_slam3_detailed_comments_amd/csrc 2>&1 | grep -E "error|warning" || true
make -s -C $R/tests/emu 2>&1 | grep -E "error" || true
F=${1:-k_fast.hip}
mkdir -p /tmp/isa
(cd $R/orb_slam3_detailed_comments_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -x hip $F -o /tmp/isa/${F%.hip}.s 2>/dev/null)
grep -E "^\s+\.(sgpr|vgpr)(_spill)?_count|\.name:" /tmp/isa/${F%.hip}.s | paste - - - - - | sed 's/\s\+/ /g' | head -20
