#!/bin/bash
# The kernel sources + the host side of the library, built for the CPU SIMT emulator with AddressSanitizer / UndefinedBehaviorSanitizer, under the emulator-based
# tests of the CPU suite.  Device memory of the emulator build is plain heap memory, so a kernel that reads or writes one element past a device buffer - silent on
# the GPU - is an ASan report here; the same holds for the host-side staging.  (The emulator runs workgroups on ucontext fibers: ASan warns that it does not fully
# follow swapcontext; stack-use-after-return detection is off for that reason.)  usage: tools/emu_sanitizers.sh [asan|ubsan]  ->  /tmp/orbx_<kind>_suite.log
set -e
cd "$(dirname "$0")/.."
KIND=${1:-asan}
CSRC=orb_slam3_detailed_comments_amd/csrc
SRCS=$(grep "^SRCS" tests/emu/Makefile | sed "s/SRCS = //; s#\$(CSRC)#$CSRC#g")
SEL="(emu or mono_init or resident or batch or local_points or kb8 or ties or undistort or sophus or simd or lifetime or multi_comm or abi) and not rccl"        # (the rccl load-order probe loads the HIP build)
if [ "$KIND" = asan ]; then
    g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address -std=c++17 -ffp-contract=off -fwrapv -fno-gnu-unique -DORBX_EMU -DHIPEMU_UCONTEXT -Itests/emu -I$CSRC -fPIC -shared -w -x c++ $SRCS -o /tmp/liborbx_emu_asan.so -lpthread
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path=/tmp/orbx_asan_report \
        ORBX_EMU_LIB=/tmp/liborbx_emu_asan.so python -m pytest tests -q -p no:xdist -m "not gpu" -k "$SEL" > /tmp/orbx_asan_suite.log 2>&1 || true
    tail -3 /tmp/orbx_asan_suite.log; grep -l "ERROR: AddressSanitizer" /tmp/orbx_asan_report.* 2>/dev/null || echo "no AddressSanitizer error reports"
else
    g++ -O1 -g -fsanitize=undefined -fno-sanitize=alignment,vptr -std=c++17 -ffp-contract=off -fwrapv -fno-gnu-unique -DORBX_EMU -DHIPEMU_UCONTEXT -Itests/emu -I$CSRC -fPIC -shared -w -x c++ $SRCS -o /tmp/liborbx_emu_ubsan.so -lpthread
    UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/orbx_ubsan_report ORBX_EMU_LIB=/tmp/liborbx_emu_ubsan.so python -m pytest tests -q -p no:xdist -m "not gpu" -k "$SEL" > /tmp/orbx_ubsan_suite.log 2>&1 || true
    tail -3 /tmp/orbx_ubsan_suite.log; ls /tmp/orbx_ubsan_report.* 2>/dev/null || echo "no UBSan reports"
fi
