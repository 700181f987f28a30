/* Exhaustive check of csrc/glibc_logf_model.h against the live libm: every positive finite float (0x00000001 .. 0x7f7fffff), both contraction
   patterns.  Build: g++ -O2 -ffp-contract=off -x c++ tools/check_logf_model.c -Iorb_slam3_detailed_comments_amd/csrc -lpthread -o /tmp/chklog && /tmp/chklog
   (about 20 s on 16 cores).  Prints the mismatch count of the plain and of the FMA pattern; the one with 0 is the libm variant this machine runs. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "glibc_logf_model.h"
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
struct Job { uint32_t lo, hi; long bad_plain, bad_fma; uint32_t first_plain, first_fma; };
static void* run(void* a) {
    Job* j = (Job*)a;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        const float x = asf(u);
        const uint32_t ref = asu(logf(x));
        if (ref != asu(orbx::glibc_logf_model<false>(x))) { if (!j->bad_plain) j->first_plain = u; j->bad_plain++; }
        if (ref != asu(orbx::glibc_logf_model<true>(x))) { if (!j->bad_fma) j->first_fma = u; j->bad_fma++; }
    }
    return 0;
}
int main() {
    const uint32_t lo = 1, hi = 0x7f800000u;
    const int T = 32; pthread_t th[T]; Job jb[T]; long bp = 0, bf = 0;
    for (int t = 0; t < T; t++) {
        memset(&jb[t], 0, sizeof jb[t]);
        jb[t].lo = lo + (uint32_t)((uint64_t)(hi - lo) * t / T); jb[t].hi = lo + (uint32_t)((uint64_t)(hi - lo) * (t + 1) / T);
        pthread_create(&th[t], 0, run, &jb[t]);
    }
    for (int t = 0; t < T; t++) { pthread_join(th[t], 0); bp += jb[t].bad_plain; bf += jb[t].bad_fma; if (jb[t].bad_plain && bp == jb[t].bad_plain) printf("first plain mismatch at 0x%08x\n", jb[t].first_plain); if (jb[t].bad_fma && bf == jb[t].bad_fma) printf("first fma mismatch at 0x%08x\n", jb[t].first_fma); }
    printf("floats checked %u: mismatches plain %ld, fma %ld\n", hi - lo, bp, bf);
    return !(bp == 0 || bf == 0);
}
