/* Exhaustive check of csrc/glibc_sincosf_model.h against the live libm for every float in [-6.5, 6.5].
   Build: g++ -O2 -ffp-contract=off -x c++ tools/check_sincosf_model.c -Iorb_slam3_detailed_comments_amd/csrc -lpthread -o /tmp/chk && /tmp/chk
   (about 3 s on 8 cores).  Result recorded in DESIGN.md. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "glibc_sincosf_model.h"
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
struct Job { uint32_t lo, hi; long bad; };
static void* run(void* a) {
    Job* j = (Job*)a;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        float x = asf(u);
        if (asu(cosf(x)) != asu(orbx::glibc_cosf(x)) || asu(sinf(x)) != asu(orbx::glibc_sinf(x))) j->bad++;
        x = -x;            /* psi = atan2f(y, x) of KannalaBrandt8::project is in [-pi, pi] */
        if (asu(cosf(x)) != asu(orbx::glibc_cosf(x)) || asu(sinf(x)) != asu(orbx::glibc_sinf(x))) j->bad++;
    }
    return 0;
}
int main() {
    const uint32_t hi = asu(6.5f);
    const int T = 8; pthread_t th[T]; Job jb[T]; long bad = 0;
    for (int t = 0; t < T; t++) { jb[t].lo = (uint32_t)((uint64_t)hi * t / T); jb[t].hi = (uint32_t)((uint64_t)hi * (t + 1) / T); jb[t].bad = 0; pthread_create(&th[t], 0, run, &jb[t]); }
    for (int t = 0; t < T; t++) { pthread_join(th[t], 0); bad += jb[t].bad; }
    printf("floats checked %u mismatches %ld\n", hi, bad);
    return bad != 0;
}
