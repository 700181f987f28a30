/* Exhaustive check of csrc/glibc_tanf_model.h against the live libm for every float in [0, 8].
   Build: g++ -O2 -ffp-contract=off -x c++ tools/check_tanf_model.c -Iorb_slam3_detailed_comments_amd/csrc -lpthread -o /tmp/chktan && /tmp/chktan
   (a few seconds on 8 cores).  Result recorded in DESIGN.md. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "glibc_tanf_model.h"
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
struct Job { uint32_t lo, hi; long bad; uint32_t first; };
static void* run(void* a) {
    Job* j = (Job*)a;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        float x = asf(u);
        if (asu(tanf(x)) != asu(orbx::glibc_tanf_model(x))) { if (!j->bad) j->first = u; j->bad++; }
    }
    return 0;
}
int main() {
    const uint32_t hi = asu(8.0f);
    const int T = 8; pthread_t th[T]; Job jb[T]; long bad = 0;
    for (int t = 0; t < T; t++) { jb[t].lo = (uint32_t)((uint64_t)hi * t / T); jb[t].hi = (uint32_t)((uint64_t)hi * (t + 1) / T); jb[t].bad = 0; jb[t].first = 0; pthread_create(&th[t], 0, run, &jb[t]); }
    for (int t = 0; t < T; t++) { pthread_join(th[t], 0); bad += jb[t].bad; if (jb[t].bad) printf("first mismatch in part %d: 0x%08x x=%.9g libm=%.9g model=%.9g\n", t, jb[t].first, asf(jb[t].first), tanf(asf(jb[t].first)), orbx::glibc_tanf_model(asf(jb[t].first))); }
    printf("floats checked %u mismatches %ld\n", hi, bad);
    return bad != 0;
}
