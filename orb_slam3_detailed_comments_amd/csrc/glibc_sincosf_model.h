// glibc_sincosf_model.h — a bit-for-bit model of glibc 2.35's cosf()/sinf() (the ARM "optimized
// routines" single-precision kernels: fast range reduction by pi/2 in double + degree-8/7 double
// polynomials, result rounded once to float), restricted to the argument range the ORB descriptor
// needs: x in [0, 2*pi] (angle*pi/180 with angle in [0,360], reference src/ORBextractor.cc:155-157).
//
// Why: the reference rotates the BRIEF pattern with (float)cos(angle), (float)sin(angle), i.e. glibc's
// cosf/sinf.  Those are not correctly rounded, so a device libm would differ in the last bit for some
// angles.  This model uses only IEEE double multiply/add (no FMA needed: verified identical with and
// without contraction) and was checked EXHAUSTIVELY against the live glibc for all 1,087,373,312 floats
// in [0, 6.5] (tools/check_sincosf_model.c: 0 mismatches); tests/test_models.py re-checks a sample.
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif

namespace orbx {

ORBX_HD inline unsigned sc_abstop12(float x) {
    union { float f; unsigned u; } c; c.f = x; return (c.u >> 20) & 0x7ff;
}
// n even: sine polynomial, n odd: cosine polynomial; neg selects the -cos table
ORBX_HD inline float sc_poly(double x, double x2, bool neg, int n) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5;
    const double C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = S2 + x2 * S3;
        const double x7 = x3 * x2;
        const double s = x + x3 * S1;
        return (float)(s + x7 * s1);
    } else {
        const double sg = neg ? -1.0 : 1.0;
        const double x4 = x2 * x2;
        const double c2 = sg * C3 + x2 * (sg * C4);
        const double c1 = sg * C0 + x2 * (sg * C1);
        const double x6 = x4 * x2;
        const double c = c1 + x4 * (sg * C2);
        return (float)(c + x6 * c2);
    }
}
ORBX_HD inline double sc_reduce(double x, int* np) {
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    const double r = x * HPI_INV;
    const int n = ((int)r + 0x800000) >> 24;
    *np = n;
    return x - n * HPI;
}
// valid for 0 <= y < 120 (the model is only *verified* on [0, 6.5])
ORBX_HD inline float glibc_cosf(float y) {
    double x = y;
    if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {
        const double x2 = x * x;
        if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) return 1.0f;
        return sc_poly(x, x2, false, 1);
    }
    int n; x = sc_reduce(x, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sc_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}
ORBX_HD inline float glibc_sinf(float y) {
    double x = y;
    if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {
        const double x2 = x * x;
        if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) return y;
        return sc_poly(x, x2, false, 0);
    }
    int n; x = sc_reduce(x, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sc_poly(x * s, x * x, (n & 2) != 0, n);
}

}  // namespace orbx
