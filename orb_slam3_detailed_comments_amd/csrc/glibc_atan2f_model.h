// glibc_atan2f_model.h — a bit-for-bit model of glibc 2.35's atanf() / atan2f() (fdlibm: sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c - argument
// reduction to one of four intervals and an 11-term odd polynomial, everything in float arithmetic; no FMA variant exists for them).
//
// Why: KannalaBrandt8::project (reference src/CameraModels/KannalaBrandt8.cpp:87-104) computes theta = atan2f(sqrtf(x^2 + y^2), z) and
// psi = atan2f(y, x).  atan2f is not correctly rounded, and the device's atan2f differs from glibc's in the last bits for most arguments, which
// moved projected pixel coordinates by a few ulp (found by the first GPU run of tests/test_local_points_rig.py).  Every operation below is an
// IEEE single-precision add / multiply / divide (the build uses -ffp-contract=off), so host, emulator and GPU agree.
// tools/check_atan2f_model.c: atanf for EVERY float, atan2f for 4e9 pairs (all sign / magnitude classes) against the live libm.
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif
#include <cstdint>
#include <cstring>

namespace orbx {

ORBX_HD inline uint32_t at_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
ORBX_HD inline float at_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

ORBX_HD inline float glibc_atanf_model(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)at_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;                    // NaN
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                    // |x| < 0.4375
        if (ix < 0x31000000) return x;                        // |x| < 2^-29
        id = -1;
    } else {
        x = at_float((uint32_t)ix);                           // fabsf
        if (ix < 0x3f980000) {                                // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }     // 7/16 <= |x| < 11/16
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                              // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }       // |x| < 2.4375
            else { id = 3; x = -1.0f / x; }                                            // 2.4375 <= |x| < 2^25
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

ORBX_HD inline float glibc_atan2f_model(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)at_bits(x), hy = (int32_t)at_bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;    // NaN
    if (hx == 0x3f800000) return glibc_atanf_model(y);        // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);        // 2 * sign(x) + sign(y)
    if (iy == 0) {                                            // y = 0
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {                                   // x = inf
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
        }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                    // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                     // |y| / x < -2^60
    else z = glibc_atanf_model(at_float(at_bits(y / x) & 0x7fffffffu));
    switch (m) {
        case 0: return z;
        case 1: return at_float(at_bits(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace orbx
