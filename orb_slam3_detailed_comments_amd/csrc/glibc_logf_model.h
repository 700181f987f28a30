// glibc_logf_model.h — a bit-for-bit model of glibc 2.35's logf() (the ARM "optimized routines" single-precision kernel,
// sysdeps/ieee754/flt-32/e_logf.c: 16-entry table of 1/c and log(c), degree-3 polynomial in double, result rounded once to float).
//
// Why: MapPoint::PredictScale (reference src/MapPoint.cc:688-731) computes ceil(log(ratio) / mfLogScaleFactor) with a FLOAT ratio inside a
// translation unit that is `using namespace std`, so the call resolves to std::log(float) = logf, followed by a float division and
// ceil(float).  logf is not correctly rounded, so neither the device's logf nor a double log reproduces it for every ratio (an fp64
// log differs from it in about 5 of 1e8 ratios: one wrong mnTrackScaleLevel every few thousand frames of 5000 local points).
// The model uses IEEE double multiply / add only; `fma` selects the contraction pattern of glibc's FMA build (the x86-64 ifunc variant
// __logf_fma, which is what runs on every FMA-capable CPU) - tools/check_logf_model.c compares both patterns with the live libm for EVERY
// positive finite float and reports which one the machine uses; the result is recorded in DESIGN.md.
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orbx {

// valid for finite x > 0 (normal or subnormal); x <= 0, inf and nan are the caller's business (PredictScale's ratio is a quotient of two
// positive distances)
template <bool FMA>
ORBX_HD inline float glibc_logf_model(float x) {
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
                             0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,
                             -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, Ln2 = 0x1.62e42fefa39efp-1;
    uint32_t ix; memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {          // subnormal: normalise (x * 2^23, exponent - 23)
        const float xs = x * 0x1p23f;
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    // x = 2^k z, z in [OFF, 2 OFF), exact
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int32_t)tmp >> 23;                               // arithmetic shift
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    float zf; memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    // log(x) = log1p(z / c - 1) + log(c) + k ln 2
    double r, y0, r2, y;
    if (FMA) {
        r = fma(z, invc[i], -1.0);
        y0 = fma((double)k, Ln2, logc[i]);
        r2 = r * r;
        y = fma(A1, r, A2);
        y = fma(A0, r2, y);
        y = fma(y, r2, y0 + r);
    } else {
        r = z * invc[i] - 1.0;
        y0 = logc[i] + (double)k * Ln2;
        r2 = r * r;
        y = A1 * r + A2;
        y = A0 * r2 + y;
        y = y * r2 + (y0 + r);
    }
    return (float)y;
}

}  // namespace orbx
