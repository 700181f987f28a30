// glibc_tanf_model.h — a bit-for-bit model of glibc 2.35's tanf() (sysdeps/ieee754/flt-32/s_tanf.c + k_tanf.c + e_rem_pio2f.c: the fdlibm
// single-precision kernel behind a double-precision reduction by pi/2; not an ifunc, one variant for every x86-64 CPU, compiled without FMA),
// restricted to 0 <= x < 120.
//
// Why: KannalaBrandt8::unproject (reference src/CameraModels/KannalaBrandt8.cpp:180-216) scales the ray by std::tan(theta) / theta_d with a
// float theta in (1e-8, pi/2], i.e. tanf.  It is not correctly rounded, the device's tanf is another algorithm, and a last-bit difference in
// the ray of a low-parallax pair moves the triangulated depth (:553-573) by ~eps32 / (1 - cos parallax).  The model uses IEEE float
// and double multiply / add / divide only (no contraction: the functions below are built with -ffp-contract=off on the host and the device) and was
// checked EXHAUSTIVELY against the live glibc for every float in [0, 8] (tools/check_tanf_model.c: 0 mismatches).
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif
#include <cstdint>
#include <cstring>

namespace orbx {

ORBX_HD inline uint32_t tanf_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
ORBX_HD inline float tanf_from_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// __kernel_tanf(x, y, iy) of k_tanf.c: tan(x + y) for iy = 1, -1 / tan(x + y) for iy = -1, |x| <~ pi/4
ORBX_HD inline float glibc_kernel_tanf(float x, float y, int iy) {
    const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f,
                T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
                T12 = 2.5907305826e-05f;
    float z, r, v, w, s;
    const int32_t hx = (int32_t)tanf_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {                                  // |x| < 2**-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return one / (x < 0 ? -x : x);
            else if (iy == 1) return x;
            else return -one / x;
        }
    }
    if (ix >= 0x3f2ca140) {                                 // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w; y = 0.0f;
        if ((x < 0 ? -x : x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
    }
    z = x * x;
    w = z * z;
    r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (x + r), computed accurately
    float a, t;
    z = tanf_from_bits(tanf_bits(w) & 0xfffff000u);
    v = r - (z - x);
    t = a = -1.0f / w;
    t = tanf_from_bits(tanf_bits(t) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}

// tanf(x) for 0 <= x < 120 (s_tanf.c; the medium-argument reduction of e_rem_pio2f.c / sincosf.h's reduce_fast: x - n pi/2 in double with
// n = round(x * 2/pi), handed to the kernel as two floats)
ORBX_HD inline float glibc_tanf_model(float x) {
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    const uint32_t ix = tanf_bits(x) & 0x7fffffffu;
    if (ix <= 0x3f490fdau) return glibc_kernel_tanf(x, 0.0f, 1);
    const double xd = (double)x;
    const double r = xd * HPI_INV;
    const int n = ((int)r + 0x800000) >> 24;
    const double xr = xd - (double)n * HPI;
    const float y0 = (float)xr;
    const float y1 = (float)(xr - (double)y0);
    return glibc_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

}  // namespace orbx
