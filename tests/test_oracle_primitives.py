"""Oracle self-validation (OpenCV is absent): every restated OpenCV primitive in
oracle/orb_primitives.h is cross-checked against an independent slow definition written in numpy/Python."""
import ctypes as C

import numpy as np
import oracle_lib as ol

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def slow_is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in RING]
    for start in range(16):
        arc = [ring[(start + k) % 16] for k in range(9)]
        if all(p > v + t for p in arc) or all(p < v - t for p in arc):
            return True
    return False


def slow_score(img, x, y):
    """largest threshold for which (x,y) is still a FAST-9 corner, -1 if never"""
    best = -1
    for t in range(256):
        if slow_is_corner(img, x, y, t):
            best = t
        else:
            break
    return best


def test_fast_corner_and_score_vs_definition():
    L = ol.oracle()
    rng = np.random.default_rng(1)
    L.orbo_prim_is_corner.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orbo_prim_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
    n_corner = 0
    for trial in range(6):
        img = rng.integers(0, 256, (24, 24), dtype=np.uint8)
        if trial % 2:   # blocky image: long arcs, real corners
            img = np.kron(rng.integers(0, 2, (6, 6)) * 120 + 40, np.ones((4, 4))).astype(np.uint8) + rng.integers(0, 6, (24, 24)).astype(np.uint8)
        for y in range(3, 21):
            for x in range(3, 21):
                p = img.ctypes.data + y * 24 + x
                for t in (7, 20):
                    c = slow_is_corner(img, x, y, t)
                    assert bool(L.orbo_prim_is_corner(p, 24, t)) == c
                    if c:
                        n_corner += 1
                        assert L.orbo_prim_corner_score(p, 24, t) == slow_score(img, x, y)
    assert n_corner > 50


def test_fast_nms_row_major_and_strict():
    L = ol.oracle()
    L.orbo_prim_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(2)
    img = (np.kron(rng.integers(0, 2, (10, 12)) * 100 + 60, np.ones((5, 5))) + rng.integers(0, 8, (50, 60))).astype(np.uint8)
    out = np.zeros((4096, 3), np.int32)
    n = L.orbo_prim_fast(img.ctypes.data, 60, 50, 60, 20, 1, out.ctypes.data, 4096)
    pts = out[:n]
    # independent: score map then strict 3x3 max
    sc = np.zeros((50, 60), np.int32)
    for y in range(3, 47):
        for x in range(3, 57):
            if slow_is_corner(img, x, y, 20):
                sc[y, x] = slow_score(img, x, y)
    exp = []
    for y in range(3, 47):
        for x in range(3, 57):
            s = sc[y, x]
            if s > 0:
                nb = sc[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                if (s > nb).all():
                    exp.append((x, y, s))
    assert n == len(exp) and n > 5
    assert [tuple(p) for p in pts.tolist()] == exp   # includes the row-major (y, then x) order


def np_resize(src, dw, dh):
    sh, sw = src.shape
    def axis(ss, ds, clamp):
        sc = 1.0 / (float(ds) / ss)
        d = np.arange(ds)
        f = ((d + 0.5) * sc - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:
            lo = s < 0; f[lo] = 0; s[lo] = 0
            hi = s >= ss - 1; f[hi] = 0; s[hi] = ss - 1
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, a0, a1
    sx, a0, a1 = axis(sw, dw, True)
    sy, b0, b1 = axis(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    r0 = H[np.clip(sy, 0, sh - 1)]; r1 = H[np.clip(sy + 1, 0, sh - 1)]
    v = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def test_resize_vs_numpy_restatement_and_properties():
    L = ol.oracle()
    L.orbo_prim_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    for (sw, sh, dw, dh) in [(752, 480, 627, 400), (627, 400, 522, 333), (97, 61, 81, 51), (64, 64, 64, 64), (40, 30, 57, 41)]:
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        L.orbo_prim_resize(src.ctypes.data, sw, sh, dst.ctypes.data, dw, dh)
        assert np.array_equal(dst, np_resize(src, dw, dh))
        if (sw, sh) == (dw, dh):
            assert np.array_equal(dst, src)            # identity
        const = np.full((sh, sw), 173, np.uint8)
        L.orbo_prim_resize(const.ctypes.data, sw, sh, dst.ctypes.data, dw, dh)
        assert (dst == 173).all()                      # weights sum to one


def test_gaussian_blur_vs_exact_convolution():
    L = ol.oracle()
    L.orbo_prim_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(4)
    for variant, taps in ((0, [18, 34, 48, 56, 48, 34, 18]), (1, [18, 34, 49, 55, 49, 34, 18])):
        img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
        out = np.zeros_like(img)
        L.orbo_prim_blur(img.ctypes.data, 53, 37, out.ctypes.data, variant)
        k = np.array(taps, np.int64)
        P = np.pad(img.astype(np.int64), 3, mode="reflect")     # numpy 'reflect' == BORDER_REFLECT_101
        H = sum(k[i] * P[:, i:i + 53] for i in range(7))
        V = sum(k[j] * H[j:j + 37, :] for j in range(7))
        exp = np.minimum((V + 32768) >> 16, 255).astype(np.uint8)
        assert np.array_equal(out, exp)
    # float Gaussian sigma=2 for orientation: fixed-point taps are its rounding
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0); g = g / g.sum() * 256
    assert np.abs(g - np.array([18, 34, 48, 56, 48, 34, 18])).max() < 1.0          # error-diffused, sums to 256
    assert np.array_equal(np.rint(g), [18, 34, 49, 55, 49, 34, 18])                # 3.2 path: plain rounding, sums to 257


def test_fast_atan2_accuracy_and_quadrants():
    L = ol.oracle()
    rng = np.random.default_rng(5)
    y = rng.integers(-200000, 200000, 20000).astype(np.float32); x = rng.integers(-200000, 200000, 20000).astype(np.float32)
    got = np.array([L.orbo_fast_atan2(float(a), float(b)) for a, b in zip(y[:4000], x[:4000])])
    ref = np.degrees(np.arctan2(y[:4000].astype(np.float64), x[:4000].astype(np.float64))) % 360.0
    err = np.abs(((got - ref + 180) % 360) - 180)
    assert err.max() < 0.3            # the documented accuracy of cv::fastAtan2
    assert L.orbo_fast_atan2(0.0, 0.0) == 0.0
    assert L.orbo_fast_atan2(1.0, 0.0) == 90.0 and L.orbo_fast_atan2(0.0, -1.0) == 180.0 and L.orbo_fast_atan2(-1.0, 0.0) == 270.0


def test_round_half_even_and_border():
    L = ol.oracle()
    L.orbo_prim_round.argtypes = [C.c_double]
    assert [L.orbo_prim_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    L.orbo_prim_border.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    img = np.arange(30 * 40, dtype=np.uint32).reshape(30, 40).astype(np.uint8)
    out = np.zeros((30 + 38, 40 + 38), np.uint8)
    L.orbo_prim_border(img.ctypes.data, 40, 30, out.ctypes.data, 19)
    assert np.array_equal(out, np.pad(img, 19, mode="reflect"))


def test_descriptor_distance_known_answers():
    rng = np.random.default_rng(6)
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert ol.oracle_hamming(z, z) == 0 and ol.oracle_hamming(z, o) == 256
    a = rng.integers(0, 256, (300, 32), dtype=np.uint8); b = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    exp = np.unpackbits(a ^ b, axis=1).sum(1)
    assert [ol.oracle_hamming(a[i], b[i]) for i in range(300)] == exp.tolist()


# ---- the input pre-step: cv::remap / cv::resize on colour frames / cv::cvtColor ------------------------------
# Call sites in the reference: System.cc:286-293 (cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) with the CV_32F maps of Settings.cc:549-574),
# System.cc:295-297 (cv::resize to Settings::newImSize), Tracking.cc:1532-1560 (cv::cvtColor RGB/BGR/RGBA/BGRA -> GRAY).  OpenCV is not
# installed, so each restatement in oracle/orb_primitives.h is checked here against an independent slow definition in exact rational
# arithmetic (fractions.Fraction), written from the operators' published definitions:
#   remap INTER_LINEAR, 8U (OpenCV 3.x and 4.x, imgproc/src/imgwarp.cpp / remap.cpp: the same fixed-point scheme in both): source
#     coordinates are quantised to 1/32 pixel (INTER_BITS = 5, cvRound(x * 32)), the four neighbours are blended with the bilinear weights
#     of that quantised position, the sum is rounded half up, taps outside the image read the constant border value 0;
#   cvtColor to grey, 8U: Y = (R * cR + G * cG + B * cB + half) >> shift with (cR, cG, cB, shift) = (9798, 19235, 3735, 15) in OpenCV 4.x
#     (variant 0) and (4899, 9617, 1868, 14) in OpenCV 3.x (variant 1) - 0.299 / 0.587 / 0.114 in 15 / 14 bits.
def _slow_remap(src, mapx, mapy):
    from fractions import Fraction
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    s3 = src.reshape(sh, sw, cn)
    dh, dw = mapx.shape
    out = np.zeros((dh, dw, cn), np.uint8)

    def rne(v):                       # cvRound: round half to even
        f = int(np.floor(v)); d = v - f
        return f + 1 if d > 0.5 or (d == 0.5 and f % 2 == 1) else f
    for y in range(dh):
        for x in range(dw):
            qx = rne(float(np.float32(mapx[y, x]) * np.float32(32.0))); qy = rne(float(np.float32(mapy[y, x]) * np.float32(32.0)))
            ix, iy = qx >> 5, qy >> 5
            fx, fy = Fraction(qx & 31, 32), Fraction(qy & 31, 32)
            for c in range(cn):
                acc = Fraction(0)
                for (dx, dy, wgt) in ((0, 0, (1 - fx) * (1 - fy)), (1, 0, fx * (1 - fy)), (0, 1, (1 - fx) * fy), (1, 1, fx * fy)):
                    xx, yy = ix + dx, iy + dy
                    if 0 <= xx < sw and 0 <= yy < sh:
                        acc += wgt * int(s3[yy, xx, c])
                out[y, x, c] = min(255, int(np.floor(acc + Fraction(1, 2))))
    return out.reshape((dh, dw) if src.ndim == 2 else (dh, dw, cn))


def test_remap_vs_exact_bilinear_definition():
    rng = np.random.default_rng(11)
    for cn in (1, 3):
        src = rng.integers(0, 256, (19, 23) if cn == 1 else (19, 23, cn), dtype=np.uint8)
        # a rectification-like map (smooth warp) plus adversarial entries: exact integers, exact 1/64 ties of the 1/32 grid, positions
        # left / right / above / below the image (constant border), one pixel outside (half-covered taps)
        yy, xx = np.mgrid[0:17, 0:21].astype(np.float32)
        mx = (xx + 1.3 + 0.04 * yy + 0.8 * np.sin(yy / 5.0)).astype(np.float32); my = (yy + 0.7 - 0.03 * xx + 0.6 * np.cos(xx / 4.0)).astype(np.float32)
        mx[0, :6] = [0.0, 5.0, 22.0, -1.0, 23.0, -0.5]; my[0, :6] = [0.0, 3.0, 18.0, 4.0, 4.0, 18.5]
        mx[1, :5] = [2.015625, 2.046875, 7.5, 21.984375, -0.015625]; my[1, :5] = [3.015625, 3.046875, 9.5, 17.984375, -0.015625]
        mx[2, :3] = [-40.0, 1000.0, 4.0]; my[2, :3] = [5.0, 5.0, -300.0]
        got = ol.oracle_remap(src, mx, my)
        assert np.array_equal(got, _slow_remap(src, mx, my))
    # properties: the identity map is the identity, an integer shift is a shift with a zero border
    img = rng.integers(0, 256, (12, 14), dtype=np.uint8)
    yy, xx = np.mgrid[0:12, 0:14].astype(np.float32)
    assert np.array_equal(ol.oracle_remap(img, xx, yy), img)
    sh = ol.oracle_remap(img, xx + 3, yy - 2)
    assert np.array_equal(sh[2:, :11], img[:10, 3:]) and (sh[:2] == 0).all() and (sh[:, 11:] == 0).all()


def test_cvtcolor_vs_definition_and_known_answers():
    from fractions import Fraction
    rng = np.random.default_rng(12)
    coef = {0: (9798, 19235, 3735, 15), 1: (4899, 9617, 1868, 14)}
    for variant, (cr, cg, cb, shift) in coef.items():
        assert cr + cg + cb == 1 << shift                      # white stays 255, grey stays grey
        for cn in (3, 4):
            img = rng.integers(0, 256, (9, 11, cn), dtype=np.uint8)
            img[0, 0, :3] = (255, 255, 255); img[0, 1, :3] = (255, 0, 0); img[0, 2, :3] = (0, 255, 0); img[0, 3, :3] = (0, 0, 255); img[0, 4, :3] = (128, 128, 128)
            for red_first in (1, 0):
                got = ol.oracle_gray(img, red_first, variant)
                exp = np.zeros((9, 11), np.uint8)
                for y in range(9):
                    for x in range(11):
                        r, g, b = (img[y, x, 0], img[y, x, 1], img[y, x, 2]) if red_first else (img[y, x, 2], img[y, x, 1], img[y, x, 0])
                        exp[y, x] = int(np.floor(Fraction(int(r) * cr + int(g) * cg + int(b) * cb, 1 << shift) + Fraction(1, 2)))
                assert np.array_equal(got, exp)
                # ITU-R BT.601 luma of the primaries: 0.299, 0.587, 0.114 of 255
                first, third = (76, 29) if red_first else (29, 76)
                assert [int(v) for v in got[0, :5]] == [255, first, 150, third, 128]


def test_resize_colour_is_per_channel_resize():
    """cv::resize on a CV_8UC3 frame interpolates every channel independently with the coefficients of the single-channel case (the
    horizontal / vertical tables depend on the geometry only); the single-channel primitive is pinned above."""
    rng = np.random.default_rng(13)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for (dw, dh) in ((41, 29), (53, 37), (60, 44), (26, 18)):
        got = ol.oracle_resize_cn(img, dw, dh)
        for c in range(3):
            assert np.array_equal(got[:, :, c], ol.oracle_resize_cn(np.ascontiguousarray(img[:, :, c]), dw, dh))
