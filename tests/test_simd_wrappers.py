"""The byte / packed-16-bit instruction wrappers of csrc/orbx_simd.h (v_mul_i32_i24, v_perm_b32, v_alignbyte_b32, v_dot4_u32_u8, v_dot2_u32_u16,
v_pk_maximum3_f16 / v_pk_minimum3_f16 on biased pixel patterns, v_pk_sub_i16) against independent numpy definitions, operand by operand:
on the GPU the instructions themselves, in the CPU suite the plain-C stand-ins the SIMT emulator runs instead of them.  The operand ranges are
those the kernels use (24-bit factors, selectors 0..7 and 0x0c, pixel patterns 0x6400 + byte in both halves for the 3-input min / max)."""
import numpy as np
import pytest

from orb_slam3_detailed_comments_amd.extractor import ORBextractor

N = 1 << 16


def _operands(rng):
    u32 = lambda: rng.integers(0, 1 << 32, N, dtype=np.uint64).astype(np.uint32)
    sets = {}
    # signed 24-bit factors (mul24): the kernels multiply row indices, pitches, fixed-point weights
    s24 = lambda: rng.integers(-(1 << 23), 1 << 23, N).astype(np.int32).view(np.uint32)
    sets["mul"] = (s24(), s24(), u32())
    sel = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 0x0c], np.uint32), (N, 4))
    sets["perm"] = (u32(), u32(), (sel[:, 0] | sel[:, 1] << 8 | sel[:, 2] << 16 | sel[:, 3] << 24).astype(np.uint32))
    sets["align"] = (u32(), u32(), rng.integers(0, 4, N).astype(np.uint32))
    sets["dot"] = (u32(), u32(), rng.integers(0, 1 << 24, N).astype(np.uint32))
    pix = lambda: (0x6400 + rng.integers(0, 256, N) | (0x6400 + rng.integers(0, 256, N)) << 16).astype(np.uint32)
    sets["pk3"] = (pix(), pix(), pix())
    i16 = lambda: (rng.integers(0, 1 << 15, N) | rng.integers(0, 1 << 15, N) << 16).astype(np.uint32)
    sets["pk"] = (i16(), i16(), rng.choice(np.array([0x64006400, 0x640064FF, 0x64FF6400, 0x64FF64FF], np.uint32), N))
    for k in sets:                                              # edge operands in front
        a, b, c = sets[k]
        a[:4] = [0, 0xFFFFFFFF if k in ("perm", "align", "dot") else a[0], a[1], a[2]]
    return sets


def _run(ex, a, b, c):
    out = np.zeros((22, N), np.uint32)
    a, b, c = (np.ascontiguousarray(x, np.uint32) for x in (a, b, c))
    ex._lib.check(ex._lib.L.orbx_debug_simd_selftest(ex._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, N, out.ctypes.data))
    return out


def _bytes(x):
    return np.stack([(x >> (8 * i)) & 0xFF for i in range(4)], 1).astype(np.uint64)


def _halves(x):
    return (x & 0xFFFF).astype(np.int64), (x >> 16).astype(np.int64)


def _check(ex):
    S = _operands(np.random.default_rng(2024))
    # mul24: product of the sign-extended low 24 bits, low 32 bits
    a, b, c = S["mul"]; out = _run(ex, a, b, c)
    exp = ((a.view(np.int32).astype(np.int64) * b.view(np.int32).astype(np.int64)) & 0xFFFFFFFF).astype(np.uint32)
    assert np.array_equal(out[0], exp) and np.array_equal(out[1], exp), "mul24"
    # v_perm_b32: result byte i = byte sel_i of {hi:lo} (lo = bytes 0..3), selector 0x0c = 0
    a, b, c = S["perm"]; out = _run(ex, a, b, c)
    src = np.concatenate([_bytes(b), _bytes(a), np.zeros((N, 8), np.uint64)], 1)            # index 12 -> 0
    selb = _bytes(c).astype(np.int64)
    exp = sum(src[np.arange(N), selb[:, i]] << np.uint64(8 * i) for i in range(4)).astype(np.uint32)
    assert np.array_equal(out[2], exp), "byte_perm"
    # v_alignbyte_b32
    a, b, c = S["align"]; out = _run(ex, a, b, c)
    v = (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)
    assert np.array_equal(out[3], ((v >> (np.uint64(8) * c.astype(np.uint64))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)), "align_byte"
    # dot products (no clamp, 32-bit wrap)
    a, b, c = S["dot"]; out = _run(ex, a, b, c)
    exp4 = ((_bytes(a) * _bytes(b)).sum(1) + c.astype(np.uint64)) & np.uint64(0xFFFFFFFF)
    assert np.array_equal(out[4], exp4.astype(np.uint32)), "dot4_u8"
    al, ah = _halves(a); bl, bh = _halves(b)
    assert np.array_equal(out[5], ((al * bl + ah * bh + c.astype(np.int64)) & 0xFFFFFFFF).astype(np.uint32)), "dot2_u16"
    # v_sad_u8: sum of absolute byte differences + c; v_mul_u32_u24: product of the low 24 bits, low 32 bits
    sad = np.abs(_bytes(a).astype(np.int64) - _bytes(b).astype(np.int64)).sum(1) + c.astype(np.int64)
    assert np.array_equal(out[13], (sad & 0xFFFFFFFF).astype(np.uint32)), "sad4_u8"
    um = ((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
    assert np.array_equal(out[14], um.astype(np.uint32)), "umul24"
    # v_mul_hi_u32_u24: bits 32.. of the product of the low 24 bits; v_add3_u32 (32-bit wrap)
    uh = ((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)) >> np.uint64(32)
    assert np.array_equal(out[20], uh.astype(np.uint32)), "mulhi_u24"
    assert np.array_equal(out[21], ((a.astype(np.uint64) + b.astype(np.uint64) + c.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)), "add3_u32"
    # packed three-input max / min on biased pixel patterns (positive binary16 numbers order like their bit patterns)
    a, b, c = S["pk3"]; out = _run(ex, a, b, c)
    hs = [_halves(x) for x in (a, b, c)]
    mx = np.maximum.reduce([h[0] for h in hs]) | np.maximum.reduce([h[1] for h in hs]) << 16
    mn = np.minimum.reduce([h[0] for h in hs]) | np.minimum.reduce([h[1] for h in hs]) << 16
    assert np.array_equal(out[6], mx.astype(np.uint32)) and np.array_equal(out[7], mn.astype(np.uint32)), "pk_max3 / pk_min3"
    # packed 16-bit subtract (wraps per half), packed xor
    a, b, c = S["pk"]; out = _run(ex, a, b, c)
    al, ah = _halves(a); bl, bh = _halves(b)
    assert np.array_equal(out[8], (((al - bl) & 0xFFFF) | ((ah - bh) & 0xFFFF) << 16).astype(np.uint32)), "pk_sub"
    assert np.array_equal(out[9], a ^ c), "pk_xor"
    # wave primitives (orbx_block.h): inclusive scan / sum of a & 0xFFFF and minimum of b over every 64 consecutive elements
    v = (a & 0xFFFF).astype(np.int64).reshape(-1, 64)
    assert np.array_equal(out[10], np.cumsum(v, 1).reshape(-1).astype(np.uint32)), "wave_incl_scan"
    assert np.array_equal(out[11], np.repeat(v.sum(1), 64).astype(np.uint32)), "wave_sum"
    assert np.array_equal(out[12], np.repeat(b.reshape(-1, 64).min(1), 64)), "wave_min_u32"
    # 64-bit scans of three 20-bit fields (on the GPU: DPP steps on both halves with carry; k_quadtree's workgroup scans)
    f = (a & 0xFFFFF).astype(np.uint64) | (b & 0xFFFFF).astype(np.uint64) << np.uint64(20) | (c & 0xFFFFF).astype(np.uint64) << np.uint64(40)
    ws = np.cumsum(f.reshape(-1, 64), 1, dtype=np.uint64).reshape(-1)
    assert np.array_equal(out[15].astype(np.uint64) | out[16].astype(np.uint64) << np.uint64(32), ws), "wave_incl_scan (64-bit)"
    fb = f.reshape(-1, 256)
    bs = (np.cumsum(fb, 1, dtype=np.uint64) - fb + (fb.sum(1, dtype=np.uint64) << np.uint64(1))[:, None]).reshape(-1)
    assert np.array_equal(out[17].astype(np.uint64) | out[18].astype(np.uint64) << np.uint64(32), bs), "block_excl_scan_n (64-bit)"
    bit = np.where((b & 7) == 0, np.uint32(1) << (a & 31), np.uint32(0)).astype(np.uint32).reshape(-1, 64)
    assert np.array_equal(out[19], np.repeat(np.bitwise_or.reduce(bit, 1), 64)), "wave_or_u32"


def test_simd_wrappers_emulated(emu_lib):
    _check(ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib))


@pytest.mark.gpu
def test_simd_wrappers_gpu(hip_lib):
    _check(ORBextractor(500, 1.2, 8, 20, 7, lib=hip_lib))
