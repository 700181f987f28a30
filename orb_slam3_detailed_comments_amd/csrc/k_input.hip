// k_input.hip — the step immediately in front of the extractor, so that raw camera frames can be handed
// over as they come off the driver:
//   k_input_remap   cv::remap(src, dst, M1, M2, cv::INTER_LINEAR) as System::TrackStereo uses it for stereo rectification
//                   (src/System.cc:286-293; maps from cv::initUndistortRectifyMap(..., CV_32F, M1, M2), src/Settings.cc:549-574).
//                   OpenCV's 8U path: sx = cvRound(mapx * 32), integer source pixel sx >> 5 (saturated to short), 5-bit fractions,
//                   weights (32-fx)(32-fy)*32, (fx)(32-fy)*32, (32-fx)(fy)*32, fx*fy*32 (BilinearTab_i, sum 32768),
//                   out = (sum + (1 << 14)) >> 15, BORDER_CONSTANT with value 0 tap by tap.
//   k_input_resize  cv::resize(src, dst, newImSize) INTER_LINEAR 8U (src/System.cc:295-297), same fixed-point arithmetic as k_resize
//   k_input_gray    cv::cvtColor(..., COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) (src/Tracking.cc:1532-1560): OpenCV's 8U path
//                   (r*RY + g*GY + b*BY + (1 << (shift-1))) >> shift with (9798, 19235, 3735, 15) [OpenCV 4.x] or
//                   (4899, 9617, 1868, 14) [OpenCV 3.x]
// All three write level 0 of the pyramid block (row pitch of the level, zeros in the pitch padding) or, for multi-channel
// frames, an interleaved intermediate that k_input_gray then converts - the order System (geometry) -> Tracking (grey) of the reference.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_simd.h"

namespace orbx {

// grid (ceil(dst_pitch_px/64), ceil(out_h/4), B), block (64,4).  One output pixel (all C channels) per thread.
// dst: pitch dst_pitch bytes per row, C bytes per pixel; image stride dst_stride.
__global__ void __launch_bounds__(256) k_input_remap(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t simg, int C,
                                                     const float* __restrict__ mapx, const float* __restrict__ mapy, int out_w, int out_h,
                                                     uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride) {
    const int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y), b = (int)blockIdx.z;
    if (y >= out_h || x * C >= dst_pitch) return;
    uint8_t* o = dst + (size_t)b * dst_stride + (size_t)y * dst_pitch + (size_t)x * C;
    if (x >= out_w) { for (int c = 0; c < C && x * C + c < dst_pitch; c++) o[c] = 0; return; }
    const float mx = mapx[(size_t)y * out_w + x], my = mapy[(size_t)y * out_w + x];
    const int fsx = __float2int_rn(__fmul_rn(mx, 32.0f)), fsy = __float2int_rn(__fmul_rn(my, 32.0f));     // cvRound(map * INTER_TAB_SIZE)
    int sx = fsx >> 5, sy = fsy >> 5;
    sx = imin(imax(sx, -32768), 32767); sy = imin(imax(sy, -32768), 32767);                                // saturate_cast<short>
    const int fx = fsx & 31, fy = fsy & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0in = sx >= 0 && sx < sw, x1in = sx + 1 >= 0 && sx + 1 < sw, y0in = sy >= 0 && sy < sh, y1in = sy + 1 >= 0 && sy + 1 < sh;
    const uint8_t* s = src + (size_t)b * simg;
    for (int c = 0; c < C; c++) {
        const int p00 = (x0in && y0in) ? s[(size_t)sy * sstride + (size_t)sx * C + c] : 0;
        const int p01 = (x1in && y0in) ? s[(size_t)sy * sstride + (size_t)(sx + 1) * C + c] : 0;
        const int p10 = (x0in && y1in) ? s[(size_t)(sy + 1) * sstride + (size_t)sx * C + c] : 0;
        const int p11 = (x1in && y1in) ? s[(size_t)(sy + 1) * sstride + (size_t)(sx + 1) * C + c] : 0;
        int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        o[c] = (uint8_t)imin(imax(v, 0), 255);
    }
}

// same launch shape.  xt/yt: taps of cv::resize for sw -> out_w and sh -> out_h (ofs = left/top source index, w = a0 | a1 << 16).
__global__ void __launch_bounds__(256) k_input_resize(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t simg, int C,
                                                      const ResizeTap* __restrict__ xt, const ResizeTap* __restrict__ yt, int out_w, int out_h,
                                                      uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride) {
    const int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y), b = (int)blockIdx.z;
    if (y >= out_h || x * C >= dst_pitch) return;
    uint8_t* o = dst + (size_t)b * dst_stride + (size_t)y * dst_pitch + (size_t)x * C;
    if (x >= out_w) { for (int c = 0; c < C && x * C + c < dst_pitch; c++) o[c] = 0; return; }
    const ResizeTap tx = xt[x], ty = yt[y];
    const int x0 = tx.ofs, x1 = imin(tx.ofs + 1, sw - 1);
    const int y0 = imin(imax(ty.ofs, 0), sh - 1), y1 = imin(imax(ty.ofs + 1, 0), sh - 1);
    const int a0 = (int)(int16_t)(tx.w & 0xFFFF), a1 = tx.w >> 16, b0 = (int)(int16_t)(ty.w & 0xFFFF), b1 = ty.w >> 16;
    const uint8_t* s = src + (size_t)b * simg;
    for (int c = 0; c < C; c++) {
        const int h0 = s[(size_t)y0 * sstride + (size_t)x0 * C + c] * a0 + s[(size_t)y0 * sstride + (size_t)x1 * C + c] * a1;
        const int h1 = s[(size_t)y1 * sstride + (size_t)x0 * C + c] * a0 + s[(size_t)y1 * sstride + (size_t)x1 * C + c] * a1;
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)imin(imax(v, 0), 255);
    }
}

// grid (ceil(dst_pitch/256), ceil(h/4), B), block (64,4).  src: C = 3 or 4 interleaved channels; ridx = index of the red channel
// (0 for RGB / RGBA, 2 for BGR / BGRA); (ry, gy, by, shift) = the OpenCV version's coefficients.
// A thread converts FOUR adjacent pixels: their 12 / 16 source bytes come in as three / four dwords (any byte address: global loads need no alignment
// on gfx950) and leave as one dword - a byte load per channel and a byte store per pixel ran this streaming kernel at 1.4 TB/s (round 4).
__global__ void __launch_bounds__(256) k_input_gray(const uint8_t* __restrict__ src, int sstride, size_t simg, int C, int ridx,
                                                    int ry, int gy, int by, int shift, int w, int h,
                                                    uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride) {
    const int x = 4 * (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y), b = (int)blockIdx.z;
    if (y >= h || x >= dst_pitch) return;                       // (the pitch is a multiple of 64: a thread's dword lies inside the row)
    uint32_t* o = (uint32_t*)(dst + (size_t)b * dst_stride + (size_t)y * dst_pitch + x);
    if (x >= w) { *o = 0u; return; }
    const uint8_t* p = src + (size_t)b * simg + (size_t)y * sstride + (size_t)x * C;
    const int half = 1 << (shift - 1);
    uint32_t out = 0;
    if (x + 4 <= w) {
        uint32_t d[4];
        d[0] = load_u32_any(p); d[1] = load_u32_any(p + 4); d[2] = load_u32_any(p + 8); d[3] = C == 4 ? load_u32_any(p + 12) : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // channel c of pixel j is byte j * C + c of the 12 / 16 loaded bytes
            auto byte_at = [&](int k) { return (int)((d[k >> 2] >> (8 * (k & 3))) & 0xFFu); };
            int r, g, bl;
            if (C == 4) { const uint32_t q = d[j]; r = (int)((q >> (8 * ridx)) & 0xFFu); g = (int)((q >> 8) & 0xFFu); bl = (int)((q >> (8 * (2 - ridx))) & 0xFFu); }
            else { r = byte_at(3 * j + ridx); g = byte_at(3 * j + 1); bl = byte_at(3 * j + 2 - ridx); }
            out |= (uint32_t)((mul24(r, ry) + mul24(g, gy) + mul24(bl, by) + half) >> shift) << (8 * j);
        }
    } else {                                                    // the row's last, partial group: pixel by pixel, zeros behind the width
        for (int j = 0; j < 4 && x + j < w; j++) {
            const uint8_t* q = p + j * C;
            out |= (uint32_t)(((int)q[ridx] * ry + (int)q[1] * gy + (int)q[2 - ridx] * by + half) >> shift) << (8 * j);
        }
    }
    *o = out;
}

}  // namespace orbx
