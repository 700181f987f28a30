// k_describe.hip — per-keypoint kernels of the ORB extractor:
//   k_layout        final output index of every keypoint: level scaling + lapping-area front/back
//                   reorder of ORBextractor::operator() (src/ORBextractor.cc:1613-1681) as prefix sums
//   k_orient_brief  a wave64 works through 8 keypoints: IC_Angle (:91-138, :580-591) on the raw level, then the
//                   256-bit steered BRIEF (:150-203) on the blurred level; 64 lanes x 4 rounds of
//                   __ballot assemble the four 64-bit descriptor words directly; the atan2 / cos / sin of
//                   the 8 keypoints are evaluated together, one keypoint per lane.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_simd.h"
#include "orbx_kernels.h"
#include "undistort_model.h"
#include "glibc_sincosf_model.h"

namespace orbx {

// rBRIEF pattern (data table of the reference, src/ORBextractor.cc:206-464), read through the scalar/L1 path
__device__ const signed char d_brief_pattern[1024] = {
#include "brief_pattern.inc"
};
#define BRIEF_PATTERN d_brief_pattern

// cv::fastAtan2 (OpenCV core mathfuncs): fp32, explicit IEEE ops, no contraction.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float s = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = 2.2204460492503131e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps)); c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps)); c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// grid (B), 256 threads.  The level-ordered keypoint slots of an image (level l owns [kp_off, kp_off + kp_cap), the first lvl_count of them
// are in use) are walked in slot order = output order of the reference: a thread takes kLayoutPer consecutive slots, so that one workgroup
// scan per 256 * kLayoutPer slots (one in all for the usual feature counts) places every keypoint, with all key loads in flight at once.
constexpr int kLayoutPer = 8;
__global__ void __launch_bounds__(256) k_layout(const LevelInfo* __restrict__ lv, int nlevels,
                                                const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                const int* __restrict__ lvl_count, int lap0, int lap1,
                                                int* __restrict__ final_idx, int* __restrict__ n_out, int* __restrict__ mono_out,
                                                int nb, int* __restrict__ row_start, int* __restrict__ row_items) {
    ORBX_DYN_SMEM(smem);                                // row index: histogram and cursors, (nb + 1) ints each
    __shared__ unsigned long long s_scan[20];
    __shared__ int s_cnt[kMaxLevels], s_off[kMaxLevels + 1];
    __shared__ float s_scale[kMaxLevels];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (tid < nlevels) { s_cnt[tid] = lvl_count[(size_t)b * nlevels + tid]; s_off[tid] = lv[tid].kp_off; s_scale[tid] = lv[tid].scale; }
    if (tid == 0) s_off[nlevels] = kp_total_cap;
    int* hist = (int*)smem; int* cursor = hist + (nb + 1);
    if (row_start != nullptr) for (int i = tid; i <= nb; i += 256) hist[i] = 0;
    __syncthreads();
    // bucket of the stereo row index (below): first row of the keypoint's candidate band
    auto bucket_of = [&](uint32_t key, int l) {
        float yf = (float)(key_y(key) + kBorder);
        if (l != 0) yf = __fmul_rn(yf, s_scale[l]);
        const int first = (int)floorf(__fsub_rn(yf, __fmul_rn(2.0f, s_scale[l])));     // = aux.x of k_orient_brief
        return imin(imax(first, 0) >> kStereoRowShift, nb - 1);
    };
    int total = 0;
    for (int l = 0; l < nlevels; l++) total += s_cnt[l];
    int mono_run = 0, lap_run = 0;
    // (declared outside the loop: with the usual feature counts the loop runs once and the row index below places the keypoints from these
    // registers instead of reading keys and output indices back)
    int valid[kLayoutPer], lvl[kLayoutPer], fin[kLayoutPer];
    uint32_t key[kLayoutPer];
    for (int base = 0; base < kp_total_cap; base += 256 * kLayoutPer) {
        const int s0 = base + tid * kLayoutPer;
        int l = 0;
        while (l + 1 < nlevels && s_off[l + 1] <= s0) l++;
#pragma unroll
        for (int k = 0; k < kLayoutPer; k++) {
            const int s = s0 + k;
            while (l + 1 < nlevels && s_off[l + 1] <= s) l++;
            lvl[k] = l;
            valid[k] = s < kp_total_cap && s - s_off[l] < s_cnt[l];
            key[k] = valid[k] ? lvl_keys[(size_t)b * kp_total_cap + s] : 0u;
        }
        int lapf[kLayoutPer], nm = 0, nl = 0;
#pragma unroll
        for (int k = 0; k < kLayoutPer; k++) {
            float xf = (float)(key_x(key[k]) + kBorder);
            if (lvl[k] != 0) xf = __fmul_rn(xf, s_scale[lvl[k]]);
            lapf[k] = valid[k] && xf >= (float)lap0 && xf <= (float)lap1;
            nm += valid[k] && !lapf[k]; nl += lapf[k];
            if (row_start != nullptr && valid[k]) atomicAdd(&hist[bucket_of(key[k], lvl[k])], 1);
        }
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan<unsigned long long>((unsigned long long)nm | ((unsigned long long)nl << 32), &tot, s_scan);
        int pm = mono_run + (int)(ex & 0xFFFFFFFFu), pl = lap_run + (int)(ex >> 32);
#pragma unroll
        for (int k = 0; k < kLayoutPer; k++) {
            fin[k] = lapf[k] ? total - 1 - pl : pm;
            if (valid[k]) { final_idx[(size_t)b * kp_total_cap + s0 + k] = fin[k]; if (lapf[k]) pl++; else pm++; }
            // an unused slot is given one of the rows [total, cap) - the i-th unused slot row total + i - which k_orient_brief clears: descriptor
            // rows beyond the count read as zero (fixed-shape blocks for collectives) without a fill launch in front of every extraction
            else if (s0 + k < kp_total_cap) final_idx[(size_t)b * kp_total_cap + s0 + k] = total + (s0 + k) - (pm + pl);
        }
        mono_run += (int)(tot & 0xFFFFFFFFu); lap_run += (int)(tot >> 32);
    }
    if (tid == 0) { n_out[b] = total; mono_out[b] = mono_run; }
    // ---- row index for the stereo search (the role of vRowIndices, src/Frame.cc:1129-1155) ----
    // The keypoints of this image bucketed by the first row of their candidate band [floor(y - r), ceil(y + r)], r = 2 * scale
    // (1 << kStereoRowShift rows per bucket, CSR over the output indices).  A left keypoint at row v then only visits the buckets that can
    // hold bands covering v.  Order inside a bucket is arbitrary (atomics): the search reduces full (distance << 16 | index) keys, so the
    // visiting order never shows in the result.  The band only needs the key and the level, so it is built here, where both are at hand,
    // instead of by a launch of its own in front of the match (a launch is 4.5 us of a single pair's latency).
    if (row_start == nullptr) return;
    __syncthreads();                                     // the histogram (counted in the loop above) is complete
    int run = 0;
    for (int c0 = 0; c0 < nb; c0 += 256) {
        const int c = c0 + tid;
        const int v = c < nb ? hist[c] : 0;
        unsigned long long tot;
        const int ex = run + (int)block_excl_scan<unsigned long long>((unsigned long long)v, &tot, s_scan);
        if (c < nb) { cursor[c] = ex; row_start[(size_t)b * (nb + 1) + c] = ex; }
        run += (int)tot;
    }
    if (tid == 0) row_start[(size_t)b * (nb + 1) + nb] = run;
    __syncthreads();
    if (kp_total_cap <= 256 * kLayoutPer) {             // one trip of the loop above: keys, levels and output indices are still in registers
#pragma unroll
        for (int k = 0; k < kLayoutPer; k++)
            if (valid[k]) row_items[(size_t)b * kp_total_cap + atomicAdd(&cursor[bucket_of(key[k], lvl[k])], 1)] = fin[k];
        return;
    }
    int l = 0;
    for (int s = tid; s < kp_total_cap; s += 256) {
        while (l + 1 < nlevels && s_off[l + 1] <= s) l++;
        if (s - s_off[l] < s_cnt[l])
            row_items[(size_t)b * kp_total_cap + atomicAdd(&cursor[bucket_of(lvl_keys[(size_t)b * kp_total_cap + s], l)], 1)] = final_idx[(size_t)b * kp_total_cap + s];
    }
}

// grid (groups_per_image * 8 * ceil(B / 8)) with groups_per_image = ceil(kp_total_cap / (4 * kKpPerWave)), 256 threads; a wave owns kKpPerWave
// consecutive keypoint slots.
//   1  lane k prepares keypoint k (key, level, addresses, output index); the wave then walks the keypoints one at a time:
//      IC_Angle moments with all 64 lanes (lanes 0..30 rows v = 0,-1..-15, lanes 32..62 rows v = 1..15), the sums go to lane k
//   2  the per-keypoint transcendental work - fastAtan2 and the fp64 cosf/sinf model - runs once for the whole wave, one keypoint
//      per lane, instead of 64-fold redundantly per keypoint
//   3  steered BRIEF, again one keypoint at a time with all lanes (4 tests per lane, 4 ballots = 4 descriptor words)
//   4  lane k writes the record of keypoint k
constexpr int kKpPerWave = 8;                       // large batches; small ones use kKpPerWaveSmall (more, shorter waves: latency)
// window of the steered-BRIEF samples: rows / columns -18 .. +18 (the pattern's radius is 18.4), 10 dwords per row
constexpr int kWinR = 18, kWinRows = 2 * kWinR + 1, kWinDw = 10, kWinTrips = (kWinRows * kWinDw + 63) / 64;
static_assert(kKpPerWave == kKpPerWaveDecl, "orbx_kernels.h out of date");
constexpr int kKpPerWaveSmall = kKpPerWaveSmallDecl;
template <int KPW>
__device__ __forceinline__ void orient_brief_impl(const LevelInfo* __restrict__ lv, int nlevels,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t pyr_stride,
                                                      const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                      const int* __restrict__ lvl_count, const int* __restrict__ final_idx,
                                                      UmaxTab umax, KeyPointRec* __restrict__ out_kps,
                                                      unsigned long long* __restrict__ out_desc, int4* __restrict__ out_aux, int B, int groups_per_image) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][kWinRows * kWinDw * 4];
    // Workgroup -> (image, slot group).  Consecutive workgroup ids go to different XCDs (id % 8), each with its own L2; the keypoints of an
    // image read overlapping windows of the same pyramid / blur levels, so all workgroups of an image are kept on one XCD: groups of 8 images,
    // image = 8 * group + id % 8.  (With the plain (slot group, image) grid every XCD fetched every image: 470 MB per 128 images against 205 MB
    // of windows.)
    const int id = (int)blockIdx.x, nx = groups_per_image;
    const int b = 8 * (id / (8 * nx)) + (id & 7), gx = (id >> 3) % nx;
    if (b >= B) return;
    const int lane = lane_id();
    const int slot0 = (gx * 4 + (int)(threadIdx.x >> 6)) * KPW;
    if (slot0 >= kp_total_cap) return;
    // ---- 1a: per-lane keypoint state (lanes >= KPW mirror lane 0 and are never read) ----
    const int myslot = slot0 + (lane < KPW ? lane : 0);
    int my_valid = 0, my_level = 0, my_pitch = 0, my_fi = 0;
    uint32_t my_key = 0;
    int my_off = 0;                                         // byte offset of the keypoint centre inside the image's pyramid / blur block (< 2^31)
    if (myslot < kp_total_cap) {
        for (int l = 1; l < nlevels; l++) if (myslot >= lv[l].kp_off) my_level = l;
        const int i = myslot - lv[my_level].kp_off;
        my_valid = i < lvl_count[(size_t)b * nlevels + my_level];
        if (!my_valid && lane < KPW) {                      // unused slot: clear the descriptor row k_layout assigned to it
            const int zrow = final_idx[(size_t)b * kp_total_cap + myslot];
            unsigned long long* z = out_desc + ((size_t)b * kp_total_cap + zrow) * 4;
            z[0] = 0ull; z[1] = 0ull; z[2] = 0ull; z[3] = 0ull;
        }
        if (my_valid) {
            my_key = lvl_keys[(size_t)b * kp_total_cap + myslot];
            my_pitch = lv[my_level].pitch;
            my_off = (int)lv[my_level].off + (key_y(my_key) + kBorder) * my_pitch + (key_x(my_key) + kBorder);
            my_fi = final_idx[(size_t)b * kp_total_cap + myslot];
        }
    }
    const unsigned long long vmask = __ballot(my_valid && lane < KPW);
    if (vmask == 0ull) return;
    const BufRsrc raw0 = buf_make(pyr + (size_t)b * pyr_stride);
    const BufRsrc blur0 = buf_make(blur + (size_t)b * pyr_stride);
    // ---- 1b: IC_Angle ----
    // The 31 x 31 patch is read as dwords: lane = (row r8 = lane >> 3, column group c = lane & 7) covers columns u = -15 + 4c .. +3 of
    // rows v = -15 + 8 * trip + r8, four trips (the 32nd row / column carries weight 0), i.e. 4 load instructions per keypoint instead
    // of 16 byte gathers.  The moments are three v_dot4_u32_u8 per trip on unsigned weights: with w' = w + 16 inside the disc and 0
    // outside, sum(u * I) = dot(I, u') - 16 * dot(I, on).  Integer sums: the order of accumulation is irrelevant.
    const int r8 = lane >> 3, cg = lane & 7;
    uint32_t wu4[4], wv4[4], on4[4];
#pragma unroll
    for (int trip = 0; trip < 4; trip++) {
        const int v = -kHalfPatch + 8 * trip + r8;
        const int av = v < 0 ? -v : v;
        uint32_t pu = 0, pv = 0, po = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u = -kHalfPatch + 4 * cg + j;
            const int au = u < 0 ? -u : u;
            const bool on = av <= kHalfPatch && au <= kHalfPatch && au <= umax.u[av <= kHalfPatch ? av : 0];
            if (on) { pu |= (uint32_t)(u + 16) << (8 * j); pv |= (uint32_t)(v + 16) << (8 * j); po |= 1u << (8 * j); }
        }
        wu4[trip] = pu; wv4[trip] = pv; on4[trip] = po;
    }
    int my_m10 = 0, my_m01 = 0;
    for (int k = 0; k < KPW; k++) {
        if (!((vmask >> k) & 1ull)) continue;               // wave-uniform
        const int pitch = ORBX_READLANE(my_pitch, k);
        // scalar offset of the patch's first row / column + per-lane (row, dword) offset: buffer addressing, no 64-bit vector arithmetic
        const uint32_t so = (uint32_t)(ORBX_READLANE(my_off, k) - kHalfPatch * pitch - kHalfPatch);
        uint32_t I4[4];
#pragma unroll
        for (int trip = 0; trip < 4; trip++)                // issued back to back: one memory round trip
            I4[trip] = buf_load_u32(raw0, (uint32_t)(__mul24(8 * trip + r8, pitch) + 4 * cg), so);
        uint32_t du = 0, dv = 0, ds = 0;
#pragma unroll
        for (int trip = 0; trip < 4; trip++) { du = dot4_u8(I4[trip], wu4[trip], du); dv = dot4_u8(I4[trip], wv4[trip], dv); ds = dot4_u8(I4[trip], on4[trip], ds); }
        int m10 = (int)du - 16 * (int)ds, m01 = (int)dv - 16 * (int)ds;
        m10 = wave_sum(m10); m01 = wave_sum(m01);
        if (lane == k) { my_m10 = m10; my_m01 = m01; }
    }
    // ---- 2: angle, cos, sin: one keypoint per lane ----
    const float my_angle = fast_atan2_deg((float)my_m01, (float)my_m10);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float my_rad = __fmul_rn(my_angle, factorPI);
    const float my_a = glibc_cosf(my_rad), my_b = glibc_sinf(my_rad);
    // ---- 3: steered BRIEF on the blurred level ----
    // this lane's 4 tests (8 pattern points), the same for every keypoint
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const signed char* p = &BRIEF_PATTERN[4 * (64 * r + lane)];
        px0[r] = (float)p[0]; py0[r] = (float)p[1]; px1[r] = (float)p[2]; py1[r] = (float)p[3];
    }
    // The 512 sample points of a keypoint fall inside the 37 x 37 window around it (pattern radius 18.4).  The wave first copies that
    // window into its LDS slice with six row-coalesced dword loads per lane-trip (lanes of a row read 40 contiguous bytes) and then picks
    // the samples out of LDS: 512 scattered single-byte gathers from global memory per keypoint were what kept the texture path busy.
    uint8_t* win = s_win[threadIdx.x >> 6];
    for (int k = 0; k < KPW; k++) {
        if (!((vmask >> k) & 1ull)) continue;
        const int pitch = ORBX_READLANE(my_pitch, k);
        const uint32_t so = (uint32_t)(ORBX_READLANE(my_off, k) - kWinR * pitch - kWinR);      // first row / column of the window
        const float a = __int_as_float(ORBX_READLANE(__float_as_int(my_a), k)), bb = __int_as_float(ORBX_READLANE(__float_as_int(my_b), k));
        const int fi = ORBX_READLANE(my_fi, k);
        uint32_t wv[kWinTrips];
#pragma unroll
        for (int t = 0; t < kWinTrips; t++) {
            const int i = lane + 64 * t;                    // dword i of the window: row i / 10, dword column i % 10
            if (i < kWinRows * kWinDw) wv[t] = buf_load_u32(blur0, (uint32_t)(__mul24(i / kWinDw, pitch) + 4 * (i % kWinDw)), so);
        }
        ORBX_WAVE_SYNC();                                   // the previous keypoint's samples have been read
#pragma unroll
        for (int t = 0; t < kWinTrips; t++) { const int i = lane + 64 * t; if (i < kWinRows * kWinDw) ((uint32_t*)win)[i] = wv[t]; }
        ORBX_WAVE_SYNC();
        // cvRound(x) for |x| < 2^22: x + 1.5 * 2^23 is rounded (to nearest, ties to even) at integer granularity, and its bit pattern is
        // 0x4B400000 + round(x).  The low 24 bits (0x400000 + r) feed the 24-bit multiply by the row pitch directly; all the constants are
        // folded into one subtraction per sample: offset = 40 * (r + 18) + (c + 18).
        constexpr float kRoundMagic = 12582912.0f;
        constexpr uint32_t kRoundFold = 0x400000u * (4u * kWinDw) + 0x4B400000u - (uint32_t)(kWinR * 4 * kWinDw + kWinR);
        int t0v[4], t1v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(px0[r], bb), __fmul_rn(py0[r], a)), kRoundMagic));
            const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(px0[r], a), __fmul_rn(py0[r], bb)), kRoundMagic));
            const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(px1[r], bb), __fmul_rn(py1[r], a)), kRoundMagic));
            const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(px1[r], a), __fmul_rn(py1[r], bb)), kRoundMagic));
            t0v[r] = win[(uint32_t)__umul24(r0, 4 * kWinDw) + c0 - kRoundFold];
            t1v[r] = win[(uint32_t)__umul24(r1, 4 * kWinDw) + c1 - kRoundFold];
        }
        unsigned long long mine = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned long long w = ORBX_BALLOT(t0v[r] < t1v[r]);
            if (lane == r) mine = w;
        }
        if (lane < 4) out_desc[((size_t)b * kp_total_cap + fi) * 4 + lane] = mine;
    }
    // ---- 4: records ----
    if (lane < KPW && my_valid) {
        const LevelInfo L = lv[my_level];
        KeyPointRec k;
        float xf = (float)(key_x(my_key) + kBorder), yf = (float)(key_y(my_key) + kBorder);
        if (my_level != 0) { xf = __fmul_rn(xf, L.scale); yf = __fmul_rn(yf, L.scale); }
        k.x = xf; k.y = yf; k.size = (float)L.patch; k.angle = my_angle; k.response = (float)key_s(my_key);
        k.octave = my_level; k.class_id = -1;
        out_kps[(size_t)b * kp_total_cap + my_fi] = k;
        // compact 16-byte record for the stereo row search (Frame::ComputeStereoMatches, src/Frame.cc:1141-1155): the band of
        // image rows [floor(y - r), ceil(y + r)], r = 2 * scale, in which this keypoint is a candidate; x; octave
        const float r = __fmul_rn(2.0f, L.scale);
        int4 aux; aux.x = (int)floorf(__fsub_rn(yf, r)); aux.y = (int)ceilf(__fadd_rn(yf, r)); aux.z = __float_as_int(xf); aux.w = my_level;
        out_aux[(size_t)b * kp_total_cap + my_fi] = aux;
    }
}

__global__ void __launch_bounds__(256) k_orient_brief(const LevelInfo* __restrict__ lv, int nlevels,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t pyr_stride,
                                                      const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                      const int* __restrict__ lvl_count, const int* __restrict__ final_idx,
                                                      UmaxTab umax, KeyPointRec* __restrict__ out_kps,
                                                      unsigned long long* __restrict__ out_desc, int4* __restrict__ out_aux, int B, int groups_per_image) {
    orient_brief_impl<kKpPerWave>(lv, nlevels, pyr, blur, pyr_stride, lvl_keys, kp_total_cap, lvl_count, final_idx, umax, out_kps, out_desc, out_aux, B, groups_per_image);
}
// the same with fewer keypoints per wave: four times as many, shorter waves (small batches, where one wave's chain of keypoints is the stage time)
__global__ void __launch_bounds__(256) k_orient_brief_small(const LevelInfo* __restrict__ lv, int nlevels,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t pyr_stride,
                                                      const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                      const int* __restrict__ lvl_count, const int* __restrict__ final_idx,
                                                      UmaxTab umax, KeyPointRec* __restrict__ out_kps,
                                                      unsigned long long* __restrict__ out_desc, int4* __restrict__ out_aux, int B, int groups_per_image) {
    orient_brief_impl<kKpPerWaveSmall>(lv, nlevels, pyr, blur, pyr_stride, lvl_keys, kp_total_cap, lvl_count, final_idx, umax, out_kps, out_desc, out_aux, B, groups_per_image);
}

// Frame::UndistortKeyPoints (src/Frame.cc:1003-1034) for the keypoints of a batch: the records of `kps` with x, y replaced by cv::undistortPoints'
// result (undistort_model.h), everything else copied.  grid (ceil(cap / 256), B).
__global__ void __launch_bounds__(256) k_undistort(const KeyPointRec* __restrict__ kps, const int* __restrict__ n_per_frame, int cap, UndistortParams U,
                                                   KeyPointRec* __restrict__ kps_un) {
    const size_t b = blockIdx.y;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= cap || i >= n_per_frame[b]) return;
    KeyPointRec k = kps[b * (size_t)cap + i];
    float x, y;
    undistort_point(U, k.x, k.y, &x, &y);
    k.x = x; k.y = y;
    kps_un[b * (size_t)cap + i] = k;
}

}  // namespace orbx
