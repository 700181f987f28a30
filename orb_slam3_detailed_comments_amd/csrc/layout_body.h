// layout_body.h — final output index of every keypoint of one image (level scaling + lapping-area front / back) and the row index of its keypoints for
// the stereo search: the body of k_layout (k_describe.hip), which small batches run at the end of k_quadtree instead (the workgroup that finishes
// an image's last tree; k_quadtree.hip).
#pragma once
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"

namespace orbx {

// One workgroup of nt threads (a multiple of 64, all of them alive) per image b.  The level-ordered keypoint slots of an image (level l owns [kp_off, kp_off + kp_cap), the first lvl_count of them
// are in use) are walked in slot order = output order of the reference: a thread takes kLayoutPer consecutive slots, so that one workgroup
// scan per nt * kLayoutPer slots (one in all for the usual feature counts) places every keypoint, with all key loads in flight at once.
constexpr int kLayoutPer = 8;
// smem: (nb + 1) ints of histogram and as many of cursors (the row index).
__device__ __forceinline__ void layout_body(const LevelInfo* __restrict__ lv, int nlevels, const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                            const int* __restrict__ lvl_count, int lap0, int lap1, int* __restrict__ final_idx, int* __restrict__ n_out,
                                            int* __restrict__ mono_out, int nb, int* __restrict__ row_start, int* __restrict__ row_items,
                                            int b, int nt, uint8_t* smem) {
    __shared__ unsigned long long s_scan[20];
    __shared__ int s_cnt[kMaxLevels], s_off[kMaxLevels + 1];
    __shared__ float s_scale[kMaxLevels];
    const int tid = (int)threadIdx.x, nw = nt >> 6;
    if (tid < nlevels) { s_cnt[tid] = lvl_count[(size_t)b * nlevels + tid]; s_off[tid] = lv[tid].kp_off; s_scale[tid] = lv[tid].scale; }
    if (tid == 0) s_off[nlevels] = kp_total_cap;
    int* hist = (int*)smem; int* cursor = hist + (nb + 1);
    if (row_start != nullptr) for (int i = tid; i <= nb; i += nt) hist[i] = 0;
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
    for (int base = 0; base < kp_total_cap; base += nt * kLayoutPer) {
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
        const unsigned long long ex = block_excl_scan_n<unsigned long long>((unsigned long long)nm | ((unsigned long long)nl << 32), &tot, s_scan, nw);
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
    for (int c0 = 0; c0 < nb; c0 += nt) {
        const int c = c0 + tid;
        const int v = c < nb ? hist[c] : 0;
        unsigned long long tot;
        const int ex = run + (int)block_excl_scan_n<unsigned long long>((unsigned long long)v, &tot, s_scan, nw);
        if (c < nb) { cursor[c] = ex; row_start[(size_t)b * (nb + 1) + c] = ex; }
        run += (int)tot;
    }
    if (tid == 0) row_start[(size_t)b * (nb + 1) + nb] = run;
    __syncthreads();
    if (kp_total_cap <= nt * kLayoutPer) {             // one trip of the loop above: keys, levels and output indices are still in registers
#pragma unroll
        for (int k = 0; k < kLayoutPer; k++)
            if (valid[k]) row_items[(size_t)b * kp_total_cap + atomicAdd(&cursor[bucket_of(key[k], lvl[k])], 1)] = fin[k];
        return;
    }
    int l = 0;
    for (int s = tid; s < kp_total_cap; s += nt) {
        while (l + 1 < nlevels && s_off[l + 1] <= s) l++;
        if (s - s_off[l] < s_cnt[l])
            row_items[(size_t)b * kp_total_cap + atomicAdd(&cursor[bucket_of(lvl_keys[(size_t)b * kp_total_cap + s], l)], 1)] = final_idx[(size_t)b * kp_total_cap + s];
    }
}

}  // namespace orbx
