// k_quadtree.hip — the reference's quadtree keypoint distribution (ORBextractor::DistributeOctTree +
// ExtractorNode::DivideNode + compareNodes, src/ORBextractor.cc:602-697, :711-1057) as ONE workgroup
// per (image, level), bit-exact including list order and the unstable std::sort tie permutation.
//
// Formulation (proved equal to the reference by oracle/orb_oracle.cpp::quadtree, which is checked
// against the reference's own source):
//   * the std::list is an array in list order; a full pass replaces it by
//       [children of the LAST divided node ... children of the FIRST divided node] ++ [undivided nodes]
//     with each child block ordered n4,n3,n2,n1 (push_front order), empty children dropped;
//   * a node's keys are a contiguous span of a key buffer, children laid out n1|n2|n3|n4.  ONE stable counting sort by
//     (root, first D child digits) up front puts every node of depth <= D in that state at once (D <= 5, <= 1024 buckets,
//     bucket code = xpart[x] + ypart[y] from two LDS tables); dividing such a node is a difference of bucket offsets.
//     Deeper nodes are divided by *stable* 4-way partitions of their span (wave ballots; whole workgroup for huge spans);
//   * "first max response wins" (:1028-1053) refers to vKeys order = FAST emission order, which is a function of the key
//     (cell row, cell column, y, x) and is recomputed in the final selection;
//   * bNoMore <=> span length 1;
//   * the final rounds (:940-1020) sort (size, node) pairs with a workgroup-parallel model of libstdc++'s std::sort
//     (partition tree level by level, then a stable rank inside every final range of <= 16), then a prefix over
//     "non-empty children - 1" finds where `lNodes.size() >= N` breaks the loop.
// Key buffers A/B live in HBM (L2-resident); node lists, bucket offsets and scratch live in LDS.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "libstdcxx_sort_model.h"

namespace orbx {

#ifndef ORBX_BIGSPAN
#define ORBX_BIGSPAN 1024
#endif
#ifndef ORBX_PRESORT_U16_MAX
#define ORBX_PRESORT_U16_MAX 65535
#endif
constexpr int kPresortU16Max = ORBX_PRESORT_U16_MAX;     // levels with more keys count in 32-bit counters (tests lower it to force that path)
constexpr int kBigSpan = ORBX_BIGSPAN;   // spans above this are partitioned by the whole workgroup (tests also build with 80)

struct SortLess {
    __device__ __forceinline__ bool operator()(unsigned long long a, unsigned long long b) const { return (a >> 16) < (b >> 16); }
};

__device__ __forceinline__ int node_cnt(const QNode& n) { return (int)(n.cnt_buf & 0x3FFFFFFFu); }
__device__ __forceinline__ int node_buf(const QNode& n) { return (int)((n.cnt_buf >> 30) & 1u); }

// Stable 4-way partition of one node's span by one wave: src span -> dst span (same offsets), children
// laid out n1|n2|n3|n4.  Returns the 4 child counts (valid in every lane).
__device__ __forceinline__ void wave_partition(const QNode nd, const uint32_t* __restrict__ src,
                                               uint32_t* __restrict__ dst, int cnt[4]) {
    const int lane = lane_id();
    const int c = node_cnt(nd);
    const int s = (int)nd.start;
    // halfX = ceil(float(w)/2) (src/ORBextractor.cc:608-609); exact in integers for w >= 0
    const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
    const int my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (c <= 64) {
        uint32_t key = 0; int q = -1;
        if (lane < c) {
            key = src[s + lane];
            const bool left = key_x(key) < mx, top = key_y(key) < my;
            q = left ? (top ? 0 : 2) : (top ? 1 : 3);
        }
        const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
        cnt[0] = __popcll(b0); cnt[1] = __popcll(b1); cnt[2] = __popcll(b2); cnt[3] = __popcll(b3);
        if (q >= 0) {
            const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
            const int base = q == 0 ? 0 : q == 1 ? cnt[0] : q == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2];
            dst[s + base + __popcll(bq & lt)] = key;
        }
        return;
    }
    int tot[4] = {0, 0, 0, 0};
    for (int i0 = 0; i0 < c; i0 += 64) {
        int q = -1;
        if (i0 + lane < c) {
            const uint32_t key = src[s + i0 + lane];
            const bool left = key_x(key) < mx, top = key_y(key) < my;
            q = left ? (top ? 0 : 2) : (top ? 1 : 3);
        }
        tot[0] += __popcll(__ballot(q == 0)); tot[1] += __popcll(__ballot(q == 1));
        tot[2] += __popcll(__ballot(q == 2)); tot[3] += __popcll(__ballot(q == 3));
    }
    int run[4] = {0, tot[0], tot[0] + tot[1], tot[0] + tot[1] + tot[2]};
    for (int i0 = 0; i0 < c; i0 += 64) {
        int q = -1; uint32_t key = 0;
        if (i0 + lane < c) {
            key = src[s + i0 + lane];
            const bool left = key_x(key) < mx, top = key_y(key) < my;
            q = left ? (top ? 0 : 2) : (top ? 1 : 3);
        }
        const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
        if (q >= 0) {
            const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
            dst[s + run[q] + __popcll(bq & lt)] = key;
        }
        run[0] += __popcll(b0); run[1] += __popcll(b1); run[2] += __popcll(b2); run[3] += __popcll(b3);
    }
    cnt[0] = tot[0]; cnt[1] = tot[1]; cnt[2] = tot[2]; cnt[3] = tot[3];
}


// Block-cooperative stable partition of the span [s, s+c) into <= 4 classes (src -> dst, same offsets): every wave owns a
// contiguous 1/nw of the span, counts its classes with ballots, the waves exchange counts through LDS once, then
// each wave writes its keys at its ordered offsets.  Two barriers per call instead of two per 256 keys.
// cls(key) in [0,4).  s_cnt: 4 << lgnw ints of LDS.  All (64 << lgnw) threads must call.  cnt[] = class totals (valid in every thread).
template <typename F>
__device__ __forceinline__ void block_partition4(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int s, int c,
                                                 F cls, int* s_cnt, int cnt[4], int lgnw) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = 1 << lgnw;
    const int seg = ((c + (64 << lgnw) - 1) >> (6 + lgnw)) << 6;
    const int beg = wave * seg, end = imin(c, beg + seg);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int tot[4] = {0, 0, 0, 0};
    for (int i0 = beg; i0 < end; i0 += 256) {          // 4 chunks per trip: the four loads are in flight together
        uint32_t key[4]; int q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u + lane; key[u] = i < end ? src[s + i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            q[u] = (i0 + 64 * u + lane < end) ? cls(key[u]) : -1;
            tot[0] += __popcll(__ballot(q[u] == 0)); tot[1] += __popcll(__ballot(q[u] == 1));
            tot[2] += __popcll(__ballot(q[u] == 2)); tot[3] += __popcll(__ballot(q[u] == 3));
        }
    }
    if (lane < 4) s_cnt[wave * 4 + lane] = lane == 0 ? tot[0] : lane == 1 ? tot[1] : lane == 2 ? tot[2] : tot[3];
    __syncthreads();
    int run[4], acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int before = 0, total = 0;
        for (int w = 0; w < nw; w++) { const int v = s_cnt[w * 4 + k]; total += v; before += w < wave ? v : 0; }
        cnt[k] = total; run[k] = acc + before; acc += total;
    }
    for (int i0 = beg; i0 < end; i0 += 256) {
        uint32_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u + lane; key[u] = i < end ? src[s + i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = (i0 + 64 * u + lane < end) ? cls(key[u]) : -1;
            const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
            if (q >= 0) {
                const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                dst[s + run[q] + __popcll(bq & lt)] = key[u];
            }
            run[0] += __popcll(b0); run[1] += __popcll(b1); run[2] += __popcll(b2); run[3] += __popcll(b3);
        }
    }
    __syncthreads();
}

struct QuadCls {    // DivideNode's key -> child test (:651-661)
    int mx, my;
    __device__ __forceinline__ int operator()(uint32_t key) const {
        const bool left = key_x(key) < mx, top = key_y(key) < my;
        return left ? (top ? 0 : 2) : (top ? 1 : 3);
    }
};


#ifndef ORBX_WAVE_SORT_RANGE
#define ORBX_WAVE_SORT_RANGE 16
#endif
constexpr int kWaveSortRange = ORBX_WAVE_SORT_RANGE;     // ranges longer than this are partitioned by a wave instead of one thread

// Lanes of one wave hand values to each other through the arrays of the tree.  In LDS the wave's accesses are performed in issue order and the
// scheduling barrier is all that is needed; when the arrays live in the global node pool (kSpill) the compiler is also told that the hand-over is a
// release / acquire at wavefront scope (no instruction: a wave's vector memory accesses go through one L1 in issue order).
template <bool kSpill>
__device__ __forceinline__ void wave_sync_mem() {
#ifdef ORBX_EMU
    ORBX_WAVE_SYNC();
#else
    if (kSpill) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    ORBX_WAVE_SYNC();
    if (kSpill) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// A pending range of the sort: first | last | depth budget.  The LDS form of the tree holds fewer than 4096 nodes (12-bit positions in a dword);
// the node-pool form takes up to 65 535 (20-bit positions in a qword).
template <bool kWide> struct SortSeg;
template <> struct SortSeg<false> {
    typedef uint32_t T;
    static __device__ __forceinline__ T pack(int first, int last, int depth) { return (uint32_t)first | ((uint32_t)last << 12) | ((uint32_t)depth << 24); }
    static __device__ __forceinline__ int first(T s) { return (int)(s & 0xFFF); }
    static __device__ __forceinline__ int last(T s) { return (int)((s >> 12) & 0xFFF); }
    static __device__ __forceinline__ int depth(T s) { return (int)(s >> 24); }
};
template <> struct SortSeg<true> {
    typedef unsigned long long T;
    static __device__ __forceinline__ T pack(int first, int last, int depth) { return (T)first | ((T)last << 20) | ((T)depth << 40); }
    static __device__ __forceinline__ int first(T s) { return (int)(s & 0xFFFFF); }
    static __device__ __forceinline__ int last(T s) { return (int)((s >> 20) & 0xFFFFF); }
    static __device__ __forceinline__ int depth(T s) { return (int)(s >> 40); }
};

// libstdc++'s __unguarded_partition (sm_unguarded_partition) of a[lo, hi) around a[pivot] by one wave, same result and same array state.
// The sequential loop alternates "first walks up to the next element >= pivot" and "last walks down to the next element <= pivot" and
// swaps the two while first < last.  A walk never revisits a swapped position except the most recent one, so with
//   L_0 < L_1 < ..   the positions of the elements >= pivot (initial values),     R_0 > R_1 > ..  those of the elements <= pivot,
// swap k exchanges a[L_k] and a[R_k], K = #{k : L_k < R_k} swaps happen (the predicate is monotone in k), and the walk of `first` that
// ends the loop stops at L_K or at R_(K-1) (which holds an element >= pivot since swap K-1), whichever comes first.
// scratch: 4 * (hi - lo) uint16.  Returns the cut in every lane.
template <bool kSpill>
__device__ __forceinline__ int wave_unguarded_partition(unsigned long long* a, int lo, int hi, int pivot, uint16_t* scratch) {
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    SortLess less;
    const unsigned long long pv = a[pivot];
    const int m = hi - lo;
    uint16_t* Lpos = scratch;
    uint16_t* Rasc = scratch + 2 * m;        // positions of the elements <= pivot in ascending order: R_k = Rasc[nR - 1 - k]
    int nL = 0, nR = 0;
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int i = lo + c0 + lane;
        const bool valid = c0 + lane < m;
        const unsigned long long v = valid ? a[i] : 0ull;
        const bool isL = valid && !less(v, pv), isR = valid && !less(pv, v);
        const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
        if (isL) Lpos[nL + __popcll(bL & lt)] = (uint16_t)i;
        if (isR) Rasc[nR + __popcll(bR & lt)] = (uint16_t)i;
        nL += __popcll(bL); nR += __popcll(bR);
    }
    wave_sync_mem<kSpill>();
    const int mn = nL < nR ? nL : nR;
    int K = 0;
    for (int k0 = 0; k0 < mn; k0 += 64) {
        const int k = k0 + lane;
        int pl = 0, pr = 0;
        bool ok = false;
        if (k < mn) { pl = Lpos[k]; pr = Rasc[nR - 1 - k]; ok = pl < pr; }
        const unsigned long long bal = __ballot(ok);
        if (ok) { const unsigned long long t = a[pl]; a[pl] = a[pr]; a[pr] = t; }
        K += __popcll(bal);
        if (bal != ~0ull) break;
    }
    wave_sync_mem<kSpill>();
    int cut = 0x7FFFFFFF;
    if (K < nL) cut = Lpos[K];
    if (K >= 1) { const int r = Rasc[nR - K]; cut = r < cut ? r : cut; }
    return cut;
}

// std::sort(a, a+n, SortLess) by the whole workgroup, same result as libstdc++ (libstdcxx_sort_model.h) but with the
// independent work run in parallel:
//  * introsort's partition tree: the sub-ranges produced by one __unguarded_partition_pivot are disjoint, so each pending
//    range is partitioned by its own thread, level by level (a range whose depth budget is exhausted is heap-sorted by
//    its thread, exactly like the recursion would);
//  * __final_insertion_sort never moves an element across a partition cut (left of a cut is <= pivot <= right of it) and
//    insertion sort is stable, so it equals a stable rank-sort inside every final range of <= 16 elements: each thread
//    ranks one element among its <= 16 range-mates.
// seg0/seg1: two range lists (SortSeg: first | last | depth), capacity >= n/8 + 8 each; flags: n bytes
// (1 = a final range starts here, 2 = a heap-sorted range starts here); tmp: n elements.  All NT threads of the workgroup must call.
template <bool kSpill>
__device__ __forceinline__ void block_sort_libstdcxx(unsigned long long* a, unsigned long long* tmp, int n,
                                                     typename SortSeg<kSpill>::T* seg0, typename SortSeg<kSpill>::T* seg1, uint8_t* flags, int* s_ctr, int NT) {
    typedef SortSeg<kSpill> Seg;
    typedef typename Seg::T SegT;
    const int tid = (int)threadIdx.x;
    SortLess less;
    for (int i = tid; i < n; i += NT) flags[i] = 0;
    if (tid == 0) {
        int lg = 0;
        for (int t = n; t > 1; t >>= 1) lg++;
        flags[0] = 1;
        s_ctr[0] = 0; s_ctr[1] = 0;
        if (n > 16) { seg0[0] = Seg::pack(0, n, lg * 2); s_ctr[0] = 1; }
    }
    __syncthreads();
    SegT* cur = seg0; SegT* nxt = seg1;
    int which = 0;
    const int lane = lane_id(), wave = tid >> 6, nwaves = NT >> 6;
    for (;;) {
        const int ncur = s_ctr[which];
        if (ncur == 0) break;
        // long ranges (the first levels of the partition tree): one wave each, wave_unguarded_partition
        for (int j = wave; j < ncur; j += nwaves) {
            const SegT sg = cur[j];
            const int first = Seg::first(sg), last = Seg::last(sg);
            int depth = Seg::depth(sg);
            if (depth == 0 || last - first <= kWaveSortRange) continue;
            --depth;
            if (lane == 0) sm_move_median_to_first(a, first, first + 1, first + (last - first) / 2, last - 1, less);
            wave_sync_mem<kSpill>();
            const int cut = wave_unguarded_partition<kSpill>(a, first + 1, last, first, (uint16_t*)(tmp + first));
            if (lane == 0) {
                if (last - cut > 16) nxt[atomicAdd(&s_ctr[which ^ 1], 1)] = Seg::pack(cut, last, depth);
                if (cut < last) flags[cut] = flags[cut] ? flags[cut] : 1;
                if (cut - first > 16) nxt[atomicAdd(&s_ctr[which ^ 1], 1)] = Seg::pack(first, cut, depth);
            }
        }
        for (int j = tid; j < ncur; j += NT) {
            const SegT sg = cur[j];
            const int first = Seg::first(sg), last = Seg::last(sg);
            int depth = Seg::depth(sg);
            if (depth == 0) {
                sm_heap_sort(a, first, last, less);
                flags[first] = 2;
            } else if (last - first <= kWaveSortRange) {
                --depth;
                const int mid = first + (last - first) / 2;
                sm_move_median_to_first(a, first, first + 1, mid, last - 1, less);
                const int cut = sm_unguarded_partition(a, first + 1, last, first, less);
                // right range [cut, last), left range [first, cut): both continue with the decremented budget
                if (last - cut > 16) nxt[atomicAdd(&s_ctr[which ^ 1], 1)] = Seg::pack(cut, last, depth);
                if (cut < last) flags[cut] = flags[cut] ? flags[cut] : 1;
                if (cut - first > 16) nxt[atomicAdd(&s_ctr[which ^ 1], 1)] = Seg::pack(first, cut, depth);
            }
        }
        __syncthreads();
        if (tid == 0) s_ctr[which] = 0;
        which ^= 1;
        { SegT* t = cur; cur = nxt; nxt = t; }
        __syncthreads();
    }
    // final insertion sort == stable rank inside each final range
    for (int i = tid; i < n; i += NT) {
        int f = i;
        while (flags[f] == 0) --f;
        const unsigned long long v = a[i];
        int pos = i;
        if (flags[f] == 1) {
            int e = i + 1;
            while (e < n && flags[e] == 0) ++e;
            int rank = 0;
            for (int j = f; j < e; j++) {
                const unsigned long long o = a[j];
                rank += (less(o, v) || (j < i && !less(v, o))) ? 1 : 0;
            }
            pos = f + rank;
        }
        tmp[pos] = v;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) a[i] = tmp[i];
    __syncthreads();
}


// One wave partitions the nodes idx_of(j), j = wave, wave+nw, ... < count, skipping spans of <= 1 key and spans above
// kBigSpan (done cooperatively elsewhere).  Spans of <= 64 keys (the common case) are software-pipelined: the keys of
// the next node are requested before the ballots of the current one, hiding most of the L2 latency.
template <typename IdxFn>
__device__ __forceinline__ void wave_partition_many(int count, IdxFn idx_of, const QNode* __restrict__ cur,
                                                    uint32_t* __restrict__ bufA, uint32_t* __restrict__ bufB,
                                                    uint32_t* __restrict__ childcnt, int D, int nw) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int j = wave;
    int idx = 0; QNode nd; nd.x0 = nd.y0 = nd.x1 = nd.y1 = 0; nd.start = 0; nd.cnt_buf = 0; nd.code = 0; nd.depth = 0;
    uint32_t key[4] = {0, 0, 0, 0};
    if (j < count) {
        idx = idx_of(j); nd = cur[idx];
        const int c = node_cnt(nd);
        if (c > 1 && c <= 256 && (int)nd.depth >= D) {
            const uint32_t* sp = (node_buf(nd) ? bufB : bufA) + nd.start;
#pragma unroll
            for (int u = 0; u < 4; u++) if (64 * u + lane < c) key[u] = sp[64 * u + lane];
        }
    }
    while (j < count) {
        const int jn = j + nw;
        int idxn = 0; QNode ndn = nd; uint32_t keyn[4] = {0, 0, 0, 0};
        if (jn < count) {           // request the next node's keys before working on this one
            idxn = idx_of(jn); ndn = cur[idxn];
            const int cn = node_cnt(ndn);
            if (cn > 1 && cn <= 256 && (int)ndn.depth >= D) {
                const uint32_t* sp = (node_buf(ndn) ? bufB : bufA) + ndn.start;
#pragma unroll
                for (int u = 0; u < 4; u++) if (64 * u + lane < cn) keyn[u] = sp[64 * u + lane];
            }
        }
        const int c = node_cnt(nd);
        if (c > 1 && c <= kBigSpan && (int)nd.depth >= D) {
            int cnt[4];
            const int bsel = node_buf(nd);
            if (c <= 256) {
                const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                unsigned long long bal[4][4];
                int q[4];
                cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    q[u] = -1;
                    if (64 * u + lane < c) { const bool left = key_x(key[u]) < mx, top = key_y(key[u]) < my; q[u] = left ? (top ? 0 : 2) : (top ? 1 : 3); }
                    if (64 * u < c) {
#pragma unroll
                        for (int k = 0; k < 4; k++) { bal[u][k] = __ballot(q[u] == k); cnt[k] += __popcll(bal[u][k]); }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) bal[u][k] = 0;
                    }
                }
                uint32_t* dp = (bsel ? bufA : bufB) + nd.start;
                int run[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (q[u] >= 0) {
                        const int k = q[u];
                        const unsigned long long bq = k == 0 ? bal[u][0] : k == 1 ? bal[u][1] : k == 2 ? bal[u][2] : bal[u][3];
                        const int base = k == 0 ? run[0] : k == 1 ? run[1] : k == 2 ? run[2] : run[3];
                        dp[base + __popcll(bq & lt)] = key[u];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) run[k] += __popcll(bal[u][k]);
                }
            } else {
                wave_partition(nd, bsel ? bufB : bufA, bsel ? bufA : bufB, cnt);
            }
            if (lane < 4) childcnt[4 * idx + lane] = (uint32_t)(lane == 0 ? cnt[0] : lane == 1 ? cnt[1] : lane == 2 ? cnt[2] : cnt[3]);
        }
        j = jn; idx = idxn; nd = ndn;
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = keyn[u];
    }
}
constexpr int kDeepSmall = 1, kDeepBig = 2;       // a node below the presorted depths with > 1 key: one wave partitions it / the workgroup does
struct IdentityIdx { __device__ __forceinline__ int operator()(int j) const { return j; } };
struct ExpandIdx { const unsigned long long* e; __device__ __forceinline__ int operator()(int j) const { return (int)(e[j] & 0xFFFF); } };

__device__ __forceinline__ QNode make_child(const QNode& p, int q, const int cnt[4], int newbuf) {
    const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1);
    const int my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
    QNode c;
    c.x0 = (q & 1) ? (int16_t)mx : p.x0;  c.x1 = (q & 1) ? p.x1 : (int16_t)mx;
    c.y0 = (q & 2) ? (int16_t)my : p.y0;  c.y1 = (q & 2) ? p.y1 : (int16_t)my;
    int off = 0;
    for (int k = 0; k < q; k++) off += cnt[k];
    c.start = p.start + (uint32_t)off;
    c.cnt_buf = (uint32_t)cnt[q] | ((uint32_t)newbuf << 30);
    c.code = p.code * 4u + (uint32_t)q;
    c.depth = p.depth + 1u;
    return c;
}
// Child counts of a node whose span is still ordered by its next child digit (depth < presort depth D): differences of
// the bucket offsets of the up-front counting sort, no key is touched.  bucket id = code digits padded to D digits.
__device__ __forceinline__ void presorted_child_counts(const QNode& nd, int D, const int* __restrict__ bucket_start, int cnt[4]) {
    const int sh = 2 * (D - (int)nd.depth - 1);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c0 = (int)((nd.code * 4u + (uint32_t)q) << sh), c1 = (int)((nd.code * 4u + (uint32_t)q + 1u) << sh);
        cnt[q] = bucket_start[c1] - bucket_start[c0];
    }
}

// position of a candidate in the reference's vToDistributeKeys: FAST cells row-major (:1097-1166), row-major inside a cell.
// Mw, Mh = (1 << 20) / cell size + 1: v / d == (v * M) >> 20 exactly for v * d < 2^20 (coordinates < 4096, cells <= 240 px: orbx_api.cpp) - the
// two integer divisions were most of the selection phase (a workgroup of 1024 threads pays ~7 ns per instruction per thread)
__device__ __forceinline__ unsigned long long vkeys_order(uint32_t key, unsigned Mw, unsigned Mh) {
    const int x = key_x(key), y = key_y(key);
    const unsigned long long ci = (unsigned long long)(((unsigned)(y - 3) * Mh) >> 20), cj = (unsigned long long)(((unsigned)(x - 3) * Mw) >> 20);
    return (ci << 36) | (cj << 24) | ((unsigned long long)y << 12) | (unsigned long long)x;
}

// One full pass (src/ORBextractor.cc:790-905) over <= 64 nodes whose divisible members all sit at a presorted depth, by one wave: lane i owns
// node i, child counts come from the bucket offsets, the three prefix sums (non-empty children, children with > 1 key, kept nodes) are
// wave scans.  Same placement as the workgroup pass: blocks of children in reverse node order, n4..n1 inside a block, kept nodes behind.
// res: {T, E, K, overflow}.
__device__ __forceinline__ void solo_full_pass(const QNode* __restrict__ cur, QNode* __restrict__ nxt, unsigned long long* __restrict__ expn, int nnodes,
                                               int D, const int* __restrict__ bucket_start, int node_cap, int* res) {
    const int lane = lane_id();
    QNode nd; nd.x0 = nd.y0 = nd.x1 = nd.y1 = 0; nd.start = 0; nd.cnt_buf = 0; nd.code = 0; nd.depth = 0;
    int c = 0, cq[4] = {0, 0, 0, 0}, m = 0, e = 0;
    unsigned long long v = 0;
    if (lane < nnodes) {
        nd = cur[lane]; c = node_cnt(nd);
        if (c > 1) {
            presorted_child_counts(nd, D, bucket_start, cq);
            for (int q = 0; q < 4; q++) { m += cq[q] > 0; e += cq[q] > 1; }
            v = (unsigned long long)m | ((unsigned long long)e << 20);
        } else v = 1ull << 40;
    }
    const unsigned long long incl = wave_incl_scan(v), tot = __shfl(incl, 63), ex = incl - v;
    const int T = (int)(tot & 0xFFFFF), E = (int)((tot >> 20) & 0xFFFFF), K = (int)(tot >> 40);
    const int overflow = T + K > node_cap;
    if (!overflow && lane < nnodes) {
        if (c > 1) {
            const int Pm = (int)(ex & 0xFFFFF), Pe = (int)((ex >> 20) & 0xFFFFF);
            const int newbuf = node_buf(nd);                      // count-only divisions move no key
            int after = 0, erank = 0;
            for (int q = 3; q >= 0; q--)
                if (cq[q] > 0) { nxt[T - Pm - m + after] = make_child(nd, q, cq, newbuf); after++; }
            for (int q = 0; q < 4; q++) {
                if (cq[q] > 1) {
                    int pos_in_block = 0;                         // position of child q inside the block = number of non-empty children with q' > q
                    for (int q2 = q + 1; q2 < 4; q2++) pos_in_block += cq[q2] > 0;
                    const int mxx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
                    const int cx0 = (q & 1) ? mxx : (int)nd.x0;
                    expn[Pe + erank] = ((unsigned long long)cq[q] << 32) | ((unsigned long long)(uint16_t)cx0 << 16) | (unsigned long long)(T - Pm - m + pos_in_block);
                    erank++;
                }
            }
        } else nxt[T + (int)(ex >> 40)] = nd;
    }
    if (lane == 0) { res[0] = T; res[1] = E; res[2] = K; res[3] = overflow; }
}

// The up-front stable counting sort of a level's keys (bufB -> bufA) by bucket code = xpart[x] + ypart[y]: the keys are cut into nseg
// contiguous segments of `seg` keys (whole 64-key chunks), wave w < nseg owns segment w.  counts[w][b] (CT = uint16_t when the level has
// < 65536 keys, else uint32_t with fewer segments in the same LDS) comes in as the segment's histogram - counted by the gather with LDS
// atomics, the order of the counts does not matter - and becomes the number of keys of bucket b in the segments before w, i.e. the wave's
// cursor relative to bucket_start[b].  The scatter finds the lanes of a 64-key chunk that share a bucket with a bit-wise match (ballots):
// rank inside the group = number of lower lanes in it.
template <typename CT>
__device__ __forceinline__ void presort_keys(const uint32_t* __restrict__ bufB, uint32_t* __restrict__ bufA, int n, int NB, int nseg, int seg, CT* counts,
                                             int* bucket_start, const uint16_t* xpart, const uint16_t* ypart, unsigned long long* s_scan, int NT) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = imin(n, wave * seg), end = wave < nseg ? imin(n, beg + seg) : beg;
    CT* mycount = counts + (wave < nseg ? wave : 0) * NB;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __syncthreads();                                               // the histogram (counted by the gather) is complete
    // exclusive scan over buckets of the totals
    int run = 0;
    for (int b0 = 0; b0 < NB; b0 += NT) {
        const int bk = b0 + tid;
        int tot_b = 0;
        if (bk < NB)
            for (int w = 0; w < nseg; w++) { const int c = (int)counts[w * NB + bk]; counts[w * NB + bk] = (CT)tot_b; tot_b += c; }
        unsigned long long tot;
        const int ex = run + (int)block_excl_scan_n<unsigned long long>((unsigned long long)tot_b, &tot, s_scan, NT >> 6);
        if (bk < NB) bucket_start[bk] = ex;
        run += (int)tot;
    }
    if (tid == 0) bucket_start[NB] = run;
    __syncthreads();
    // stable scatter: same match; rank inside the group = number of lower lanes in it
    // (the keys of the next four chunks are requested before the ballots of the current four: the chunks of a segment depend on each other
    // through the cursors, so nothing else hides the L2 latency of a wave's three or four trips)
    uint32_t next[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = beg + 64 * u + lane; next[u] = i < end ? bufB[i] : 0u; }
    for (int i0 = beg; i0 < end; i0 += 256) {
        uint32_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = next[u];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + 256 + 64 * u + lane; next[u] = i < end ? bufB[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = i0 + 64 * u + lane < end;
            const uint32_t code = in ? ((uint32_t)xpart[key_x(key[u])] + (uint32_t)ypart[key_y(key[u])]) : 0xFFFFFFFFu;
            // lanes with the same code: a match over the bits in which the codes of this chunk differ at all - its 64 keys are neighbours in
            // one or two cells, i.e. a handful of buckets that differ in two or three bits of the nine or ten
            unsigned long long same = __ballot(in);
            const uint32_t cref = (uint32_t)ORBX_READLANE((int)code, 0);                   // lane 0 is in whenever any lane is
            for (uint32_t diff = wave_or_u32(in ? code ^ cref : 0u); diff != 0u; diff &= diff - 1u) {
                const int bit = __ffsll((unsigned long long)diff) - 1;
                const unsigned long long bb = __ballot((code >> bit) & 1u);
                same &= ((code >> bit) & 1u) ? bb : ~bb;
            }
            int cursor = 0;
            if (in) { cursor = (int)mycount[code]; bufA[bucket_start[code] + cursor + __popcll(same & lt)] = key[u]; }
            ORBX_WAVE_SYNC();           // every lane of a group has read the cursor ...
            if (in && (same >> lane) <= 1ull) mycount[code] = (CT)(cursor + __popcll(same));     // ... before its highest lane advances it
            ORBX_WAVE_SYNC();
        }
    }
}

// One tree = one workgroup: kQuadtreeThreads threads of which the first L.qt_threads (256 or 1024, by the size of the level) work on the level
// and the rest exits at once.
// kSpill = false: node lists, child counts, expand lists and flags are carved from dynamic LDS (81 bytes per node) beside the bucket tables - the
// form every level of the reference's stereo / RGB-D settings takes.
// kSpill = true: the same arrays live in a per-tree slice of a global node pool (L2-resident: a level of 10 000 nodes is 0.8 MB) and LDS only
// holds the bucket tables; this is what levels take whose node lists exceed the LDS of the device - the monocular initialisation extractor,
// ORBextractor(5 * nFeatures, ..) (src/Tracking.cc:634-635, :1331-1332): 10 000 features at 1241 x 376 (KITTI), 7 500 at 512 x 512 (TUM-VI).
// The reference's std::list has no bound (src/ORBextractor.cc:711-1057), and neither has this form below the 65 535 keypoints per image of the
// 16-bit node indices.
template <bool kSpill>
__device__ __forceinline__ void quadtree_tree(unsigned char* smem, unsigned char* pool, const int level, const int b,
                                              const LevelInfo* __restrict__ lv,
                                              const CellInfo* __restrict__ cells, int ncells,
                                              const int* __restrict__ cell_count,
                                              const uint32_t* __restrict__ slots, size_t slots_stride,
                                              uint32_t* __restrict__ candA, uint32_t* __restrict__ candB, size_t cand_stride,
                                              uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                              int* __restrict__ lvl_count, int nlevels, int node_cap, int nb_cap, int lut_x, int lut_y,
                                              int* __restrict__ status, long long* __restrict__ qt_prof, int wide, int counter_bytes) {
    __shared__ unsigned long long s_scan[20];
    __shared__ int s_i[80];
    int* s_part = s_i;                                  // block_partition4: 4 counts per wave
    int* s_ndiv = s_i + 64;
    int* s_sortctr = s_i + 66;
    int* s_kinds = s_i + 68;                            // kinds of deep nodes met in the current pass (kDeepSmall | kDeepBig)
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LevelInfo L = lv[level];
    // wide: the large levels get L.qt_threads threads (small batches: the latency of one tree is what counts); otherwise every level runs on
    // four waves, which packs more trees on a CU (large batches)
    const int NT = wide ? L.qt_threads : 256, NW = NT >> 6, lgNW = NT == 256 ? 2 : NT == 512 ? 3 : 4;
    if (tid >= NT) return;
    const int N = L.quota;
#ifndef ORBX_QT_STAMP_LEVEL
#define ORBX_QT_STAMP_LEVEL 0      // the tree whose phases orbx_debug_quadtree_profile reports (experimental builds look at other levels)
#endif
#ifdef ORBX_EMU
#define QT_STAMP(i)
#else
#define QT_STAMP(i) if (qt_prof && tid == 0 && level == ORBX_QT_STAMP_LEVEL && b == 0) qt_prof[i] = wall_clock64();
#endif
    QT_STAMP(0)
    if (tid == 0) *s_kinds = 0;
    // carve: nodes[2][cap] (24 B) | childcnt[cap][4] (u32) | expand[2][cap] (u64) | erased[cap] (u8) - from LDS, or from this tree's slice of the
    // node pool (kSpill) - then, always in LDS: bucket_start[nb+1] | counts (counter_bytes per bucket: u16 or u32 per segment) | code LUTs
    unsigned char* node_mem = kSpill ? pool : smem;
    QNode* nodes0 = (QNode*)node_mem;
    QNode* nodes1 = nodes0 + node_cap;
    uint32_t* childcnt = (uint32_t*)(nodes1 + node_cap);
    unsigned long long* exp0 = (unsigned long long*)(childcnt + 4 * (size_t)node_cap);
    unsigned long long* exp1 = exp0 + node_cap;
    uint8_t* erased = (uint8_t*)(exp1 + node_cap);
    int* bucket_start = kSpill ? (int*)smem : (int*)(smem + (((size_t)node_cap * 81 + 15) & ~(size_t)15));
    int* counts = bucket_start + (nb_cap + 2);
    uint16_t* xpart = (uint16_t*)(counts + (size_t)(counter_bytes >> 2) * nb_cap);     // bucket code = xpart[x] + ypart[y]
    uint16_t* ypart = xpart + lut_x;
    const int D = L.presort_depth, NBr = 1 << (2 * D), NB = L.nini * NBr;     // buckets per root / in total
    uint32_t* bufA = candA + (size_t)b * cand_stride + L.cand_off;
    uint32_t* bufB = candB + (size_t)b * cand_stride + L.cand_off;
    const int* ccount = cell_count + (size_t)b * ncells + L.cell_begin;
    const uint32_t* slot_base = slots + (size_t)b * slots_stride;

    // ---- P0: gather the level's candidates in the reference order (cells row-major) into bufB, and count them per (sort segment, bucket) on
    // the way (the histogram pass of the counting sort below would read every key again) ----
    // bucket(key) = root index (int)(x / hX) (:763), then the D child digits DivideNode (:602-674) would assign on the way down.  The child
    // digit at every depth is (right ? 1 : 0) + (bottom ? 2 : 0): the x half depends only on x (and the root the x falls in), the y half only
    // on y, so the bucket code splits into two small LDS tables built once per workgroup.
    for (int x = tid; x < L.bw; x += NT) {
        const int r = __float2int_rz(__fdiv_rn((float)x, L.hX));
        int x0 = __float2int_rz(__fmul_rn(L.hX, (float)r)), x1 = __float2int_rz(__fmul_rn(L.hX, (float)(r + 1)));
        int code = r;
        for (int d = 0; d < D; d++) {
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            const bool left = x < mx;
            code = code * 4 + (left ? 0 : 1);
            if (left) x1 = mx; else x0 = mx;
        }
        xpart[x] = (uint16_t)code;
    }
    for (int y = tid; y < L.bh; y += NT) {
        int y0 = 0, y1 = L.bh, code = 0;
        for (int d = 0; d < D; d++) {
            const int my = y0 + ((y1 - y0 + 1) >> 1);
            const bool top = y < my;
            code = code * 4 + (top ? 0 : 2);
            if (top) y1 = my; else y0 = my;
        }
        ypart[y] = (uint16_t)code;
    }
    // Lanes per cell: as many as the workgroup has for every cell of the level at once (a power of two, at most a wave), so that the usual level
    // is gathered in ONE trip - counts and slot offsets in one round trip, the keys in a second one with up to 16 loads per lane in flight -
    // and a level of a few large cells (the top of the pyramid) still uses every lane.  Levels with more cells than threads take trips of NT cells.
    // (One lane per KEY with a search for its cell coalesces better and is slower: a workgroup of 1024 threads on one CU pays 6.7 ns per
    // instruction per thread, and the search is ~50 of them per key - profiles/r04/quadtree_gather_key_parallel_variant.patch.txt.)
    int lgt = 0;
    while (lgt < 6 && (L.cell_count << (lgt + 1)) <= NT) lgt++;
    const int tpc = 1 << lgt, sub = tid & (tpc - 1), cpt = NT >> lgt;          // lanes per cell, lane within the cell, cells per trip
    const bool one_trip = L.cell_count <= cpt;
    // number of keys first: it decides the counter width and the segment length
    int n = 0, cnt0 = 0, soff0 = 0, pos0 = 0;
    if (one_trip) {
        const int c = tid >> lgt;
        if (c < L.cell_count) { cnt0 = ccount[c]; soff0 = cells[L.cell_begin + c].slot_off; }      // (travels with the count: one round trip, not two)
        unsigned long long tot;
        pos0 = (int)block_excl_scan_n<unsigned long long>((unsigned long long)(sub == 0 ? cnt0 : 0), &tot, s_scan, NW);
        pos0 = __shfl(pos0, lane & ~(tpc - 1));
        n = (int)tot;
    } else {
        for (int c0 = 0; c0 < L.cell_count; c0 += NT) {
            const int c = c0 + tid;
            unsigned long long tot;
            (void)block_excl_scan_n<unsigned long long>((unsigned long long)(c < L.cell_count ? ccount[c] : 0), &tot, s_scan, NW);
            n += (int)tot;
        }
    }
    const bool narrow_counters = n <= kPresortU16Max;
    const int nseg = narrow_counters ? imin(NW, counter_bytes >> 1) : imin(NW, counter_bytes >> 2);
    const int seg = ((n + 64 * nseg - 1) / (64 * nseg)) << 6;     // keys per sort segment (whole 64-key chunks)
    for (int i = tid; i < (narrow_counters ? (nseg * NB + 1) >> 1 : nseg * NB); i += NT) counts[i] = 0;
    __syncthreads();
    {
        int run = 0;
        for (int c0 = 0; c0 < L.cell_count; c0 += cpt) {
            int cnt = cnt0, soff = soff0, pos = pos0;
            unsigned long long tot = 0;
            if (!one_trip) {
                const int c = c0 + (tid >> lgt);
                cnt = c < L.cell_count ? ccount[c] : 0;
                soff = c < L.cell_count ? cells[L.cell_begin + c].slot_off : 0;
                pos = run + (int)block_excl_scan_n<unsigned long long>((unsigned long long)(sub == 0 ? cnt : 0), &tot, s_scan, NW);
                pos = __shfl(pos, lane & ~(tpc - 1));
            }
            if (cnt > 0) {
                const uint32_t* sp = slot_base + soff;
                int sidx = (pos + sub) / seg, send = (sidx + 1) * seg;       // segment of this lane's first key, and where it ends
                auto count_key = [&](uint32_t key, int p) {
                    while (p >= send) { sidx++; send += seg; }
                    const int slot = sidx * NB + (int)xpart[key_x(key)] + (int)ypart[key_y(key)];
                    if (narrow_counters) atomicAdd((unsigned*)counts + (slot >> 1), 1u << (16 * (slot & 1)));
                    else atomicAdd(counts + slot, 1);
                };
                // (16-byte accesses - four keys per load and store - change nothing: what this loop waits for is the LDS pipe, 12 k table reads
                // and atomics per level-0 tree)
                for (int k = sub; k < cnt; k += 16 * tpc) {        // sixteen loads in flight, the tail of a cell included
                    uint32_t v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) if (k + u * tpc < cnt) v[u] = sp[k + u * tpc];
#pragma unroll
                    for (int u = 0; u < 16; u++) if (k + u * tpc < cnt) { bufB[pos + k + u * tpc] = v[u]; count_key(v[u], pos + k + u * tpc); }
                }
            }
            run += (int)tot;
        }
    }
    __syncthreads();
    QT_STAMP(1)
    // ---- roots + first D tree levels in ONE stable counting sort (bufB -> bufA) --------------------------------------
    // After the sort every node of depth <= D is a contiguous span, its children are laid out n1|n2|n3|n4 and keys keep
    // their vKeys order inside a bucket (stable), which is exactly the state D partition passes would have produced.
    if (narrow_counters) presort_keys<uint16_t>(bufB, bufA, n, NB, nseg, seg, (uint16_t*)counts, bucket_start, xpart, ypart, s_scan, NT);
    else presort_keys<uint32_t>(bufB, bufA, n, NB, nseg, seg, (uint32_t*)counts, bucket_start, xpart, ypart, s_scan, NT);
    __syncthreads();
    int nnodes = 0;
    for (int r = 0; r < L.nini; r++) {
        const int s0 = bucket_start[r * NBr], c = bucket_start[(r + 1) * NBr] - s0;
        if (c > 0) {
            if (tid == 0) {
                QNode nd;
                nd.x0 = (int16_t)__float2int_rz(__fmul_rn(L.hX, (float)r));
                nd.x1 = (int16_t)__float2int_rz(__fmul_rn(L.hX, (float)(r + 1)));
                nd.y0 = 0; nd.y1 = (int16_t)L.bh;
                nd.start = (uint32_t)s0; nd.cnt_buf = (uint32_t)c;   // buffer A = 0
                nd.code = (uint32_t)r; nd.depth = 0;
                nodes0[nnodes] = nd;
            }
            nnodes++;
        }
    }
    __syncthreads();

    QT_STAMP(2)
    QNode* cur = nodes0; QNode* nxt = nodes1;
    unsigned long long* expc = exp0; unsigned long long* expn = exp1;
    bool finish = (n == 0);
    int overflow = 0;
    int passno = 0;
    while (!finish) {
        const int prevSize = nnodes;
        int T = 0, E = 0, K = 0;
        // ---- full pass: divide every node with more than one key (:790-905) ----
        // The first passes of a level - few nodes, every divisible node at depth passno < D, so that child counts are differences of bucket
        // offsets and no key moves - are run by ONE wave with wave scans: a workgroup pass costs eight barriers for a microsecond of work.
        if (nnodes <= 64 && passno < D) {
            int* res = s_i + 72 + 4 * (passno & 1);
            if (wave == 0) solo_full_pass(cur, nxt, expn, nnodes, D, bucket_start, node_cap, res);
            __syncthreads();
            T = res[0]; E = res[1]; K = res[2]; overflow = res[3];
        } else {
        // presorted depths: child counts come from the bucket offsets; deeper: big spans are partitioned by the whole
        // workgroup one after another, small spans by one wave each
        // (which kinds of deep nodes exist is found on the way: most passes of a level move no key at all and skip the partition loops)
        {
            int kinds = 0;
            for (int i = tid; i < nnodes; i += NT) {
                const QNode nd = cur[i];
                const int c = node_cnt(nd);
                if (c > 1) {
                    if ((int)nd.depth < D) {
                        int cnt[4];
                        presorted_child_counts(nd, D, bucket_start, cnt);
                        childcnt[4 * i] = (uint32_t)cnt[0]; childcnt[4 * i + 1] = (uint32_t)cnt[1]; childcnt[4 * i + 2] = (uint32_t)cnt[2]; childcnt[4 * i + 3] = (uint32_t)cnt[3];
                    } else kinds |= c > kBigSpan ? kDeepBig : kDeepSmall;
                }
            }
            if (kinds) atomicOr(s_kinds, kinds);
        }
        __syncthreads();
        const int kinds = *s_kinds;
        if (kinds & kDeepBig) {
            for (int i = 0; i < nnodes; i++) {
                const QNode nd = cur[i];
                if (node_cnt(nd) > kBigSpan && (int)nd.depth >= D) {
                    int cnt[4];
                    QuadCls cls; cls.mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1); cls.my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                    const int bsel = node_buf(nd);
                    block_partition4(bsel ? bufB : bufA, bsel ? bufA : bufB, (int)nd.start, node_cnt(nd), cls, s_part, cnt, lgNW);
                    if (tid < 4) childcnt[4 * i + tid] = (uint32_t)(tid == 0 ? cnt[0] : tid == 1 ? cnt[1] : tid == 2 ? cnt[2] : cnt[3]);
                }
            }
        }
        if (kinds & kDeepSmall) wave_partition_many(nnodes, IdentityIdx(), cur, bufA, bufB, childcnt, D, NW);
        __syncthreads();
        if (tid == 0) *s_kinds = 0;
        {
            // exclusive scans over the list of (m = non-empty children, e = children with >1 key, k = kept)
            unsigned long long run = 0, total_all = 0;
            // first compute totals (needed for the reversed block placement)
            for (int i0 = 0; i0 < nnodes; i0 += NT) {
                const int i = i0 + tid;
                unsigned long long v = 0;
                if (i < nnodes) {
                    const int c = node_cnt(cur[i]);
                    if (c > 1) {
                        int m = 0, e = 0;
                        for (int q = 0; q < 4; q++) { const int cq = (int)childcnt[4 * i + q]; m += cq > 0; e += cq > 1; }
                        v = (unsigned long long)m | ((unsigned long long)e << 20);
                    } else v = 1ull << 40;
                }
                unsigned long long tot;
                (void)block_excl_scan_n<unsigned long long>(v, &tot, s_scan, NW);
                total_all += tot;
            }
            T = (int)(total_all & 0xFFFFF); E = (int)((total_all >> 20) & 0xFFFFF); K = (int)(total_all >> 40);
            if (T + K > node_cap) { overflow = 1; }
            if (!overflow) {
                for (int i0 = 0; i0 < nnodes; i0 += NT) {
                    const int i = i0 + tid;
                    unsigned long long v = 0; int c = 0; int cq[4] = {0, 0, 0, 0};
                    QNode nd;
                    if (i < nnodes) {
                        nd = cur[i]; c = node_cnt(nd);
                        if (c > 1) {
                            int m = 0, e = 0;
                            for (int q = 0; q < 4; q++) { cq[q] = (int)childcnt[4 * i + q]; m += cq[q] > 0; e += cq[q] > 1; }
                            v = (unsigned long long)m | ((unsigned long long)e << 20);
                        } else v = 1ull << 40;
                    }
                    unsigned long long tot;
                    const unsigned long long ex = run + block_excl_scan_n<unsigned long long>(v, &tot, s_scan, NW);
                    run += tot;
                    if (i < nnodes) {
                        if (c > 1) {
                            const int Pm = (int)(ex & 0xFFFFF), Pe = (int)((ex >> 20) & 0xFFFFF);
                            const int m = (int)(v & 0xFFFFF);
                            const int newbuf = node_buf(nd) ^ ((int)nd.depth >= D ? 1 : 0);   // count-only divisions move no key
                            int after = 0;      // non-empty children with larger q come first in the block
                            int erank = 0;
                            for (int q = 3; q >= 0; q--) {
                                if (cq[q] > 0) { nxt[T - Pm - m + after] = make_child(nd, q, cq, newbuf); after++; }
                            }
                            int seen = 0;
                            for (int q = 0; q < 4; q++) {
                                if (cq[q] > 0) {
                                    // position of child q inside the block = number of non-empty children with q' > q
                                    int pos_in_block = 0;
                                    for (int q2 = q + 1; q2 < 4; q2++) pos_in_block += cq[q2] > 0;
                                    if (cq[q] > 1) {
                                        const int mxx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
                                        const int cx0 = (q & 1) ? mxx : (int)nd.x0;
                                        expn[Pe + erank] = ((unsigned long long)cq[q] << 32) | ((unsigned long long)(uint16_t)cx0 << 16) |
                                                           (unsigned long long)(T - Pm - m + pos_in_block);
                                        erank++;
                                    }
                                    seen++;
                                }
                            }
                            (void)seen;
                        } else {
                            const int Pk = (int)(ex >> 40);
                            nxt[T + Pk] = nd;
                        }
                    }
                }
            }
        }
        }
        __syncthreads();
        if (overflow) break;
        passno++;
        nnodes = T + K;
        int nexp = E;
        { QNode* t = cur; cur = nxt; nxt = t; }
        { unsigned long long* t = expc; expc = expn; expn = t; }
        if (nnodes >= N || nnodes == prevSize) { finish = true; }
        else if (nnodes + nexp * 3 > N) {
            QT_STAMP(3)
            if (qt_prof && tid == 0 && level == ORBX_QT_STAMP_LEVEL && b == 0) { qt_prof[10] = n; qt_prof[11] = nnodes; qt_prof[12] = nexp; }
            // ---- final rounds (:940-1020): split largest-first until the quota is reached ----
            while (!finish) {
                const int prev2 = nnodes;
                QT_STAMP(4)
                // scratch: the child-count table is not live here (range lists), `erased` doubles as the range-start flags,
                // the other expand array is the rank-sort target
                block_sort_libstdcxx<kSpill>(expc, expn, nexp, (typename SortSeg<kSpill>::T*)childcnt, (typename SortSeg<kSpill>::T*)(childcnt + 2 * (size_t)node_cap),
                                             erased, s_sortctr, NT);
                QT_STAMP(5)
                for (int i = tid; i < nnodes; i += NT) erased[i] = 0;
                if (tid == 0) *s_ndiv = nexp;
                // children counts of every candidate (partition into the other buffer; harmless if the
                // node ends up not being divided: its own buffer is untouched)
                {
                    int kinds = 0;
                    for (int j = tid; j < nexp; j += NT) {    // presorted depths: counts from the bucket offsets
                        const int idx = (int)(expc[j] & 0xFFFF);
                        const QNode nd = cur[idx];
                        if ((int)nd.depth < D) {
                            int cnt[4];
                            presorted_child_counts(nd, D, bucket_start, cnt);
                            childcnt[4 * idx] = (uint32_t)cnt[0]; childcnt[4 * idx + 1] = (uint32_t)cnt[1]; childcnt[4 * idx + 2] = (uint32_t)cnt[2]; childcnt[4 * idx + 3] = (uint32_t)cnt[3];
                        } else kinds |= node_cnt(nd) > kBigSpan ? kDeepBig : kDeepSmall;
                    }
                    if (kinds) atomicOr(s_kinds, kinds);
                }
                __syncthreads();
                const int kinds = *s_kinds;
                if (kinds & kDeepBig) {
                    for (int j = 0; j < nexp; j++) {           // (rare) spans too big for one wave
                        const int idx = (int)(expc[j] & 0xFFFF);
                        const QNode nd = cur[idx];
                        if (node_cnt(nd) > kBigSpan && (int)nd.depth >= D) {
                            int cnt[4];
                            QuadCls cls; cls.mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1); cls.my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                            const int bsel = node_buf(nd);
                            block_partition4(bsel ? bufB : bufA, bsel ? bufA : bufB, (int)nd.start, node_cnt(nd), cls, s_part, cnt, lgNW);
                            if (tid < 4) childcnt[4 * idx + tid] = (uint32_t)(tid == 0 ? cnt[0] : tid == 1 ? cnt[1] : tid == 2 ? cnt[2] : cnt[3]);
                        }
                    }
                }
                if (kinds & kDeepSmall) { ExpandIdx ei; ei.e = expc; wave_partition_many(nexp, ei, cur, bufA, bufB, childcnt, D, NW); }
                __syncthreads();
                if (tid == 0) *s_kinds = 0;
                QT_STAMP(6)
                // how many of the sorted candidates get divided before `size >= N` breaks the loop (:995-1012): candidate t (sorted index
                // nexp - 1 - t) leaves the list with prev2 + sum over t' <= t of (non-empty children - 1) nodes; the first t that reaches N
                // is the last one divided
                {
                    int run = prev2;
                    for (int t0 = 0; t0 < nexp; t0 += NT) {
                        const int t = t0 + tid;
                        int v = 0;
                        if (t < nexp) {
                            const int idx = (int)(expc[nexp - 1 - t] & 0xFFFF);
                            int m = 0;
                            for (int q = 0; q < 4; q++) m += childcnt[4 * idx + q] > 0;
                            v = m - 1;
                        }
                        unsigned long long tot;
                        const int incl = run + (int)block_excl_scan_n<unsigned long long>((unsigned long long)v, &tot, s_scan, NW) + v;
                        if (t < nexp && incl >= N) atomicMin(s_ndiv, t + 1);
                        run += (int)tot;
                        if (run >= N) break;            // uniform
                    }
                }
                __syncthreads();
                QT_STAMP(7)
                const int ndiv = *s_ndiv;
                // totals over the divided set (division order t = 0..ndiv-1 <-> sorted index nexp-1-t): a pass of its own only when the set
                // takes more than one trip - otherwise the scan of the placement loop below delivers them
                const bool one_trip = ndiv <= NT;
                unsigned long long total_all = 0;
                for (int t0 = 0; t0 < ndiv && !one_trip; t0 += NT) {
                    const int t = t0 + tid;
                    unsigned long long v = 0;
                    if (t < ndiv) {
                        const int idx = (int)(expc[nexp - 1 - t] & 0xFFFF);
                        int m = 0, e = 0;
                        for (int q = 0; q < 4; q++) { const int cq = (int)childcnt[4 * idx + q]; m += cq > 0; e += cq > 1; }
                        v = (unsigned long long)m | ((unsigned long long)e << 20);
                        erased[idx] = 1;
                    }
                    unsigned long long tot;
                    (void)block_excl_scan_n<unsigned long long>(v, &tot, s_scan, NW);
                    total_all += tot;
                }
                int T2 = (int)(total_all & 0xFFFFF), E2 = (int)((total_all >> 20) & 0xFFFFF);
                if (!one_trip) {
                    __syncthreads();
                    if (T2 + (prev2 - ndiv) > node_cap) { overflow = 1; break; }
                }
                unsigned long long run = 0;
                for (int t0 = 0; t0 < ndiv; t0 += NT) {
                    const int t = t0 + tid;
                    unsigned long long v = 0; int cq[4] = {0, 0, 0, 0}; QNode nd; int m = 0;
                    if (t < ndiv) {
                        const int idx = (int)(expc[nexp - 1 - t] & 0xFFFF);
                        nd = cur[idx];
                        int e = 0;
                        for (int q = 0; q < 4; q++) { cq[q] = (int)childcnt[4 * idx + q]; m += cq[q] > 0; e += cq[q] > 1; }
                        v = (unsigned long long)m | ((unsigned long long)e << 20);
                        if (one_trip) erased[idx] = 1;             // (read by the kept-list loop below, behind the barriers of this scan)
                    }
                    unsigned long long tot;
                    const unsigned long long ex = run + block_excl_scan_n<unsigned long long>(v, &tot, s_scan, NW);
                    run += tot;
                    if (one_trip) {
                        T2 = (int)(tot & 0xFFFFF); E2 = (int)((tot >> 20) & 0xFFFFF);
                        if (T2 + (prev2 - ndiv) > node_cap) { overflow = 1; break; }      // uniform, before anything is written
                    }
                    if (t < ndiv) {
                        const int Pm = (int)(ex & 0xFFFFF), Pe = (int)((ex >> 20) & 0xFFFFF);
                        const int newbuf = node_buf(nd) ^ ((int)nd.depth >= D ? 1 : 0);
                        int after = 0, erank = 0;
                        for (int q = 3; q >= 0; q--)
                            if (cq[q] > 0) { nxt[T2 - Pm - m + after] = make_child(nd, q, cq, newbuf); after++; }
                        for (int q = 0; q < 4; q++) {
                            if (cq[q] > 1) {
                                int pos_in_block = 0;
                                for (int q2 = q + 1; q2 < 4; q2++) pos_in_block += cq[q2] > 0;
                                const int mxx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
                                const int cx0 = (q & 1) ? mxx : (int)nd.x0;
                                expn[Pe + erank] = ((unsigned long long)cq[q] << 32) | ((unsigned long long)(uint16_t)cx0 << 16) |
                                                   (unsigned long long)(T2 - Pm - m + pos_in_block);
                                erank++;
                            }
                        }
                    }
                }
                if (overflow) break;
                // old list minus the erased parents, in old order, after the new blocks
                int kept_run = 0;
                for (int i0 = 0; i0 < prev2; i0 += NT) {
                    const int i = i0 + tid;
                    const int keep = (i < prev2 && !erased[i]) ? 1 : 0;
                    unsigned long long tot;
                    const int pos = (int)block_excl_scan_n<unsigned long long>((unsigned long long)keep, &tot, s_scan, NW);
                    if (keep) nxt[T2 + kept_run + pos] = cur[i];
                    kept_run += (int)tot;
                }
                __syncthreads();
                nnodes = T2 + kept_run;
                nexp = E2;
                { QNode* t = cur; cur = nxt; nxt = t; }
                { unsigned long long* t = expc; expc = expn; expn = t; }
                if (nnodes >= N || nnodes == prev2) finish = true;
            }
            break;
        }
    }
    __syncthreads();
    QT_STAMP(8)
    // ---- result (:1028-1053): per node, the first key with the largest response ----
    uint32_t* outk = lvl_keys + (size_t)b * kp_total_cap + L.kp_off;
    if (overflow || nnodes > L.kp_cap) {
        if (tid == 0) { atomicOr(status, 1); lvl_count[(size_t)b * nlevels + level] = 0; }
        return;
    }
    // "first key with the largest response" in vKeys order (:1028-1053).  vKeys order = FAST emission order = cells row-major, then y, then
    // x, which is a function of the key itself; spans that were never physically partitioned are ordered by bucket instead, so the order is
    // recomputed rather than read off the position.  A power of two of lanes shares a node - as many as there are for all nodes at once (a
    // level that fills its quota of 257 with four lanes per node on 1024 threads made a second trip for its last node), at most 16 - with
    // keys k = r, r + lanes, .. and eight loads in flight per lane; the lanes combine their candidates with the same rule.
    const unsigned Mw = (1u << 20) / (unsigned)L.wcell + 1u, Mh = (1u << 20) / (unsigned)L.hcell + 1u;
    int lgl = 0;
    while (lgl < 4 && (nnodes << (lgl + 1)) <= NT) lgl++;
    const int lpn = 1 << lgl;
    for (int i0 = 0; i0 < nnodes; i0 += NT >> lgl) {
        const int i = i0 + (tid >> lgl), r = tid & (lpn - 1);
        uint32_t best = 0; int bs = -1; unsigned long long bo = ~0ull;
        if (i < nnodes) {
            const QNode nd = cur[i];
            const uint32_t* kb = (node_buf(nd) ? bufB : bufA) + nd.start;
            const int c = node_cnt(nd);
            for (int k0 = r; k0 < c; k0 += 8 * lpn) {
                uint32_t key[8];
#pragma unroll
                for (int u = 0; u < 8; u++) key[u] = k0 + lpn * u < c ? kb[k0 + lpn * u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k0 + lpn * u >= c) continue;
                    const int ds = key_s(key[u]) - bs;
                    if (ds < 0) continue;
                    const unsigned long long ko = vkeys_order(key[u], Mw, Mh);
                    if (ds > 0 || ko < bo) { best = key[u]; bs = key_s(key[u]); bo = ko; }
                }
            }
        }
        for (int d = 1; d < lpn; d <<= 1) {
            const uint32_t ob = __shfl_xor(best, d); const int os = __shfl_xor(bs, d); const unsigned long long oo = __shfl_xor(bo, d);
            if (os > bs || (os == bs && oo < bo)) { best = ob; bs = os; bo = oo; }
        }
        if (i < nnodes && r == 0) outk[i] = best;
    }
    if (tid == 0) lvl_count[(size_t)b * nlevels + level] = nnodes;
    QT_STAMP(9)
}

// grid (B, number of levels of this launch): workgroups are dispatched image-fastest, i.e. every image's largest level (the longest tree by far)
// starts first and the short trees of the small levels fill the remaining slots.  The launch covers levels level0 .. level0 + gridDim.y - 1.
// Dynamic LDS: node arrays + tables (host: quadtree_lds_bytes).
__global__ void __launch_bounds__(kQuadtreeThreads) k_quadtree(const LevelInfo* __restrict__ lv,
                                                  const CellInfo* __restrict__ cells, int ncells,
                                                  const int* __restrict__ cell_count,
                                                  const uint32_t* __restrict__ slots, size_t slots_stride,
                                                  uint32_t* __restrict__ candA, uint32_t* __restrict__ candB, size_t cand_stride,
                                                  uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                  int* __restrict__ lvl_count, int nlevels, int node_cap, int nb_cap, int lut_x, int lut_y,
                                                  int* __restrict__ status, long long* __restrict__ qt_prof, int wide, int counter_bytes, int level0) {
    ORBX_DYN_SMEM(smem);
    quadtree_tree<false>(smem, nullptr, (int)blockIdx.y + level0, (int)blockIdx.x, lv, cells, ncells, cell_count, slots, slots_stride, candA, candB, cand_stride,
                         lvl_keys, kp_total_cap, lvl_count, nlevels, node_cap, nb_cap, lut_x, lut_y, status, qt_prof, wide, counter_bytes);
}
// The levels whose node lists do not fit the LDS: levels 0 .. gridDim.y - 1, tree (b, level) works in pool + (b * gridDim.y + level) * pool_stride.
// Dynamic LDS: the tables only.
__global__ void __launch_bounds__(kQuadtreeThreads) k_quadtree_spill(const LevelInfo* __restrict__ lv,
                                                  const CellInfo* __restrict__ cells, int ncells,
                                                  const int* __restrict__ cell_count,
                                                  const uint32_t* __restrict__ slots, size_t slots_stride,
                                                  uint32_t* __restrict__ candA, uint32_t* __restrict__ candB, size_t cand_stride,
                                                  uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                                                  int* __restrict__ lvl_count, int nlevels, int node_cap, int nb_cap, int lut_x, int lut_y,
                                                  int* __restrict__ status, long long* __restrict__ qt_prof, int wide, int counter_bytes,
                                                  unsigned char* __restrict__ pool, size_t pool_stride) {
    ORBX_DYN_SMEM(smem);
    quadtree_tree<true>(smem, pool + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * pool_stride, (int)blockIdx.y, (int)blockIdx.x, lv, cells, ncells, cell_count,
                        slots, slots_stride, candA, candB, cand_stride, lvl_keys, kp_total_cap, lvl_count, nlevels, node_cap, nb_cap, lut_x, lut_y, status,
                        qt_prof, wide, counter_bytes);
}

}  // namespace orbx
