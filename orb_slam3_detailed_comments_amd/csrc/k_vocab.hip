// k_vocab.hip — DBoW2 vocabulary transform: what Frame::ComputeBoW / KeyFrame::ComputeBoW
// (src/Frame.cc:984-997) ask of ORBVocabulary::transform(features, BowVector&, FeatureVector&, levelsup = 4)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1195), on the device:
//   k_voc_descend   the per-feature tree descent (TemplatedVocabulary.h:1218-1259): at every level the child with the smallest
//                   Hamming distance (FORB::distance, FORB.cpp:81-101), first child on ties (strict `d < best_d`), until a node
//                   without children; records the word id, the word weight and the ancestor at level L - levelsup.
//                   A group of 16 lanes serves one feature (one child per lane and trip; k = 10 in ORBvoc), 4 features per wave;
//                   the children of a node sit in consecutive slots so the group reads one contiguous 32*k-byte run.
//   k_voc_assemble  one workgroup per image: BowVector (std::map<WordId, WordValue>: ascending word ids, value accumulated by
//                   addWeight in feature order, then L1/L2-normalised in id order, BowVector.cpp:27-84) and FeatureVector
//                   (std::map<NodeId, vector<unsigned>>: ascending node ids, feature indices in insertion order) as sorted
//                   arrays; bitonic sort of (id << 16 | feature) keys in LDS, run detection with a workgroup scan, and the
//                   order-dependent double-precision sums done sequentially by one thread exactly as the map iteration does.
#include "orbx_types.h"
#include "orbx_block.h"

namespace orbx {

constexpr int kVocGroup = 16;      // lanes per feature in k_voc_descend

// grid (ceil(n_total / 16)), 256 threads.  Features of image b are fdesc[(b*cap + i)*4 .. +3], i < n_feat[b] (n_feat == nullptr:
// one image with n_fixed features).  Outputs are indexed like the features.
__global__ void __launch_bounds__(256) k_voc_descend(const unsigned long long* __restrict__ fdesc, const int* __restrict__ n_feat, int n_fixed,
                                                     int cap, int B, const unsigned long long* __restrict__ slot_desc,
                                                     const VocSlot* __restrict__ slots, const double* __restrict__ slot_weight,
                                                     int root_children, int nid_level, unsigned* __restrict__ out_word,
                                                     unsigned* __restrict__ out_node, double* __restrict__ out_weight) {
    const int g = (int)((blockIdx.x * 256u + threadIdx.x) / kVocGroup), sub = (int)(threadIdx.x & (kVocGroup - 1));
    const int b = g / cap, i = g - b * cap;
    // whole groups leave together (g is uniform inside a group), so the width-16 shuffles below never wait for a missing lane
    if (b >= B) return;
    const int n = n_feat ? n_feat[b] : n_fixed;
    if (i >= n) return;
    const unsigned long long* f = fdesc + 4 * ((size_t)b * cap + i);
    const unsigned long long f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
    int cs = 0, cc = root_children, level = 0, slot = -1;
    unsigned nid = 0;
    bool have_nid = nid_level <= 0;          // "if(nid_level <= 0 && nid != NULL) *nid = 0; // root"
    // the loop runs while any group of the wave still descends (groups whose leaf is shallower idle with cc == 0), which keeps the
    // width-16 exchanges convergent
    while (__ballot(cc > 0) != 0ull) {
        level++;
        unsigned best = 0xFFFFFFFFu;         // dist << 8 | child: minimum = smallest distance, first child on ties
        for (int c0 = 0; c0 < cc; c0 += kVocGroup) {
            const int c = c0 + sub;
            unsigned key = 0xFFFFFFFFu;
            if (c < cc) {
                const unsigned long long* d = slot_desc + 4 * (size_t)(cs + c);
                const int dist = __popcll(f0 ^ d[0]) + __popcll(f1 ^ d[1]) + __popcll(f2 ^ d[2]) + __popcll(f3 ^ d[3]);
                key = ((unsigned)dist << 8) | (unsigned)c;
            }
            best = key < best ? key : best;
        }
#pragma unroll
        for (int d = kVocGroup / 2; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(best, d, kVocGroup); best = o < best ? o : best; }
        if (cc > 0) {
            slot = cs + (int)(best & 0xFFu);
            const VocSlot s = slots[slot];
            if (level == nid_level) { nid = (unsigned)s.node_id; have_nid = true; }
            cs = s.child_start; cc = s.child_cnt;
        }
    }
    if (sub == 0) {
        const size_t o = (size_t)b * cap + i;
        if (slot < 0) { out_word[o] = 0; out_node[o] = 0; out_weight[o] = 0.0; return; }   // empty vocabulary
        const VocSlot s = slots[slot];
        out_word[o] = (unsigned)s.word_id;
        // a leaf above level L - levelsup: the reference leaves *nid untouched (an uninitialised local, TemplatedVocabulary.h:1150);
        // here that case yields the leaf itself
        out_node[o] = have_nid ? nid : (unsigned)s.node_id;
        out_weight[o] = slot_weight[slot];
    }
}

// in-LDS bitonic sort of P (power of two) 64-bit keys, ascending; all 256 threads
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* key, int P, int tid) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P; t += 256) {
                const int p = t ^ j;
                if (p > t) {
                    const unsigned long long a = key[t], b2 = key[p];
                    const bool up = (t & k) == 0;
                    if ((a > b2) == up) { key[t] = b2; key[p] = a; }
                }
            }
            __syncthreads();
        }
}

// run starts of the sorted keys (id = key >> 16): ids[u], start[u] for every distinct id, start[nu] = nvalid; returns nu (all threads)
__device__ __forceinline__ int emit_runs(const unsigned long long* key, int nvalid, unsigned* __restrict__ ids, int* __restrict__ start,
                                         int tid, unsigned long long* scan_scratch) {
    int base = 0;
    for (int i0 = 0; i0 < nvalid; i0 += 256) {
        const int i = i0 + tid;
        int flag = 0;
        if (i < nvalid) flag = (i == 0) || ((key[i] >> 16) != (key[i - 1] >> 16));
        unsigned long long tot;
        const int ex = (int)block_excl_scan<unsigned long long>((unsigned long long)flag, &tot, scan_scratch);
        if (flag) { ids[base + ex] = (unsigned)(key[i] >> 16); start[base + ex] = i; }
        base += (int)tot;
    }
    if (tid == 0) start[base] = nvalid;
    return base;
}

// grid (B), 256 threads, dynamic LDS = P * 8 bytes (P = pow2 >= max features per image).
// weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY;  norm: 0 none, 1 L1, 2 L2  (BowVector.h:29-50, ScoringObject.h:74-89)
// Per image b (regions of `cap` entries): bow_id/bow_val/[n_out[2b]], fv_node/fv_start(cap+1)/fv_feat/[n_out[2b+1]];
// bow_start: scratch, cap+1 ints per image.
__global__ void __launch_bounds__(256) k_voc_assemble(const unsigned* __restrict__ word, const unsigned* __restrict__ node,
                                                      const double* __restrict__ weight, const int* __restrict__ n_feat, int n_fixed, int cap,
                                                      int P, int weighting, int norm, unsigned* __restrict__ bow_id,
                                                      double* __restrict__ bow_val, int* __restrict__ bow_start,
                                                      unsigned* __restrict__ fv_node, int* __restrict__ fv_start,
                                                      unsigned* __restrict__ fv_feat, int* __restrict__ n_out) {
    ORBX_DYN_SMEM(smem);
    __shared__ unsigned long long s_scan[20];
    __shared__ int s_nvalid;
    __shared__ double s_norm;
    unsigned long long* key = (unsigned long long*)smem;
    const int tid = (int)threadIdx.x, b = (int)blockIdx.x;
    const int n = n_feat ? n_feat[b] : n_fixed;
    const size_t off = (size_t)b * cap, offs = (size_t)b * (cap + 1);
    const unsigned* w_ = word + off; const unsigned* nd_ = node + off; const double* wt_ = weight + off;
    const unsigned long long kEmpty = ~0ull;
    for (int pass = 0; pass < 2; pass++) {
        // "if(w > 0) // not stopped": stopped features enter neither vector
        if (tid == 0) s_nvalid = 0;
        __syncthreads();
        int mine = 0;
        for (int i = tid; i < P; i += 256) {
            unsigned long long k = kEmpty;
            if (i < n && wt_[i] > 0) { k = ((unsigned long long)(pass == 0 ? w_[i] : nd_[i]) << 16) | (unsigned long long)i; mine++; }
            key[i] = k;
        }
        if (mine) atomicAdd(&s_nvalid, mine);
        __syncthreads();
        bitonic_sort_u64(key, P, tid);
        const int nvalid = s_nvalid;
        if (pass == 0) {
            const int nu = emit_runs(key, nvalid, bow_id + off, bow_start + offs, tid, s_scan);
            __syncthreads();
            // value of each word: addWeight adds w once per feature in feature order (TF_IDF / TF); addIfNotExist keeps the first (IDF / BINARY)
            for (int u = tid; u < nu; u += 256) {
                const int s = bow_start[offs + u], e = bow_start[offs + u + 1];
                const double w = wt_[(int)(key[s] & 0xFFFFu)];
                double v = w;
                if (weighting <= 1) for (int t = s + 1; t < e; t++) v = v + w;
                bow_val[off + u] = v;
            }
            __syncthreads();
            if (weighting <= 1 && norm == 0 && nu > 0) {           // "unnecessary when normalizing": vit->second /= nd
                const double ndv = (double)nu;
                for (int u = tid; u < nu; u += 256) bow_val[off + u] = bow_val[off + u] / ndv;
            }
            if (norm != 0) {                                        // BowVector::normalize, in ascending id order
                if (tid == 0) {
                    double acc = 0.0;
                    if (norm == 1) for (int u = 0; u < nu; u++) acc = acc + fabs(bow_val[off + u]);
                    else { for (int u = 0; u < nu; u++) acc = acc + bow_val[off + u] * bow_val[off + u]; acc = sqrt(acc); }
                    s_norm = acc;
                }
                __syncthreads();
                const double nv = s_norm;
                if (nv > 0.0) for (int u = tid; u < nu; u += 256) bow_val[off + u] = bow_val[off + u] / nv;
            }
            if (tid == 0) n_out[2 * b] = nu;
        } else {
            const int nu = emit_runs(key, nvalid, fv_node + off, fv_start + offs, tid, s_scan);
            for (int i = tid; i < nvalid; i += 256) fv_feat[off + i] = (unsigned)(key[i] & 0xFFFFu);
            if (tid == 0) n_out[2 * b + 1] = nu;
        }
        __syncthreads();
    }
}

}  // namespace orbx
