// libstdcxx_sort_model.h — a bit-for-bit model of libstdc++'s std::sort (GCC bits/stl_algo.h:
// __introsort_loop + __final_insertion_sort, threshold 16, median-of-3 pivot, heapsort fallback at
// depth 2*floor(log2 n)), runnable by a single GPU lane.
//
// Why: the reference's quadtree (src/ORBextractor.cc:950) orders the nodes it splits in its final
// rounds with an *unstable* std::sort whose comparator (compareNodes, :676-697) only looks at
// (#keys, UL.x).  Nodes that tie on both are permuted in an implementation-defined way, and that
// permutation decides which nodes are split before the quota is reached and the order of the output
// keypoints.  The oracle is the reference compiled with libstdc++, so the device quadtree reproduces
// libstdc++'s permutation exactly.  tests/test_sort_model.py checks this model against std::sort on
// tie-heavy, sorted, reversed and median-of-3-killer inputs (host build of the same header).
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif

namespace orbx {

template <typename T, typename Less>
ORBX_HD inline void sm_unguarded_linear_insert(T* a, int last, Less less) {
    T val = a[last];
    int next = last - 1;
    while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}
template <typename T, typename Less>
ORBX_HD inline void sm_insertion_sort(T* a, int first, int last, Less less) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (less(a[i], a[first])) {
            T val = a[i];
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = val;
        } else sm_unguarded_linear_insert(a, i, less);
    }
}
template <typename T, typename Less>
ORBX_HD inline void sm_push_heap(T* a, int first, int hole, int top, T value, Less less) {
    int parent = (hole - 1) / 2;
    while (hole > top && less(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
template <typename T, typename Less>
ORBX_HD inline void sm_adjust_heap(T* a, int first, int hole, int len, T value, Less less) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(a[first + child], a[first + (child - 1)])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + (child - 1)];
        hole = child - 1;
    }
    sm_push_heap(a, first, hole, top, value, less);
}
// std::__partial_sort(first, last, last): make_heap + sort_heap
template <typename T, typename Less>
ORBX_HD inline void sm_heap_sort(T* a, int first, int last, Less less) {
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            T value = a[first + parent];
            sm_adjust_heap(a, first, parent, len, value, less);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {
        --l;
        T value = a[l];
        a[l] = a[first];
        sm_adjust_heap(a, first, 0, l - first, value, less);
    }
}
template <typename T, typename Less>
ORBX_HD inline void sm_move_median_to_first(T* a, int result, int ia, int ib, int ic, Less less) {
    int pick;
    if (less(a[ia], a[ib])) {
        if (less(a[ib], a[ic])) pick = ib;
        else if (less(a[ia], a[ic])) pick = ic;
        else pick = ia;
    } else if (less(a[ia], a[ic])) pick = ia;
    else if (less(a[ib], a[ic])) pick = ic;
    else pick = ib;
    T t = a[result]; a[result] = a[pick]; a[pick] = t;
}
template <typename T, typename Less>
ORBX_HD inline int sm_unguarded_partition(T* a, int first, int last, int pivot, Less less) {
    for (;;) {
        while (less(a[first], a[pivot])) ++first;
        --last;
        while (less(a[pivot], a[last])) --last;
        if (!(first < last)) return first;
        T t = a[first]; a[first] = a[last]; a[last] = t;
        ++first;
    }
}

// The same partition in the data-parallel form the quadtree kernel uses for long ranges (wave_unguarded_partition in k_quadtree.hip):
// L_0 < L_1 < .. = positions of the elements >= pivot, R_0 > R_1 > .. = positions of the elements <= pivot (initial values); swap k
// exchanges a[L_k], a[R_k]; K = #{k : L_k < R_k} swaps happen; the cut is min(L_K, R_(K-1)).  Host-only (tests compare it with the loop above).
#if !defined(__HIP_DEVICE_COMPILE__)
template <typename T, typename Less>
inline int sm_unguarded_partition_lists(T* a, int lo, int hi, int pivot, Less less) {
    const int m = hi - lo;
    int* Lp = new int[2 * (m > 0 ? m : 1)];
    int* Rp = Lp + (m > 0 ? m : 1);
    int nL = 0, nR = 0;
    for (int i = lo; i < hi; i++) if (!less(a[i], a[pivot])) Lp[nL++] = i;
    for (int i = hi - 1; i >= lo; i--) if (!less(a[pivot], a[i])) Rp[nR++] = i;
    int K = 0;
    while (K < nL && K < nR && Lp[K] < Rp[K]) { T t = a[Lp[K]]; a[Lp[K]] = a[Rp[K]]; a[Rp[K]] = t; K++; }
    int cut = 0x7FFFFFFF;
    if (K < nL) cut = Lp[K];
    if (K >= 1 && Rp[K - 1] < cut) cut = Rp[K - 1];
    delete[] Lp;
    return cut;
}
#endif

// std::sort(a, a+n, less); LISTS selects the data-parallel form of the partition step (host tests)
template <typename T, typename Less>
ORBX_HD inline void libstdcxx_sort(T* a, int n, Less less, bool lists = false) {
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    // __introsort_loop with an explicit stack (sub-ranges are disjoint, so their processing order is
    // immaterial; each carries its own depth budget exactly like the recursion does)
    int stk_first[64], stk_last[64], stk_depth[64];
    int sp = 0;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { sm_heap_sort(a, first, last, less); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            sm_move_median_to_first(a, first, first + 1, mid, last - 1, less);
#if !defined(__HIP_DEVICE_COMPILE__)
            const int cut = lists ? sm_unguarded_partition_lists(a, first + 1, last, first, less) : sm_unguarded_partition(a, first + 1, last, first, less);
#else
            const int cut = sm_unguarded_partition(a, first + 1, last, first, less);
#endif
            stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth; sp++;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        sm_insertion_sort(a, 0, 16, less);
        for (int i = 16; i != n; ++i) sm_unguarded_linear_insert(a, i, less);
    } else sm_insertion_sort(a, 0, n, less);
}

}  // namespace orbx
