// orbv_api.cpp — host side of the vocabulary transform (include/orbx.h "Vocabulary"): flattens an ORBVocabulary
// (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ORBVocabulary.h:28-29) into slot arrays whose children are
// contiguous, uploads them once, and runs k_voc_descend + k_voc_assemble (csrc/k_vocab.hip) per batch of descriptors.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include "orbx_internal.h"

using namespace orbx;

struct orbv_vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0, device = 0;
    int n_nodes = 0;          // without the root
    int n_words = 0;
    int root_children = 0;
    DevBuf<unsigned long long> d_desc; DevBuf<VocSlot> d_slots; DevBuf<double> d_weight;
    // per-call scratch / results (sized by reserve())
    int cap = 0, maxB = 0, lastB = 0, run_cap = 0;      // run_cap: per-image stride of the results of the last run
    int run_first = -1; const void* run_handle = nullptr; uint64_t run_extract_gen = 0;   // orbv_transform_extracted: which images of which extractor the results belong to
    DevBuf<unsigned long long> d_fdesc;
    DevBuf<unsigned> d_word, d_node, d_bow_id, d_fv_node, d_fv_feat;
    DevBuf<double> d_wt, d_bow_val;
    DevBuf<int> d_bow_start, d_fv_start, d_nout, d_nfeat;
};

namespace {

int norm_of(int scoring) {      // ScoringObject.h:74-89: (mustNormalize, norm) of each scoring class
    switch (scoring) { case 0: return 1; case 1: return 2; case 2: case 3: case 4: return 1; default: return 0; }
}

int reserve(orbv_vocabulary* v, int cap, int B) {
    if (cap <= v->cap && B <= v->maxB) return 0;
    cap = std::max(cap, v->cap); B = std::max(B, v->maxB);
    const size_t t = (size_t)cap * B;
    int e = v->d_word.ensure(t) | v->d_node.ensure(t) | v->d_wt.ensure(t) | v->d_bow_id.ensure(t) | v->d_bow_val.ensure(t) |
            v->d_fv_node.ensure(t) | v->d_fv_feat.ensure(t) | v->d_bow_start.ensure(t + B) | v->d_fv_start.ensure(t + B) |
            v->d_nout.ensure(2 * (size_t)B) | v->d_nfeat.ensure(B);
    if (e) return -1;
    v->cap = cap; v->maxB = B;
    return 0;
}

// launches the two kernels over B images whose descriptors sit at fdesc[(b*cap + i)*4]; n_feat: device counts or nullptr (n_fixed)
int run(orbv_vocabulary* v, orbx_extractor* h, const unsigned long long* fdesc, const int* n_feat, int n_fixed, int cap, int B, int levelsup) {
    if (cap > 16384) return fail(ORBX_E_CAPACITY, "more than 16384 features per image");
    if (reserve(v, cap, B)) return fail(ORBX_E_DEVICE, "vocabulary scratch allocation failed");
    int P = 64; while (P < cap) P <<= 1;
    // k_voc_assemble sorts one image's (id, feature) keys in LDS: 8 bytes per key slot (+ its static scan scratch)
    if ((size_t)P * 8 + 1024 > rt::lds_limit(h->device))
        return fail(ORBX_E_CAPACITY, "%d features per image need %zu bytes of LDS for the vocabulary transform, the device allows %zu per workgroup", cap, (size_t)P * 8 + 1024, rt::lds_limit(h->device));
    const long groups = (long)cap * B;
    dim3 g1((unsigned)((groups * 16 + 255) / 256), 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_voc_descend, g1, blk, 0, h->s0, fdesc, n_feat, n_fixed, cap, B, (const unsigned long long*)v->d_desc.p,
                (const VocSlot*)v->d_slots.p, (const double*)v->d_weight.p, v->root_children, v->L - levelsup, v->d_word.p, v->d_node.p, v->d_wt.p);
    dim3 g2((unsigned)B, 1, 1);
    ORBX_LAUNCH(k_voc_assemble, g2, blk, (size_t)P * 8, h->s0, (const unsigned*)v->d_word.p, (const unsigned*)v->d_node.p, (const double*)v->d_wt.p,
                n_feat, n_fixed, cap, P, v->weighting, norm_of(v->scoring), v->d_bow_id.p, v->d_bow_val.p, v->d_bow_start.p, v->d_fv_node.p,
                v->d_fv_start.p, v->d_fv_feat.p, v->d_nout.p);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "vocabulary kernels failed to launch: %s", rt::last_error());
    v->lastB = B; v->run_cap = cap; v->run_first = -1; v->run_handle = nullptr;
    return ORBX_OK;
}

}  // namespace

namespace orbx {
int orbv_frame_arrays(const orbv_vocabulary* v, VocFrameArrays* out) {
    if (!v || v->lastB <= 0) return -1;
    out->fv_node = (const uint32_t*)v->d_fv_node.p; out->fv_start = v->d_fv_start.p; out->fv_feat = (const int*)v->d_fv_feat.p; out->nout = v->d_nout.p;
    out->cap = v->run_cap; out->lastB = v->lastB; out->device = v->device; out->first = v->run_first; out->handle = v->run_handle; out->extract_gen = v->run_extract_gen;
    return 0;
}
}  // namespace orbx

extern "C" {

int orbv_create(orbx_extractor* h, int k, int L, int scoring, int weighting, int n_nodes, const int* parent, const uint8_t* is_leaf,
                const uint8_t* desc, const double* weight, orbv_vocabulary** out) {
    if (!h || !out || n_nodes < 0 || (n_nodes > 0 && (!parent || !is_leaf || !desc || !weight))) return fail(ORBX_E_ARG, "null");
    if (scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return fail(ORBX_E_ARG, "unknown scoring / weighting type");
    rt::set_device(h->device);
    // node i+1 of the reference = entry i here (loadFromTextFile numbers the nodes by line, TemplatedVocabulary.h:1385-1392); the
    // children of a node keep their line order; word ids are handed out to the nodes flagged as leaves, in line order (:1408-1416)
    std::vector<std::vector<int>> children((size_t)n_nodes + 1);
    for (int i = 0; i < n_nodes; i++) {
        const int p = parent[i];
        if (p < 0 || p > i) return fail(ORBX_E_ARG, "node %d: parent %d must be an earlier node", i + 1, p);
        children[p].push_back(i + 1);
    }
    std::vector<int> word_of((size_t)n_nodes + 1, 0);
    int n_words = 0;
    for (int i = 0; i < n_nodes; i++) if (is_leaf[i]) word_of[i + 1] = n_words++;
    // slots: breadth-first, so that the children of every node are consecutive
    std::vector<int> slot_node; slot_node.reserve(n_nodes);
    std::vector<int> first_child_slot((size_t)n_nodes + 1, 0);
    std::vector<int> queue; queue.push_back(0);
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const int nd = queue[qi];
        first_child_slot[nd] = (int)slot_node.size();
        for (int c : children[nd]) { slot_node.push_back(c); queue.push_back(c); }
    }
    const int ns = (int)slot_node.size();      // == n_nodes (every node hangs below the root)
    std::vector<VocSlot> slots(ns > 0 ? ns : 1);
    std::vector<unsigned long long> sdesc((size_t)(ns > 0 ? ns : 1) * 4);
    std::vector<double> sw(ns > 0 ? ns : 1);
    for (int s = 0; s < ns; s++) {
        const int nd = slot_node[s];
        slots[s].node_id = nd; slots[s].child_start = first_child_slot[nd]; slots[s].child_cnt = (int)children[nd].size();
        slots[s].word_id = word_of[nd];
        if (slots[s].child_cnt > 255) return fail(ORBX_E_ARG, "node %d has more than 255 children", nd);
        memcpy(&sdesc[(size_t)s * 4], desc + 32 * (size_t)(nd - 1), 32);
        sw[s] = weight[nd - 1];
    }
    if (children[0].size() > 255) return fail(ORBX_E_ARG, "the root has more than 255 children");
    orbv_vocabulary* v = new orbv_vocabulary();
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->device = h->device;
    v->n_nodes = n_nodes; v->n_words = n_words; v->root_children = (int)children[0].size();
    int e = v->d_desc.ensure(sdesc.size()) | v->d_slots.ensure(slots.size()) | v->d_weight.ensure(sw.size());
    if (!e) e = rt::copy_h2d(v->d_desc.p, sdesc.data(), sdesc.size() * 8, h->s0) | rt::copy_h2d(v->d_slots.p, slots.data(), slots.size() * sizeof(VocSlot), h->s0) |
                rt::copy_h2d(v->d_weight.p, sw.data(), sw.size() * 8, h->s0) | rt::stream_sync(h->s0);
    if (e) { orbv_destroy(v); return fail(ORBX_E_DEVICE, "vocabulary upload failed"); }
    *out = v;
    return ORBX_OK;
}

// ORBvoc.txt: "k L scoring weighting" then one line per node "parent isLeaf d0 .. d31 weight" (TemplatedVocabulary.h:1338-1430)
int orbv_load_text(orbx_extractor* h, const char* path, orbv_vocabulary** out) {
    if (!h || !path || !out) return fail(ORBX_E_ARG, "null");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(ORBX_E_ARG, "cannot open %s", path);
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    const char* p = text.c_str();
    char* end = nullptr;
    long hdr[4];
    for (int i = 0; i < 4; i++) { hdr[i] = strtol(p, &end, 10); if (end == p) return fail(ORBX_E_ARG, "bad vocabulary header"); p = end; }
    if (hdr[0] < 0 || hdr[0] > 20 || hdr[1] < 1 || hdr[1] > 10 || hdr[2] < 0 || hdr[2] > 5 || hdr[3] < 0 || hdr[3] > 3)
        return fail(ORBX_E_ARG, "not a vocabulary text file");                       // the reference's own check (:1359)
    std::vector<int> parent; std::vector<uint8_t> leaf, desc; std::vector<double> weight;
    for (;;) {
        while (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t') p++;
        if (!*p) break;
        long v0 = strtol(p, &end, 10); if (end == p) return fail(ORBX_E_ARG, "bad node line %zu", parent.size() + 1); p = end;
        long v1 = strtol(p, &end, 10); if (end == p) return fail(ORBX_E_ARG, "bad node line %zu", parent.size() + 1); p = end;
        parent.push_back((int)v0); leaf.push_back(v1 > 0 ? 1 : 0);
        for (int i = 0; i < 32; i++) { long d = strtol(p, &end, 10); if (end == p) return fail(ORBX_E_ARG, "bad descriptor in node line %zu", parent.size()); p = end; desc.push_back((uint8_t)d); }
        double w = strtod(p, &end); if (end == p) return fail(ORBX_E_ARG, "bad weight in node line %zu", parent.size()); p = end;
        weight.push_back(w);
    }
    return orbv_create(h, (int)hdr[0], (int)hdr[1], (int)hdr[2], (int)hdr[3], (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

void orbv_destroy(orbv_vocabulary* v) {
    if (!v) return;
    rt::set_device(v->device);
    v->d_desc.release(); v->d_slots.release(); v->d_weight.release(); v->d_fdesc.release(); v->d_word.release(); v->d_node.release();
    v->d_bow_id.release(); v->d_fv_node.release(); v->d_fv_feat.release(); v->d_wt.release(); v->d_bow_val.release(); v->d_bow_start.release();
    v->d_fv_start.release(); v->d_nout.release(); v->d_nfeat.release();
    delete v;
}

int orbv_words(const orbv_vocabulary* v) { return v ? v->n_words : 0; }

int orbv_transform_extracted(orbv_vocabulary* v, orbx_extractor* h, int first, int B, int levelsup) {
    if (!v || !h) return fail(ORBX_E_ARG, "null");
    if (v->device != h->device) return fail(ORBX_E_ARG, "vocabulary and extractor live on different devices");
    if (first < 0 || B <= 0 || first + B > h->lastB) return fail(ORBX_E_ARG, "images [%d, %d) are not in the last batch of %d", first, first + B, h->lastB);
    rt::set_device(h->device);
    const int cap = h->kp_total_cap;
    const int rc = run(v, h, (const unsigned long long*)(h->d_desc.p + (size_t)first * cap * 4), (const int*)(h->d_nm.p + first), 0, cap, B, levelsup);
    if (rc == ORBX_OK) { v->run_first = first; v->run_handle = h; v->run_extract_gen = h->extract_gen; }
    return rc;
}

int orbv_fetch(orbv_vocabulary* v, orbx_extractor* h, int b, uint32_t* word_id, uint32_t* node_id, int n_features, uint32_t* bow_id, double* bow_val,
               int* n_bow, uint32_t* fv_node, int* fv_start, uint32_t* fv_feat, int* n_fv) {
    if (!v || !h) return fail(ORBX_E_ARG, "null");
    if (b < 0 || b >= v->lastB) return fail(ORBX_E_ARG, "image %d is not in the last transformed batch of %d", b, v->lastB);
    rt::set_device(h->device);
    int nout[2] = {0, 0};
    const int cap = v->run_cap;
    const size_t off = (size_t)b * cap, offs = (size_t)b * (cap + 1);
    int e = rt::copy_d2h(nout, v->d_nout.p + 2 * b, sizeof nout, h->s0) | rt::stream_sync(h->s0);
    if (e) return fail(ORBX_E_DEVICE, "vocabulary fetch failed: %s", rt::last_error());
    if (n_features > cap) n_features = cap;
    if (word_id && n_features > 0) e |= rt::copy_d2h(word_id, v->d_word.p + off, 4 * (size_t)n_features, h->s0);
    if (node_id && n_features > 0) e |= rt::copy_d2h(node_id, v->d_node.p + off, 4 * (size_t)n_features, h->s0);
    if (bow_id && nout[0] > 0) e |= rt::copy_d2h(bow_id, v->d_bow_id.p + off, 4 * (size_t)nout[0], h->s0);
    if (bow_val && nout[0] > 0) e |= rt::copy_d2h(bow_val, v->d_bow_val.p + off, 8 * (size_t)nout[0], h->s0);
    if (fv_node && nout[1] > 0) e |= rt::copy_d2h(fv_node, v->d_fv_node.p + off, 4 * (size_t)nout[1], h->s0);
    if (fv_start) e |= rt::copy_d2h(fv_start, v->d_fv_start.p + offs, 4 * (size_t)(nout[1] + 1), h->s0);
    if (fv_feat) {
        int total = 0;
        e |= rt::copy_d2h(&total, v->d_fv_start.p + offs + nout[1], 4, h->s0) | rt::stream_sync(h->s0);
        if (total > 0) e |= rt::copy_d2h(fv_feat, v->d_fv_feat.p + off, 4 * (size_t)total, h->s0);
    }
    e |= rt::stream_sync(h->s0);
    if (e) return fail(ORBX_E_DEVICE, "vocabulary fetch failed: %s", rt::last_error());
    if (n_bow) *n_bow = nout[0];
    if (n_fv) *n_fv = nout[1];
    return ORBX_OK;
}

int orbv_transform(orbv_vocabulary* v, orbx_extractor* h, const uint8_t* desc, int n, int levelsup, uint32_t* word_id, uint32_t* node_id,
                   uint32_t* bow_id, double* bow_val, int* n_bow, uint32_t* fv_node, int* fv_start, uint32_t* fv_feat, int* n_fv) {
    if (!v || !h || n < 0 || (n > 0 && !desc)) return fail(ORBX_E_ARG, "null");
    if (v->device != h->device) return fail(ORBX_E_ARG, "vocabulary and extractor live on different devices");
    if (n > 16384) return fail(ORBX_E_CAPACITY, "more than 16384 features");
    rt::set_device(h->device);
    const int cap = std::max(n, 1);
    if (v->d_fdesc.ensure((size_t)cap * 4)) return fail(ORBX_E_DEVICE, "allocation failed");
    if (n > 0 && rt::copy_h2d(v->d_fdesc.p, desc, 32 * (size_t)n, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    int rc = run(v, h, v->d_fdesc.p, nullptr, n, cap, 1, levelsup); if (rc) return rc;
    return orbv_fetch(v, h, 0, word_id, node_id, n, bow_id, bow_val, n_bow, fv_node, fv_start, fv_feat, n_fv);
}

}  // extern "C"
