// oracle/ref_dbow2_driver.cpp — TEST INFRASTRUCTURE ONLY.
// C wrapper around the REFERENCE's own DBoW2 (Thirdparty/DBoW2/DBoW2/{TemplatedVocabulary.h,FORB.cpp,BowVector.cpp,FeatureVector.cpp,
// ScoringObject.cpp}), compiled unmodified and in place from /root/reference by oracle/Makefile against oracle/opencv_shim (cv::Mat,
// stub FileStorage) and oracle/boost_shim (two empty serialization headers).  Used to pin the vocabulary transform
// (Frame::ComputeBoW, src/Frame.cc:984-997 -> TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup),
// TemplatedVocabulary.h:1127-1195) and MapPoint-side Hamming work against the reference implementation itself.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "DBoW2/FORB.h"
#include "DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary;   // include/ORBVocabulary.h:28-29
struct VocProbe : ORBVocabulary { using ORBVocabulary::transform; };                          // the single-feature descent is protected

extern "C" {

void* ref_voc_load_text(const char* path) {
    ORBVocabulary* v = new VocProbe();
    if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
    return v;
}
void ref_voc_destroy(void* h) { delete (ORBVocabulary*)h; }
int ref_voc_size(void* h) { return (int)((ORBVocabulary*)h)->size(); }

// transform(features, bow, fv, levelsup).  Outputs: bow_id/bow_val (ascending word id, *n_bow entries, capacity n each);
// fv as CSR: fv_node (ascending), fv_start (*n_fv + 1), fv_feat (n entries max).
void ref_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, unsigned* bow_id, double* bow_val, int* n_bow,
                       unsigned* fv_node, int* fv_start, unsigned* fv_feat, int* n_fv) {
    ORBVocabulary* voc = (ORBVocabulary*)h;
    std::vector<cv::Mat> feats(n);
    for (int i = 0; i < n; i++) { feats[i].create(1, 32, CV_8U); memcpy(feats[i].data, desc + 32 * (size_t)i, 32); }   // Converter::toDescriptorVector
    DBoW2::BowVector bv; DBoW2::FeatureVector fv;
    voc->transform(feats, bv, fv, levelsup);
    int k = 0;
    for (auto it = bv.begin(); it != bv.end(); ++it, ++k) { bow_id[k] = it->first; bow_val[k] = it->second; }
    *n_bow = k;
    int m = 0, pos = 0;
    for (auto it = fv.begin(); it != fv.end(); ++it, ++m) {
        fv_node[m] = it->first; fv_start[m] = pos;
        for (unsigned f : it->second) fv_feat[pos++] = f;
    }
    fv_start[m] = pos; *n_fv = m;
}

// single-feature transform (TemplatedVocabulary.h:1218-1259): word id, weight and the ancestor `levelsup` levels above the leaf
void ref_voc_transform_one(void* h, const uint8_t* desc, int levelsup, unsigned* word, double* weight, unsigned* node) {
    VocProbe* voc = (VocProbe*)h;
    cv::Mat f(1, 32, CV_8U); memcpy(f.data, desc, 32);
    DBoW2::WordId w; DBoW2::WordValue wt; DBoW2::NodeId nid = 0;
    voc->transform(f, w, wt, &nid, levelsup);
    *word = w; *weight = wt; *node = nid;
}

// FORB::distance (Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101), the reference's second copy of ORBmatcher::DescriptorDistance
// (src/ORBmatcher.cc:2383-2403, identical SWAR popcount): pins the Hamming kernels to reference code.
int ref_forb_distance(const uint8_t* a, const uint8_t* b) {
    cv::Mat A(1, 32, CV_8U), B(1, 32, CV_8U);
    memcpy(A.data, a, 32); memcpy(B.data, b, 32);
    return DBoW2::FORB::distance(A, B);
}

double ref_voc_score(void* h, const unsigned* id1, const double* v1, int n1, const unsigned* id2, const double* v2, int n2) {
    ORBVocabulary* voc = (ORBVocabulary*)h;
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; i++) a.insert(a.end(), DBoW2::BowVector::value_type(id1[i], v1[i]));
    for (int i = 0; i < n2; i++) b.insert(b.end(), DBoW2::BowVector::value_type(id2[i], v2[i]));
    return voc->score(a, b);
}

}  // extern "C"
