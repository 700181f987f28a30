// include/orb_slam3_amd/ORBVocabulary.h — drop-in for the one use the front-end makes of ORB_SLAM3::ORBVocabulary
// (= DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>, /root/reference/include/ORBVocabulary.h:28-29):
//     mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);          // Frame::ComputeBoW, src/Frame.cc:984-997
// The class below keeps that signature and DBoW2's own result types (BowVector = std::map<WordId, WordValue>, FeatureVector =
// std::map<NodeId, std::vector<unsigned>>, Thirdparty/DBoW2/DBoW2/BowVector.h:58, FeatureVector.h:25), fills them from the device
// transform of include/orbx.h (orbv_*), and loads the same ORBvoc.txt (loadFromTextFile, TemplatedVocabulary.h:1338).
// Requires DBoW2's BowVector.h / FeatureVector.h on the include path (they are part of ORB-SLAM3's tree).
#ifndef ORB_SLAM3_AMD_ORBVOCABULARY_H
#define ORB_SLAM3_AMD_ORBVOCABULARY_H

#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
#include "DBoW2/BowVector.h"
#include "DBoW2/FeatureVector.h"
#include "../orbx.h"

namespace ORB_SLAM3 {

class ORBVocabularyAmd {
public:
    // `h` supplies the GPU stream and scratch memory (any extractor handle of the device the vocabulary should live on)
    explicit ORBVocabularyAmd(orbx_extractor* h) : h_(h), v_(nullptr) {}
    ~ORBVocabularyAmd() { if (v_) orbv_destroy(v_); }
    ORBVocabularyAmd(const ORBVocabularyAmd&) = delete;
    ORBVocabularyAmd& operator=(const ORBVocabularyAmd&) = delete;

    bool loadFromTextFile(const std::string& filename) {
        if (v_) { orbv_destroy(v_); v_ = nullptr; }
        return orbv_load_text(h_, filename.c_str(), &v_) == ORBX_OK;
    }
    unsigned int size() const { return (unsigned int)orbv_words(v_); }
    bool empty() const { return v_ == nullptr || orbv_words(v_) == 0; }

    // TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup), TemplatedVocabulary.h:1127-1195
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
        v.clear(); fv.clear();
        if (empty()) return;
        const int n = (int)features.size();
        std::vector<uint8_t> desc((size_t)n * 32 + 1);
        for (int i = 0; i < n; i++) memcpy(&desc[(size_t)i * 32], features[i].ptr(), 32);     // Converter::toDescriptorVector rows
        std::vector<uint32_t> bow_id(n + 1), fv_node(n + 1), fv_feat(n + 1);
        std::vector<double> bow_val(n + 1);
        std::vector<int> fv_start(n + 2);
        int n_bow = 0, n_fv = 0;
        if (orbv_transform(v_, h_, desc.data(), n, levelsup, nullptr, nullptr, bow_id.data(), bow_val.data(), &n_bow, fv_node.data(), fv_start.data(),
                           fv_feat.data(), &n_fv) != ORBX_OK) return;
        for (int k = 0; k < n_bow; k++) v.insert(v.end(), DBoW2::BowVector::value_type(bow_id[k], bow_val[k]));          // ascending ids
        for (int m = 0; m < n_fv; m++) {
            DBoW2::FeatureVector::iterator it = fv.insert(fv.end(), DBoW2::FeatureVector::value_type(fv_node[m], std::vector<unsigned int>()));
            it->second.assign(fv_feat.begin() + fv_start[m], fv_feat.begin() + fv_start[m + 1]);
        }
    }

private:
    orbx_extractor* h_;
    orbv_vocabulary* v_;
};

}  // namespace ORB_SLAM3
#endif
