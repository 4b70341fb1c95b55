// Host-side bindings of the two bag-of-words / map-point rows, over the reference's own types by name
// (templates: compile inside MS-SLAM; tests/dropin_bow_main.cc compiles them against stand-ins):
//
//   msorb_host::Vocabulary                          RAII handle of the device-resident DBoW2 tree; load it once next to
//                                                   mpVocabulary->loadFromTextFile(strVocFile) (src/System.cc)
//   msorb_host::ComputeBoW(voc, F)                  body of Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:670-677):
//                                                   fills F.mBowVec (DBoW2::BowVector = std::map<WordId, WordValue>) and
//                                                   F.mFeatVec (DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned>>)
//   msorb_host::DistinctiveDescriptorIndices(...)   the choice made by MapPoint::ComputeDistinctiveDescriptors
//                                                   (src/MapPoint.cc:395-424) for many map points in one device call: the
//                                                   caller collects each point's vDescriptors exactly as :358-391 does and
//                                                   assigns mDescriptor = vDescriptors[best].clone() under mMutexFeatures.
#ifndef MSORB_BOW_DEVICE_H
#define MSORB_BOW_DEVICE_H

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "msorb.h"

namespace ORB_SLAM3 {
namespace msorb_host {
#ifndef MSORB_HOST_FAIL_CALL
#define MSORB_HOST_FAIL_CALL
// a failed call of the C ABI: the application's fatal-error callback first (msorb_set_fatal_callback), then std::runtime_error
[[noreturn]] inline void fail_call(const char* what) {
    const std::string msg = std::string(what) + ": " + msorb_last_error();
    msorb_notify_fatal(MSORB_E_HIP, msg.c_str());
    throw std::runtime_error(msg);
}
#endif

class Vocabulary {
public:
    Vocabulary(const std::string& text_file, int device = 0) {
        if (msorb_vocabulary_load_text(device, text_file.c_str(), &h_) != MSORB_OK)
            fail_call("msorb_vocabulary_load_text");
    }
    ~Vocabulary() { msorb_vocabulary_destroy(h_); }
    Vocabulary(const Vocabulary&) = delete;
    Vocabulary& operator=(const Vocabulary&) = delete;
    msorb_vocabulary* get() const { return h_; }

private:
    msorb_vocabulary* h_ = nullptr;
};

// if(mBowVec.empty()) { vCurrentDesc = toDescriptorVector(mDescriptors); mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4); }
template <class FrameT>
void ComputeBoW(const Vocabulary& voc, FrameT& F, int levelsup = 4) {
    if (!F.mBowVec.empty()) return;
    const int n = F.mDescriptors.rows;
    F.mBowVec.clear();
    F.mFeatVec.clear();
    if (n == 0) return;
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(i), 32);
    std::vector<int> word(n), node(n), fbeg(n + 1), feat(n);
    std::vector<double> value(n);
    int nb = 0, nf = 0;
    if (msorb_bow_transform(voc.get(), desc.data(), n, levelsup, word.data(), value.data(), &nb, node.data(), fbeg.data(),
                            feat.data(), &nf, nullptr, nullptr, nullptr) != MSORB_OK)
        fail_call("msorb_bow_transform");
    using BowValue = typename std::remove_reference<decltype(F.mBowVec)>::type::value_type;
    using FeatValue = typename std::remove_reference<decltype(F.mFeatVec)>::type::value_type;
    for (int i = 0; i < nb; i++)  // ascending word id: hinted insertion at the end is O(1)
        F.mBowVec.insert(F.mBowVec.end(), BowValue((typename BowValue::first_type)word[i], (typename BowValue::second_type)value[i]));
    for (int r = 0; r < nf; r++) {
        typename FeatValue::second_type list(feat.begin() + fbeg[r], feat.begin() + fbeg[r + 1]);
        F.mFeatVec.insert(F.mFeatVec.end(), FeatValue((typename FeatValue::first_type)node[r], std::move(list)));
    }
}

// perPoint[p] = the vDescriptors of map point p (cv::Mat rows of 32 bytes, in the order MapPoint.cc:372-391 pushes them).
// Returns BestIdx per point (-1 for a point without descriptors: the reference returns early, :393-394).
template <class MatT>
std::vector<int> DistinctiveDescriptorIndices(const std::vector<std::vector<MatT>>& perPoint, int device = 0) {
    const int P = (int)perPoint.size();
    std::vector<int> begin(P + 1, 0);
    for (int p = 0; p < P; p++) begin[p + 1] = begin[p] + (int)perPoint[p].size();
    std::vector<uint8_t> desc((size_t)begin[P] * 32);
    for (int p = 0; p < P; p++)
        for (size_t k = 0; k < perPoint[p].size(); k++)
            std::memcpy(&desc[(size_t)(begin[p] + (int)k) * 32], perPoint[p][k].template ptr<unsigned char>(0), 32);
    std::vector<int> best(P, -1);
    if (P && msorb_distinctive_descriptors(device, desc.data(), begin.data(), P, best.data(), nullptr, nullptr) != MSORB_OK)
        fail_call("msorb_distinctive_descriptors");
    return best;
}

}  // namespace msorb_host
}  // namespace ORB_SLAM3

#endif
