// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h header).  PARITY UNPINNED: the reference holds no golden
// vectors for these paths and cannot be built here (OpenCV absent); this file restates the published code.
//
// (1) DBoW2 vocabulary transform as ORB-SLAM3 calls it from Frame::ComputeBoW (src/Frame.cc:670-677) and
//     KeyFrame::ComputeBoW: TemplatedVocabulary<FORB::TDescriptor,FORB>::transform(features, BowVector&,
//     FeatureVector&, levelsup) — /root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1123-1191,
//     the per-feature descent :1218-1259, BowVector::addWeight / addIfNotExist / normalize
//     (BowVector.cpp:36-94), FeatureVector::addFeature (FeatureVector.cpp:30-45), FORB::distance
//     (FORB.cpp:77-98), tree construction as loadFromTextFile builds it (TemplatedVocabulary.h:1338-1423).
//     Containers are the reference's own (std::map), so iteration / accumulation order is the reference's.
//     One documented deviation from "as compiled": when a leaf is reached above level L-levelsup the
//     reference leaves `NodeId nid` unwritten (:1153,1245-1246 — an uninitialised read); here, and in the
//     device path, nid then keeps the value of the previous feature (0 for the first).
// (2) MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:351-429): least-median descriptor choice.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <climits>
#include <map>
#include <vector>

namespace {

// FORB::distance, FORB.cpp:77-98 (same SWAR popcount as ORBmatcher::DescriptorDistance)
int forb_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct Node {  // TemplatedVocabulary.h:297-329
    double weight = 0;
    std::vector<int> children;
    int parent = 0;
    uint8_t descriptor[32] = {0};
    int word_id = 0;
    bool isLeaf() const { return children.empty(); }
};

struct Vocabulary {
    int k, L, scoring, weighting;
    std::vector<Node> nodes;
    int n_words = 0;
};

}  // namespace

extern "C" {

// Tree exactly as loadFromTextFile leaves it: node i (1-based file line) has parent[i], the leaf flag decides
// whether a word id is assigned (in file order), children lists fill in node-id order.
void* orc_vocab_create(int k, int L, int scoring, int weighting, int n_nodes, const int* parent,
                       const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights) {
    Vocabulary* v = new Vocabulary;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->nodes.resize(n_nodes);
    for (int i = 1; i < n_nodes; i++) {
        Node& n = v->nodes[i];
        n.parent = parent[i];
        v->nodes[parent[i]].children.push_back(i);
        std::memcpy(n.descriptor, descriptors + (size_t)i * 32, 32);
        n.weight = weights[i];
        if (is_leaf[i]) n.word_id = v->n_words++;
    }
    return v;
}
void orc_vocab_destroy(void* h) { delete (Vocabulary*)h; }

// transform(features, v, fv, levelsup).  Outputs: BowVector as ascending (word, value), FeatureVector as
// ascending node ids with CSR feature lists (features in insertion = ascending index order).
// Also per-feature word / node / weight (debug & device cross-check).
int orc_bow_transform(void* h, const uint8_t* desc, int n, int levelsup, int* bow_word, double* bow_value, int* n_bow,
                      int* fv_node, int* fv_begin, int* fv_feat, int* n_fv, int* feat_word, int* feat_node,
                      double* feat_weight) {
    const Vocabulary& V = *(const Vocabulary*)h;
    std::map<unsigned, double> v;                       // BowVector
    std::map<unsigned, std::vector<unsigned>> fv;       // FeatureVector
    *n_bow = *n_fv = 0;
    fv_begin[0] = 0;
    if (V.n_words == 0) return 0;                       // empty(): m_words.empty() (:1132-1135)
    // mustNormalize (ScoringObject.h:74-89): all scorings normalise (L2 scoring with L2, others L1) but DOT_PRODUCT
    const bool must = V.scoring != 5;
    const bool l2 = V.scoring == 1;
    unsigned nid = 0;                                   // see header note on the uninitialised read
    for (int i = 0; i < n; i++) {
        const uint8_t* f = desc + (size_t)i * 32;
        // :1218-1259
        const int nid_level = V.L - levelsup;
        if (nid_level <= 0) nid = 0;
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const std::vector<int>& nodes = V.nodes[final_id].children;
            final_id = nodes[0];
            double best_d = forb_distance(f, V.nodes[final_id].descriptor);
            for (size_t c = 1; c < nodes.size(); c++) {
                const int id = nodes[c];
                const double d = forb_distance(f, V.nodes[id].descriptor);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!V.nodes[final_id].isLeaf());
        const unsigned id = V.nodes[final_id].word_id;
        const double w = V.nodes[final_id].weight;
        if (feat_word) feat_word[i] = (int)id;
        if (feat_node) feat_node[i] = (int)nid;
        if (feat_weight) feat_weight[i] = w;
        if (w > 0) {
            if (V.weighting == 0 || V.weighting == 1) {  // TF_IDF, TF: addWeight (:1157)
                auto it = v.lower_bound(id);
                if (it != v.end() && !(id < it->first)) it->second += w;
                else v.insert(it, {id, w});
            } else {                                     // IDF, BINARY: addIfNotExist (:1183)
                auto it = v.lower_bound(id);
                if (it == v.end() || id < it->first) v.insert(it, {id, w});
            }
            fv[nid].push_back((unsigned)i);
        }
    }
    if ((V.weighting == 0 || V.weighting == 1) && !v.empty() && !must) {  // :1162-1168
        const double nd = (double)v.size();
        for (auto& e : v) e.second /= nd;
    }
    if (must) {  // BowVector::normalize, BowVector.cpp:64-88
        double norm = 0.0;
        if (!l2) {
            for (auto& e : v) norm += std::fabs(e.second);
        } else {
            // the reference is built -O3 -march=native (GCC contracts a*b+c): written as the fma it compiles to
            for (auto& e : v) norm = std::fma(e.second, e.second, norm);
            norm = std::sqrt(norm);
        }
        if (norm > 0.0)
            for (auto& e : v) e.second /= norm;
    }
    int nb = 0;
    for (auto& e : v) { bow_word[nb] = (int)e.first; bow_value[nb] = e.second; nb++; }
    *n_bow = nb;
    int nf = 0, pos = 0;
    for (auto& e : fv) {
        fv_node[nf] = (int)e.first;
        fv_begin[nf] = pos;
        for (unsigned x : e.second) fv_feat[pos++] = (int)x;
        nf++;
    }
    fv_begin[nf] = pos;
    *n_fv = nf;
    return 0;
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:351-429) for n_points map points; point p owns the
// descriptors [obs_begin[p], obs_begin[p+1]) in vDescriptors order.  best_idx = index inside the point's list
// (-1: no descriptor, the reference returns early :393-394), best_median = BestMedian.
void orc_distinctive_descriptors(const uint8_t* desc, const int* obs_begin, int n_points, int* best_idx,
                                 int* best_median) {
    for (int p = 0; p < n_points; p++) {
        const int b = obs_begin[p];
        const size_t N = (size_t)(obs_begin[p + 1] - b);
        if (N == 0) { best_idx[p] = -1; if (best_median) best_median[p] = INT_MAX; continue; }
        std::vector<float> D(N * N);  // float Distances[N][N] (:399)
        for (size_t i = 0; i < N; i++) {
            D[i * N + i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                const int dij = forb_distance(desc + (size_t)(b + i) * 32, desc + (size_t)(b + j) * 32);
                D[i * N + j] = dij;
                D[j * N + i] = dij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(D.begin() + i * N, D.begin() + (i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[0.5 * (N - 1)];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best_idx[p] = BestIdx;
        if (best_median) best_median[p] = BestMedian;
    }
}

}  // extern "C"
