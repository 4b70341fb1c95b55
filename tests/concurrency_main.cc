// Concurrency parity (SURVEY.md §8b threading): the reference enters the extractor from two FRESH threads per frame
// (Frame.cc:122-125), one extractor object per eye, while three long-lived threads (Tracking, LocalMapping, LoopClosing)
// run ORBmatcher calls.  This program first computes every result once, single-threaded (the Python driver checks THOSE
// against the oracle), then repeats the reference's threading shape: per iteration two new std::threads call the drop-in
// ORB_SLAM3::ORBextractor objects while three worker threads hammer msorb_search_by_projection_mps / msorb_search_by_bow /
// msorb_fuse_search on frames of their own; every concurrent result must equal the single-threaded one bit for bit.
// usage: concurrency <in.bin> <out.bin> <iterations>      exit code = number of mismatching results (0 = pass)
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "msorb.h"

template <class T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(99); }
    return v;
}
template <class T>
static void wr(FILE* f, const std::vector<T>& v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }

struct EyeOut {
    int mono = 0;
    std::vector<cv::KeyPoint> kps;
    std::vector<unsigned char> desc;
    bool operator==(const EyeOut& o) const {
        return mono == o.mono && kps.size() == o.kps.size() && desc == o.desc &&
               (kps.empty() || !memcmp(kps.data(), o.kps.data(), kps.size() * sizeof(cv::KeyPoint)));
    }
};
static EyeOut run_eye(ORB_SLAM3::ORBextractor& ex, const cv::Mat& im) {
    EyeOut e;
    cv::Mat d;
    std::vector<int> lap = {0, 0};
    e.mono = ex(im, cv::Mat(), e.kps, d, lap);
    e.desc.resize(e.kps.size() * 32);
    for (size_t i = 0; i < e.kps.size(); i++) memcpy(&e.desc[i * 32], d.ptr<unsigned char>((int)i), 32);
    return e;
}

int main(int argc, char** argv) {
    if (argc < 4) return 98;
    const int iters = atoi(argv[3]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 97;
    const auto hdr = rd<int>(f, 8);   // rows cols N nlevels M(sbp) Mf(fuse) n1 n2(bow)
    const int rows = hdr[0], cols = hdr[1], N = hdr[2], nl = hdr[3], M = hdr[4], Mf = hdr[5], n1 = hdr[6], n2 = hdr[7];
    auto imgL = rd<unsigned char>(f, (size_t)rows * cols), imgR = rd<unsigned char>(f, (size_t)rows * cols);
    // the matcher threads' frame
    const auto kps = rd<msorb_keypoint>(f, N);
    const auto desc = rd<unsigned char>(f, (size_t)32 * N);
    const auto ur = rd<float>(f, N);
    const auto scale = rd<float>(f, nl), inv_sigma2 = rd<float>(f, nl);
    // SearchByProjection table
    const auto inView = rd<unsigned char>(f, M), bad = rd<unsigned char>(f, M), spars = rd<unsigned char>(f, M);
    const auto px = rd<float>(f, M), py = rd<float>(f, M), pxr = rd<float>(f, M), depth = rd<float>(f, M);
    const auto level = rd<int>(f, M);
    const auto vcos = rd<float>(f, M);
    const auto mdesc = rd<unsigned char>(f, (size_t)32 * M);
    const auto obs = rd<int>(f, M);
    const auto frame_mp0 = rd<int>(f, N);
    // Fuse queries
    const auto fvalid = rd<unsigned char>(f, Mf);
    const auto fu = rd<float>(f, Mf), fv = rd<float>(f, Mf), fur = rd<float>(f, Mf);
    const auto flevel = rd<int>(f, Mf);
    const auto fradius = rd<float>(f, Mf);
    const auto fdesc = rd<unsigned char>(f, (size_t)32 * Mf);
    // BoW pair
    const auto bd1 = rd<unsigned char>(f, (size_t)32 * n1), bd2 = rd<unsigned char>(f, (size_t)32 * n2);
    const auto bvalid1 = rd<unsigned char>(f, n1), bavail2 = rd<unsigned char>(f, n2);
    const int nn1 = rd<int>(f, 1)[0];
    const auto b1node = rd<int>(f, nn1), b1begin = rd<int>(f, nn1 + 1);
    const auto b1feat = rd<int>(f, b1begin.back());
    const int nn2 = rd<int>(f, 1)[0];
    const auto b2node = rd<int>(f, nn2), b2begin = rd<int>(f, nn2 + 1);
    const auto b2feat = rd<int>(f, b2begin.back());
    const auto bang1 = rd<float>(f, n1), bang2 = rd<float>(f, n2);
    fclose(f);

    cv::Mat imL(rows, cols, CV_8UC1, imgL.data(), (size_t)cols), imR(rows, cols, CV_8UC1, imgR.data(), (size_t)cols);
    ORB_SLAM3::ORBextractor exL(2000, 1.2f, 8, 20, 7), exR(2000, 1.2f, 8, 20, 7);   // Tracking.cc:595-596

    // every matcher thread owns its frame handle (one handle per thread of use, include/msorb.h)
    auto sbp = [&](msorb_frame* fr, std::vector<int>& frame_mp, int& nm) {
        frame_mp = frame_mp0;
        return msorb_search_by_projection_mps(fr, M, inView.data(), bad.data(), spars.data(), px.data(), py.data(), pxr.data(), depth.data(),
                                              level.data(), vcos.data(), mdesc.data(), obs.data(), frame_mp.data(), 3.0f, 1, 60.0f, 0.8f, &nm);
    };
    auto fuse = [&](msorb_frame* fr, std::vector<int>& bi, std::vector<int>& bdist) {
        bi.assign(Mf, -1); bdist.assign(Mf, 0);
        return msorb_fuse_search(fr, inv_sigma2.data(), nl, Mf, fvalid.data(), fu.data(), fv.data(), fur.data(), flevel.data(),
                                 fradius.data(), fdesc.data(), bi.data(), bdist.data());
    };
    auto bow = [&](std::vector<int>& m12, std::vector<int>& m21, int& nm) {
        m12.assign(n1, -1); m21.assign(n2, -1);
        msorb_bow_pair P{};
        P.n1 = n1; P.n2 = n2; P.desc1 = bd1.data(); P.desc2 = bd2.data(); P.valid1 = bvalid1.data(); P.avail2 = bavail2.data();
        P.fv1_nodes = nn1; P.fv1_node = b1node.data(); P.fv1_begin = b1begin.data(); P.fv1_feat = b1feat.data();
        P.fv2_nodes = nn2; P.fv2_node = b2node.data(); P.fv2_begin = b2begin.data(); P.fv2_feat = b2feat.data();
        P.angle1 = bang1.data(); P.angle2 = bang2.data(); P.match12 = m12.data(); P.match21 = m21.data();
        const int rc = msorb_search_by_bow(0, &P, 1, 50, 1, 0.7f, 1, nullptr);
        nm = P.nmatches;
        return rc;
    };
    auto make_frame = [&]() {
        msorb_frame* fr = nullptr;
        if (msorb_frame_create(0, &fr) ||
            msorb_frame_set(fr, kps.data(), N, desc.data(), ur.data(), 0.f, (float)cols, 0.f, (float)rows, scale.data(), nl)) {
            fprintf(stderr, "frame: %s\n", msorb_last_error());
            exit(96);
        }
        return fr;
    };

    // ---- single-threaded baselines
    const EyeOut baseL = run_eye(exL, imL), baseR = run_eye(exR, imR);
    msorb_frame* fr0 = make_frame();
    std::vector<int> base_mp, base_bi, base_bd, base_m12, base_m21;
    int base_nm = 0, base_bn = 0;
    if (sbp(fr0, base_mp, base_nm) || fuse(fr0, base_bi, base_bd) || bow(base_m12, base_m21, base_bn)) { fprintf(stderr, "baseline: %s\n", msorb_last_error()); return 95; }
    msorb_frame_destroy(fr0);
    FILE* o = fopen(argv[2], "wb");
    for (const EyeOut* e : {&baseL, &baseR}) {
        const int n = (int)e->kps.size();
        fwrite(&e->mono, 4, 1, o); fwrite(&n, 4, 1, o);
        wr(o, e->kps); wr(o, e->desc);
    }
    fwrite(&base_nm, 4, 1, o); wr(o, base_mp);
    wr(o, base_bi); wr(o, base_bd);
    fwrite(&base_bn, 4, 1, o); wr(o, base_m12); wr(o, base_m21);

    // ---- the reference's threading shape
    std::atomic<int> mismatches{0}, errors{0}, stop{0};
    std::atomic<long> matcher_calls{0};
    std::thread tracking([&] {        // Tracking thread: SearchLocalPoints
        msorb_frame* fr = make_frame();
        std::vector<int> mp; int nm;
        while (!stop.load()) {
            if (sbp(fr, mp, nm)) errors++;
            else if (nm != base_nm || mp != base_mp) mismatches++;
            matcher_calls++;
        }
        msorb_frame_destroy(fr);
    });
    std::thread mapping([&] {         // LocalMapping thread: SearchInNeighbors -> Fuse
        msorb_frame* fr = make_frame();
        std::vector<int> bi, bd;
        while (!stop.load()) {
            if (fuse(fr, bi, bd)) errors++;
            else if (bi != base_bi || bd != base_bd) mismatches++;
            matcher_calls++;
        }
        msorb_frame_destroy(fr);
    });
    std::thread looping([&] {         // LoopClosing thread: SearchByBoW
        std::vector<int> m12, m21; int nm;
        while (!stop.load()) {
            if (bow(m12, m21, nm)) errors++;
            else if (nm != base_bn || m12 != base_m12 || m21 != base_m21) mismatches++;
            matcher_calls++;
        }
    });
    for (int it = 0; it < iters; it++) {
        EyeOut l, r;
        std::thread threadLeft([&] { l = run_eye(exL, imL); });      // Frame.cc:122-125: fresh threads every frame
        std::thread threadRight([&] { r = run_eye(exR, imR); });
        threadLeft.join();
        threadRight.join();
        if (!(l == baseL)) mismatches++;
        if (!(r == baseR)) mismatches++;
    }
    stop.store(1);
    tracking.join(); mapping.join(); looping.join();
    const int mm = mismatches.load(), ee = errors.load();
    const long calls = matcher_calls.load();
    fwrite(&mm, 4, 1, o); fwrite(&ee, 4, 1, o); fwrite(&calls, 8, 1, o);
    fclose(o);
    fprintf(stderr, "concurrency: %d extraction iterations x 2 eyes, %ld matcher calls on 3 threads, %d mismatches, %d errors\n", iters, calls, mm, ee);
    return mm + ee > 90 ? 90 : mm + ee;
}
