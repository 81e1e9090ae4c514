// Drives the drop-in classes exactly like the reference's callers do (Tracking.cc:119-125 constructs the extractor,
// Frame::Frame calls ExtractORB, Tracking.cc:599-600 calls SearchForInitialization) and dumps the results for pytest.
// usage: test_dropin W H nfeatures in0.raw in1.raw out.bin [vocabulary.txt [threads|- [maps.bin]]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "ORBVocabulary.h"
#include "Frame.h"

float ORB_SLAM2::Frame::mnMinX, ORB_SLAM2::Frame::mnMaxX, ORB_SLAM2::Frame::mnMinY, ORB_SLAM2::Frame::mnMaxY;

static void dump(FILE* f, const ORB_SLAM2::Frame& F)
{
    int n = F.N; fwrite(&n, 4, 1, f);
    if (n) { fwrite(&F.mvKeys[0], sizeof(cv::KeyPoint), n, f); for (int i = 0; i < n; i++) fwrite(F.mDescriptors.ptr(i), 1, 32, f); }
}

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const int W = atoi(argv[1]), H = atoi(argv[2]), N = atoi(argv[3]);
    cv::Mat im[2];
    for (int k = 0; k < 2; k++) { im[k].create(H, W, CV_8UC1); FILE* f = fopen(argv[4 + k], "rb"); if (!f || fread(im[k].data, 1, (size_t)W * H, f) != (size_t)W * H) return 3; fclose(f); }
    ORB_SLAM2::ORBextractor* ex = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
    ORB_SLAM2::Frame F1(im[0], ex), F2(im[1], ex);
    std::vector<cv::Point2f> prev(F1.mvKeysUn.size());
    for (size_t i = 0; i < prev.size(); i++) prev[i] = F1.mvKeysUn[i].pt;          // Tracking.cc:590-592
    std::vector<int> matches;
    ORB_SLAM2::ORBmatcher matcher(0.9, true);
    const int nm = matcher.SearchForInitialization(F1, F2, prev, matches, 100);     // Tracking.cc:599-600
    FILE* f = fopen(argv[6], "wb");
    dump(f, F1); dump(f, F2);
    fwrite(&nm, 4, 1, f);
    if (!matches.empty()) fwrite(&matches[0], 4, matches.size(), f);
    if (!prev.empty()) fwrite(&prev[0], 8, prev.size(), f);
    int lv = ex->GetLevels(); fwrite(&lv, 4, 1, f);
    for (int l = 0; l < lv; l++) { int wh[2] = {ex->mvImagePyramid[l].cols, ex->mvImagePyramid[l].rows}; fwrite(wh, 4, 2, f); fwrite(ex->mvImagePyramid[l].data, 1, (size_t)wh[0] * wh[1], f); }
    {   // a fork that walks the pyramid instead of indexing it: range-for, at(), a copy into a std::vector<cv::Mat> - the same planes
        int l = 0; bool same = true;
        for (const cv::Mat& im : ex->mvImagePyramid) { same = same && im.data == ex->mvImagePyramid[l].data && im.cols == ex->mvImagePyramid.at(l).cols; l++; }
        const std::vector<cv::Mat>& asVector = ex->mvImagePyramid;
        if (!same || l != lv || (int)asVector.size() != lv || asVector[lv - 1].rows != ex->mvImagePyramid.back().rows) { fprintf(stderr, "mvImagePyramid iteration differs from indexing\n"); return 3; }
    }
    const int d = ORB_SLAM2::ORBmatcher::DescriptorDistance(F1.mDescriptors.row(0), F2.mDescriptors.row(0)); fwrite(&d, 4, 1, f);
    // stereo: two extractor instances like Tracking.cc:119-122, Frame::ComputeStereoMatches replaced by the GPU entry point
    ORB_SLAM2::ORBextractor* exL = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
    ORB_SLAM2::ORBextractor* exR = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
    ORB_SLAM2::Frame FL(im[0], exL), FR(im[1], exR);
    std::vector<float> uRight, depth;
    exL->ComputeStereoMatches(*exR, 386.1448f, 386.1448f / 718.856f, FL.N, uRight, depth);
    int ns = (int)uRight.size(); fwrite(&ns, 4, 1, f);
    if (ns) { fwrite(&uRight[0], 4, ns, f); fwrite(&depth[0], 4, ns, f); }
    // colour: BGR frame with channels (im[0], im[1], im[0]) like a Camera.RGB: 0 sequence entering Tracking::GrabImageMonocular
    std::vector<unsigned char> bgr((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; i++) { bgr[3 * i] = im[0].data[i]; bgr[3 * i + 1] = im[1].data[i]; bgr[3 * i + 2] = im[0].data[i]; }
    std::vector<cv::KeyPoint> kc; cv::Mat dc;
    ex->ExtractColor(&bgr[0], 3 * W, W, H, 3, false, kc, dc);
    int nc = (int)kc.size(); fwrite(&nc, 4, 1, f);
    if (nc) { fwrite(&kc[0], sizeof(cv::KeyPoint), nc, f); for (int i = 0; i < nc; i++) fwrite(dc.ptr(i), 1, 32, f); }
    fwrite(ex->mvImagePyramid[0].data, 1, (size_t)W * H, f);
    // bag of words like System.cc:68 + Frame::ComputeBoW (Frame.cc:395-402) + KeyFrameDatabase.cc:133
    if (argc > 7) {
        ORB_SLAM2::ORBVocabulary voc;
        if (!voc.loadFromTextFile(argv[7])) return 4;
        DBoW2::BowVector b1, b2, b3; DBoW2::FeatureVector f1, f2, f3;
        std::vector<cv::Mat> rows(F1.mDescriptors.rows);                                   // Converter::toDescriptorVector
        for (int i = 0; i < F1.mDescriptors.rows; i++) rows[i] = F1.mDescriptors.row(i);
        voc.transform(rows, b1, f1, 4);
        voc.transform(F2.mDescriptors, b2, f2, 4);
        std::vector<cv::KeyPoint> kk; cv::Mat dd;
        (*ex)(im[1], cv::Mat(), kk, dd);
        voc.ComputeBoW(*ex, b3, f3, 4);                                                    // descriptors of im[1] still on the device
        const DBoW2::BowVector* bs[3] = {&b1, &b2, &b3}; const DBoW2::FeatureVector* fs[3] = {&f1, &f2, &f3};
        for (int q = 0; q < 3; q++) {
            int nb = (int)bs[q]->size(); fwrite(&nb, 4, 1, f);
            for (DBoW2::BowVector::const_iterator it = bs[q]->begin(); it != bs[q]->end(); ++it) { fwrite(&it->first, 4, 1, f); fwrite(&it->second, 8, 1, f); }
            int nf = (int)fs[q]->size(); fwrite(&nf, 4, 1, f);
            for (DBoW2::FeatureVector::const_iterator it = fs[q]->begin(); it != fs[q]->end(); ++it) {
                int cnt = (int)it->second.size(); fwrite(&it->first, 4, 1, f); fwrite(&cnt, 4, 1, f); fwrite(&it->second[0], 4, cnt, f);
            }
        }
        const double sc = voc.score(b1, b2); fwrite(&sc, 8, 1, f);
        const unsigned int nw = voc.size(); fwrite(&nw, 4, 1, f);
    }
    // the reference extracts the left and the right image on two std::threads (Frame.cc:78-81): two extractor objects, two device
    // contexts, concurrent calls into the library.
    if (argc > 8 && !strncmp(argv[8], "threads", 7)) {
        int ok = 1;
        const int reps = argv[8][7] ? atoi(argv[8] + 7) : 20;                 // "threads" or "threads<N>"
        for (int rep = 0; rep < reps && ok; rep++) {
            std::vector<cv::KeyPoint> kL, kR; cv::Mat dL, dR;
            std::thread tl([&]() { (*exL)(im[0], cv::Mat(), kL, dL); }), tr([&]() { (*exR)(im[1], cv::Mat(), kR, dR); });
            tl.join(); tr.join();
            ok = (int)kL.size() == F1.N && (int)kR.size() == F2.N && !memcmp(&kL[0], &F1.mvKeys[0], sizeof(cv::KeyPoint) * kL.size()) && !memcmp(&kR[0], &F2.mvKeys[0], sizeof(cv::KeyPoint) * kR.size());
            for (int i = 0; i < F1.N && ok; i++) ok = !memcmp(dL.ptr(i), F1.mDescriptors.ptr(i), 32);
            for (int i = 0; i < F2.N && ok; i++) ok = !memcmp(dR.ptr(i), F2.mDescriptors.ptr(i), 32);
        }
        // ... and the same pair as ONE call on one extractor (ORBextractor::ExtractStereo: one context with two camera slots, the stereo matcher queued
        // behind the extraction): key points and descriptors of both images and mvuRight / mvDepth must equal the two-extractor path, call after call
        {
            const float mbf = 386.1448f, mb = 386.1448f / 718.856f;
            std::vector<float> uTwo, dTwo, uOne, dOne;
            exL->ComputeStereoMatches(*exR, mbf, mb, F1.N, uTwo, dTwo);        // exL / exR processed im[0] / im[1] last
            ORB_SLAM2::ORBextractor* exP = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
            for (int rep = 0; rep < 3 && ok; rep++) {
                std::vector<cv::KeyPoint> kL, kR, un; cv::Mat dL, dR;
                exP->ExtractStereo(im[0], im[1], kL, dL, kR, dR, mbf, mb);
                exP->UndistortKeyPoints(un);
                exP->ComputeStereoMatches(*exR, mbf, mb, (int)kL.size(), uOne, dOne);
                ok = (int)kL.size() == F1.N && (int)kR.size() == F2.N && !memcmp(&kL[0], &F1.mvKeys[0], sizeof(cv::KeyPoint) * kL.size()) && !memcmp(&kR[0], &F2.mvKeys[0], sizeof(cv::KeyPoint) * kR.size())
                     && un.size() == kL.size() && !memcmp(&un[0], &kL[0], sizeof(cv::KeyPoint) * kL.size()) && exP->HoldsStereoColumns()
                     && uOne.size() == uTwo.size() && !memcmp(&uOne[0], &uTwo[0], 4 * uOne.size()) && !memcmp(&dOne[0], &dTwo[0], 4 * dOne.size());
                for (int i = 0; i < F1.N && ok; i++) ok = !memcmp(dL.ptr(i), F1.mDescriptors.ptr(i), 32);
                for (int i = 0; i < F2.N && ok; i++) ok = !memcmp(dR.ptr(i), F2.mDescriptors.ptr(i), 32);
                // the frame-resident forms belong to the thread that bound the frame: another thread is told "not held" (and takes the host path)
                exP->BindFrame(100 + rep);
                bool other = true; std::thread th([&] { other = exP->HoldsFrame(100 + rep, (int)kL.size()); }); th.join();
                ok = ok && exP->HoldsFrame(100 + rep, (int)kL.size()) && !exP->HoldsFrame(99, (int)kL.size()) && !other;
                if (rep == 1) { std::vector<cv::KeyPoint> k1; cv::Mat d1; (*exP)(im[1], cv::Mat(), k1, d1); ok = ok && (int)k1.size() == F2.N && !exP->HoldsStereoColumns(); }      // a single-image call in between
            }
            // the pair's columns belong to the pair: a Submit in between takes them away - ComputeStereoMatches then says that there is no
            // frame state instead of handing out the pair before's mvuRight / mvDepth
            if (ok) {
                std::vector<cv::KeyPoint> kL, kR; cv::Mat dL, dR;
                exP->ExtractStereo(im[0], im[1], kL, dL, kR, dR, mbf, mb);
                std::vector<cv::Mat> b(2); b[0] = im[0]; b[1] = im[1];
                const int t = exP->Submit(b, 2);
                bool refused = false;
                try { exP->ComputeStereoMatches(*exR, mbf, mb, (int)kL.size(), uOne, dOne); } catch (const ORB_SLAM2::ORBhipError&) { refused = true; }
                std::vector<std::vector<cv::KeyPoint> > kb; std::vector<cv::Mat> db;
                exP->Collect(t, kb, db);
                ok = refused && (int)kb[0].size() == F1.N;
            }
            delete exP;
        }
        fwrite(&ok, 4, 1, f);
    }
    // distorted monocular camera (TUM1.yaml scaled to the test image) and raw stereo input rectified on the device
    if (argc > 9) {
        cv::Mat K(3, 3, CV_32F), D(5, 1, CV_32F);
        for (int i = 0; i < 9; i++) K.at<float>(i / 3, i % 3) = 0.0f;
        K.at<float>(0, 0) = 517.306408f * W / 640; K.at<float>(1, 1) = 516.469215f * H / 480; K.at<float>(0, 2) = 318.643040f * W / 640; K.at<float>(1, 2) = 255.313989f * H / 480; K.at<float>(2, 2) = 1.0f;
        const float d5[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
        for (int i = 0; i < 5; i++) D.at<float>(i, 0) = d5[i];
        ORB_SLAM2::ORBextractor* exD = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
        ORB_SLAM2::Frame D1(im[0], exD, K, D), D2(im[1], exD, K, D);
        std::vector<cv::Point2f> prevD(D1.mvKeysUn.size());
        for (size_t i = 0; i < prevD.size(); i++) prevD[i] = D1.mvKeysUn[i].pt;
        std::vector<int> matchesD;
        const int nmD = matcher.SearchForInitialization(D1, D2, prevD, matchesD, 100);
        int nd1 = D1.N, nd2 = D2.N; fwrite(&nd1, 4, 1, f); fwrite(&nd2, 4, 1, f);
        if (nd1) fwrite(&D1.mvKeysUn[0], sizeof(cv::KeyPoint), nd1, f);
        if (nd2) fwrite(&D2.mvKeysUn[0], sizeof(cv::KeyPoint), nd2, f);
        const float bnd[4] = {ORB_SLAM2::Frame::mnMinX, ORB_SLAM2::Frame::mnMinY, ORB_SLAM2::Frame::mnMaxX, ORB_SLAM2::Frame::mnMaxY}; fwrite(bnd, 4, 4, f);
        fwrite(&nmD, 4, 1, f);
        if (!matchesD.empty()) fwrite(&matchesD[0], 4, matchesD.size(), f);
        // RGB-D: the 16-bit depth map of a TUM sequence (5000 units per metre, 0 = hole) read at the key points of the distorted camera
        cv::Mat dm(H, W, CV_16U);
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) dm.at<unsigned short>(y, x) = (unsigned short)(((x + y) % 7 == 0) ? 0 : (x * 7 + y * 13) % 9000 + 2000);
        std::vector<float> uR, dZ;
        exD->ComputeStereoFromRGBD(dm, 1.0f / 5000.0f, 40.0f, D2.N, uR, dZ);          // exD processed im[1] last
        if (nd2) { fwrite(&uR[0], 4, nd2, f); fwrite(&dZ[0], 4, nd2, f); }
        delete exD;
        // maps file: W*H floats of M1 then W*H floats of M2; the raw image is im[0]
        cv::Mat M1(H, W, CV_32F), M2(H, W, CV_32F);
        FILE* mf = fopen(argv[9], "rb");
        if (!mf || fread(M1.data, 4, (size_t)W * H, mf) != (size_t)W * H || fread(M2.data, 4, (size_t)W * H, mf) != (size_t)W * H) return 5;
        fclose(mf);
        ORB_SLAM2::ORBextractor* exM = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
        exM->SetRectification(M1, M2, W, H);
        std::vector<cv::KeyPoint> kr; cv::Mat dr;
        exM->ExtractRectified(im[0], kr, dr);
        int nr = (int)kr.size(); fwrite(&nr, 4, 1, f);
        if (nr) { fwrite(&kr[0], sizeof(cv::KeyPoint), nr, f); for (int i = 0; i < nr; i++) fwrite(dr.ptr(i), 1, 32, f); }
        fwrite(exM->mvImagePyramid[0].data, 1, (size_t)W * H, f);
        delete exM;
        // Submit / Collect: two batches in flight deliver what operator() delivers; afterwards the per-frame follow-ups refuse (the
        // context holds a batch), a Submit that would re-create the context with a ticket in flight refuses, and a single-image call
        // makes the follow-ups valid again.  One int per expectation.
        {
            ORB_SLAM2::ORBextractor* exB = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
            std::vector<cv::Mat> b01(2), b10(2); b01[0] = im[0]; b01[1] = im[1]; b10[0] = im[1]; b10[1] = im[0];
            const int t0 = exB->Submit(b01, 2), t1 = exB->Submit(b10);
            std::vector<std::vector<cv::KeyPoint> > k0, k1; std::vector<cv::Mat> d0, d1;
            int refuse_grow = 0;
            std::vector<cv::Mat> three(3, im[0]);
            try { exB->Submit(three); } catch (const ORB_SLAM2::ORBhipError&) { refuse_grow = 1; }      // would destroy tickets t0, t1
            exB->Collect(t0, k0, d0); exB->Collect(t1, k1, d1);
            int same = (int)k0[0].size() == F1.N && (int)k0[1].size() == F2.N && (int)k1[0].size() == F2.N && (int)k1[1].size() == F1.N;
            same = same && !memcmp(&k0[0][0], &F1.mvKeys[0], sizeof(cv::KeyPoint) * F1.N) && !memcmp(&k1[0][0], &F2.mvKeys[0], sizeof(cv::KeyPoint) * F2.N)
                        && !memcmp(&k0[1][0], &F2.mvKeys[0], sizeof(cv::KeyPoint) * F2.N) && !memcmp(&k1[1][0], &F1.mvKeys[0], sizeof(cv::KeyPoint) * F1.N);
            for (int i = 0; i < F1.N && same; i++) same = !memcmp(d0[0].ptr(i), F1.mDescriptors.ptr(i), 32) && !memcmp(d1[1].ptr(i), F1.mDescriptors.ptr(i), 32);
            int refuse_un = 0, refuse_pyr = 0;
            std::vector<cv::KeyPoint> un;
            try { exB->UndistortKeyPoints(un); } catch (const ORB_SLAM2::ORBhipError&) { refuse_un = 1; }
            try { (void)exB->mvImagePyramid[0].cols; } catch (const ORB_SLAM2::ORBhipError&) { refuse_pyr = 1; }
            std::vector<cv::KeyPoint> ks; cv::Mat ds;
            (*exB)(im[0], cv::Mat(), ks, ds);                                                           // grows nothing: same size, batch 1 <= 2
            int valid_again = 1;
            try { exB->UndistortKeyPoints(un); valid_again = (int)un.size() == F1.N && exB->mvImagePyramid[0].cols == W; } catch (const ORB_SLAM2::ORBhipError&) { valid_again = 0; }
            const int flags[5] = {same, refuse_grow, refuse_un, refuse_pyr, valid_again};
            fwrite(flags, 4, 5, f);
            delete exB;
        }
        // Per-call image-size freedom (the reference re-allocates per call and takes any size, ORBextractor.cc:1043-1056): ONE extractor object
        // alternates between the full image and a crop of it (a cv::Mat view: same data, same row step) for three rounds.  Every round delivers the
        // same bytes per size, the follow-ups (UndistortKeyPoints, mvImagePyramid) belong to the last call's size, and from the second round on a
        // size change is a switch between two kept contexts, not a context creation (the times are reported, not asserted).
        {
            ORB_SLAM2::ORBextractor* exS = new ORB_SLAM2::ORBextractor(N, 1.2f, 8, 20, 7);
            const cv::Mat crop = im[0].rowRange(0, H - 24).colRange(0, W - 32);
            std::vector<cv::KeyPoint> kFull0, kCrop0, k; cv::Mat dFull0, dCrop0, d;
            int same = 1; double ms[6];
            for (int round = 0; round < 3; round++)
                for (int which = 0; which < 2; which++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    (*exS)(which ? crop : im[0], cv::Mat(), k, d);
                    ms[2 * round + which] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    std::vector<cv::KeyPoint>& k0 = which ? kCrop0 : kFull0; cv::Mat& d0 = which ? dCrop0 : dFull0;
                    if (round == 0) { k0 = k; d0 = d.clone(); }
                    else {
                        same = same && k.size() == k0.size() && !memcmp(&k[0], &k0[0], sizeof(cv::KeyPoint) * k.size());
                        for (size_t i = 0; i < k.size() && same; i++) same = !memcmp(d.ptr((int)i), d0.ptr((int)i), 32);
                    }
                    std::vector<cv::KeyPoint> un; exS->UndistortKeyPoints(un);
                    same = same && un.size() == k.size() && exS->mvImagePyramid[0].cols == (which ? W - 32 : W) && exS->mvImagePyramid[0].rows == (which ? H - 24 : H);
                }
            same = same && (int)kFull0.size() == F1.N && !memcmp(&kFull0[0], &F1.mvKeys[0], sizeof(cv::KeyPoint) * F1.N);
            fwrite(&same, 4, 1, f);
            int ncrop = (int)kCrop0.size(); fwrite(&ncrop, 4, 1, f);
            if (ncrop) { fwrite(&kCrop0[0], sizeof(cv::KeyPoint), ncrop, f); for (int i = 0; i < ncrop; i++) fwrite(dCrop0.ptr(i), 1, 32, f); }
            fwrite(ms, 8, 6, f);
            delete exS;
        }
    }
    fclose(f);
    delete ex; delete exL; delete exR;
    return 0;
}
