// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference visual-odometry hot path (JeanElsner/practical-multi-view).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (practical-multi-view_amd/) never links, imports or calls anything under oracle/.
//
// PARITY STATUS
//   * pinned by in-repo reference source (arithmetic fully visible in /root/reference):
//       ShiTomasi extractor, ProjectionResidual, Feature3D transforms/projectPoint,
//       grid/ROI geometry, hasNeighbor, BA window/schedule, motionHeuristics.
//   * PARITY UNPINNED (arithmetic lives in OpenCV >=3.4 / Ceres >=1.13, which are not vendored
//     in the reference and not installed in this image; the reference ships no tests, fixtures
//     or golden vectors): goodFeaturesToTrack, calcOpticalFlowPyrLK, solvePnPRansac/EPnP,
//     findEssentialMat/recoverPose, ceres::Solve.  These are restated from the published
//     algorithms; every place where the published algorithm leaves a
//     floating-point summation order or tie-break open, this restatement fixes one and says so.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <algorithm>

namespace orc {

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// cvRound(float): round-half-to-even (SSE cvtss2si semantics)
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }

struct Image8 {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image8() {}
    Image8(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    inline uint8_t px(int x, int y) const { return d[(size_t)y * w + x]; }
    // padded access as in an OpenCV pyramid level padded with BORDER_REFLECT_101
    inline uint8_t pxr(int x, int y) const { return d[(size_t)reflect101(y, h) * w + reflect101(x, w)]; }
};

}  // namespace orc
