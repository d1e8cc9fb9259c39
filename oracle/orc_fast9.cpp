// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED (OpenCV internals).
//
// Restates cv::FAST(img, keypoints, threshold = 10, nonmaxSuppression = true) (type TYPE_9_16) as called at
// /root/reference/OpenCVFASTFeatureExtractor.cpp:8 from the published OpenCV 3.4 algorithm (features2d/fast.cpp FAST_t<16>,
// fast_score.cpp cornerScore<16>): a pixel is a corner if 9 contiguous pixels of the 16-pixel Bresenham circle of radius 3 are all
// darker than v - t or all brighter than v + t; its score is the largest threshold for which it stays a corner (max over the
// 9-arcs of the min |difference|, minus 1 on the way out); with non-maximum suppression a corner survives if its score is strictly
// greater than the scores of its 8 neighbours; rows 3..rows-4 and columns 3..cols-4 of the (sub-)image are examined; keypoints come
// out in raster order with response = score. The adapter (:10-19) keeps the first `max` of them.
#include "orc_api.h"
#include <cstring>

namespace orc {

static const int FAST_OFF[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_corner_score(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    const int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]); a = std::min(a, (int)d[k + 5]); a = std::min(a, (int)d[k + 6]); a = std::min(a, (int)d[k + 7]); a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]); b = std::max(b, (int)d[k + 4]); b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]); b = std::max(b, (int)d[k + 7]); b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// img: full image, row stride W; the (cx0, cy0, cw, ch) view is treated as the image (cv::FAST never reads outside its Mat).
// Returns the number of keypoints written (<= max_kp when max_kp > 0; the reference's loop takes nothing for max <= 0).
int fast9_cell(const uint8_t* img, int W, int cx0, int cy0, int cw, int ch, int threshold, bool nonmax, int max_kp, int* out_xy, float* out_response) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = FAST_OFF[k][0] + FAST_OFF[k][1] * W;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    std::vector<uint8_t> bufmem((size_t)cw * 3, 0);
    uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cw, bufmem.data() + 2 * cw};
    std::vector<int> cp[3];
    int n = 0;
    const uint8_t* base = img + (size_t)cy0 * W + cx0;
    for (int i = 3; i < ch - 2; i++) {
        const uint8_t* ptr = base + (size_t)i * W + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        std::vector<int>& cornerpos = cp[(i - 3) % 3];
        memset(curr, 0, cw);
        cornerpos.clear();
        if (i < ch - 3) {
            for (int j = 3; j < cw - 3; j++, ptr++) {
                const int v = ptr[0];
                const uint8_t* t = tab - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
                if (d & 1) {
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        const int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos.push_back(j);
                                if (nonmax) curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
                if (d & 2) {
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        const int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos.push_back(j);
                                if (nonmax) curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
            }
        }
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        const std::vector<int>& pc = cp[(i - 4 + 3) % 3];
        for (int j : pc) {
            const int score = prev[j];
            if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                            score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                if (max_kp <= 0 || n >= max_kp) return n;   // the adapter's `if (i >= max) break`
                out_xy[2 * n] = j; out_xy[2 * n + 1] = i - 1;
                out_response[n] = (float)score;
                n++;
            }
        }
    }
    return n;
}

}  // namespace orc

extern "C" {
int orc_fast9_cell(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, int threshold, int nonmax, int max_kp, int* out_xy, float* out_response) {
    (void)H;
    return orc::fast9_cell(img, W, cx0, cy0, cw, ch, threshold, nonmax != 0, max_kp, out_xy, out_response);
}
}
