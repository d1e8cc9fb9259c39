// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// kNNFeatureMatcher's arithmetic, PINNED by in-repo source (/root/reference/kNNFeatureMatcher.cpp:3-122, include/kNNFeatureMatcher.h:
// window = 15, threshold = 2, n = 7 neighbours): for every source feature
//   getNearestNeighbors (:63-101)  n passes over the candidate list; a pass takes the FIRST candidate with the smallest Chebyshev
//                                  distance (Feature::distance, Feature.cpp:9-15) among those whose coordinates differ from the source
//                                  and from every neighbour chosen so far; `dist == 0` means "nothing yet"; a pass that finds
//                                  nothing pushes the previous pick again (initially the default Feature at (0,0));
//   compareFeatures (:103-122)     17x17 window (ceil(15/2) = 8 to each side), pixels outside either image skipped, float
//                                  accumulator that takes each squared difference through a double addition, then
//                                  sqrt(err) / pow(window, 2);
//   best fit (:19-31)              sequential `_err < err || err == 0`.
// FIXED CHOICE: the unqualified `pow` / `sqrt` calls of the reference resolve to the double versions of <math.h> (the squared
// difference is added in double and rounded to the float accumulator per term; the final sqrt and the division by 225 are done in
// double and rounded to float once).
#include "orc_api.h"
#include <cmath>

namespace orc {

float knn_compare(const uint8_t* src, const uint8_t* cmp, int w, int h, int src_x, int src_y, int cmp_x, int cmp_y, int window) {
    const int _win = (int)std::ceil((float)window / 2.f);
    float err = 0;
    for (int x = -_win; x < _win + 1; x++)
        for (int y = -_win; y < _win + 1; y++) {
            if (src_x + x < 0 || src_y + y < 0 || cmp_x + x < 0 || cmp_y + y < 0 || src_x + x >= w || src_y + y >= h || cmp_x + x >= w || cmp_y + y >= h)
                continue;
            const float d = (float)src[(size_t)(src_y + y) * w + src_x + x] - (float)cmp[(size_t)(cmp_y + y) * w + cmp_x + x];
            err = (float)((double)err + std::pow((double)d, 2.0));
        }
    return (float)(std::sqrt((double)err) / std::pow((double)window, 2.0));
}

// out_best[i]: index into cmp of the best fit, or -1 for the default Feature (0,0); out_err[i]: its window error
void knn_match(const uint8_t* src, const uint8_t* cmp, int w, int h, const int* src_xy, int n, const int* cmp_xy, int m, int n_nn, int window,
               int* out_best, float* out_err) {
    std::vector<int> nn(n_nn);
    for (int i = 0; i < n; i++) {
        const int fx = src_xy[2 * i], fy = src_xy[2 * i + 1];
        int nearest = -1;   // -1 = default Feature: column 0, row 0
        auto cx = [&](int j) { return j < 0 ? 0 : cmp_xy[2 * j]; };
        auto cy = [&](int j) { return j < 0 ? 0 : cmp_xy[2 * j + 1]; };
        for (int k = 0; k < n_nn; k++) {
            float dist = 0;
            for (int j = 0; j < m; j++) {
                if (fx == cx(j) && fy == cy(j)) continue;                      // f != ff
                bool b = true;
                for (int q = 0; q < k; q++) if (cx(j) == cx(nn[q]) && cy(j) == cy(nn[q])) b = false;   // ff == fff
                if (!b) continue;
                const int dx = std::abs(fx - cx(j)), dy = std::abs(fy - cy(j));
                const float _dist = (float)(dx > dy ? dx : dy);
                if (_dist < dist || dist == 0) { dist = _dist; nearest = j; }
            }
            nn[k] = nearest;
        }
        int best = -1;
        float err = 0;
        for (int k = 0; k < n_nn; k++) {
            const float _err = knn_compare(src, cmp, w, h, fx, fy, cx(nn[k]), cy(nn[k]), window);
            if (_err < err || err == 0) { err = _err; best = nn[k]; }
        }
        out_best[i] = best;
        out_err[i] = err;
    }
}

}  // namespace orc

extern "C" {
void orc_knn_match(const uint8_t* src, const uint8_t* cmp, int w, int h, const int* src_xy, int n, const int* cmp_xy, int m, int n_nn, int window,
                   int* out_best, float* out_err) {
    orc::knn_match(src, cmp, w, h, src_xy, n, cmp_xy, m, n_nn, window, out_best, out_err);
}
float orc_knn_compare(const uint8_t* src, const uint8_t* cmp, int w, int h, int sx, int sy, int cx, int cy, int window) {
    return orc::knn_compare(src, cmp, w, h, sx, sy, cx, cy, window);
}
}
