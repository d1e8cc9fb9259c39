// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// (1) gftt_cell: restates cv::goodFeaturesToTrack(cell, corners, max, 0.01, 5, noMask, 3, 3, false, .04)
//     as called at /root/reference/OpenCVGoodFeatureExtractor.cpp:7 on a <=255x255 grid cell that is a
//     sub-view of the full gray image (/root/reference/OdometryPipeline.cpp:674-693, Frame.cpp:95-117).
//     PARITY UNPINNED (OpenCV internals, restated from the published 3.4 algorithm):
//       Sobel 3x3 (u8->f32, scale 1/3060 folded into the smoothing kernel, REFLECT_101 on the PARENT
//       image because the cell is a non-isolated ROI), cov=(dx^2,dxdy,dy^2), un-normalised 3x3 box
//       (REFLECT_101 on the CELL), eig=(a+c)-sqrt((a-c)^2+b^2), threshold 0.01*max (TOZERO),
//       3x3 dilate NMS, sort by (value desc, address desc), greedy min-distance 5 px, <=max corners.
//     FIXED CHOICES: float32 ops are evaluated exactly as written below with no FMA contraction;
//       the box sum is a plain 9-term double sum in raster order (OpenCV uses running double sums).
// (2) shitomasi_cell: restates /root/reference/ShiTomasiFeatureExtractor.cpp:5-75 on top of
//     /root/reference/Frame.cpp:58-86 (signed-char central differences, quirk Q1) and Frame.cpp:119-138
//     (cv::blur 3x3 of the structure tensor, REFLECT_101).  Pinned by in-repo source except cv::blur's
//     summation order and std::sort's tie order (FIXED: 9-term raster-order sum * (1/9); ties keep raster order).
#include "orc_api.h"
#include <cstring>
#include <cfloat>

namespace orc {

// ---- GFTT -----------------------------------------------------------------------------------
// img: full image (W x H, stride W). cell rect (cx0, cy0, cw, ch). eig: cw*ch floats.
void gftt_eig(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, float* eig) {
    const double dscale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)(1.0 * dscale), k2 = (float)(2.0 * dscale);  // smoothing kernel [1 2 1]*scale as CV_32F
    auto P = [&](int x, int y) -> float { return (float)img[(size_t)reflect101(y, H) * W + reflect101(x, W)]; };
    std::vector<float> cov((size_t)cw * ch * 3);
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            const int gx = cx0 + x, gy = cy0 + y;
            // Dx: row filter [-1 0 1] (exact), column filter SymmColumnSmall: (top+bot)*f1 + mid*f0
            float rt = P(gx + 1, gy - 1) - P(gx - 1, gy - 1);
            float rm = P(gx + 1, gy) - P(gx - 1, gy);
            float rb = P(gx + 1, gy + 1) - P(gx - 1, gy + 1);
            float dx = (rt + rb) * k1 + rm * k2;
            // Dy: row filter [1 2 1]*scale generic left-to-right, column filter bot - top
            float st = k1 * P(gx - 1, gy - 1); st += k2 * P(gx, gy - 1); st += k1 * P(gx + 1, gy - 1);
            float sb = k1 * P(gx - 1, gy + 1); sb += k2 * P(gx, gy + 1); sb += k1 * P(gx + 1, gy + 1);
            float dy = sb - st;
            float* c = &cov[((size_t)y * cw + x) * 3];
            c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
        }
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            double s[3] = {0, 0, 0};
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    const float* c = &cov[((size_t)reflect101(y + j, ch) * cw + reflect101(x + i, cw)) * 3];
                    s[0] += c[0]; s[1] += c[1]; s[2] += c[2];
                }
            const float a = (float)s[0] * 0.5f, b = (float)s[1], c2 = (float)s[2] * 0.5f;
            eig[(size_t)y * cw + x] = (float)((a + c2) - std::sqrt((a - c2) * (a - c2) + b * b));
        }
}

int gftt_cell(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, int max_corners,
              double quality, double min_dist, int* out_xy, float* eig_out) {
    std::vector<float> eig((size_t)cw * ch);
    gftt_eig(img, W, H, cx0, cy0, cw, ch, eig.data());
    if (eig_out) memcpy(eig_out, eig.data(), eig.size() * sizeof(float));
    // minMaxLoc (NaN never wins)
    double maxVal = -DBL_MAX;
    for (float v : eig) if ((double)v > maxVal) maxVal = v;
    const float thr = (float)(maxVal * quality);
    for (float& v : eig) v = v > thr ? v : 0.f;  // THRESH_TOZERO
    struct Cand { float v; int idx; };
    std::vector<Cand> cand;
    for (int y = 1; y < ch - 1; y++)
        for (int x = 1; x < cw - 1; x++) {
            const float v = eig[(size_t)y * cw + x];
            if (v == 0.f) continue;
            float m = v;  // dilate 3x3 (outside ignored; interior pixels have full neighbourhoods)
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) m = std::max(m, eig[(size_t)(y + j) * cw + x + i]);
            if (v == m) cand.push_back({v, y * cw + x});
        }
    // greaterThanPtr: value desc, then higher address first
    std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
        return (a.v > b.v) ? true : (a.v < b.v) ? false : (a.idx > b.idx);
    });
    int n = 0;
    if (min_dist >= 1) {
        const int cell = cv_round(min_dist);
        const int gw = (cw + cell - 1) / cell, gh = (ch + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
        const double md2 = min_dist * min_dist;
        for (const Cand& c : cand) {
            const int y = c.idx / cw, x = c.idx - y * cw;
            bool good = true;
            const int xc = x / cell, yc = y / cell;
            const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (auto& m : grid[(size_t)yy * gw + xx]) {
                        const float dx = x - m.first, dy = y - m.second;
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
            if (good) {
                grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
                out_xy[2 * n] = x; out_xy[2 * n + 1] = y;
                n++;
                if (max_corners > 0 && n == max_corners) break;
            }
        }
    } else {
        for (const Cand& c : cand) {
            const int y = c.idx / cw, x = c.idx - y * cw;
            out_xy[2 * n] = x; out_xy[2 * n + 1] = y;
            n++;
            if (max_corners > 0 && n == max_corners) break;
        }
    }
    return n;
}

// ---- ShiTomasi (in-repo arithmetic) -----------------------------------------------------------
void shitomasi_response(const uint8_t* img, int W, int cx0, int cy0, int cw, int ch, double* R) {
    // Frame.cpp:58-86: zeros on the cell border, signed-char central differences inside
    std::vector<double> gx((size_t)cw * ch, 0.0), gy((size_t)cw * ch, 0.0);
    auto S = [&](int x, int y) -> double { return (double)(int8_t)img[(size_t)(cy0 + y) * W + cx0 + x]; };
    for (int r = 1; r < ch - 1; r++)
        for (int c = 1; c < cw - 1; c++) {
            gx[(size_t)r * cw + c] = 1. / 2. * S(c + 1, r) - 1. / 2. * S(c - 1, r);
            gy[(size_t)r * cw + c] = 1. / 2. * S(c, r + 1) - 1. / 2. * S(c, r - 1);
        }
    // Frame.cpp:119-138: channels (Ixx, Iyy, Ixy), cv::blur 3x3 normalised, REFLECT_101
    std::vector<double> hx((size_t)cw * ch * 3);
    for (size_t i = 0; i < (size_t)cw * ch; i++) {
        hx[i * 3 + 0] = gx[i] * gx[i];
        hx[i * 3 + 1] = gy[i] * gy[i];
        hx[i * 3 + 2] = gx[i] * gy[i];
    }
    const double inv9 = 1.0 / 9.0;
    for (int r = 0; r < ch; r++)
        for (int c = 0; c < cw; c++) {
            R[(size_t)r * cw + c] = 0.0;  // Mat::zeros; last column stays 0 (ShiTomasiFeatureExtractor.cpp:58)
            if (c >= cw - 1) continue;
            double s[3] = {0, 0, 0};
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    const double* h = &hx[((size_t)reflect101(r + j, ch) * cw + reflect101(c + i, cw)) * 3];
                    s[0] += h[0]; s[1] += h[1]; s[2] += h[2];
                }
            const double Ixx = s[0] * inv9, Iyy = s[1] * inv9, Ixy = s[2] * inv9;
            // ShiTomasiFeatureExtractor.cpp:64-71
            const double B = -Ixx - Iyy;
            const double C = Ixx * Iyy - Ixy * Ixy;
            const double disc = std::sqrt(B * B - 4 * C);
            const double l1 = (-B + disc) / 2, l2 = (-B - disc) / 2;
            R[(size_t)r * cw + c] = std::min(l1, l2);  // std::min(a,b) = (b<a)?b:a  (NaN propagates as in the reference)
        }
}

int shitomasi_cell(const uint8_t* img, int W, int cx0, int cy0, int cw, int ch, int max_feats,
                   double quality, int* out_xy, double* out_score, double* R_out) {
    std::vector<double> R((size_t)cw * ch);
    shitomasi_response(img, W, cx0, cy0, cw, ch, R.data());
    if (R_out) memcpy(R_out, R.data(), R.size() * sizeof(double));
    double rmax = -DBL_MAX;
    for (double v : R) if (v > rmax) rmax = v;  // minMaxLoc; NaN never wins
    const double thr = rmax * quality;
    struct Cand { double v; int idx; };
    std::vector<Cand> cand;
    for (int j = 0; j < ch; j++)
        for (int i = 0; i < cw; i++) {
            const double v = R[(size_t)j * cw + i];
            if (v > thr) cand.push_back({v, j * cw + i});  // THRESH_BINARY: src > thresh
        }
    std::stable_sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) { return a.v > b.v; });
    int n = 0;
    for (const Cand& c : cand) {
        if (n >= max_feats) break;
        out_xy[2 * n] = c.idx % cw; out_xy[2 * n + 1] = c.idx / cw;
        if (out_score) out_score[n] = c.v;
        n++;
    }
    return n;
}

}  // namespace orc

extern "C" {
int orc_gftt_cell(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, int max_corners,
                  double quality, double min_dist, int* out_xy, float* eig_out) {
    return orc::gftt_cell(img, W, H, cx0, cy0, cw, ch, max_corners, quality, min_dist, out_xy, eig_out);
}
int orc_shitomasi_cell(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, int max_feats,
                       double quality, int* out_xy, double* out_score, double* R_out) {
    (void)H;
    return orc::shitomasi_cell(img, W, cx0, cy0, cw, ch, max_feats, quality, out_xy, out_score, R_out);
}
}
