// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED (OpenCV internals).
//
// Restates cv::calcOpticalFlowPyrLK as called by the reference at
//   /root/reference/OpenCVLucasKanadeFM.cpp:15   (winSize 32x32, maxLevel 4, default criteria
//   COUNT+EPS 30/0.01, flags 0, minEigThreshold 1e-4)
// following the published OpenCV 3.4 algorithm (video/lkpyramid.cpp, imgproc/pyramids.cpp):
//   buildOpticalFlowPyramid -> pyrDown ([1 4 6 4 1]^2, (sum+128)>>8, REFLECT_101)
//   calcSharrDeriv          -> int16 (dx,dy), Scharr 3/10/3, REFLECT_101 inside, 0 outside
//   LKTrackerInvoker        -> 14-bit integer bilinear weights, int16 patches with 5 fractional
//                              bits, 2x2 normal equations in float32, <=30 iterations.
//
// FIXED CHOICE (documented deviation from any particular OpenCV build): OpenCV accumulates the
// integer products ix*ix, ix*iy, iy*iy, diff*ix, diff*iy in float32 in a build-dependent order
// (scalar / SSE2 / NEON differ bitwise).  This restatement accumulates them EXACTLY in int64 and
// rounds once to float32, which is order-free and therefore reproducible on any device.
#include "orc_api.h"
#include <cfloat>
#include <thread>
#include <cstring>

namespace orc {

// cv::pyrDown for CV_8UC1, BORDER_DEFAULT (REFLECT_101); dst size ((w+1)/2, (h+1)/2)
void pyr_down(const Image8& src, Image8& dst) {
    const int dw = (src.w + 1) / 2, dh = (src.h + 1) / 2;
    dst = Image8(dw, dh);
    static const int k[5] = {1, 4, 6, 4, 1};
    std::vector<int> rowbuf((size_t)5 * dw);
    for (int y = 0; y < dh; y++) {
        for (int i = 0; i < 5; i++) {
            const int sy = reflect101(2 * y + i - 2, src.h);
            const uint8_t* s = &src.d[(size_t)sy * src.w];
            int* r = &rowbuf[(size_t)i * dw];
            for (int x = 0; x < dw; x++) {
                int acc = 0;
                for (int j = 0; j < 5; j++) acc += k[j] * s[reflect101(2 * x + j - 2, src.w)];
                r[x] = acc;
            }
        }
        for (int x = 0; x < dw; x++) {
            int acc = 0;
            for (int i = 0; i < 5; i++) acc += k[i] * rowbuf[(size_t)i * dw + x];
            dst.d[(size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
}

// buildOpticalFlowPyramid level count rule: stop when the NEXT level would be <= winSize
int build_pyramid(const Image8& img, int win, int max_level, std::vector<Image8>& pyr) {
    pyr.clear();
    pyr.push_back(img);
    int w = img.w, h = img.h;
    for (int level = 0; level <= max_level; level++) {
        if (level != 0) {
            Image8 nxt;
            pyr_down(pyr[level - 1], nxt);
            pyr.push_back(nxt);
        }
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) return level;
    }
    return max_level;
}

// calcSharrDeriv: out[(y*w+x)*2+0] = dx, +1 = dy
void scharr_deriv(const Image8& src, std::vector<int16_t>& out) {
    const int w = src.w, h = src.h;
    out.assign((size_t)w * h * 2, 0);
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = &src.d[(size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * w];
        const uint8_t* r1 = &src.d[(size_t)y * w];
        const uint8_t* r2 = &src.d[(size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * w];
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        const int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        for (int x = 0; x < w; x++) {
            out[((size_t)y * w + x) * 2 + 0] = (int16_t)(t0[x + 2] - t0[x]);
            out[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
}

static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }


// One pyramid level of LKTrackerInvoker for one point.  pts are level-0 coordinates in/out.
static void lk_level(const Image8& I, const std::vector<int16_t>& dI, const Image8& J, int level,
                     int max_level, const LKParams& P, const float prev_xy[2], float next_xy[2],
                     uint8_t* status, float* err) {
    const int W = P.win;
    const float half = (W - 1) * 0.5f;
    const float lscale = (float)(1. / (1 << level));
    float prevx = prev_xy[0] * lscale, prevy = prev_xy[1] * lscale;
    float nx, ny;
    if (level == max_level) { nx = prevx; ny = prevy; }
    else { nx = next_xy[0] * 2.f; ny = next_xy[1] * 2.f; }
    next_xy[0] = nx; next_xy[1] = ny;

    prevx -= half; prevy -= half;
    const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
    if (ipx < -W || ipx >= I.w || ipy < -W || ipy >= I.h) {
        if (level == 0) { *status = 0; *err = 0; }
        return;
    }
    float a = prevx - ipx, b = prevy - ipy;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    std::vector<int16_t> Iw((size_t)W * W), dIw((size_t)W * W * 2);
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    auto dval = [&](int x, int y, int c) -> int {
        if (x < 0 || x >= I.w || y < 0 || y >= I.h) return 0;  // BORDER_CONSTANT(0) outside
        return dI[((size_t)y * I.w + x) * 2 + c];
    };
    for (int y = 0; y < W; y++)
        for (int x = 0; x < W; x++) {
            const int gx = ipx + x, gy = ipy + y;
            int ival = descale(I.pxr(gx, gy) * iw00 + I.pxr(gx + 1, gy) * iw01 +
                               I.pxr(gx, gy + 1) * iw10 + I.pxr(gx + 1, gy + 1) * iw11, W_BITS - 5);
            int ixval = descale(dval(gx, gy, 0) * iw00 + dval(gx + 1, gy, 0) * iw01 +
                                dval(gx, gy + 1, 0) * iw10 + dval(gx + 1, gy + 1, 0) * iw11, W_BITS);
            int iyval = descale(dval(gx, gy, 1) * iw00 + dval(gx + 1, gy, 1) * iw01 +
                                dval(gx, gy + 1, 1) * iw10 + dval(gx + 1, gy + 1, 1) * iw11, W_BITS);
            Iw[(size_t)y * W + x] = (int16_t)ival;
            dIw[((size_t)y * W + x) * 2] = (int16_t)ixval;
            dIw[((size_t)y * W + x) * 2 + 1] = (int16_t)iyval;
            sA11 += (int64_t)ixval * ixval;
            sA12 += (int64_t)ixval * iyval;
            sA22 += (int64_t)iyval * iyval;
        }
    const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * W * W);
    if (minEig < P.min_eig || D < FLT_EPSILON) {
        if (level == 0) *status = 0;
        return;
    }
    D = 1.f / D;
    nx -= half; ny -= half;
    float pdx = 0, pdy = 0;
    const double eps2 = P.eps * P.eps;
    for (int j = 0; j < P.max_iter; j++) {
        const int inx = cv_floor(nx), iny = cv_floor(ny);
        if (inx < -W || inx >= J.w || iny < -W || iny >= J.h) {
            if (level == 0) *status = 0;
            break;
        }
        a = nx - inx; b = ny - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t sb1 = 0, sb2 = 0;
        for (int y = 0; y < W; y++)
            for (int x = 0; x < W; x++) {
                const int gx = inx + x, gy = iny + y;
                int diff = descale(J.pxr(gx, gy) * iw00 + J.pxr(gx + 1, gy) * iw01 +
                                   J.pxr(gx, gy + 1) * iw10 + J.pxr(gx + 1, gy + 1) * iw11, W_BITS - 5) -
                           Iw[(size_t)y * W + x];
                sb1 += (int64_t)diff * dIw[((size_t)y * W + x) * 2];
                sb2 += (int64_t)diff * dIw[((size_t)y * W + x) * 2 + 1];
            }
        const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        const float dx = (float)((A12 * b2 - A22 * b1) * D);
        const float dy = (float)((A12 * b1 - A11 * b2) * D);
        nx += dx; ny += dy;
        next_xy[0] = nx + half; next_xy[1] = ny + half;
        if ((double)dx * dx + (double)dy * dy <= eps2) break;
        if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
            next_xy[0] -= dx * 0.5f;
            next_xy[1] -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
    if (*status && level == 0) {
        const float fx = next_xy[0] - half, fy = next_xy[1] - half;
        const int inx = cv_floor(fx), iny = cv_floor(fy);
        if (inx < -W || inx >= J.w || iny < -W || iny >= J.h) {
            *status = 0;
            return;
        }
        const float aa = fx - inx, bb = fy - iny;
        iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t e = 0;  // |diff| <= 8160, 1024 terms: exact in float32 too
        for (int y = 0; y < W; y++)
            for (int x = 0; x < W; x++) {
                const int gx = inx + x, gy = iny + y;
                int diff = descale(J.pxr(gx, gy) * iw00 + J.pxr(gx + 1, gy) * iw01 +
                                   J.pxr(gx, gy + 1) * iw10 + J.pxr(gx + 1, gy + 1) * iw11, W_BITS - 5) -
                           Iw[(size_t)y * W + x];
                e += diff < 0 ? -diff : diff;
            }
        *err = (float)e * (1.f / (32 * W * W));
    }
}

void lk_track(const Image8& prev, const Image8& next, const float* prev_xy, int n, const LKParams& P,
              float* out_xy, uint8_t* out_status, float* out_err, int* levels_used, int nthreads) {
    std::vector<Image8> pp, np;
    int ml = build_pyramid(prev, P.win, P.max_level, pp);
    int ml2 = build_pyramid(next, P.win, P.max_level, np);
    (void)ml2;  // same size images => same level count
    if (levels_used) *levels_used = ml;
    for (int i = 0; i < n; i++) { out_status[i] = 1; out_err[i] = 0; out_xy[2 * i] = out_xy[2 * i + 1] = 0; }
    std::vector<int16_t> dI;
    for (int level = ml; level >= 0; level--) {
        scharr_deriv(pp[level], dI);
        // points are independent (OpenCV: parallel_for_ over points); any split gives identical results
        auto work = [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                lk_level(pp[level], dI, np[level], level, ml, P, prev_xy + 2 * i, out_xy + 2 * i, out_status + i, out_err + i);
        };
        if (nthreads <= 1 || n < 2 * nthreads) work(0, n);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nthreads; t++) th.emplace_back(work, (int)((long)n * t / nthreads), (int)((long)n * (t + 1) / nthreads));
            for (auto& x : th) x.join();
        }
    }
}

}  // namespace orc

extern "C" {

// out must hold ((w+1)/2)*((h+1)/2) bytes
// cv::cvtColor(BGR2GRAY) on 8-bit pixels (Frame::init, Frame.cpp:40-41) [mem: OpenCV 3.4 color.cpp, RGB2Gray<uchar>: 14-bit fixed point,
// B2Y = 1868, G2Y = 9617, R2Y = 4899, CV_DESCALE(.., 14)]. Parity unpinned like the other OpenCV rows; identity for B = G = R.
void orc_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* out) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = bgr + (size_t)y * stride + 3 * x;
            out[(size_t)y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
        }
}
void orc_pyr_down(const uint8_t* src, int w, int h, uint8_t* out) {
    orc::Image8 s(w, h);
    memcpy(s.d.data(), src, (size_t)w * h);
    orc::Image8 d;
    orc::pyr_down(s, d);
    memcpy(out, d.d.data(), d.d.size());
}

void orc_scharr(const uint8_t* src, int w, int h, int16_t* out) {
    orc::Image8 s(w, h);
    memcpy(s.d.data(), src, (size_t)w * h);
    std::vector<int16_t> d;
    orc::scharr_deriv(s, d);
    memcpy(out, d.data(), d.size() * sizeof(int16_t));
}

int orc_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n,
                 int win, int max_level, int max_iter, double eps, float min_eig, float* out_xy,
                 uint8_t* out_status, float* out_err) {
    orc::Image8 a(w, h), b(w, h);
    memcpy(a.d.data(), prev, (size_t)w * h);
    memcpy(b.d.data(), next, (size_t)w * h);
    orc::LKParams P;
    P.win = win; P.max_level = max_level; P.max_iter = max_iter; P.eps = eps; P.min_eig = min_eig;
    int lv = 0;
    orc::lk_track(a, b, prev_xy, n, P, out_xy, out_status, out_err, &lv, 1);
    return lv;
}
}
