// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// Speed-oriented twins of the restatements, for bench.py's `cpu_baseline` leg ONLY: same arithmetic, same results bit for bit
// (tests/test_oracle_frontend.py compares them with the plain versions), organised the way a production CPU library organises
// it so that the baseline the GPU path is quoted against is not a strawman:
//   * lk_track_fast: padded pyramid levels and padded (zero-bordered) Scharr images, so the inner loops are branch-free
//     contiguous int32 row loops the compiler vectorises (AVX2/AVX-512 with -march=native); point-major work split over a
//     persistent thread pool (OpenCV: parallel_for_ over points inside calcOpticalFlowPyrLK);
//   * pyr_down_fast: separable 5-tap filter with every source row filtered once;
//   * Pool: persistent worker threads (no thread creation per call);
//   * BA residual/Jacobian evaluation split over 4 threads (CeresBundleAdjustment.cpp:58 num_threads = 4), per-observation
//     results combined in observation order -> identical sums.
#include "orc_api.h"
#include "orc_fast.h"
#include <cfloat>
#include <cstring>

namespace orc {

// ---- persistent pool ---------------------------------------------------------------------------------------------------
Pool::Pool(int workers) {
    for (int i = 0; i < workers; i++) th_.emplace_back([this, i] { worker(i); });
}
Pool::~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
}
void Pool::worker(int) {
    unsigned seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
        }
        drain(seen);
    }
}
void Pool::drain(unsigned gen) {
    for (;;) {
        unsigned long long t = ticket_.load(std::memory_order_acquire);
        if ((unsigned)(t >> 32) != gen) return;             // a later batch: not ours
        const int c = (int)(t & 0xffffffffu);
        if (c >= nchunk_) return;
        if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
        const int lo = (int)((long)n_ * c / nchunk_), hi = (int)((long)n_ * (c + 1) / nchunk_);
        if (hi > lo) (*fn_)(lo, hi);
        done_.fetch_add(1, std::memory_order_release);
    }
}
void Pool::parallel_for(int n, int max_chunks, const std::function<void(int, int)>& fn) {
    const int nthreads = (int)th_.size() + 1;
    const int chunks = std::min(std::min(n, max_chunks > 0 ? max_chunks : 4 * nthreads), 4 * nthreads);
    if (chunks <= 1 || th_.empty()) { if (n > 0) fn(0, n); return; }
    unsigned g;
    { std::lock_guard<std::mutex> lk(mu_); g = gen_ + 1; }
    fn_ = &fn; n_ = n; nchunk_ = chunks;
    done_.store(0, std::memory_order_relaxed);
    ticket_.store((unsigned long long)g << 32, std::memory_order_release);   // publishes fn_/n_/nchunk_ to whoever reads this generation
    { std::lock_guard<std::mutex> lk(mu_); gen_ = g; }
    cv_.notify_all();
    drain(g);
    while (done_.load(std::memory_order_acquire) < chunks) std::this_thread::yield();
}

// ---- pyramid -----------------------------------------------------------------------------------------------------------
void pyr_down_fast(const Image8& src, Image8& dst) {
    const int sw = src.w, sh = src.h, dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dst = Image8(dw, dh);
    // horizontal pass of every source row once: hrow[y][x] = sum_j k[j] * src(y, reflect(2x + j - 2))
    std::vector<int> hbuf((size_t)sh * dw);
    std::vector<int> xi((size_t)dw * 5);
    for (int x = 0; x < dw; x++) for (int j = 0; j < 5; j++) xi[(size_t)x * 5 + j] = reflect101(2 * x + j - 2, sw);
    const int x_lo = 1, x_hi = std::max(x_lo, (sw - 3) / 2);   // interior: 2x-2 >= 0 and 2x+2 <= sw-1
    for (int y = 0; y < sh; y++) {
        const uint8_t* s = &src.d[(size_t)y * sw];
        int* r = &hbuf[(size_t)y * dw];
        for (int x = 0; x < std::min(x_lo, dw); x++) { const int* q = &xi[(size_t)x * 5]; r[x] = s[q[0]] + 4 * s[q[1]] + 6 * s[q[2]] + 4 * s[q[3]] + s[q[4]]; }
        for (int x = x_lo; x < std::min(x_hi, dw); x++) { const uint8_t* p = s + 2 * x - 2; r[x] = p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]; }
        for (int x = std::max(x_lo, std::min(x_hi, dw)); x < dw; x++) { const int* q = &xi[(size_t)x * 5]; r[x] = s[q[0]] + 4 * s[q[1]] + 6 * s[q[2]] + 4 * s[q[3]] + s[q[4]]; }
    }
    for (int y = 0; y < dh; y++) {
        const int* r0 = &hbuf[(size_t)reflect101(2 * y - 2, sh) * dw];
        const int* r1 = &hbuf[(size_t)reflect101(2 * y - 1, sh) * dw];
        const int* r2 = &hbuf[(size_t)reflect101(2 * y, sh) * dw];
        const int* r3 = &hbuf[(size_t)reflect101(2 * y + 1, sh) * dw];
        const int* r4 = &hbuf[(size_t)reflect101(2 * y + 2, sh) * dw];
        uint8_t* d = &dst.d[(size_t)y * dw];
        for (int x = 0; x < dw; x++) d[x] = (uint8_t)((r0[x] + 4 * r1[x] + 6 * r2[x] + 4 * r3[x] + r4[x] + 128) >> 8);
    }
}

// ---- padded level data -------------------------------------------------------------------------------------------------------
namespace {
constexpr int FPAD = 36;   // >= winSize + 2: every window access of lk_level stays inside the padded buffers

struct PaddedLevel {
    int w = 0, h = 0, stride = 0;
    std::vector<uint8_t> img;      // REFLECT_101-padded gray level
    std::vector<int16_t> dxy;      // (dx, dy) interleaved, 0 outside the image (BORDER_CONSTANT), same padding
    const uint8_t* I(int x, int y) const { return &img[(size_t)(y + FPAD) * stride + (x + FPAD)]; }
    const int16_t* D(int x, int y) const { return &dxy[((size_t)(y + FPAD) * stride + (x + FPAD)) * 2]; }
};

void pad_level(const Image8& src, PaddedLevel& L, bool with_deriv) {
    L.w = src.w; L.h = src.h; L.stride = src.w + 2 * FPAD;
    L.img.resize((size_t)L.stride * (src.h + 2 * FPAD));
    std::vector<int> xm(L.stride);
    for (int x = 0; x < L.stride; x++) xm[x] = reflect101(x - FPAD, src.w);
    for (int y = 0; y < src.h + 2 * FPAD; y++) {
        const uint8_t* s = &src.d[(size_t)reflect101(y - FPAD, src.h) * src.w];
        uint8_t* d = &L.img[(size_t)y * L.stride];
        for (int x = 0; x < FPAD; x++) d[x] = s[xm[x]];
        memcpy(d + FPAD, s, (size_t)src.w);
        for (int x = FPAD + src.w; x < L.stride; x++) d[x] = s[xm[x]];
    }
    if (!with_deriv) return;
    L.dxy.assign((size_t)L.stride * (src.h + 2 * FPAD) * 2, 0);
    // calcSharrDeriv on the image with REFLECT_101 at its edges == Scharr evaluated on the padded buffer at interior positions
    for (int y = 0; y < src.h; y++) {
        const uint8_t* r0 = L.I(0, y - 1);
        const uint8_t* r1 = L.I(0, y);
        const uint8_t* r2 = L.I(0, y + 1);
        int16_t* d = &L.dxy[((size_t)(y + FPAD) * L.stride + FPAD) * 2];
        for (int x = 0; x < src.w; x++) {
            const int t0m = (r0[x - 1] + r2[x - 1]) * 3 + r1[x - 1] * 10, t0p = (r0[x + 1] + r2[x + 1]) * 3 + r1[x + 1] * 10;
            const int t1m = r2[x - 1] - r0[x - 1], t1c = r2[x] - r0[x], t1p = r2[x + 1] - r0[x + 1];
            d[2 * x] = (int16_t)(t0p - t0m);
            d[2 * x + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
        }
    }
}

inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
inline void weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = cv_round((1.f - a) * (1.f - b) * 16384.f);
    w01 = cv_round(a * (1.f - b) * 16384.f);
    w10 = cv_round((1.f - a) * b * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

// one point through all levels (same statements as lk_level in orc_lk.cpp, window loops over contiguous padded rows)
void lk_point(const std::vector<PaddedLevel>& PI, const std::vector<PaddedLevel>& PJ, int ml, const LKParams& P, const float prev_xy[2],
              float next_xy[2], uint8_t* status, float* err) {
    constexpr int W = 32;
    const float half = (W - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    alignas(64) int32_t Iw[W * W];
    alignas(64) int32_t Ixw[W * W], Iyw[W * W];
    for (int level = ml; level >= 0; level--) {
        const PaddedLevel& I = PI[level];
        const PaddedLevel& J = PJ[level];
        const float lscale = (float)(1. / (1 << level));
        float prevx = prev_xy[0] * lscale, prevy = prev_xy[1] * lscale;
        float nx, ny;
        if (level == ml) { nx = prevx; ny = prevy; }
        else { nx = next_xy[0] * 2.f; ny = next_xy[1] * 2.f; }
        next_xy[0] = nx; next_xy[1] = ny;
        prevx -= half; prevy -= half;
        const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
        if (ipx < -W || ipx >= I.w || ipy < -W || ipy >= I.h) {
            if (level == 0) { *status = 0; *err = 0; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        weights(prevx - ipx, prevy - ipy, iw00, iw01, iw10, iw11);
        int64_t sA11 = 0, sA12 = 0, sA22 = 0;
        for (int y = 0; y < W; y++) {
            const uint8_t* p0 = I.I(ipx, ipy + y);
            const uint8_t* p1 = p0 + I.stride;
            const int16_t* d0 = I.D(ipx, ipy + y);
            const int16_t* d1 = d0 + 2 * I.stride;
            int32_t a11 = 0, a12 = 0, a22 = 0;   // 32 terms of at most 8160^2: fits int32
            for (int x = 0; x < W; x++) {
                const int iv = descale(p0[x] * iw00 + p0[x + 1] * iw01 + p1[x] * iw10 + p1[x + 1] * iw11, 9);
                const int ix = descale(d0[2 * x] * iw00 + d0[2 * x + 2] * iw01 + d1[2 * x] * iw10 + d1[2 * x + 2] * iw11, 14);
                const int iy = descale(d0[2 * x + 1] * iw00 + d0[2 * x + 3] * iw01 + d1[2 * x + 1] * iw10 + d1[2 * x + 3] * iw11, 14);
                Iw[y * W + x] = iv; Ixw[y * W + x] = ix; Iyw[y * W + x] = iy;
                a11 += ix * ix; a12 += ix * iy; a22 += iy * iy;
            }
            sA11 += a11; sA12 += a12; sA22 += a22;
        }
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * W * W);
        if (minEig < P.min_eig || D < FLT_EPSILON) {
            if (level == 0) *status = 0;
            continue;
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
            weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            int64_t sb1 = 0, sb2 = 0;
            for (int y = 0; y < W; y++) {
                const uint8_t* p0 = J.I(inx, iny + y);
                const uint8_t* p1 = p0 + J.stride;
                int32_t b1 = 0, b2 = 0;
                for (int x = 0; x < W; x++) {
                    const int diff = descale(p0[x] * iw00 + p0[x + 1] * iw01 + p1[x] * iw10 + p1[x + 1] * iw11, 9) - Iw[y * W + x];
                    b1 += diff * Ixw[y * W + x]; b2 += diff * Iyw[y * W + x];
                }
                sb1 += b1; sb2 += b2;
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
            if (inx < -W || inx >= J.w || iny < -W || iny >= J.h) { *status = 0; continue; }
            weights(fx - inx, fy - iny, iw00, iw01, iw10, iw11);
            int64_t e = 0;
            for (int y = 0; y < W; y++) {
                const uint8_t* p0 = J.I(inx, iny + y);
                const uint8_t* p1 = p0 + J.stride;
                int32_t er = 0;
                for (int x = 0; x < W; x++) {
                    const int diff = descale(p0[x] * iw00 + p0[x + 1] * iw01 + p1[x] * iw10 + p1[x + 1] * iw11, 9) - Iw[y * W + x];
                    er += diff < 0 ? -diff : diff;
                }
                e += er;
            }
            *err = (float)e * (1.f / (32 * W * W));
        }
    }
}
}  // namespace

void lk_track_fast(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, const LKParams& P,
                   float* out_xy, uint8_t* out_status, float* out_err, Pool* pool) {
    // cv::calcOpticalFlowPyrLK on two images builds both pyramids on every call (the reference passes cv::Mat, not pyramids)
    std::vector<Image8> pp(1), np(1);
    pp[0] = Image8(w, h); np[0] = Image8(w, h);
    memcpy(pp[0].d.data(), prev, (size_t)w * h);
    memcpy(np[0].d.data(), next, (size_t)w * h);
    int ml = 0;
    {
        int lw = w, lh = h;
        for (int level = 0; level <= P.max_level; level++) {
            if (level != 0) {
                pp.emplace_back(); np.emplace_back();
                pyr_down_fast(pp[level - 1], pp[level]);
                pyr_down_fast(np[level - 1], np[level]);
            }
            ml = level;
            lw = (lw + 1) / 2; lh = (lh + 1) / 2;
            if (lw <= P.win || lh <= P.win) break;
        }
    }
    std::vector<PaddedLevel> PI(ml + 1), PJ(ml + 1);
    auto prep = [&](int lo, int hi) {
        for (int k = lo; k < hi; k++) {
            if (k <= ml) pad_level(pp[k], PI[k], true);
            else pad_level(np[k - ml - 1], PJ[k - ml - 1], false);
        }
    };
    if (pool) pool->parallel_for(2 * (ml + 1), 2 * (ml + 1), prep); else prep(0, 2 * (ml + 1));
    for (int i = 0; i < n; i++) { out_status[i] = 1; out_err[i] = 0; out_xy[2 * i] = out_xy[2 * i + 1] = 0; }
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++) lk_point(PI, PJ, ml, P, prev_xy + 2 * i, out_xy + 2 * i, out_status + i, out_err + i);
    };
    if (pool) pool->parallel_for(n, 0, work); else work(0, n);
}

}  // namespace orc

extern "C" {
// same contract as orc_lk_track; nthreads > 1 uses a pool created for the call (tests); the pipeline keeps one pool per run
int orc_lk_track_fast(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, int win, int max_level,
                      int max_iter, double eps, float min_eig, float* out_xy, uint8_t* out_status, float* out_err, int nthreads) {
    if (win != 32) return -1;
    orc::LKParams P;
    P.win = win; P.max_level = max_level; P.max_iter = max_iter; P.eps = eps; P.min_eig = min_eig;
    std::unique_ptr<orc::Pool> pool;
    if (nthreads > 1) pool.reset(new orc::Pool(nthreads - 1));
    orc::lk_track_fast(prev, next, w, h, prev_xy, n, P, out_xy, out_status, out_err, pool.get());
    return 0;
}
void orc_pyr_down_fast(const uint8_t* src, int w, int h, uint8_t* out) {
    orc::Image8 s(w, h);
    memcpy(s.d.data(), src, (size_t)w * h);
    orc::Image8 d;
    orc::pyr_down_fast(s, d);
    memcpy(out, d.d.data(), d.d.size());
}
}
