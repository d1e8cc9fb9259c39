// Deterministic synthetic KITTI-like monocular sequence (SURVEY.md §8d "Inputs").
// No KITTI data ships with the repo or exists on the GPU box, so every BASELINE config is
// instantiated on this procedural scene: a textured corridor (ground plane, two side walls,
// ceiling) rendered through a pinhole camera that drives forward ~0.9 m/frame with gentle yaw.
// Ground-truth poses are emitted in KITTI's 12-float row format (camera-to-world [R|t], x right,
// y down, z forward), i.e. what /root/reference/OdometryPipeline.cpp:525-594 parses.
//
// Pure integer hashing + IEEE double arithmetic (no FMA contraction, see build flags) so the same
// seed gives the same bytes on any x86-64 host.  This is input generation, not part of the hot path.
#include <cstdint>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

inline uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
inline uint64_t hash4(uint64_t seed, int64_t a, int64_t b, int64_t c) {
    uint64_t h = mix64(seed ^ 0x51ed270b1a2b3c4dULL);
    h = mix64(h ^ (uint64_t)a);
    h = mix64(h ^ ((uint64_t)b * 0x9e3779b97f4a7c15ULL));
    h = mix64(h ^ ((uint64_t)c * 0xc2b2ae3d27d4eb4fULL));
    return h;
}
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

inline double smooth(double t) { return t * t * (3.0 - 2.0 * t); }

// value noise in [-1,1] on an integer lattice
double vnoise(uint64_t seed, int chan, double u, double v) {
    const double fu = std::floor(u), fv = std::floor(v);
    const int64_t iu = (int64_t)fu, iv = (int64_t)fv;
    const double a = smooth(u - fu), b = smooth(v - fv);
    const double n00 = u01(hash4(seed, chan, iu, iv)), n10 = u01(hash4(seed, chan, iu + 1, iv));
    const double n01 = u01(hash4(seed, chan, iu, iv + 1)), n11 = u01(hash4(seed, chan, iu + 1, iv + 1));
    const double top = n00 + (n10 - n00) * a, bot = n01 + (n11 - n01) * a;
    return (top + (bot - top) * b) * 2.0 - 1.0;
}

inline double clamp01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

// high-contrast rectangles ("windows", road markings): one candidate per lattice cell
double blobs(uint64_t seed, int chan, double u, double v, double cell, double fp) {
    const double cu = u / cell, cv = v / cell;
    const double fu = std::floor(cu), fv = std::floor(cv);
    const uint64_t h = hash4(seed, chan, (int64_t)fu, (int64_t)fv);
    if ((h & 3) == 0) return 0.0;  // 25% of the cells are empty
    const double x0 = 0.10 + 0.30 * u01(mix64(h ^ 1)), x1 = 0.60 + 0.30 * u01(mix64(h ^ 2));
    const double y0 = 0.10 + 0.30 * u01(mix64(h ^ 3)), y1 = 0.60 + 0.30 * u01(mix64(h ^ 4));
    const double amp = (u01(mix64(h ^ 5)) < 0.5 ? -1.0 : 1.0) * (55.0 + 45.0 * u01(mix64(h ^ 6)));
    const double lu = cu - fu, lv = cv - fv;
    const double w = fp / cell;  // footprint in cell units (edge anti-aliasing)
    const double inv = 1.0 / (w > 1e-9 ? w : 1e-9);
    const double cx = clamp01((lu - x0) * inv + 0.5) * clamp01((x1 - lu) * inv + 0.5);
    const double cy = clamp01((lv - y0) * inv + 0.5) * clamp01((y1 - lv) * inv + 0.5);
    return amp * cx * cy;
}

double texture(uint64_t seed, int plane, double u, double v, double fp) {
    double val = 128.0;
    double lambda = 4.0, amp = 26.0;
    for (int o = 0; o < 7; o++) {
        const double fade = clamp01((lambda / fp - 2.0) * 0.5);
        if (fade > 0) val += amp * fade * vnoise(seed, plane * 16 + o, u / lambda, v / lambda);
        lambda *= 0.5;
        amp *= 0.8;
    }
    val += blobs(seed, plane * 16 + 8, u, v, 1.3, fp) * clamp01((1.3 / fp - 2.0) * 0.5);
    val += 0.6 * blobs(seed, plane * 16 + 9, u + 0.37, v + 0.11, 0.45, fp) * clamp01((0.45 / fp - 2.0) * 0.5);
    return val;
}

struct Pose { double R[9]; double t[3]; };

// Closed-form trajectory: z advances ~0.9 m/frame, x weaves inside the corridor, yaw follows the path.
void pose_at(uint64_t seed, int frame, Pose& P) {
    const double ph = 6.283185307179586 * u01(mix64(seed ^ 0xabcdefULL));
    auto path = [&](double f, double& x, double& z) {
        z = 0.9 * f + 8.0 * (std::sin(0.013 * f + ph) - std::sin(ph));
        x = 1.2 * (std::sin(0.02 * f + ph) - std::sin(ph)) + 0.5 * (std::sin(0.07 * f + 2.0 * ph) - std::sin(2.0 * ph));
    };
    double x0, z0, x1, z1, xa, za;
    path(0.0, xa, za);
    path((double)frame, x0, z0);
    path((double)frame + 1e-3, x1, z1);
    const double yaw = std::atan2(x1 - x0, z1 - z0);
    const double c = std::cos(yaw), s = std::sin(yaw);
    P.R[0] = c; P.R[1] = 0; P.R[2] = s;
    P.R[3] = 0; P.R[4] = 1; P.R[5] = 0;
    P.R[6] = -s; P.R[7] = 0; P.R[8] = c;
    P.t[0] = x0 - xa; P.t[1] = 0; P.t[2] = z0 - za;
}

// world (KITTI-like, x right / y down / z forward): ground y=+1.65, ceiling y=-9, walls x=-9 / +10.5
void render(uint64_t seed, int frame, int w, int h, double fx, double fy, double cx, double cy, uint8_t* out, int stride) {
    Pose P0, P;
    pose_at(seed, 0, P0);
    pose_at(seed, frame, P);
    // express the pose relative to frame 0 (KITTI: first pose is identity)
    double Rw[9], tw[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P0.R[k * 3 + i] * P.R[k * 3 + j];
            Rw[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += P0.R[k * 3 + i] * (P.t[k] - P0.t[k]);
        tw[i] = s;
    }
    struct Plane { int axis; double off; int id; };  // axis 0: x = off, axis 1: y = off
    const Plane planes[4] = {{1, 1.65, 0}, {1, -9.0, 1}, {0, -9.0, 2}, {0, 10.5, 3}};
    for (int v = 0; v < h; v++) {
        for (int u = 0; u < w; u++) {
            const double dcx = (u - cx) / fx, dcy = (v - cy) / fy, dcz = 1.0;
            double d[3];
            for (int i = 0; i < 3; i++) d[i] = Rw[i * 3] * dcx + Rw[i * 3 + 1] * dcy + Rw[i * 3 + 2] * dcz;
            const double dn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double best_t = 1e300;
            int best = -1;
            for (int p = 0; p < 4; p++) {
                const double denom = d[planes[p].axis];
                if (std::fabs(denom) < 1e-12) continue;
                const double tt = (planes[p].off - tw[planes[p].axis]) / denom;
                if (tt > 1e-6 && tt < best_t) { best_t = tt; best = p; }
            }
            double val = 128.0;
            if (best >= 0) {
                const double X = tw[0] + best_t * d[0], Y = tw[1] + best_t * d[1], Z = tw[2] + best_t * d[2];
                const double cosi = std::fabs(d[planes[best].axis]) / dn;
                const double range = best_t * dn;
                const double fp = range / fx / (cosi > 1e-3 ? cosi : 1e-3);
                const double tu = planes[best].axis == 1 ? X : Y;
                val = texture(seed, planes[best].id, tu, Z, fp);
            }
            // sensor noise: +-2 gray levels, per frame and pixel
            val += 4.0 * (u01(hash4(seed ^ 0x5e5e5e5eULL, frame, u, v)) - 0.5);
            int iv = (int)std::floor(val + 0.5);
            out[(size_t)v * stride + u] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
        }
    }
}

}  // namespace

extern "C" {

// KITTI pose row for `frame` relative to frame 0: r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz
void pmv_synth_pose(uint64_t seed, int frame, double* pose12) {
    Pose P0, P;
    pose_at(seed, 0, P0);
    pose_at(seed, frame, P);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P0.R[k * 3 + i] * P.R[k * 3 + j];
            pose12[i * 4 + j] = s;
        }
        double s = 0;
        for (int k = 0; k < 3; k++) s += P0.R[k * 3 + i] * (P.t[k] - P0.t[k]);
        pose12[i * 4 + 3] = s;
    }
}

void pmv_synth_render(uint64_t seed, int frame, int w, int h, double fx, double fy, double cx, double cy,
                      uint8_t* out, int stride) {
    render(seed, frame, w, h, fx, fy, cx, cy, out, stride);
}

// frames [first, first+n) into out (n * w * h bytes, tightly packed), rows split over nthreads
void pmv_synth_sequence(uint64_t seed, int first, int n, int w, int h, double fx, double fy, double cx, double cy,
                        uint8_t* out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([=]() {
            for (int f = t; f < n; f += nthreads)
                render(seed, first + f, w, h, fx, fy, cx, cy, out + (size_t)f * w * h, w);
        });
    for (auto& x : th) x.join();
}
}
