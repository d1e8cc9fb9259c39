// Host data model of the VO pipeline: OpenCV-free mirrors of the reference's Feature / Feature3D / Frame and of its
// five plugin interfaces (the drop-in boundary).  Same names, argument meaning and container types as
//   /root/reference/include/Feature.h, Feature3D.h, Frame.h, Base{FeatureExtractor,FeatureMatcher,PnPSolver,
//   Triangulator,Optimizer}.h
// so that iteration orders (libstdc++ unordered_map + Feature::Hasher + the same insertion sequence) are reproduced
// exactly.  cv::Mat is replaced by Mat3 / Vec3 (row-major doubles).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include <list>

namespace vo {

struct Mat3 {
    double m[9];
    static Mat3 eye() { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0; return r; }
    double& operator()(int i, int j) { return m[i * 3 + j]; }
    double operator()(int i, int j) const { return m[i * 3 + j]; }
    Mat3 t() const { Mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i * 3 + j] = m[j * 3 + i]; return r; }
};
struct Vec3 {
    double v[3];
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
inline Mat3 operator*(const Mat3& a, const Mat3& b) {
    Mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
inline Vec3 operator*(const Mat3& a, const Vec3& b) {
    Vec3 r;
    for (int i = 0; i < 3; i++) r.v[i] = a.m[i * 3] * b.v[0] + a.m[i * 3 + 1] * b.v[1] + a.m[i * 3 + 2] * b.v[2];
    return r;
}
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3{{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; }
inline Vec3 operator-(const Vec3& a) { return Vec3{{-a.v[0], -a.v[1], -a.v[2]}}; }
inline Vec3 operator*(double s, const Vec3& a) { return Vec3{{s * a.v[0], s * a.v[1], s * a.v[2]}}; }

// std::hash<std::string>{}(std::to_string(v)) memoised for the coordinate range features can take
size_t coord_hash(int v);

class Feature3D;
// Feature.h:10-86
class Feature {
public:
    enum extractor { shi_tomasi, cv_good };
    int row = 0;
    int column = 0;
    extractor detector = cv_good;
    bool tracked = true;
    double score = 0;
    double displacement = 0;
    // Host-side shortcuts (not in the reference, no effect on any result): where this feature's entries live in its frame's
    // containers, so the back-end adapters do not hash/compare weak_ptr keys (several atomic reference-count round trips per
    // lookup). Unordered-map nodes are address-stable; a null slot means "look it up as the reference does".
    std::weak_ptr<Feature3D>* map_slot = nullptr;    // &frame.map[this] (set by the adapter that is about to use it)
    std::weak_ptr<Feature>* corr_slot = nullptr;     // &frame.feat_corr[this] when THIS object is the entry's key (set by the front-end)
    Feature* corr_feat = nullptr;                    // the feature *corr_slot refers to (frames, and with them their features, live for the whole run)
    const void* map_owner = nullptr;                 // the container each slot points into: a slot is only used for that container
    const void* corr_owner = nullptr;

    Feature(int column_, int row_) : row(row_), column(column_) {}
    Feature() { tracked = false; }

    struct Hasher {   // Feature.h:28-48
        std::size_t operator()(const std::weak_ptr<Feature>& f) const {
            const std::shared_ptr<Feature> p = f.lock();   // one atomic round trip instead of expired() + lock()
            if (!p) return 0;
            return coord_hash(p->column) ^ (coord_hash(p->row) << 1);
        }
        std::size_t operator()(const std::shared_ptr<Feature>& f) const {
            return coord_hash(f->column) ^ (coord_hash(f->row) << 1);
        }
    };
    // Feature.cpp:48-55: coordinate equality for weak_ptr keys (expired keys never compare equal)
    struct WeakEq {
        bool operator()(const std::weak_ptr<Feature>& a, const std::weak_ptr<Feature>& b) const {
            const std::shared_ptr<Feature> pa = a.lock();
            if (!pa) return false;
            const std::shared_ptr<Feature> pb = b.lock();
            if (!pb) return false;
            return pa->column == pb->column && pa->row == pb->row;
        }
    };
    // Feature.cpp:9-15 (Chebyshev distance)
    float distance(const Feature& f) const {
        int x = std::abs(column - f.column), y = std::abs(row - f.row);
        return (float)(x > y ? x : y);
    }
};

// Feature3D.h:6-146 — the point is float32 at rest (cv::Point3f), arithmetic in double (quirk Q7)
class Feature3D {
public:
    float x, y, z;
    int id = -1;   // creation order (explicit landmark id); not used by any arithmetic
    std::list<std::shared_ptr<Feature3D>>::iterator self;   // position in OdometryPipeline::feats3d (O(1) erase)
    Feature3D(double x_, double y_, double z_) : x((float)x_), y((float)y_), z((float)z_) {}
    void rotate(const Mat3& R) {   // Feature3D.cpp:125-139
        double x0 = R.m[0] * x + R.m[1] * y + R.m[2] * z;
        double y0 = R.m[3] * x + R.m[4] * y + R.m[5] * z;
        double z0 = R.m[6] * x + R.m[7] * y + R.m[8] * z;
        x = (float)x0; y = (float)y0; z = (float)z0;
    }
    void translate(const Vec3& t) {   // Feature3D.cpp:104-109
        x = (float)(x + t.v[0]); y = (float)(y + t.v[1]); z = (float)(z + t.v[2]);
    }
    void transform(const Mat3& R, const Vec3& t) { rotate(R); translate(t); }                 // :85-89
    void transformInv(const Mat3& R, const Vec3& t) { Mat3 inv = R.t(); translate(-t); rotate(inv); }   // :91-97
    void update(double x_, double y_, double z_) { x = (float)x_; y = (float)y_; z = (float)z_; }       // :111-116
    // Feature3D.cpp:18-33 (known-answer twin of ProjectionResidual)
    static void projectPoint(const double* R, const double* t, const double* camera, const double* p3, double* p2) {
        double xp = p3[0] - t[0], yp = p3[1] - t[1], zp = p3[2] - t[2];
        double xr = R[0] * xp + R[3] * yp + R[6] * zp, yr = R[1] * xp + R[4] * yp + R[7] * zp, zr = R[2] * xp + R[5] * yp + R[8] * zp;
        zr *= -1;
        double mz = zr ? 1. / zr : 1;
        p2[0] = xr * mz * camera[0] + camera[2];
        p2[1] = yr * mz * camera[4] + camera[5];
    }
};

// Gray image view: host pixels (CPU plugins) and/or a device frame slot (HIP plugins). A grid cell is a sub-view that
// shares the parent's storage (Frame::regionOfInterest, Frame.cpp:95-117).
struct ImageView {
    const uint8_t* host = nullptr;   // full image, row stride = full_w
    int slot = -1;                   // device frame slot of the full image
    int full_w = 0, full_h = 0;
    int x0 = 0, y0 = 0, w = 0, h = 0;   // this view inside the full image
};

// Frame.h:12-105
// One entry of Frame::map as OpenCVEPnPSolver's gather loop (OpenCVEPnPSolver.cpp:13-28) will meet it, as far as that is known once
// the next frame exists (everything but whether the landmark is still alive): built by the front-end thread, which has the
// features of both frames in cache, so that the back-end's loop is one pass over a flat list instead of a walk through the
// hash-map nodes and feature objects of two frames (host-side shortcut like Feature::map_slot: no effect on any result).
struct PnPLink {
    std::weak_ptr<Feature3D>* src_val = nullptr;          // &entry.second: the landmark of the source feature
    std::weak_ptr<Feature3D>* next_slot = nullptr;        // kind 1: &next.map[corresponding feature] (the node operator[] would reach)
    const Feature* f = nullptr;                           // kind 1: the corresponding feature (its coordinates are the image point)
    const std::shared_ptr<Feature>* key = nullptr;        // &entry.first (kind 2 runs the reference's lookups on it)
    int key_column = 0, key_row = 0;                      // the entry's own feature (the observation CeresBundleAdjustment.cpp:28-33 reads)
    unsigned char kind = 2;                               // 0 no correspondence; 1 as above; 2 same-pixel twin / absent node: the slow path, in sequence
};
// One entry of Frame::feat_corr as OpenCVFivePointTri's gather loop (OpenCVFivePointTri.cpp:9-22) will meet it: both features and the
// nodes of the two frames' maps that receive the triangulated landmark (:47-50). Built by the front-end like PnPLink.
struct TriLink {
    Feature* fst = nullptr; Feature* sec = nullptr;
    std::weak_ptr<Feature3D>* src_slot = nullptr; std::weak_ptr<Feature3D>* next_slot = nullptr;
};

class Frame {
public:
    std::unordered_map<std::shared_ptr<Feature>, std::weak_ptr<Feature3D>, Feature::Hasher> map;
    std::unordered_map<std::weak_ptr<Feature>, std::weak_ptr<Feature>, Feature::Hasher, Feature::WeakEq> feat_corr;
    ImageView bw;
    int frame = 0;
    std::vector<PnPLink> pnp_links;       // one per map entry, in iteration order; valid for solvePnP(*this, next) iff ...
    const void* pnp_links_for = nullptr;  // ... == &next.map (a moved or copied frame has another address: self-invalidating)
    std::vector<TriLink> tri_links;       // one per live feat_corr entry, in iteration order; valid under the same condition + ...
    const void* tri_links_src = nullptr;  // ... == &map (of this frame)
    bool links_cover_map() const { return pnp_links_for != nullptr && pnp_links.size() == map.size() && tri_links_src == (const void*)&map; }

    Frame() {}
    explicit Frame(const ImageView& img) : bw(img) {}
    bool isEmpty() const { return bw.w == 0; }
    Frame regionOfInterest(int rx, int ry, int rw, int rh) const {   // Frame.cpp:95-117
        Frame f;
        f.bw = bw;
        f.bw.x0 = bw.x0 + rx; f.bw.y0 = bw.y0 + ry; f.bw.w = rw; f.bw.h = rh;
        return f;
    }
    int count3DPoints() const {   // Frame.cpp:14-24
        int c = 0;
        if (links_cover_map()) { for (const PnPLink& L : pnp_links) if (!L.src_val->expired()) c++; }   // the same entries, from the flat list
        else for (auto& p : map) if (!p.second.expired()) c++;
        return c;
    }
    bool hasNeighbor(const Feature& f, int dist = 5) const {   // Frame.cpp:3-12
        for (auto& p : map) if (f.distance(*p.first) < dist) return true;
        return false;
    }
};

typedef std::unordered_map<std::weak_ptr<Feature>, std::weak_ptr<Feature>, Feature::Hasher, Feature::WeakEq> fmap;

// Frame::hasNeighbor for many candidates against one (growing) feature set: the reference runs its O(N) scan per candidate
// (OdometryPipeline.cpp:359-366, 2000 x 1350 distance evaluations per re-detection at configs[3]); an occupancy grid with
// `dist`-sized buckets answers the same question — is there a feature with Chebyshev distance < dist — from the 3 x 3 buckets around
// the candidate. Same boolean for every candidate, including features added while the loop runs (add()).
class NeighborGrid {
public:
    explicit NeighborGrid(const Frame& fr, int dist_ = 5) : dist(dist_) {
        cells.reserve(fr.map.size() * 2 + 16);
        for (auto& p : fr.map) add(p.first->column, p.first->row);
    }
    void add(int column, int row) { cells[key(bucket(column), bucket(row))].push_back({column, row}); }
    bool hasNeighbor(int column, int row) const {
        const int bx = bucket(column), by = bucket(row);
        for (int j = -1; j <= 1; j++)
            for (int i = -1; i <= 1; i++) {
                auto it = cells.find(key(bx + i, by + j));
                if (it == cells.end()) continue;
                for (auto& q : it->second) {
                    const int x = std::abs(column - q.first), y = std::abs(row - q.second);
                    if ((float)(x > y ? x : y) < dist) return true;   // Feature::distance(f) < dist
                }
            }
        return false;
    }
private:
    int dist;
    std::unordered_map<long long, std::vector<std::pair<int, int>>> cells;
    int bucket(int v) const { return v >= 0 ? v / dist : -((-v + dist - 1) / dist); }   // floor division
    static long long key(int bx, int by) { return ((long long)bx << 32) ^ (long long)(unsigned)by; }
};

// ---- plugin interfaces (Base*.h) --------------------------------------------------------------------------
class BaseFeatureExtractor {
public:
    virtual ~BaseFeatureExtractor() {}
    // BaseFeatureExtractor.h:21 — src is one grid cell (or any sub-view); features are returned in cell coordinates
    virtual std::vector<Feature> extractFeatures(Frame& src, int max) = 0;
    // Batched form used by the pipeline for the whole grid (OdometryPipeline.cpp:357 / :450 loop); the default is the
    // reference's per-cell loop, device plugins override it to issue one launch for all cells.
    virtual std::vector<std::vector<Feature>> extractGrid(std::vector<Frame>& cells, int max) {
        std::vector<std::vector<Feature>> out;
        for (auto& c : cells) out.push_back(extractFeatures(c, max));
        return out;
    }
};
class BaseFeatureMatcher {
public:
    virtual ~BaseFeatureMatcher() {}
    virtual fmap matchFeatures(Frame& src, Frame& next) = 0;   // BaseFeatureMatcher.h:22
};
class BasePnPSolver {
public:
    virtual ~BasePnPSolver() {}
    virtual void solvePnP(Frame& src, Frame& next, Mat3& R, Vec3& t) = 0;   // BasePnPSolver.h:19
};
class BaseTriangulator {
public:
    virtual ~BaseTriangulator() {}
    virtual void triangulate(Frame& src, Frame& next, Mat3& R, Vec3& t) = 0;   // BaseTriangulator.h:20
    // Called by the front-end thread of the two-thread pipeline as soon as `prev.feat_corr` (the correspondences prev -> the frame
    // just tracked) is final. A triangulator may start whatever depends on nothing but these correspondences; the back-end reaches
    // this frame pair at least one frame later. Default: nothing.
    virtual void prefetch(const Frame& prev) { (void)prev; }
    // End of the run (both pipeline threads are done): whatever prefetch() started must be finished or dropped before this returns.
    virtual void finish() {}
};
class BaseOptimizer {
public:
    virtual ~BaseOptimizer() {}
    virtual void apply(Frame& src) = 0;   // BaseOptimizer.h:15
};

}  // namespace vo
