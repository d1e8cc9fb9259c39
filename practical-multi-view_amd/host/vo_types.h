// Host data model of the VO pipeline: OpenCV-free mirrors of the reference's Feature / Feature3D / Frame and of its
// five plugin interfaces (the drop-in boundary).  Same names and argument meaning as
//   /root/reference/include/Feature.h, Feature3D.h, Frame.h, Base{FeatureExtractor,FeatureMatcher,PnPSolver,
//   Triangulator,Optimizer}.h
// The reference keeps a frame's features in std::unordered_map<shared_ptr<Feature>, weak_ptr<Feature3D>> / <weak_ptr, weak_ptr>
// and its landmarks as shared_ptr<Feature3D>: a heap node, a control block and several atomic reference-count round trips per
// feature and frame. Here they are TRACK TABLES - plain arrays per frame (coordinates, landmark id) and per run (landmark xyz,
// liveness) - walked in exactly the order libstdc++ would walk the reference's containers (HashOrder: the same hash codes, the
// same insertion sequence, the same bucket growth), because that order is part of the result. The tables are
// what a device kernel can read as they are.  cv::Mat is replaced by Mat3 / Vec3 (row-major doubles).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

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
// Feature::Hasher (Feature.h:28-48) of a feature at (column, row): h1 ^ (h2 << 1)
inline size_t feature_hash(int column, int row) { return coord_hash(column) ^ (coord_hash(row) << 1); }

// Feature.h:10-86 as a plain value: what an extractor returns. (Inside a Frame a feature is a row of the frame's tables, below.)
class Feature {
public:
    enum extractor { shi_tomasi, cv_good };
    int row = 0;
    int column = 0;
    extractor detector = cv_good;
    bool tracked = true;
    double score = 0;
    double displacement = 0;
    Feature(int column_, int row_) : row(row_), column(column_) {}
    Feature() { tracked = false; }
    // Feature.cpp:9-15 (Chebyshev distance)
    float distance(const Feature& f) const {
        int x = std::abs(column - f.column), y = std::abs(row - f.row);
        return (float)(x > y ? x : y);
    }
};

// Feature3D.h:6-146 - the point is float32 at rest (cv::Point3f), arithmetic in double (quirk Q7). The pipeline keeps its landmarks as
// rows of a LandmarkTable (below); this value type carries the arithmetic of Feature3D.cpp:85-139 for one row.
class Feature3D {
public:
    float x, y, z;
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

// The landmarks of a run: OdometryPipeline::feats3d (a vector of shared_ptr<Feature3D> in the reference) as a table. A landmark id
// is its creation number; `alive[id]` is what `weak_ptr<Feature3D>::expired()` asks in the reference - the only owner of a landmark is
// feats3d, and the only erase is the RANSAC-outlier erase of OpenCVEPnPSolver.cpp:40-49. Rows are plain arrays so that the same table
// can be mirrored in HBM.
struct LandmarkTable {
    std::vector<float> xyz;          // 3 per landmark, float32 at rest (cv::Point3f)
    std::vector<uint8_t> alive;
    int n_alive = 0;                 // feats3d.size()
    int size() const { return (int)alive.size(); }
    int create(const Feature3D& p) { xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z); alive.push_back(1); n_alive++; return (int)alive.size() - 1; }
    bool expired(int id) const { return id < 0 || !alive[(size_t)id]; }
    void erase(int id) { if (alive[(size_t)id]) { alive[(size_t)id] = 0; n_alive--; } }
    Feature3D get(int id) const { return Feature3D(xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2]); }
    void put(int id, const Feature3D& p) { xyz[3 * (size_t)id] = p.x; xyz[3 * (size_t)id + 1] = p.y; xyz[3 * (size_t)id + 2] = p.z; }
};

// Iteration order of a libstdc++ std::unordered_map (unique keys) for a given sequence of insertions, without the map: the
// reference's results depend on the order in which Frame::map and Frame::feat_corr are WALKED (the order of the LK points, of the
// PnP object points - RANSAC samples index into it - and of the BA residual blocks), and that order is a function of
// the hash codes, the insertion sequence and the growth of the bucket array alone. This class replays exactly that function -
// _Hashtable::_M_insert_unique_node / _M_insert_bucket_begin / _M_rehash_aux and the bucket counts of _Prime_rehash_policy, which
// are read off a real std::unordered_map at start-up - over node indices: node i is the i-th inserted element; a lookup walks the
// bucket's chain like _M_find_before_node. No allocation per node, no reference counts; the per-frame tables it orders are plain
// arrays. Checked against the real container in tests/test_host_logic.py (random codes, collisions, growth through every prime)
// and, end to end, against tests/twin/ref_twin.cpp, which keeps the reference's own containers.
class HashOrder {
public:
    int size() const { return (int)next_.size(); }
    int head() const { return head_; }                       // first node in iteration order, -1 when empty
    int next(int node) const { return next_[(size_t)node]; }   // -1 after the last
    size_t bucket_count() const { return nb_; }
    void reserve_nodes(size_t n) { next_.reserve(n); code_.reserve(n); }
    // an empty table again (one bucket, like a fresh container) that keeps its allocations: the tables of a retired frame serve the next
    void clear() { next_.clear(); code_.clear(); bprev_.clear(); head_ = -1; nb_ = 1; }
    int insert(size_t code);                                 // a new node (the caller has made sure no equal key exists); returns its index
    template <class Eq> int find(size_t code, Eq eq) const {  // node with this hash code for which eq(node) holds, or -1
        const size_t bkt = code % nb_;
        const int prev = bprev_.empty() ? EMPTY : bprev_[bkt];
        if (prev == EMPTY) return -1;
        int p = prev == BEFORE_BEGIN ? head_ : next_[(size_t)prev];
        for (;;) {
            if (code_[(size_t)p] == code && eq(p)) return p;
            const int nx = next_[(size_t)p];
            if (nx < 0 || code_[(size_t)nx] % nb_ != bkt) return -1;
            p = nx;
        }
    }
    // the nodes in iteration order
    template <class F> void for_each(F f) const { for (int p = head_; p >= 0; p = next_[(size_t)p]) f(p); }
    // bucket count of a std::unordered_map that started empty and now holds n elements (n insertions, no erase)
    static size_t buckets_for(size_t n);
private:
    static constexpr int EMPTY = -1, BEFORE_BEGIN = -2;
    std::vector<int> next_;       // per node
    std::vector<size_t> code_;    // per node: cached hash code
    std::vector<int> bprev_;      // per bucket: the node BEFORE the bucket's first node (BEFORE_BEGIN = the list head), EMPTY = no node
    std::vector<int> bscratch_;   // rehash builds the new bucket array here and swaps: no allocation once both have grown
    int head_ = -1;
    size_t nb_ = 1;
    void rehash(size_t n);
};

// Gray image view: host pixels (CPU plugins) and/or a device frame slot (HIP plugins). A grid cell is a sub-view that
// shares the parent's storage (Frame::regionOfInterest, Frame.cpp:95-117).
struct ImageView {
    const uint8_t* host = nullptr;   // full image, row stride = full_w
    int slot = -1;                   // device frame slot of the full image
    int full_w = 0, full_h = 0;
    int x0 = 0, y0 = 0, w = 0, h = 0;   // this view inside the full image
};

// BaseFeatureMatcher::fmap = unordered_map<weak_ptr<Feature>, weak_ptr<Feature>, Feature::Hasher> (BaseFeatureMatcher.h:13) and
// Frame::feat_corr: source feature -> feature of the next frame. Keys compare by COORDINATES (Feature.cpp:48-55), so two source
// features on the same pixel share one entry: the first stays the key, the last value wins. Entry n: key[n] = index of the source
// feature in the source frame's table, val[n] = index in the next frame's table, -1 for the empty value operator[] leaves (quirk Q10).
struct FeatureCorr {
    std::vector<int> key, val;
    HashOrder order;
    size_t size() const { return key.size(); }
    void clear() { key.clear(); val.clear(); order.clear(); }
};
typedef FeatureCorr fmap;

// Frame.h:12-105. Frame::map - unordered_map<shared_ptr<Feature>, weak_ptr<Feature3D>, Feature::Hasher> with POINTER equality, so
// every inserted feature is its own entry, same pixel or not - as a table: feature e = (column[e], row[e]) -> landmark id lm[e]
// (-1: the empty weak_ptr), walked in map_order's order.
class Frame {
public:
    std::vector<int> column, row, lm;
    HashOrder map_order;
    FeatureCorr feat_corr;
    ImageView bw;
    int frame = 0;

    Frame() {}
    explicit Frame(const ImageView& img) : bw(img) {}
    // a fresh Frame(img) in the storage of an old one (OdometryPipeline recycles the tables of frames nothing reads any more)
    void reset(const ImageView& img) { column.clear(); row.clear(); lm.clear(); map_order.clear(); feat_corr.clear(); bw = img; frame = 0; }
    bool isEmpty() const { return bw.w == 0; }
    int n_features() const { return (int)lm.size(); }                         // map.size()
    int add_feature(int column_, int row_, int landmark) {                    // map[make_shared<Feature>(..)] = landmark
        column.push_back(column_); row.push_back(row_); lm.push_back(landmark);
        return map_order.insert(feature_hash(column_, row_));
    }
    template <class F> void for_each_feature(F f) const { map_order.for_each(f); }   // for (auto& p : map), p = feature index
    // feat_corr.find(feature e of this frame): the entry whose key has e's coordinates, or -1
    int corr_find(const FeatureCorr& c, int e) const {
        const int col = column[(size_t)e], rw = row[(size_t)e];
        return c.order.find(feature_hash(col, rw), [&](int n) { const int k = c.key[(size_t)n]; return column[(size_t)k] == col && row[(size_t)k] == rw; });
    }
    // feat_corr[feature e]: the entry (inserted with an empty value when absent: operator[])
    int corr_at(FeatureCorr& c, int e) const {
        int n = corr_find(c, e);
        if (n < 0) { c.key.push_back(e); c.val.push_back(-1); n = c.order.insert(feature_hash(column[(size_t)e], row[(size_t)e])); }
        return n;
    }
    Frame regionOfInterest(int rx, int ry, int rw, int rh) const {   // Frame.cpp:95-117 (a view of the pixels; no features)
        Frame f;
        f.bw = bw;
        f.bw.x0 = bw.x0 + rx; f.bw.y0 = bw.y0 + ry; f.bw.w = rw; f.bw.h = rh;
        return f;
    }
    int count3DPoints(const LandmarkTable& L) const {   // Frame.cpp:14-24
        int c = 0;
        for (int id : lm) if (!L.expired(id)) c++;
        return c;
    }
    bool hasNeighbor(const Feature& f, int dist = 5) const {   // Frame.cpp:3-12
        for (size_t e = 0; e < lm.size(); e++) if (f.distance(Feature(column[e], row[e])) < dist) return true;
        return false;
    }
};

// Frame::hasNeighbor for many candidates against one (growing) feature set: the reference runs its O(N) scan per candidate
// (OdometryPipeline.cpp:359-366, 2000 x 1350 distance evaluations per re-detection at configs[3]); an occupancy grid with
// `dist`-sized buckets answers the same question - is there a feature with Chebyshev distance < dist - from the 3 x 3 buckets around
// the candidate. Same boolean for every candidate, including features added while the loop runs (add()).
class NeighborGrid {
public:
    explicit NeighborGrid(const Frame& fr, int dist_ = 5) : dist(dist_) {
        cells.reserve(fr.lm.size() * 2 + 16);
        for (size_t e = 0; e < fr.lm.size(); e++) add(fr.column[e], fr.row[e]);
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
    // Called by the front-end thread of the two-thread pipeline as soon as `prev.feat_corr` (the correspondences prev -> `next`, the
    // frame just tracked) is final. A triangulator may start whatever depends on nothing but these correspondences; the back-end
    // reaches this frame pair at least one frame later. Default: nothing.
    virtual void prefetch(const Frame& prev, const Frame& next) { (void)prev; (void)next; }
    // End of the run (both pipeline threads are done): whatever prefetch() started must be finished or dropped before this returns.
    virtual void finish() {}
};
class BaseOptimizer {
public:
    virtual ~BaseOptimizer() {}
    virtual void apply(Frame& src) = 0;   // BaseOptimizer.h:15
};

}  // namespace vo
