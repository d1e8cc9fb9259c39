// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
// CPU plugin set: the oracle's restatements plugged behind the SAME Base* adapters / pipeline as the product
// (practical-multi-view_amd/host/vo_pipeline.*), giving the reference-equivalent CPU pipeline that
//   * tests compare the HIP pipeline against (tracks bit-exact, poses within tolerance), and
//   * bench.py times as `cpu_baseline` (kind "port") on the GPU box's host cores, with the reference's threading model
//     (front-end + back-end threads, LK parallel over tracks).
#include "orc_api.h"
#include "orc_fast.h"
#include "vo_capi_impl.h"
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <cstdlib>
#include <stdexcept>

namespace {
using namespace vo;

struct CpuGftt : GoodFeatureExtractorBase {
    orc::Pool* pool = nullptr;   // fast mode: the grid cells are independent calls -> side by side (same results)
    void gftt(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out) override {
        out.assign(cells.size(), {});
        const size_t cap = (size_t)(max > 0 ? max : 65536);   // max <= 0: no limit (at most one corner per pixel of a 255x255 cell)
        auto work = [&](int lo, int hi) {
            std::vector<int> xy(cap * 2);
            for (int k = lo; k < hi; k++) {
                const ImageView& c = cells[k];
                const int n = orc::gftt_cell(c.host, c.full_w, c.full_h, c.x0, c.y0, c.w, c.h, max, quality, min_distance, xy.data(), nullptr);
                for (int i = 0; i < n; i++) out[k].push_back({xy[2 * i], xy[2 * i + 1]});
            }
        };
        if (pool) pool->parallel_for((int)cells.size(), (int)cells.size(), work); else work(0, (int)cells.size());
    }
};
struct CpuShiTomasi : ShiTomasiExtractorBase {
    void shitomasi(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
                   std::vector<std::vector<double>>& score) override {
        out.clear(); score.clear();
        std::vector<int> xy((size_t)std::max(max, 1) * 2);
        std::vector<double> sc((size_t)std::max(max, 1));
        for (auto& c : cells) {
            const int n = orc::shitomasi_cell(c.host, c.full_w, c.x0, c.y0, c.w, c.h, max, quality, xy.data(), sc.data(), nullptr);
            std::vector<std::pair<int, int>> v;
            for (int i = 0; i < n; i++) v.push_back({xy[2 * i], xy[2 * i + 1]});
            out.push_back(v);
            score.push_back(std::vector<double>(sc.begin(), sc.begin() + n));
        }
    }
};
struct CpuFast : FastExtractorBase {
    void fast(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
              std::vector<std::vector<float>>& response) override {
        out.assign(cells.size(), {});
        response.assign(cells.size(), {});
        if (max < 1) return;
        std::vector<int> xy((size_t)max * 2);
        std::vector<float> rs((size_t)max);
        for (size_t k = 0; k < cells.size(); k++) {
            const ImageView& c = cells[k];
            const int n = orc::fast9_cell(c.host, c.full_w, c.x0, c.y0, c.w, c.h, threshold, nonmax, max, xy.data(), rs.data());
            for (int i = 0; i < n; i++) { out[k].push_back({xy[2 * i], xy[2 * i + 1]}); response[k].push_back(rs[i]); }
        }
    }
};
struct CpuKnn : KnnFeatureMatcherBase {
    void knn(const ImageView& src, const ImageView& next, const int* src_xy, int n, const int* cmp_xy, int m, int* best, float* err) override {
        orc::knn_match(src.host, next.host, src.full_w, src.full_h, src_xy, n, cmp_xy, m, neighbours, window, best, err);
    }
};
struct CpuLK : LucasKanadeFMBase {
    int nthreads = 1;
    orc::Pool* pool = nullptr;   // fast mode (cpu_baseline timing): padded-buffer LK on a persistent pool, bit-identical results
    void pyrlk(const ImageView& prev, const ImageView& next, const float* prev_xy, int n, float* next_xy, uint8_t* status,
               float* err) override {
        // EXPERIMENT (ORC_EXPERIMENT_LK_ROUND=1): the adapter truncates the sub-pixel LK result toward zero
        // (OpenCVLucasKanadeFM.cpp:25); + 0.5 here turns that truncation into rounding, to see how much of the trajectory's
        // heading drift the truncation explains. Never set in tests or benchmarks.
        static const bool round_exp = getenv("ORC_EXPERIMENT_LK_ROUND") && atoi(getenv("ORC_EXPERIMENT_LK_ROUND")) != 0;
        struct Rounder { float* xy; int n; bool on; ~Rounder() { if (on) for (int i = 0; i < 2 * n; i++) xy[i] += 0.5f; } } rounder{next_xy, n, round_exp};
        if (pool) {
            orc::LKParams P;
            orc::lk_track_fast(prev.host, next.host, prev.full_w, prev.full_h, prev_xy, n, P, next_xy, status, err, pool);
            return;
        }
        orc::Image8 a(prev.full_w, prev.full_h), b(next.full_w, next.full_h);
        memcpy(a.d.data(), prev.host, a.d.size());
        memcpy(b.d.data(), next.host, b.d.size());
        orc::LKParams P;
        orc::lk_track(a, b, prev_xy, n, P, next_xy, status, err, nullptr, nthreads);
    }
};
struct CpuPnP : EPnPSolverBase {
    bool pnp_ransac(const float* obj, const float* img, int m, const double* K, double* rvec, double* tvec,
                    std::vector<int>& inliers) override {
        inliers.assign(std::max(m, 1), 0);
        if (m < 6) { inliers.clear(); return false; }   // same guard as the product ABI (the reference never gets here: m >= tol)
        const int n = orc::pnp_ransac(obj, img, m, K, rvec, tvec, 100, 8.f, .99, inliers.data(), nullptr);
        inliers.resize(n > 0 ? n : 0);
        return n > 0;
    }
};
struct CpuBA : BundleAdjustmentBase {
    orc::Pool* pool = nullptr;   // fast mode: residual evaluation on 4 threads (CeresBundleAdjustment.cpp:58), this run's own pool
    int calls = 0;
    void ba_solve(double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
                  const double* K, double huber, int max_iterations) override {
        orc::BASummary s;
        // test hook: ORC_FAIL_BA_CALL=k makes the k-th solve of a run throw, like a capacity error of a device plugin would (the
        // error path of OdometryPipeline::run_threaded has no other way to be exercised on the CPU)
        if (const char* e = getenv("ORC_FAIL_BA_CALL")) if (++calls == atoi(e)) throw std::runtime_error("oracle: forced BA failure (ORC_FAIL_BA_CALL)");
        orc::ba_set_pool(pool, 4);   // per calling thread: the back-end thread of THIS pipeline
        orc::ba_solve(cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, &s);
    }
};
}  // namespace

extern "C" {
void* orc_pipeline_run(const vo::PipelineParams* P, const uint8_t* frames, const double* K9, const double* gt_poses12) {
    auto* run = new vo::PipelineRun();
    vo::pipeline_setup(*run, *P, frames, K9, gt_poses12);
    // reserved bit 0 = "fast": the speed-oriented twins of orc_fast.cpp (bench.py's cpu_baseline); results are bit-identical
    // (two pools: a Pool serves one caller at a time, and the front-end and back-end threads run concurrently)
    std::unique_ptr<orc::Pool> pool, ba_pool;
    // (n_threads == 1 in fast mode: no helper threads at all - the single-thread form bench.py runs 16 of side by side)
    if (P->reserved & 1) { pool.reset(new orc::Pool(std::max(0, P->n_threads - 1))); if (P->n_threads > 1) ba_pool.reset(new orc::Pool(std::min(3, P->n_threads - 1))); }
    vo::BaseFeatureExtractor* ex;
    if (P->extractor == 1) ex = new CpuShiTomasi(); else if (P->extractor == 2) ex = new CpuFast(); else { auto* g = new CpuGftt(); g->pool = pool.get(); ex = g; }
    run->owned_ex.push_back(ex);
    vo::BaseFeatureMatcher* lk;
    if (P->matcher == 1) { auto* k = new CpuKnn(); k->extractor = ex; lk = k; }
    else { auto* l = new CpuLK(); l->nthreads = P->n_threads; l->pool = pool.get(); lk = l; }
    auto* pnp = new CpuPnP(); pnp->tracker = &run->pipe;
    auto* tri = new vo::FivePointTri(); tri->tracker = &run->pipe; tri->workers = std::max(1, std::min(P->n_threads, 8));
    tri->prefetch_threads = P->n_threads > 1 ? 2 : 0;   // same host code as the product: essential matrices ahead of time
    auto* ba = new CpuBA(); ba->tracker = &run->pipe; ba->pool = ba_pool.get();
    run->m = lk; run->p = pnp; run->tr = tri; run->b = ba;
    run->pipe.extractor = ex; run->pipe.matcher = lk; run->pipe.pnpsolver = pnp; run->pipe.triangulator = tri; run->pipe.ba = ba;
    try {
        vo::pipeline_execute(*run, *P);
    } catch (const std::exception&) {   // a plugin error ended the run (both pipeline threads are joined by then): no result
        delete run;
        return nullptr;
    }
    orc::ba_set_pool(nullptr, 1);
    ba->pool = nullptr;   // the pools end with this call
    return run;
}
void orc_pipeline_free(void* h) { delete (vo::PipelineRun*)h; }
int orc_pipeline_num_poses(void* h) { return vo::pipeline_num_poses(*(vo::PipelineRun*)h); }
void orc_pipeline_get_poses(void* h, double* out) { vo::pipeline_get_poses(*(vo::PipelineRun*)h, out); }
int orc_pipeline_num_frames(void* h) { return vo::pipeline_num_frames(*(vo::PipelineRun*)h); }
int orc_pipeline_frame_feature_count(void* h, int k) { return vo::pipeline_frame_feature_count(*(vo::PipelineRun*)h, k); }
int orc_pipeline_frame_corr_count(void* h, int k) { return vo::pipeline_frame_corr_count(*(vo::PipelineRun*)h, k); }
void orc_pipeline_get_frame_features(void* h, int k, int* out) { vo::pipeline_get_frame_features(*(vo::PipelineRun*)h, k, out); }
int orc_pipeline_stats_count(void) { return vo::PIPELINE_STATS_COUNT; }
void orc_pipeline_get_stats(void* h, double* out25) { vo::pipeline_get_stats(*(vo::PipelineRun*)h, out25); }
}

// ---- host-logic probes for the CPU test-suite (KA5/KA6/KA9/KA10 of SURVEY.md §8c) ------------------------------------------
extern "C" {
uint64_t orc_host_coord_hash(int v) { return (uint64_t)vo::coord_hash(v); }
// getGridROI: returns the number of cells; out: (x0, y0, w, h, gx, gy) per cell
int orc_host_grid(int w, int h, int* out) {
    vo::OdometryPipeline pl;
    vo::ImageView v; v.full_w = w; v.full_h = h; v.w = w; v.h = h;
    vo::Frame fr(v);
    auto roi = pl.getGridROI(fr);
    for (size_t i = 0; i < roi.size(); i++) {
        out[6 * i] = roi[i].frame.bw.x0; out[6 * i + 1] = roi[i].frame.bw.y0; out[6 * i + 2] = roi[i].frame.bw.w; out[6 * i + 3] = roi[i].frame.bw.h;
        out[6 * i + 4] = roi[i].x; out[6 * i + 5] = roi[i].y;
    }
    return (int)roi.size();
}
double orc_host_stddev(const double* v, int n) { return vo::OdometryPipeline::standardDeviation(std::vector<double>(v, v + n)); }
double orc_host_yrot(const double* R9, int flip) { vo::Mat3 R; memcpy(R.m, R9, 72); return vo::OdometryPipeline::calcYRotation(R, flip != 0); }
// insert n features (col,row) into a Frame::map in the given order; out = iteration order as indices into the input
void orc_host_map_order(const int* cols, const int* rows, int n, int* out) {
    vo::Frame fr;
    for (int i = 0; i < n; i++) fr.add_feature(cols[i], rows[i], -1);
    int k = 0;
    fr.for_each_feature([&](int e) { out[k++] = e; });
}
// the same insertions into the reference's own container type (std::unordered_map<shared_ptr<Feature>, weak_ptr<Feature3D>,
// Feature::Hasher>, pointer equality) written out here: what vo::HashOrder has to reproduce
}  // extern "C"
namespace {
struct RefFeature { int row, column; };
struct RefHasher {   // Feature.h:28-48
    std::size_t operator()(const std::shared_ptr<RefFeature> f) const {
        size_t const h1(std::hash<std::string>{}(std::to_string(f->column)));
        size_t const h2(std::hash<std::string>{}(std::to_string(f->row)));
        return h1 ^ (h2 << 1);
    }
};
}
extern "C" {
void orc_host_map_order_stdlib(const int* cols, const int* rows, int n, int* out) {
    std::unordered_map<std::shared_ptr<RefFeature>, int, RefHasher> m;
    for (int i = 0; i < n; i++) m[std::make_shared<RefFeature>(RefFeature{rows[i], cols[i]})] = i;
    int k = 0;
    for (auto& p : m) out[k++] = p.second;
}
// vo::HashOrder against std::unordered_map on arbitrary hash codes (collisions, equal codes, growth through many bucket counts):
// n keys with the given codes are inserted in order into both; returns the number of positions where the iteration orders differ
// (0 = identical), and also checks find() for every key and for n absent codes
}  // extern "C"
namespace {
struct CodeKey { size_t code; int id; bool operator==(const CodeKey& o) const { return id == o.id; } };
struct CodeHash { size_t operator()(const CodeKey& k) const { return k.code; } };   // (not noexcept, like Feature::Hasher)
}
extern "C" {
int orc_host_hashorder_check(const uint64_t* codes, int n) {
    std::unordered_map<CodeKey, int, CodeHash> real;
    vo::HashOrder emu;
    int bad = 0;
    for (int i = 0; i < n; i++) {
        real[CodeKey{(size_t)codes[i], i}] = i;
        if (emu.insert((size_t)codes[i]) != i) bad++;
        if (real.bucket_count() != emu.bucket_count()) bad++;
        if ((i & (i + 1)) == 0 || i == n - 1) {   // compare the whole order at sizes 1, 2, 4, ... and at the end
            int p = emu.head();
            for (auto& kv : real) { if (p != kv.second) bad++; p = p >= 0 ? emu.next(p) : -1; }
            if (p != -1) bad++;
        }
    }
    for (int i = 0; i < n; i++) {
        if (emu.find((size_t)codes[i], [&](int node) { return node == i; }) != i) bad++;
        if (emu.find((size_t)codes[i] + 0x9e3779b97f4a7c15ull, [&](int) { return true; }) != -1 && real.find(CodeKey{(size_t)codes[i] + 0x9e3779b97f4a7c15ull, -1}) == real.end()) {
            // an absent code may still share a bucket with present nodes; a hit needs an equal code, which the predicate-less probe accepts
            bool exists = false;
            for (int j = 0; j < n; j++) if ((size_t)codes[j] == (size_t)codes[i] + 0x9e3779b97f4a7c15ull) exists = true;
            if (!exists) bad++;
        }
    }
    return bad;
}
// feat_corr semantics: keys compare by coordinates (same-pixel sources collapse, last value wins). Returns the number of
// entries; out_key / out_val = indices of the surviving key feature and of its value, in iteration order
int orc_host_corr_order(const int* cols, const int* rows, int n, int* out_key, int* out_val) {
    vo::Frame src, next;
    vo::fmap corr;
    for (int i = 0; i < n; i++) {
        const int a = src.add_feature(cols[i], rows[i], -1);
        const int b = next.add_feature(cols[i] + 1000, rows[i] + 1000, -1);
        corr.val[(size_t)src.corr_at(corr, a)] = b;   // corr[a] = b
    }
    int k = 0;
    corr.order.for_each([&](int c) { out_key[k] = corr.key[(size_t)c]; out_val[k] = corr.val[(size_t)c]; k++; });
    return k;
}
// the same through the reference's own container type: unordered_map<weak_ptr<Feature>, weak_ptr<Feature>, Hasher> with the
// coordinate operator== of Feature.cpp:48-55
}  // extern "C"
namespace refcorr {
struct Feature { int row, column; };
struct Hasher {
    std::size_t operator()(const std::weak_ptr<Feature> f) const {
        if (f.expired()) return 0;
        std::shared_ptr<Feature> p = f.lock();
        size_t const h1(std::hash<std::string>{}(std::to_string(p->column)));
        size_t const h2(std::hash<std::string>{}(std::to_string(p->row)));
        return h1 ^ (h2 << 1);
    }
};
bool operator==(const std::weak_ptr<Feature> lhs, const std::weak_ptr<Feature> rhs) {
    if (lhs.expired() || rhs.expired()) return false;
    std::shared_ptr<Feature> a = lhs.lock(), b = rhs.lock();
    return a->column == b->column && a->row == b->row;
}
}
extern "C" {
int orc_host_corr_order_stdlib(const int* cols, const int* rows, int n, int* out_key, int* out_val) {
    typedef refcorr::Feature RF;
    std::unordered_map<std::weak_ptr<RF>, std::weak_ptr<RF>, refcorr::Hasher> corr;
    std::vector<std::shared_ptr<RF>> src, dst;
    std::unordered_map<RF*, int> si, di;
    for (int i = 0; i < n; i++) {
        auto a = std::make_shared<RF>(RF{rows[i], cols[i]});
        auto b = std::make_shared<RF>(RF{rows[i] + 1000, cols[i] + 1000});
        src.push_back(a); dst.push_back(b); si[a.get()] = i; di[b.get()] = i;
        corr[a] = b;
    }
    int k = 0;
    for (auto& p : corr) { out_key[k] = si[p.first.lock().get()]; out_val[k] = di[p.second.lock().get()]; k++; }
    return k;
}
// OdometryPipeline.cpp:407 trigger and CeresBundleAdjustment.cpp:7-8,20-23 window, PROBED on the real code: a pipeline of
// n_frames frames (every frame observes one common landmark, so every window camera enters the problem; t[i] = (i, 0, 0) tags
// the frames) is driven through OdometryPipeline::estimatePose with a pose plugin that does nothing and an optimizer hook that
// records what BundleAdjustmentBase::apply hands to the solver. out_trig[j] = 1 if estimatePose(frames[j], frames[j+1]) ran BA;
// out_first[j] / out_count[j] = first window frame / number of cameras of that call.
namespace {
struct NullPnP : vo::BasePnPSolver { void solvePnP(vo::Frame&, vo::Frame&, vo::Mat3&, vo::Vec3&) override {} };
struct RecordingBA : vo::BundleAdjustmentBase {
    int calls = 0, last_nc = 0, last_first = -1;
    void ba_solve(double* cams, int nc, double*, int, const double*, const int*, const int*, int, const double*, double, int) override {
        calls++; last_nc = nc;
        last_first = (int)std::lround(-cams[3]);   // cams = [aa(R^T), -t], t[i] = (i, 0, 0)
    }
};
}
void orc_host_ba_schedule(int bundle_size, int n_frames, int* out_trig, int* out_first, int* out_count) {
    vo::OdometryPipeline pl;
    pl.cfg.bundle_size = bundle_size;
    pl.cfg.tracked_features_tol = 0;   // count3DPoints() >= 0: always the PnP branch
    NullPnP pnp; RecordingBA ba; ba.tracker = &pl;
    pl.pnpsolver = &pnp; pl.ba = &ba;
    const int lm = pl.landmarks.create(vo::Feature3D(1.0, 2.0, -9.0));
    for (int i = 0; i < n_frames; i++) {
        auto fr = std::make_shared<vo::Frame>();
        fr->frame = i;
        fr->add_feature(10 + i, 20, lm);
        pl.frames.push_back(fr);
    }
    pl.scale = 1.0;
    pl.R.push_back(vo::Mat3::eye()); pl.t.push_back(vo::Vec3{{0, 0, 0}});
    pl.R_s.push_back(vo::Mat3::eye()); pl.t_s.push_back(vo::Vec3{{0, 0, 0}});
    for (int j = 0; j + 1 < n_frames; j++) {
        const int before = ba.calls;
        pl.estimatePose(*pl.frames[j], *pl.frames[j + 1]);
        // motionHeuristics appended pose j+1; retag it so the next window can be read off the camera block
        pl.R[j + 1] = vo::Mat3::eye(); pl.t[j + 1] = vo::Vec3{{(double)(j + 1), 0, 0}};
        out_trig[j] = ba.calls > before;
        out_first[j] = out_trig[j] ? ba.last_first : -1;
        out_count[j] = out_trig[j] ? ba.last_nc : 0;
    }
}
// OdometryPipeline::motionHeuristics (:171-208) on a pipeline whose pose history is given: R/t/R_s/t_s hold n entries (9 / 3
// doubles each); (_R, _t) is the relative motion handed in; j the source frame. Outputs: the absolute pose appended to R/t, the
// relative pose appended to R_s/t_s, and whether the fallback branch was taken.
int orc_host_motion_heuristics(int n, const double* R, const double* t, const double* Rs, const double* ts, double scale, int j,
                               const double* R_in, const double* t_in, double* R_abs, double* t_abs, double* R_rel, double* t_rel) {
    vo::OdometryPipeline pl;
    pl.scale = scale;
    for (int i = 0; i < n; i++) {
        vo::Mat3 a, b; memcpy(a.m, R + 9 * i, 72); memcpy(b.m, Rs + 9 * i, 72);
        pl.R.push_back(a); pl.R_s.push_back(b);
        pl.t.push_back(vo::Vec3{{t[3 * i], t[3 * i + 1], t[3 * i + 2]}});
        pl.t_s.push_back(vo::Vec3{{ts[3 * i], ts[3 * i + 1], ts[3 * i + 2]}});
    }
    vo::Mat3 _R; memcpy(_R.m, R_in, 72);
    vo::Vec3 _t{{t_in[0], t_in[1], t_in[2]}};
    pl.motionHeuristics(_R, _t, j);
    memcpy(R_abs, pl.R.back().m, 72); memcpy(t_abs, pl.t.back().v, 24);
    memcpy(R_rel, pl.R_s.back().m, 72); memcpy(t_rel, pl.t_s.back().v, 24);
    return (int)pl.stats.heuristic_motion;
}
// NeighborGrid (the occupancy-grid form of the re-detection's hasNeighbor loop) against Frame::hasNeighbor itself: the map holds the
// n_map given features; candidate i is tested by both, and - as the pipeline does - added (with its `add` coordinates) when it has no
// neighbour. out[i] = grid answer | (scan answer << 1).
void orc_host_neighbor_grid(const int* map_xy, int n_map, const int* cand_xy, const int* add_xy, int n_cand, int* out) {
    vo::Frame fr;
    for (int i = 0; i < n_map; i++) fr.add_feature(map_xy[2 * i], map_xy[2 * i + 1], -1);
    vo::NeighborGrid grid(fr);
    for (int i = 0; i < n_cand; i++) {
        vo::Feature f(cand_xy[2 * i], cand_xy[2 * i + 1]);
        const bool a = grid.hasNeighbor(f.column, f.row), b = fr.hasNeighbor(f);
        out[i] = (a ? 1 : 0) | (b ? 2 : 0);
        if (!b) {
            fr.add_feature(add_xy[2 * i], add_xy[2 * i + 1], -1);
            grid.add(add_xy[2 * i], add_xy[2 * i + 1]);
        }
    }
}
// ---- two-view geometry probes (vo_fivepoint.cpp; the host triangulator is shared by product and oracle pipelines, so its
// known-answer tests go straight at the functions, not through a pipeline) ----
int orc_host_five_point(const double* q1, const double* q2, double* E_out) { return vo::five_point_essentials(q1, q2, E_out); }
void orc_host_five_point_samples(int n, int count, int* out5) { vo::five_point_sample_stream(n, count, out5); }
int orc_host_update_num_iters(double p, double ep, int mp, int mx) { return vo::five_point_update_num_iters(p, ep, mp, mx); }
int orc_host_find_essential(const double* p1, const double* p2, int n, const double* K, double prob, double threshold, double* E,
                            unsigned char* mask_out, int* samples_drawn, int workers) {
    std::vector<uint8_t> mask;
    int drawn = 0;
    std::shared_ptr<vo::SpinPool> pool;
    if (workers > 1) pool = vo::make_spin_pool(workers - 1);
    const bool ok = vo::find_essential_mat(p1, p2, n, K, prob, threshold, E, mask, &drawn, pool.get(), workers);
    memcpy(mask_out, mask.data(), (size_t)n);
    if (samples_drawn) *samples_drawn = drawn;
    return ok ? 1 : 0;
}
int orc_host_recover_pose(const double* E, const double* p1, const double* p2, int n, const double* K, double* R, double* t,
                          unsigned char* mask_io, double* tri4) {
    vo::FivePointTri tri;   // host DLT hook
    std::vector<uint8_t> mask(mask_io, mask_io + n);
    std::vector<double> q;
    const int good = vo::recover_pose(&tri, E, p1, p2, n, K, R, t, mask, q);
    memcpy(mask_io, mask.data(), (size_t)n);
    memcpy(tri4, q.data(), (size_t)4 * n * sizeof(double));
    return good;
}
// the per-point part of cv::recoverPose (OpenCVFivePointTri.cpp:27): the checker for pmv_triangulate_candidates
void orc_triangulate_candidates(const double* q1, const double* q2, int n, const double* P1x4, const unsigned char* mask_in, double* out_Q,
                                unsigned char* out_mask, int* out_good) {
    vo::dlt_candidates_host(q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good);
}
void orc_host_project_point(const double* R, const double* t, const double* camera, const double* p3, double* p2) {
    vo::Feature3D::projectPoint(R, t, camera, p3, p2);
}
// Feature3D::transform / transformInv on a float32 point (quirk Q7)
void orc_host_f3d_roundtrip(const double* R9, const double* t3, float* xyz, int inverse_first) {
    vo::Mat3 R; memcpy(R.m, R9, 72);
    vo::Vec3 t{{t3[0], t3[1], t3[2]}};
    vo::Feature3D f(xyz[0], xyz[1], xyz[2]);
    if (inverse_first) { f.transformInv(R, t); f.transform(R, t); } else { f.transform(R, t); f.transformInv(R, t); }
    xyz[0] = f.x; xyz[1] = f.y; xyz[2] = f.z;
}
}
