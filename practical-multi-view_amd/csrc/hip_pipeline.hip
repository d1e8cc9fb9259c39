// HIP plugin set: the reference's five plugin roles served through the C ABI of this library, plugged into the host
// pipeline mirror (host/vo_pipeline.*). This is the drop-in: OdometryPipeline/addFrame/estimatePose stay host C++,
// every numerically heavy call crosses into the gfx950 kernels. No CPU fallback exists on this path (the triangulator,
// which the north star does not move to the device, is the host implementation in host/vo_fivepoint.cpp).
#include "pmv_ctx.h"
#include "batch_engine.h"
#include <mutex>
#include <thread>
#include "vo_capi_impl.h"
#include <cstring>
#include <stdexcept>
#include <string>
#include <malloc.h>

namespace {
using namespace vo;

struct HipError : std::runtime_error { int code; HipError(int c, const char* m) : std::runtime_error(m), code(c) {} };
inline void ck(pmv_ctx* ctx, int rc) { (void)ctx; if (rc != PMV_OK) throw HipError(rc, pmv::thread_error()); }   // this thread's own message

static void cells_of(const std::vector<ImageView>& cells, std::vector<int>& out) {
    out.clear();
    for (auto& c : cells) { out.push_back(c.x0); out.push_back(c.y0); out.push_back(c.w); out.push_back(c.h); }
}

struct HipGftt : GoodFeatureExtractorBase {
    pmv_ctx* ctx;
    std::vector<int> rect, xy, cnt;
    void gftt(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out) override {
        out.assign(cells.size(), {});
        if (cells.empty()) return;
        cells_of(cells, rect);
        const size_t cap = max > 0 ? (size_t)max : (size_t)PMV_GFTT_UNLIMITED_CAP;   // max <= 0: no limit (cv::goodFeaturesToTrack)
        xy.resize(cells.size() * cap * 2);
        cnt.resize(cells.size());
        ck(ctx, pmv_detect_gftt(ctx, cells[0].slot, rect.data(), (int)cells.size(), max, quality, min_distance, xy.data(), cnt.data()));
        for (size_t c = 0; c < cells.size(); c++)
            for (int i = 0; i < cnt[c]; i++) out[c].push_back({xy[(c * cap + i) * 2], xy[(c * cap + i) * 2 + 1]});
    }
};
struct HipShiTomasi : ShiTomasiExtractorBase {
    pmv_ctx* ctx;
    std::vector<int> rect, xy, cnt;
    std::vector<double> sc;
    void shitomasi(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
                   std::vector<std::vector<double>>& score) override {
        out.assign(cells.size(), {});
        score.assign(cells.size(), {});
        if (cells.empty() || max < 1) return;
        cells_of(cells, rect);
        xy.resize(cells.size() * (size_t)max * 2);
        sc.resize(cells.size() * (size_t)max);
        cnt.resize(cells.size());
        ck(ctx, pmv_detect_shitomasi(ctx, cells[0].slot, rect.data(), (int)cells.size(), max, quality, xy.data(), sc.data(), cnt.data()));
        for (size_t c = 0; c < cells.size(); c++)
            for (int i = 0; i < cnt[c]; i++) {
                out[c].push_back({xy[(c * max + i) * 2], xy[(c * max + i) * 2 + 1]});
                score[c].push_back(sc[c * max + i]);
            }
    }
};
struct HipFast : FastExtractorBase {
    pmv_ctx* ctx;
    std::vector<int> rect, xy, cnt;
    std::vector<float> rs;
    void fast(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
              std::vector<std::vector<float>>& response) override {
        out.assign(cells.size(), {});
        response.assign(cells.size(), {});
        if (cells.empty() || max < 1) return;
        cells_of(cells, rect);
        xy.resize(cells.size() * (size_t)max * 2); rs.resize(cells.size() * (size_t)max); cnt.resize(cells.size());
        ck(ctx, pmv_detect_fast(ctx, cells[0].slot, rect.data(), (int)cells.size(), max, threshold, nonmax ? 1 : 0, xy.data(), rs.data(), cnt.data()));
        for (size_t c = 0; c < cells.size(); c++)
            for (int i = 0; i < cnt[c]; i++) {
                out[c].push_back({xy[(c * max + i) * 2], xy[(c * max + i) * 2 + 1]});
                response[c].push_back(rs[c * max + i]);
            }
    }
};
struct HipKnn : KnnFeatureMatcherBase {
    pmv_ctx* ctx;
    void knn(const ImageView& src, const ImageView& next, const int* src_xy, int n, const int* cmp_xy, int m, int* best, float* err) override {
        ck(ctx, pmv_knn_match(ctx, src.slot, next.slot, src_xy, n, cmp_xy, m, neighbours, window, best, err));
    }
};
struct HipLK : LucasKanadeFMBase {
    pmv_ctx* ctx;
    void pyrlk(const ImageView& prev, const ImageView& next, const float* prev_xy, int n, float* next_xy, uint8_t* status,
               float* err) override {
        ck(ctx, pmv_lk_track(ctx, prev.slot, next.slot, prev_xy, n, next_xy, status, err));
    }
};
struct HipPnP : EPnPSolverBase {
    pmv_ctx* ctx;
    bool pnp_ransac(const float* obj, const float* img, int m, const double* K, double* rvec, double* tvec,
                    std::vector<int>& inliers) override {
        inliers.assign(m > 0 ? m : 1, 0);
        int n = 0;
        const int rc = pmv_pnp_ransac(ctx, obj, img, m, K, rvec, tvec, 100, 8.f, .99, inliers.data(), &n);
        if (rc == PMV_ERR_DEGENERATE) { inliers.clear(); return false; }   // cv::Exception in the reference
        ck(ctx, rc);
        inliers.resize(n);
        return n > 0;
    }
};
struct HipTri : vo::FivePointTri {   // five-point RANSAC hypotheses on host threads (default) or on the GPU, recoverPose's DLT + cheirality on the GPU
    pmv_ctx* ctx;
    bool essential_hypotheses(const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models, int* n_models,
                              int* counts) override {
        ck(ctx, pmv_fivepoint_hypotheses(ctx, q1, q2, n, samples, n_hyp, thr, models, n_models, counts));
        return true;
    }
    void dlt_candidates(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                        uint8_t* out_mask, int* out_good) override {
        ck(ctx, pmv_triangulate_candidates(ctx, q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good));
    }
    void dlt_candidates_ahead(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                              uint8_t* out_mask, int* out_good) override {
        ck(ctx, pmv_triangulate_candidates_ahead(ctx, q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good));
    }
};
struct HipBA : BundleAdjustmentBase {
    pmv_ctx* ctx;
    void ba_solve(double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
                  const double* K, double huber, int max_iterations) override {
        ck(ctx, pmv_ba_solve(ctx, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, nullptr));
    }
};
// ---- the same five roles served through the batch engine (several sequences per launch, batch_engine.hip) --------------------
struct BatchGftt : GoodFeatureExtractorBase {
    pmv_ctx* ctx; pmv::BatchEngine* eng;
    std::vector<int> rect, xy, cnt;
    void gftt(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out) override {
        out.assign(cells.size(), {});
        if (cells.empty()) return;
        cells_of(cells, rect);
        const size_t cap = max > 0 ? (size_t)max : (size_t)PMV_GFTT_UNLIMITED_CAP;
        xy.resize(cells.size() * cap * 2);
        cnt.resize(cells.size());
        ck(ctx, pmv::engine_detect(eng, 1, cells[0].slot, rect.data(), (int)cells.size(), max, quality, min_distance, xy.data(), nullptr, cnt.data()));
        for (size_t c = 0; c < cells.size(); c++)
            for (int i = 0; i < cnt[c]; i++) out[c].push_back({xy[(c * cap + i) * 2], xy[(c * cap + i) * 2 + 1]});
    }
};
struct BatchShiTomasi : ShiTomasiExtractorBase {
    pmv_ctx* ctx; pmv::BatchEngine* eng;
    std::vector<int> rect, xy, cnt;
    std::vector<double> sc;
    void shitomasi(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
                   std::vector<std::vector<double>>& score) override {
        out.assign(cells.size(), {});
        score.assign(cells.size(), {});
        if (cells.empty() || max < 1) return;
        cells_of(cells, rect);
        xy.resize(cells.size() * (size_t)max * 2);
        sc.resize(cells.size() * (size_t)max);
        cnt.resize(cells.size());
        ck(ctx, pmv::engine_detect(eng, 2, cells[0].slot, rect.data(), (int)cells.size(), max, quality, 0.0, xy.data(), sc.data(), cnt.data()));
        for (size_t c = 0; c < cells.size(); c++)
            for (int i = 0; i < cnt[c]; i++) {
                out[c].push_back({xy[(c * max + i) * 2], xy[(c * max + i) * 2 + 1]});
                score[c].push_back(sc[c * max + i]);
            }
    }
};
struct BatchLK : LucasKanadeFMBase {
    pmv_ctx* ctx; pmv::BatchEngine* eng;
    // Ordering hint for the batched launch (no effect on any result): a track that needed many LK iterations in the last frame pair tends
    // to need many again (weak texture, an edge along the motion), and a launch ends with its slowest track - so the engine starts the
    // expensive ones first. The cost of a track is remembered under the integer pixel it was tracked TO, which is where the next call
    // starts it from (the adapter truncates exactly like this: OpenCVLucasKanadeFM.cpp:25); re-detected features are new: default cost.
    static constexpr int TBL = 2048, DEFAULT_COST = 24;   // open addressing, > 2 x the tracks of a frame (1024 max)
    std::vector<uint32_t> tkey = std::vector<uint32_t>(TBL, 0xffffffffu);
    std::vector<uint8_t> tval = std::vector<uint8_t>(TBL, 0);
    std::vector<uint8_t> pred, iters;
    static uint32_t key_of(float x, float y) { return ((uint32_t)(int)y << 16) ^ (uint32_t)(int)x; }
    static uint32_t slot_of(uint32_t k) { return (k * 2654435761u) >> 21; }   // 11 bits
    void pyrlk(const ImageView& prev, const ImageView& next, const float* prev_xy, int n, float* next_xy, uint8_t* status,
               float* err) override {
        pred.resize((size_t)n); iters.resize((size_t)n);
        const bool use = n <= TBL / 2;
        for (int i = 0; i < n; i++) {
            uint8_t c = DEFAULT_COST;
            if (use) {
                const uint32_t k = key_of(prev_xy[2 * i], prev_xy[2 * i + 1]);
                for (uint32_t h = slot_of(k);; h = (h + 1) & (TBL - 1)) {
                    if (tkey[h] == k) { c = tval[h]; break; }
                    if (tkey[h] == 0xffffffffu) break;
                }
            }
            pred[(size_t)i] = c;
        }
        ck(ctx, pmv::engine_lk(eng, prev.slot, next.slot, prev_xy, n, next_xy, status, err, pred.data(), iters.data()));
        std::fill(tkey.begin(), tkey.end(), 0xffffffffu);
        if (!use) return;
        for (int i = 0; i < n; i++) {
            if (!status[i]) continue;
            const uint32_t k = key_of(next_xy[2 * i], next_xy[2 * i + 1]);
            uint32_t h = slot_of(k);
            while (tkey[h] != 0xffffffffu && tkey[h] != k) h = (h + 1) & (TBL - 1);
            if (tkey[h] == k) { if (iters[(size_t)i] > tval[h]) tval[h] = iters[(size_t)i]; }   // two tracks on one pixel: the dearer one
            else { tkey[h] = k; tval[h] = iters[(size_t)i]; }
        }
    }
};
struct BatchPnP : EPnPSolverBase {
    pmv_ctx* ctx; pmv::BatchEngine* eng; int seq;
    bool pnp_ransac(const float* obj, const float* img, int m, const double* K, double* rvec, double* tvec,
                    std::vector<int>& inliers) override {
        inliers.assign(m > 0 ? m : 1, 0);
        int n = 0;
        const int rc = pmv::engine_pnp(eng, seq, obj, img, m, K, rvec, tvec, 100, 8.f, .99, inliers.data(), &n);
        if (rc == PMV_ERR_DEGENERATE) { inliers.clear(); return false; }
        ck(ctx, rc);
        inliers.resize(n);
        return n > 0;
    }
};
struct BatchTri : vo::FivePointTri {
    pmv_ctx* ctx; pmv::BatchEngine* eng; int seq;
    bool essential_hypotheses(const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models, int* n_models,
                              int* counts) override {
        ck(ctx, pmv::engine_fivepoint(eng, seq, q1, q2, n, samples, n_hyp, thr, models, n_models, counts));
        return true;
    }
    void dlt_candidates(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                        uint8_t* out_mask, int* out_good) override {
        ck(ctx, pmv::engine_dlt(eng, seq, q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good));
    }
};
struct BatchBA : BundleAdjustmentBase {
    pmv_ctx* ctx; pmv::BatchEngine* eng; int seq;
    void ba_solve(double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
                  const double* K, double huber, int max_iterations) override {
        ck(ctx, pmv::engine_ba(eng, seq, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations));
    }
};
}  // namespace

struct pmv_pipeline_result { vo::PipelineRun run; };

// The host threads of a run allocate and free per-frame tables all the time, each in its own glibc arena; with the default tunables an
// arena gives its top back as soon as 128 KB are free and grows again a frame later - mprotect / madvise calls that take the process's
// address-space lock. Sampled on a B = 192 run (scripts/hostprof): 30 % of the host CPU time of the timed region in mprotect,
// __default_morecore, madvise and malloc itself. Keep freed memory in the arenas instead (once per process; PMV_KEEP_MALLOC_DEFAULTS=1
// leaves the process's settings alone).
static void host_allocator_setup() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("PMV_KEEP_MALLOC_DEFAULTS")) return;
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 16 << 20);
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
    });
}

extern "C" {

// Same run from HOST frames (n_frames * w * h gray bytes, pageable or pinned): the frames are streamed into slots 0..n_frames-1 by
// the ingest thread (ingest.hip) while the pipeline is already tracking the first ones. Results are identical to
// pmv_frames_stage + pmv_pipeline_run(build_pyramids = 1).
int pmv_pipeline_run_streamed(pmv_ctx* ctx, const pmv_pipeline_params* P, const double* K9, const double* gt_poses12,
                              const uint8_t* host_frames, pmv_pipeline_result** out) {
    if (!ctx || !P || !host_frames) { pmv::set_err(ctx, "pmv_pipeline_run_streamed: null argument"); return PMV_ERR_INVALID; }
    if (P->n_frames > ctx->n_slots) { pmv::set_err(ctx, "pmv_pipeline_run_streamed: n_frames=%d exceeds the %d frame slots", P->n_frames, ctx->n_slots); return PMV_ERR_CAPACITY; }
    int rc = pmv_frames_stream_begin(ctx, 0, P->n_frames, host_frames, P->w, P->h);
    if (rc != PMV_OK) return rc;
    pmv_pipeline_params Q = *P;
    Q.build_pyramids = 0;   // the ingest stream builds them chunk by chunk
    rc = pmv_pipeline_run(ctx, &Q, K9, gt_poses12, out);
    const int rc2 = pmv_frames_stream_end(ctx);   // joins the ingest thread (also after a failed run: the source buffer is the caller's)
    if (rc == PMV_OK && rc2 != PMV_OK) { pmv_pipeline_free(*out); *out = nullptr; return rc2; }
    return rc;
}

int pmv_pipeline_run(pmv_ctx* ctx, const pmv_pipeline_params* P, const double* K9, const double* gt_poses12,
                     pmv_pipeline_result** out) {
    if (!ctx || !P || !K9 || !gt_poses12 || !out) { pmv::set_err(ctx, "pmv_pipeline_run: null argument"); return PMV_ERR_INVALID; }
    host_allocator_setup();
    if (P->n_frames < P->init_frames + 2 || P->n_frames > ctx->n_slots || P->init_frames < 1) {
        pmv::set_err(ctx, "pmv_pipeline_run: n_frames=%d (slots %d, init_frames %d)", P->n_frames, ctx->n_slots, P->init_frames);
        return PMV_ERR_CAPACITY;
    }
    if (P->bundle_size != 0 && P->bundle_size < 3) { pmv::set_err(ctx, "pmv_pipeline_run: bundle_size 1..2 divides by zero in the reference (OdometryPipeline.cpp:407)"); return PMV_ERR_INVALID; }
    if (P->bundle_size > ctx->max_ba_cams) { pmv::set_err(ctx, "pmv_pipeline_run: bundle_size exceeds max_ba_cams"); return PMV_ERR_CAPACITY; }
    auto* res = new pmv_pipeline_result();
    vo::PipelineRun& run = res->run;
    vo::PipelineParams vp;
    vp.n_frames = P->n_frames; vp.w = P->w; vp.h = P->h;
    vp.min_tracked_features = P->min_tracked_features; vp.tracked_features_tol = P->tracked_features_tol;
    vp.init_frames = P->init_frames; vp.bundle_size = P->bundle_size; vp.ba_iterations = P->ba_iterations;
    vp.extractor = P->extractor; vp.threaded = P->threaded; vp.n_threads = P->n_threads; vp.reserved = 0; vp.matcher = P->matcher;
    try {
        if (P->build_pyramids) ck(ctx, pmv_frames_build(ctx, 0, P->n_frames));
        vo::pipeline_setup(run, vp, nullptr, K9, gt_poses12);
        vo::BaseFeatureExtractor* ex;
        if (P->extractor == 1) { auto* e = new HipShiTomasi(); e->ctx = ctx; ex = e; }
        else if (P->extractor == 2) { auto* e = new HipFast(); e->ctx = ctx; ex = e; }
        else { auto* e = new HipGftt(); e->ctx = ctx; ex = e; }
        run.owned_ex.push_back(ex);
        vo::BaseFeatureMatcher* lk;
        if (P->matcher == 1) {   // kNNFeatureMatcher(extractor): the reference's alternative matcher; it calls the extractor on whole frames
            if (P->extractor != 2) throw std::runtime_error("the kNN matcher needs an extractor that accepts whole frames: extractor = 2 (FAST)");
            auto* k = new HipKnn(); k->ctx = ctx; k->extractor = ex; lk = k;
        } else { auto* l = new HipLK(); l->ctx = ctx; lk = l; }
        auto* pnp = new HipPnP(); pnp->ctx = ctx; pnp->tracker = &run.pipe;
        auto* tri = new HipTri(); tri->ctx = ctx; tri->tracker = &run.pipe; tri->workers = std::max(1, P->n_threads);
        tri->use_hypothesis_hook = P->device_fivepoint != 0;
        tri->prefetch_threads = (P->n_threads > 1 && !tri->use_hypothesis_hook) ? 2 : 0;   // only the two-thread pipeline calls prefetch()
        auto* ba = new HipBA(); ba->ctx = ctx; ba->tracker = &run.pipe;
        run.m = lk; run.p = pnp; run.tr = tri; run.b = ba;
        run.pipe.extractor = ex; run.pipe.matcher = lk; run.pipe.pnpsolver = pnp; run.pipe.triangulator = tri; run.pipe.ba = ba;
        vo::pipeline_execute(run, vp);
    } catch (const HipError& e) {
        pmv::set_err(ctx, "pmv_pipeline_run: %s", e.what());
        const int code = e.code;
        delete res;
        return code;
    } catch (const std::exception& e) {
        pmv::set_err(ctx, "pmv_pipeline_run: %s", e.what());
        delete res;
        return PMV_ERR_INVALID;
    }
    *out = res;
    return PMV_OK;
}
// B independent sequences in one go. Sequence b uses frame slots first_slot[b] .. first_slot[b] + params[b].n_frames - 1
// (staged with pmv_frames_stage; all sequences share the frame size), its own K9 (9 doubles at K9 + 9 b) and ground-truth rows.
// Every sequence runs the reference's two host threads (front-end / back-end) with the unchanged adapters; their plugin calls are
// merged by the context's batch engine into batched launches. out[b] receives sequence b's result, bit-identical to its own
// pmv_pipeline_run. On error every result that exists is freed and the first error is returned.
int pmv_pipeline_run_batch(pmv_ctx* ctx, int B, const pmv_pipeline_params* params, const double* K9, const double* const* gt_poses12,
                           const int* first_slot, pmv_pipeline_result** out) {
    if (!ctx || !params || !K9 || !gt_poses12 || !first_slot || !out || B < 1) { pmv::set_err(ctx, "pmv_pipeline_run_batch: bad argument"); return PMV_ERR_INVALID; }
    host_allocator_setup();
    for (int b = 0; b < B; b++) {
        const pmv_pipeline_params& P = params[b];
        out[b] = nullptr;
        if (P.n_frames < P.init_frames + 2 || P.init_frames < 1 || first_slot[b] < 0 || first_slot[b] + P.n_frames > ctx->n_slots) {
            pmv::set_err(ctx, "pmv_pipeline_run_batch: sequence %d: frames [%d, %d) outside the %d slots / too short", b, first_slot[b], first_slot[b] + P.n_frames, ctx->n_slots);
            return PMV_ERR_CAPACITY;
        }
        if (P.bundle_size != 0 && P.bundle_size < 3) { pmv::set_err(ctx, "pmv_pipeline_run_batch: bundle_size 1..2 divides by zero in the reference"); return PMV_ERR_INVALID; }
        if (P.bundle_size > ctx->max_ba_cams) { pmv::set_err(ctx, "pmv_pipeline_run_batch: bundle_size exceeds max_ba_cams"); return PMV_ERR_CAPACITY; }
        if (P.w != params[0].w || P.h != params[0].h) { pmv::set_err(ctx, "pmv_pipeline_run_batch: all sequences must share the frame size"); return PMV_ERR_INVALID; }
        for (int i = 0; i < P.n_frames; i++) {   // the geometry actually staged in the slots, not only the parameter structs
            const pmv::PyrLayout& Ls = ctx->slot_layout[first_slot[b] + i];
            if (Ls.n_levels == 0 || Ls.w[0] != P.w || Ls.h[0] != P.h) {
                pmv::set_err(ctx, "pmv_pipeline_run_batch: sequence %d: slot %d holds %s (%dx%d), the run is %dx%d", b, first_slot[b] + i,
                             Ls.n_levels == 0 ? "no frame" : "a frame of another size", Ls.w[0], Ls.h[0], P.w, P.h);
                return PMV_ERR_INVALID;
            }
        }
        if (P.matcher != 0 || P.extractor > 1) { pmv::set_err(ctx, "pmv_pipeline_run_batch: the batch engine serves the reference's default plugins (LK; GFTT or ShiTomasi)"); return PMV_ERR_INVALID; }
    }
    pmv::BatchEngine* eng = nullptr;
    int rc = pmv::batch_engine_get(ctx, B, &eng);
    if (rc != PMV_OK) return rc;
    rc = pmv_sync(ctx);
    if (rc != PMV_OK) return rc;
    {   // the pyramids are built in the background, round by round, while the sequences already track
        std::vector<int> nf((size_t)B), bd((size_t)B);
        for (int b = 0; b < B; b++) { nf[(size_t)b] = params[b].n_frames; bd[(size_t)b] = params[b].build_pyramids; }
        rc = pmv::engine_build_begin(eng, B, first_slot, nf.data(), bd.data());
        if (rc != PMV_OK) return rc;
    }
    std::vector<int> codes(B, PMV_OK);
    std::vector<std::string> msgs(B);
    std::vector<std::thread> th;
    for (int b = 0; b < B; b++)
        th.emplace_back([&, b] {
            const pmv_pipeline_params* P = &params[b];
            auto* res = new pmv_pipeline_result();
            vo::PipelineRun& run = res->run;
            vo::PipelineParams vp;
            vp.n_frames = P->n_frames; vp.w = P->w; vp.h = P->h;
            vp.min_tracked_features = P->min_tracked_features; vp.tracked_features_tol = P->tracked_features_tol;
            vp.init_frames = P->init_frames; vp.bundle_size = P->bundle_size; vp.ba_iterations = P->ba_iterations;
            vp.extractor = P->extractor; vp.threaded = P->threaded; vp.n_threads = 1; vp.reserved = 0; vp.matcher = 0;
            try {
                vo::pipeline_setup(run, vp, nullptr, K9 + 9 * b, gt_poses12[b]);
                for (auto& im : run.pipe.images) im.slot += first_slot[b];
                vo::BaseFeatureExtractor* ex;
                if (P->extractor == 1) { auto* e = new BatchShiTomasi(); e->ctx = ctx; e->eng = eng; ex = e; }
                else { auto* e = new BatchGftt(); e->ctx = ctx; e->eng = eng; ex = e; }
                run.owned_ex.push_back(ex);
                auto* lk = new BatchLK(); lk->ctx = ctx; lk->eng = eng;
                auto* pnp = new BatchPnP(); pnp->ctx = ctx; pnp->eng = eng; pnp->seq = b; pnp->tracker = &run.pipe;
                auto* tri = new BatchTri(); tri->ctx = ctx; tri->eng = eng; tri->seq = b; tri->tracker = &run.pipe; tri->workers = 1;
                tri->use_hypothesis_hook = P->device_fivepoint != 0;
                auto* ba = new BatchBA(); ba->ctx = ctx; ba->eng = eng; ba->seq = b; ba->tracker = &run.pipe;
                run.m = lk; run.p = pnp; run.tr = tri; run.b = ba;
                run.pipe.extractor = ex; run.pipe.matcher = lk; run.pipe.pnpsolver = pnp; run.pipe.triangulator = tri; run.pipe.ba = ba;
                vo::pipeline_execute(run, vp);
                out[b] = res;
            } catch (const HipError& e) {
                codes[b] = e.code; msgs[b] = e.what(); delete res;
            } catch (const std::exception& e) {
                codes[b] = PMV_ERR_INVALID; msgs[b] = e.what(); delete res;
            }
        });
    for (auto& t : th) t.join();
    rc = pmv::engine_build_end(eng);
    if (rc != PMV_OK) { pmv::set_err(ctx, "pmv_pipeline_run_batch: the background pyramid build failed"); for (int k = 0; k < B; k++) { delete out[k]; out[k] = nullptr; } return rc; }
    for (int b = 0; b < B; b++)
        if (codes[b] != PMV_OK) {
            pmv::set_err(ctx, "pmv_pipeline_run_batch: sequence %d: %s", b, msgs[b].c_str());
            for (int k = 0; k < B; k++) { delete out[k]; out[k] = nullptr; }
            return codes[b];
        }
    return PMV_OK;
}
// diagnostic: what the five combiners (LK, detectors, PnP, BA, DLT) have served so far
int pmv_batch_stats(pmv_ctx* ctx, long long* counts10, double* times15) {
    if (!ctx || !counts10) return PMV_ERR_INVALID;
    for (int i = 0; i < 10; i++) counts10[i] = 0;
    if (times15) for (int i = 0; i < 15; i++) times15[i] = 0;
    if (ctx->engine) pmv::batch_engine_stats(ctx->engine, counts10, times15);
    return PMV_OK;
}
void pmv_pipeline_free(pmv_pipeline_result* r) { delete r; }
// Tearing down ~10^6 host container nodes of a 1100-frame run takes ~40 ms and is not part of the path: a result can be
// handed to a background thread instead. pmv_pipeline_drain() waits for all outstanding releases.
static std::mutex g_release_mu;
static std::vector<std::thread> g_release_threads;
void pmv_pipeline_release(pmv_pipeline_result* r) {
    if (!r) return;
    std::lock_guard<std::mutex> lk(g_release_mu);
    g_release_threads.emplace_back([r] { delete r; });
}
void pmv_pipeline_drain(void) {
    std::vector<std::thread> th;
    { std::lock_guard<std::mutex> lk(g_release_mu); th.swap(g_release_threads); }
    for (auto& t : th) t.join();
}
int pmv_pipeline_num_poses(const pmv_pipeline_result* r) { return vo::pipeline_num_poses(r->run); }
void pmv_pipeline_get_poses(const pmv_pipeline_result* r, double* out) { vo::pipeline_get_poses(r->run, out); }
int pmv_pipeline_num_frames(const pmv_pipeline_result* r) { return vo::pipeline_num_frames(r->run); }
int pmv_pipeline_frame_feature_count(const pmv_pipeline_result* r, int k) { return vo::pipeline_frame_feature_count(r->run, k); }
void pmv_pipeline_get_frame_features(const pmv_pipeline_result* r, int k, int* out) { vo::pipeline_get_frame_features(r->run, k, out); }
int pmv_pipeline_stats_count(void) { return vo::PIPELINE_STATS_COUNT; }
void pmv_pipeline_get_stats(const pmv_pipeline_result* r, double* out25) { vo::pipeline_get_stats(r->run, out25); }
}
