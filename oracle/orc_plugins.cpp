// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
// CPU plugin set: the oracle's restatements plugged behind the SAME Base* adapters / pipeline as the product
// (practical-multi-view_amd/host/vo_pipeline.*), giving the reference-equivalent CPU pipeline that
//   * tests compare the HIP pipeline against (tracks bit-exact, poses within tolerance), and
//   * bench.py times as `cpu_baseline` (kind "port") on the GPU box's host cores, with the reference's threading model
//     (front-end + back-end threads, LK parallel over tracks).
#include "orc_api.h"
#include "vo_capi_impl.h"
#include <cstring>

namespace {
using namespace vo;

struct CpuGftt : GoodFeatureExtractorBase {
    void gftt(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out) override {
        out.clear();
        std::vector<int> xy((size_t)(max > 0 ? max : 65536) * 2);   // max <= 0: no limit (at most one corner per pixel of a 255x255 cell)
        for (auto& c : cells) {
            const int n = orc::gftt_cell(c.host, c.full_w, c.full_h, c.x0, c.y0, c.w, c.h, max, quality, min_distance, xy.data(), nullptr);
            std::vector<std::pair<int, int>> v;
            for (int i = 0; i < n; i++) v.push_back({xy[2 * i], xy[2 * i + 1]});
            out.push_back(v);
        }
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
struct CpuLK : LucasKanadeFMBase {
    int nthreads = 1;
    void pyrlk(const ImageView& prev, const ImageView& next, const float* prev_xy, int n, float* next_xy, uint8_t* status,
               float* err) override {
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
    void ba_solve(double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
                  const double* K, double huber, int max_iterations) override {
        orc::BASummary s;
        orc::ba_solve(cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, &s);
    }
};
}  // namespace

extern "C" {
void* orc_pipeline_run(const vo::PipelineParams* P, const uint8_t* frames, const double* K9, const double* gt_poses12) {
    auto* run = new vo::PipelineRun();
    vo::pipeline_setup(*run, *P, frames, K9, gt_poses12);
    vo::BaseFeatureExtractor* ex;
    if (P->extractor == 1) ex = new CpuShiTomasi(); else ex = new CpuGftt();
    run->owned_ex.push_back(ex);
    auto* lk = new CpuLK(); lk->nthreads = P->n_threads;
    auto* pnp = new CpuPnP(); pnp->tracker = &run->pipe;
    auto* tri = new vo::FivePointTri(); tri->tracker = &run->pipe; tri->workers = std::max(1, std::min(P->n_threads, 8));
    auto* ba = new CpuBA(); ba->tracker = &run->pipe;
    run->m = lk; run->p = pnp; run->tr = tri; run->b = ba;
    run->pipe.extractor = ex; run->pipe.matcher = lk; run->pipe.pnpsolver = pnp; run->pipe.triangulator = tri; run->pipe.ba = ba;
    vo::pipeline_execute(*run, *P);
    return run;
}
void orc_pipeline_free(void* h) { delete (vo::PipelineRun*)h; }
int orc_pipeline_num_poses(void* h) { return vo::pipeline_num_poses(*(vo::PipelineRun*)h); }
void orc_pipeline_get_poses(void* h, double* out) { vo::pipeline_get_poses(*(vo::PipelineRun*)h, out); }
int orc_pipeline_num_frames(void* h) { return vo::pipeline_num_frames(*(vo::PipelineRun*)h); }
int orc_pipeline_frame_feature_count(void* h, int k) { return vo::pipeline_frame_feature_count(*(vo::PipelineRun*)h, k); }
void orc_pipeline_get_frame_features(void* h, int k, int* out) { vo::pipeline_get_frame_features(*(vo::PipelineRun*)h, k, out); }
int orc_pipeline_stats_count(void) { return vo::PIPELINE_STATS_COUNT; }
void orc_pipeline_get_stats(void* h, double* out24) { vo::pipeline_get_stats(*(vo::PipelineRun*)h, out24); }
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
    std::vector<std::shared_ptr<vo::Feature>> keep;
    std::unordered_map<vo::Feature*, int> idx;
    for (int i = 0; i < n; i++) {
        auto f = std::make_shared<vo::Feature>(cols[i], rows[i]);
        keep.push_back(f); idx[f.get()] = i;
        fr.map[f] = std::weak_ptr<vo::Feature3D>();
    }
    int k = 0;
    for (auto& p : fr.map) out[k++] = idx[p.first.get()];
}
// feat_corr semantics: keys compare by coordinates (same-pixel sources collapse, last value wins). Returns the number of
// entries; out_key / out_val = indices of the surviving key feature and of its value, in iteration order
int orc_host_corr_order(const int* cols, const int* rows, int n, int* out_key, int* out_val) {
    vo::fmap corr;
    std::vector<std::shared_ptr<vo::Feature>> src, dst;
    std::unordered_map<vo::Feature*, int> si, di;
    for (int i = 0; i < n; i++) {
        auto a = std::make_shared<vo::Feature>(cols[i], rows[i]);
        auto b = std::make_shared<vo::Feature>(cols[i] + 1000, rows[i] + 1000);
        src.push_back(a); dst.push_back(b); si[a.get()] = i; di[b.get()] = i;
        corr[a] = b;
    }
    int k = 0;
    for (auto& p : corr) { out_key[k] = si[p.first.lock().get()]; out_val[k] = di[p.second.lock().get()]; k++; }
    return k;
}
// OdometryPipeline.cpp:407 trigger and CeresBundleAdjustment.cpp:7-8,20-23 window for estimatePose(src_frame, src_frame+1)
int orc_host_ba_schedule(int bundle_size, int src_frame, int* win_first, int* win_count) {
    const int trig = bundle_size && src_frame && src_frame % (bundle_size / 3 * 2) == 0;
    const int fn = src_frame + 1 + 1, n = std::min(bundle_size, fn);
    int cnt = 0, first = -1;
    for (int i = fn - n; i < fn; i++) { if (i == 0) continue; if (first < 0) first = i; cnt++; }
    *win_first = first; *win_count = cnt;
    return trig;
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
