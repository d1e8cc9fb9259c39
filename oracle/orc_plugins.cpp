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
        std::vector<int> xy((size_t)std::max(max, 1) * 2);
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
    auto* tri = new vo::FivePointTri(); tri->tracker = &run->pipe;
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
void orc_pipeline_get_stats(void* h, double* out16) { vo::pipeline_get_stats(*(vo::PipelineRun*)h, out16); }
}
