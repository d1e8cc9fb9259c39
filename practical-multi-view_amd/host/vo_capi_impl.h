// Shared body of the "run a whole sequence" C entry points. The product (hip_pipeline.hip -> pmv_pipeline_*) and the
// oracle (orc_plugins.cpp -> orc_pipeline_*) instantiate it with their own plugin set; the orchestration is the same
// host code (vo_pipeline.cpp), which is exactly the drop-in contract: only the plugin kernels differ.
#pragma once
#include "vo_pipeline.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace vo {

struct PipelineParams {          // plain C struct passed through ctypes
    int n_frames, w, h;
    int min_tracked_features, tracked_features_tol, init_frames, bundle_size, ba_iterations;
    int extractor;               // 0 GFTT, 1 ShiTomasi, 2 FAST
    int threaded;                // 0 sequential schedule, 1 front-end/back-end threads
    int n_threads;               // CPU plugins only: worker threads for LK (the reference: OpenCV parallel_for_)
    int reserved;
    int matcher;                 // 0 pyramidal LK, 1 kNN over `extractor`
};

struct PipelineRun {
    OdometryPipeline pipe;
    std::vector<BaseFeatureExtractor*> owned_ex;
    BaseFeatureMatcher* m = nullptr;
    BasePnPSolver* p = nullptr;
    BaseTriangulator* tr = nullptr;
    BaseOptimizer* b = nullptr;
    double seconds = 0;
    ~PipelineRun() {
        for (auto* e : owned_ex) delete e;
        delete m; delete p; delete tr; delete b;
    }
};

// frames: n*w*h gray bytes (host) — slot i of the device context (if any) must already hold frame i.
inline void pipeline_setup(PipelineRun& run, const PipelineParams& P, const uint8_t* frames, const double* K9,
                           const double* gt_poses12) {
    OdometryPipeline& pl = run.pipe;
    pl.cfg.min_tracked_features = P.min_tracked_features;
    pl.cfg.tracked_features_tol = P.tracked_features_tol;
    pl.cfg.init_frames = P.init_frames;
    pl.cfg.bundle_size = P.bundle_size;
    pl.cfg.ba_iterations = P.ba_iterations;
    pl.cfg.extractor = P.extractor;
    pl.cfg.stop = P.n_frames;
    if (const char* e = getenv("PMV_PIPE_DEPTH")) pl.cfg.pipe_depth = atoi(e) > 0 ? atoi(e) : pl.cfg.pipe_depth;   // (a knob that changes no result)
    memcpy(pl.camera, K9, sizeof(double) * 9);
    pl.images.resize(P.n_frames);
    pl.gt_t.resize(P.n_frames);
    for (int i = 0; i < P.n_frames; i++) {
        ImageView v;
        v.host = frames ? frames + (size_t)i * P.w * P.h : nullptr;
        v.slot = i;
        v.full_w = P.w; v.full_h = P.h; v.x0 = 0; v.y0 = 0; v.w = P.w; v.h = P.h;
        pl.images[i] = v;
        pl.gt_t[i] = Vec3{{gt_poses12[i * 12 + 3], gt_poses12[i * 12 + 7], gt_poses12[i * 12 + 11]}};   // parsePoses :525-594
    }
    pl.frames.reserve((size_t)P.n_frames + 8);
}

inline void pipeline_execute(PipelineRun& run, const PipelineParams& P) {
    auto t0 = std::chrono::steady_clock::now();
    if (P.threaded) run.pipe.run_threaded();
    else run.pipe.run();
    run.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("PMV_HOST_PROF")) {
        fprintf(stderr, "[host-prof] run %.4f s:", run.seconds);
        for (int i = 0; i < HostProf::N; i++) fprintf(stderr, " %s=%.4f", HostProf::name(i), run.pipe.stats.hp.t[i]);
        fprintf(stderr, "\n");
    }
}

// poses: for i in [0, n_poses): 12 doubles = R (row-major 9) then t (3)
inline int pipeline_num_poses(const PipelineRun& run) { return (int)run.pipe.R.size(); }
inline void pipeline_get_poses(const PipelineRun& run, double* out) {
    for (size_t i = 0; i < run.pipe.R.size(); i++) {
        memcpy(out + i * 12, run.pipe.R[i].m, 9 * sizeof(double));
        memcpy(out + i * 12 + 9, run.pipe.t[i].v, 3 * sizeof(double));
    }
}
inline int pipeline_num_frames(const PipelineRun& run) { return (int)run.pipe.frames.size(); }
inline int pipeline_frame_feature_count(const PipelineRun& run, int k) { return run.pipe.frame_feature_count(k); }
inline int pipeline_frame_corr_count(const PipelineRun& run, int k) { return run.pipe.frame_corr_count(k); }   // incl. quirk Q10's empty entries
// (column, row, landmark id or -1) per map entry, in the container's iteration order
inline void pipeline_get_frame_features(const PipelineRun& run, int k, int* out) { run.pipe.frame_features(k, out); }
// PIPELINE_STATS_COUNT doubles: counters, run seconds, and per-stage wall seconds of the calling host threads (the field list is
// documented at pmv_pipeline_get_stats in include/pmv_hip.h)
constexpr int PIPELINE_STATS_COUNT = 25;
inline void pipeline_get_stats(const PipelineRun& run, double* out25) {
    const Stats& s = run.pipe.stats;
    const double v[PIPELINE_STATS_COUNT] = {(double)s.lk_calls, (double)s.lk_points, (double)s.detect_calls, (double)s.pnp_calls, (double)s.pnp_points,
                          (double)s.tri_calls, (double)s.ba_calls, (double)s.ba_obs, (double)s.ba_points, (double)s.heuristic_motion,
                          run.seconds, (double)run.pipe.init_offset, (double)run.pipe.landmarks.n_alive, run.pipe.scale,
                          s.t_lk, s.t_detect, s.t_pnp, s.t_tri, s.t_ba, s.t_pnp_kernel, s.t_ba_kernel, s.t_tri_essential, s.t_tri_pose, s.tri_hypotheses, (double)s.tri_ahead};
    memcpy(out25, v, sizeof(v));
}

}  // namespace vo
