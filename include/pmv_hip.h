/* pmv_hip.h — C ABI of the MI355X (gfx950) visual-odometry hot path.
 *
 * Drop-in boundary: each entry point replaces the third-party call that one of the
 * reference's plugin implementations makes behind its Base* interface.  Citations are into the
 * reference tree (JeanElsner/practical-multi-view):
 *
 *   pmv_frame_upload / pmv_frames_upload   Frame::Frame + cv::buildOpticalFlowPyramid   (Frame.cpp:31-42,
 *                                          implicit inside OpenCVLucasKanadeFM.cpp:15)
 *   pmv_detect_gftt                        cv::goodFeaturesToTrack per grid cell         (OpenCVGoodFeatureExtractor.cpp:7,
 *                                          called from OdometryPipeline.cpp:357 / :450)  -> BaseFeatureExtractor.h:21
 *   pmv_detect_shitomasi                   ShiTomasiFeatureExtractor::extractFeatures    (ShiTomasiFeatureExtractor.cpp:5-75,
 *                                          Frame.cpp:58-86,119-138)                      -> BaseFeatureExtractor.h:21
 *   pmv_lk_track                           cv::calcOpticalFlowPyrLK                      (OpenCVLucasKanadeFM.cpp:15) -> BaseFeatureMatcher.h:22
 *   pmv_knn_match                          kNNFeatureMatcher::matchFeatures' arithmetic   (kNNFeatureMatcher.cpp:13-31,63-122) -> BaseFeatureMatcher.h:22
 *   pmv_detect_fast                        cv::FAST                                      (OpenCVFASTFeatureExtractor.cpp:8) -> BaseFeatureExtractor.h:21
 *   pmv_pnp_ransac                         cv::solvePnPRansac                            (OpenCVEPnPSolver.cpp:35-36) -> BasePnPSolver.h:19
 *   pmv_fivepoint_hypotheses               cv::findEssentialMat (RANSAC hypotheses)      (OpenCVFivePointTri.cpp:24) -> BaseTriangulator.h
 *   pmv_triangulate_candidates             cv::recoverPose (triangulation + cheirality)  (OpenCVFivePointTri.cpp:27) -> BaseTriangulator.h
 *   pmv_ba_residuals / pmv_ba_solve        ProjectionResidual + ceres::Solve             (ProjectionResidual.h:38-58,
 *                                          CeresBundleAdjustment.cpp:50-61)              -> BaseOptimizer.h:15
 *
 * Conventions: plain pointers and sizes only; every in/out buffer is caller-allocated HOST memory
 * unless a parameter is documented as a device frame slot; the opaque context owns all device
 * memory and two HIP streams (front-end: detect/LK, back-end: PnP/BA — the reference's two threads,
 * OdometryPipeline.cpp:261-262).  All functions return 0 on success or a negative pmv_status;
 * pmv_last_error() gives the text.  Nothing throws across this boundary.  There is NO CPU fallback:
 * if no gfx950 device / code object is available pmv_ctx_create fails.
 */
#ifndef PMV_HIP_H
#define PMV_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmv_ctx pmv_ctx;

enum pmv_status {
    PMV_OK = 0,
    PMV_ERR_NO_DEVICE = -1,   /* no HIP device / wrong arch */
    PMV_ERR_INVALID = -2,     /* bad argument (null, size, slot out of range) */
    PMV_ERR_CAPACITY = -3,    /* exceeds the capacity given at pmv_ctx_create */
    PMV_ERR_HIP = -4,         /* HIP runtime error, see pmv_last_error */
    PMV_ERR_DEGENERATE = -5,  /* e.g. fewer than 5 PnP points (cv::Exception in the reference) */
    PMV_ERR_OVERFLOW = -6     /* more corners than PMV_GFTT_UNLIMITED_CAP in a cell of a no-limit goodFeaturesToTrack call */
};

/* ---- context ------------------------------------------------------------------------------- */
/* max_w/max_h: largest frame; n_slots: device frame slots (each holds a padded 8-bit pyramid);
 * max_tracks: largest N for lk_track / M for pnp; max_ba_cams / max_ba_points / max_ba_obs: BA caps. */
int pmv_ctx_create(pmv_ctx** out, int device, int max_w, int max_h, int n_slots, int max_tracks,
                   int max_ba_cams, int max_ba_points, int max_ba_obs);
void pmv_ctx_destroy(pmv_ctx* ctx);
const char* pmv_last_error(pmv_ctx* ctx); /* ctx may be NULL for create-time errors */
int pmv_sync(pmv_ctx* ctx);               /* waits for both streams */

/* ---- frames / pyramids ----------------------------------------------------------------------- */
/* Copy one 8-bit gray frame (host) into `slot` and build its LK pyramid (levels as cv::buildOpticalFlowPyramid
 * with winSize 32, maxLevel 4). */
int pmv_frame_upload(pmv_ctx* ctx, int slot, const uint8_t* gray, int w, int h, int stride);
/* The same from a colour image as Frame::Frame(file) reads it (Frame.cpp:33: imread(IMREAD_COLOR) = 8-bit BGR, `stride` bytes per row):
 * cv::cvtColor(BGR2GRAY) (Frame.cpp:40-41) runs on the device, then the pyramid as above. Identity for B = G = R (KITTI's gray PNGs). */
int pmv_frame_upload_bgr(pmv_ctx* ctx, int slot, const uint8_t* bgr, int w, int h, int stride);
/* Batch form: n frames, tightly packed (n*w*h bytes), into slots first_slot..first_slot+n-1. The gray data
 * is staged to HBM first (pmv_frames_stage: level 0 of each slot; a slot keeps no second copy of the frame), the pyramids are
 * built by pmv_frames_build (level 0's REFLECT_101 frame in place + the levels above) so that a benchmark can time the build with
 * inputs already resident in HBM. */
int pmv_frames_stage(pmv_ctx* ctx, int first_slot, int n, const uint8_t* gray, int w, int h);
int pmv_frames_build(pmv_ctx* ctx, int first_slot, int n);
/* Streamed ingest (Frame::Frame / Frame::init + the front-end's per-frame load, Frame.cpp:31-42, OdometryPipeline.cpp:212-220):
 * n tightly packed gray frames in HOST memory (pageable, or pinned / registered: then DMA'ed in place) are copied into slots
 * first_slot.. by an ingest thread on its own HIP stream, chunk by chunk, each chunk's pyramids built as soon as it lands.
 * pmv_frames_stream_begin returns at once; until pmv_frames_stream_end, pmv_lk_track / pmv_detect_* on a slot of the range first
 * make the front-end stream wait for that slot's chunk (nothing else blocks), so tracking starts while later frames are still
 * on their way. `gray` must stay valid until pmv_frames_stream_end, which joins the ingest thread. One stream per context. */
int pmv_frames_stream_begin(pmv_ctx* ctx, int first_slot, int n, const uint8_t* gray, int w, int h);
int pmv_frames_stream_end(pmv_ctx* ctx);
/* Debug/parity: copy pyramid level `level` of `slot` (unpadded, tightly packed) back to host. Returns level dims. */
int pmv_frame_get_level(pmv_ctx* ctx, int slot, int level, uint8_t* out, int* w, int* h);
int pmv_frame_num_levels(pmv_ctx* ctx, int slot); /* maxLevel actually built (>=0) or <0 */
/* Debug/parity: the same level WITH its 64-pixel BORDER_REFLECT_101 frame (what cv::buildOpticalFlowPyramid keeps around every level,
 * lkpyramid.cpp, here PMV_PYR_PAD wide): (w + 128) x (h + 128) bytes, tightly packed; returns the padded dims. */
#define PMV_PYR_PAD 64
int pmv_frame_get_level_padded(pmv_ctx* ctx, int slot, int level, uint8_t* out, int* pw, int* ph);

/* ---- feature extraction ------------------------------------------------------------------------ */
/* cells: n_cells * 4 ints (x0, y0, w, h), each <= 255x255, sub-views of the frame in `slot`.
 * out_xy: n_cells * max_per_cell * 2 ints, CELL-LOCAL (x, y) in descending-response order as OpenCV returns them;
 * out_count: n_cells ints. max_per_cell <= 0 means "no limit" as in cv::goodFeaturesToTrack: out_xy must then hold
 * n_cells * PMV_GFTT_UNLIMITED_CAP * 2 ints (cell stride PMV_GFTT_UNLIMITED_CAP corners); a cell with more corners than that
 * returns PMV_ERR_OVERFLOW. Status bits of one call never leak into the next. */
#define PMV_GFTT_UNLIMITED_CAP 4096
int pmv_detect_gftt(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, double quality,
                    double min_dist, int* out_xy, int* out_count);
/* Same geometry; out_score: n_cells * max_per_cell doubles (the reference fills Feature::score). max_per_cell <= 0 returns no
 * features (ShiTomasiFeatureExtractor.cpp:37-44). */
int pmv_detect_shitomasi(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, double quality,
                         int* out_xy, double* out_score, int* out_count);
/* Debug/parity: response map of one cell (GFTT: float32 min-eigenvalue map before thresholding). */
int pmv_debug_gftt_response(pmv_ctx* ctx, int slot, const int* cell, float* out);
int pmv_debug_shitomasi_response(pmv_ctx* ctx, int slot, const int* cell, double* out);

/* cv::FAST(cell, kp, threshold, nonmax) (OpenCVFASTFeatureExtractor.cpp:8; 9_16 pattern) on sub-views of the frame in `slot`; a
 * "cell" here may be as large as the frame (kNNFeatureMatcher.cpp:11 calls the extractor on the whole next frame). out_xy:
 * n_cells * max_per_cell * 2 ints, cell-local (x, y) in cv::FAST's raster order, first max_per_cell kept as the adapter does
 * (:10-18; max_per_cell <= 0 keeps nothing); out_response: n_cells * max_per_cell floats (the keypoint response = corner score,
 * 0 without non-maximum suppression). */
int pmv_detect_fast(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, int threshold, int nonmax, int* out_xy,
                    float* out_response, int* out_count);

/* ---- feature matching ----------------------------------------------------------------------------- */
/* The arithmetic of kNNFeatureMatcher::matchFeatures (kNNFeatureMatcher.cpp:13-31): for each of the n source features (x, y in
 * src_slot's frame) the n_neighbours nearest of the m candidates (x, y in cmp_slot's frame; getNearestNeighbors :63-101 incl. its
 * repeat-the-last-pick and default-(0,0) quirks) are compared through compareFeatures' pixel window (:103-122) and the best fit is
 * chosen by the sequential `_err < err || err == 0` rule. out_best: n indices into the candidates (-1 = the default Feature at
 * (0,0)); out_err: n floats. Thresholding, displacement statistics and the maps stay in the caller's adapter (:32-60).
 * The reference's constants: n_neighbours 7, window 15. */
int pmv_knn_match(pmv_ctx* ctx, int src_slot, int cmp_slot, const int* src_xy, int n, const int* cmp_xy, int m, int n_neighbours, int window,
                  int* out_best, float* out_err);
/* Pyramidal LK from frame slot `prev_slot` to `next_slot`. prev_xy: n*2 floats. out_xy n*2 floats,
 * out_status n bytes, out_err n floats (exactly the three outputs of cv::calcOpticalFlowPyrLK). */
int pmv_lk_track(pmv_ctx* ctx, int prev_slot, int next_slot, const float* prev_xy, int n, float* out_xy,
                 uint8_t* out_status, float* out_err);

/* ---- PnP ------------------------------------------------------------------------------------------------ */
/* obj_xyz m*3 float32, img_xy m*2 float32, K 9 doubles row-major, rvec/tvec 3 doubles in/out
 * (useExtrinsicGuess=true semantics of the reference call), out_inliers: capacity m ints. */
int pmv_pnp_ransac(pmv_ctx* ctx, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec,
                   double* tvec, int iterations, float reproj_err, double confidence, int* out_inliers,
                   int* out_n_inliers);

/* Debug/parity: models (n*6: rvec,tvec) and inlier counts of the first n hypotheses of the last pmv_pnp_ransac call. */
int pmv_debug_pnp_hypotheses(pmv_ctx* ctx, int n, double* models, int* counts);
/* diagnostic: 32 accumulated shader-clock phase timers of the back-end kernels (recorded only with PMV_BA_STAMPS=1) */
int pmv_debug_ba_stamps(pmv_ctx* ctx, unsigned long long* out32);
/* diagnostic: 16 phase timers of the LK kernel, track 0 (recorded only with PMV_LK_STAMPS=1) */
int pmv_debug_lk_stamps(pmv_ctx* ctx, unsigned long long* out16);

/* ---- bundle adjustment ------------------------------------------------------------------------------------- */
typedef struct pmv_ba_summary {
    double initial_cost, final_cost;
    int iterations;       /* LM iterations executed (successful + unsuccessful) */
    int successful_steps;
    int termination;      /* 0 = max iterations, 1 = function tol, 2 = gradient tol, 3 = parameter tol, 4 = failure */
} pmv_ba_summary;

/* Per-observation residuals (n_obs*2) and Jacobians (n_obs*2*9: d r / d cam[6], d r / d point[3]) of
 * ProjectionResidual (ProjectionResidual.h:38-58). cams nc*6 = [angle-axis(R^T), -t], pts np*3 doubles. */
int pmv_ba_residuals(pmv_ctx* ctx, const double* cams, int nc, const double* pts, int np, const double* obs_xy,
                     const int* cam_idx, const int* pt_idx, int n_obs, const double* K, double* out_r,
                     double* out_J);
/* Levenberg–Marquardt with Huber(huber_delta) loss and Schur elimination of the points; cams/pts updated in place. */
/* Which of the two LM implementations pmv_ba_solve / the pipeline's BA plugin use on this context (default 0):
 *   0  the multi-kernel launch chain: every phase of an LM iteration spread over many CUs - the shortest latency for ONE solve;
 *   1  the whole solve in one workgroup per problem: one launch per solve, and ONE launch per round of B solves in
 *      pmv_pipeline_run_batch - what a GPU shared by many sequences wants (a launch chain pays for wave slots 23 times).
 * Same algorithm, same parity bars against the CPU restatement; the floating-point sums are ordered differently, so two runs
 * are bitwise comparable only under the same mode. */
int pmv_set_ba_mode(pmv_ctx* ctx, int mode);
int pmv_ba_solve(pmv_ctx* ctx, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx,
                 const int* pt_idx, int n_obs, const double* K, double huber_delta, int max_iterations,
                 pmv_ba_summary* summary);

/* ---- five-point RANSAC round -----------------------------------------------------------------------------
 * The hypothesis half of cv::findEssentialMat(p1, p2, K, RANSAC, 0.99, 1.0) (OpenCVFivePointTri.cpp:24): for n_hyp (<= 64) samples of
 * five correspondence indices each (the caller draws them from cv::RNG as RANSACPointSetRegistrator::getSubset does), Nister's
 * solver gives up to 10 essential matrices per sample (models: n_hyp x 10 x 9 doubles, n_models: n_hyp) and every model is scored
 * on all n normalised correspondences q1, q2 (x, y each) by the float32 Sampson distance <= thr (counts: n_hyp x 10). The caller
 * replays the sequential bookkeeping (best so far, RANSACUpdateNumIters) in sample order. One thread per hypothesis: a latency
 * chain, slower than a host core for one sequence, useful when many sequences' rounds share a launch (pmv_pipeline_run_batch). */
int pmv_fivepoint_hypotheses(pmv_ctx* ctx, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models,
                             int* n_models, int* counts);

/* ---- two-view triangulation ------------------------------------------------------------------ */
/* The per-point part of cv::recoverPose(E, p1, p2, K, R, t, HUGE_VAL, mask, tri) (OpenCVFivePointTri.cpp:27): for each of
 * the four (R, t) candidates of decomposeEssentialMat, DLT-triangulate every correspondence (cv::triangulatePoints) and apply
 * the cheirality tests. q1, q2: n normalised image points (x, y) each; P1x4: four row-major 3x4 matrices [R | t];
 * mask_in: n bytes (RANSAC inlier mask of findEssentialMat). out_Q: [4][4][n] homogeneous points, out_mask: [4][n],
 * out_good: [4] number of points passing all tests. n <= max_tracks. */
int pmv_triangulate_candidates(pmv_ctx* ctx, const double* q1, const double* q2, int n, const double* P1x4,
                               const uint8_t* mask_in, double* out_Q, uint8_t* out_mask, int* out_good);
/* The same call on an auxiliary lane of the context (own workspace and stream, created on first use, one call at a time): for a
 * helper thread that evaluates cv::recoverPose of a frame pair AHEAD of the back-end thread (the candidates depend on the 2-D
 * correspondences only), concurrently with pmv_pnp_ransac / pmv_ba_solve / pmv_triangulate_candidates on the main lane. */
int pmv_triangulate_candidates_ahead(pmv_ctx* ctx, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in,
                                     double* out_Q, uint8_t* out_mask, int* out_good);

/* ---- call log (parity tooling) -------------------------------------------------------------------------------------------
 * While recording is on, every pmv_pnp_ransac / pmv_ba_solve / pmv_triangulate_candidates call of this context (also those
 * made from inside pmv_pipeline_run) appends one blob holding its inputs and outputs, so that a test can replay the calls
 * of a whole run, one by one, through another implementation ("teacher forcing"). Little-endian, tightly packed:
 *   PnP:  int32 {0, m, iterations, n_inliers}; f32 obj[3m], img[2m]; f64 K[9], rvec_in[3], tvec_in[3], reproj_err, confidence;
 *         f64 rvec_out[3], tvec_out[3]; int32 inliers[n_inliers]
 *   BA:   int32 {1, nc, np, n_obs, max_iterations}; f64 cams_in[6nc], pts_in[3np], obs[2n_obs], K[9], huber;
 *         int32 cam_idx[n_obs], pt_idx[n_obs]; f64 cams_out[6nc], pts_out[3np], summary[5] (initial, final cost, iterations,
 *         successful steps, termination)
 *   DLT:  int32 {2, n}; f64 q1[2n], q2[2n], P1x4[48]; u8 mask_in[n]; f64 Q[16n]; u8 mask[4n]; int32 good[4]
 * pmv_record_enable(ctx, 1) clears the log and starts, (ctx, 0) stops (the log stays readable). Not thread-safe against a
 * running pipeline: read after the run. */
int pmv_record_enable(pmv_ctx* ctx, int on);
int pmv_record_count(pmv_ctx* ctx);
long long pmv_record_size(pmv_ctx* ctx, int i);
int pmv_record_get(pmv_ctx* ctx, int i, void* out, long long capacity);

/* ---- per-kernel timing (HIP events on the launching stream; used by bench.py for the roofline object) -------------- */
int pmv_prof_enable(pmv_ctx* ctx, int on);  /* on != 0: reset counters and start recording; 0: stop */
int pmv_prof_select(pmv_ctx* ctx, unsigned mask); /* after pmv_prof_enable(1): record only classes whose bit (= id) is set */
int pmv_prof_kernel_count(void);
/* k_lk / k_lk_batch work since context creation / last reset: out3 = {LK iterations, (track, level) passes, tracks}; summed on the
 * host from a 16-bit word per track that the kernels write next to their results (no device-side atomics) */
int pmv_lk_counters(pmv_ctx* ctx, unsigned long long* out3, int reset);
const char* pmv_prof_kernel_name(int id);
int pmv_prof_read(pmv_ctx* ctx, int id, int* launches, double* total_ms, double* max_ms);

/* ---- whole-sequence driver -------------------------------------------------------------------------------------- */
/* Runs the reference's OdometryPipeline schedule (initialise, addFrame per frame, estimatePose with lag 2, BA every
 * bundle_size/3*2 frames; OdometryPipeline.cpp:247-264, :329-426) with every plugin call served by the kernels above.
 * Frames 0..n_frames-1 must already be staged in slots 0..n_frames-1 (pmv_frames_stage); with build_pyramids != 0 the
 * pyramids of all frames are (re)built first, inside the call, so that a benchmark times HBM-resident gray frames ->
 * poses. gt_poses12: n_frames KITTI pose rows (only the translation column is used, for the monocular scale, quirk Q11). */
typedef struct pmv_pipeline_params {
    int n_frames, w, h;
    int min_tracked_features, tracked_features_tol, init_frames, bundle_size, ba_iterations;
    int extractor;      /* 0 = goodFeaturesToTrack (reference default), 1 = ShiTomasi, 2 = FAST (OpenCVFASTFeatureExtractor) */
    int threaded;       /* 0 = sequential schedule, 1 = front-end / back-end host threads (the reference's two threads) */
    int n_threads;      /* host threads that evaluate the triangulator's five-point RANSAC hypotheses side by side (>= 1; results do not depend on it) */
    int build_pyramids; /* rebuild the pyramids of slots 0..n_frames-1 inside the call */
    int matcher;        /* 0 = pyramidal LK (reference default), 1 = kNNFeatureMatcher over `extractor` */
    int device_fivepoint; /* 0 = five-point RANSAC hypotheses on host threads, 1 = on the GPU (pmv_fivepoint_hypotheses); same results */
} pmv_pipeline_params;
typedef struct pmv_pipeline_result pmv_pipeline_result;

int pmv_pipeline_run(pmv_ctx* ctx, const pmv_pipeline_params* params, const double* K9, const double* gt_poses12,
                     pmv_pipeline_result** out);
/* The same run from n_frames gray frames in HOST memory: pmv_frames_stream_begin(ctx, 0, n_frames, host_frames, w, h), the run
 * (build_pyramids ignored), pmv_frames_stream_end. Identical results; copies and pyramid builds overlap the tracking. */
int pmv_pipeline_run_streamed(pmv_ctx* ctx, const pmv_pipeline_params* params, const double* K9, const double* gt_poses12,
                              const uint8_t* host_frames, pmv_pipeline_result** out);
/* B independent sequences through batched launches (SURVEY.md §8e "same kernels with a leading batch dimension"): sequence b =
 * frame slots first_slot[b] .. + params[b].n_frames - 1 (pmv_frames_stage; one frame size for all), intrinsics K9 + 9 b, ground
 * truth gt_poses12[b]. Each sequence keeps the reference's front-end / back-end host threads; their plugin calls are merged into
 * one k_lk_batch / detector / k_pnp_*_batch / k_bamB_* / k_tri_dlt_batch launch per kernel class by that class's combiner
 * thread (one HIP stream each); the combiners are the only threads that talk to the HIP runtime. out[b] is bit-identical to the same sequence's own pmv_pipeline_run. */
int pmv_pipeline_run_batch(pmv_ctx* ctx, int B, const pmv_pipeline_params* params, const double* K9, const double* const* gt_poses12,
                           const int* first_slot, pmv_pipeline_result** out);
/* diagnostic, per combiner in the order LK, detectors, PnP, BA, DLT: counts10 = {launch rounds, requests served} x 5; times15 (may
 * be NULL) = seconds spent {CPU time of the combiner thread, wall time processing batches, of that waiting for the GPU} x 5 */
int pmv_batch_stats(pmv_ctx* ctx, long long* counts10, double* times15);
void pmv_pipeline_free(pmv_pipeline_result* r);
/* Same, but the (host-container) teardown runs on a background thread; pmv_pipeline_drain() joins all of them. */
void pmv_pipeline_release(pmv_pipeline_result* r);
void pmv_pipeline_drain(void);
int pmv_pipeline_num_poses(const pmv_pipeline_result* r);
void pmv_pipeline_get_poses(const pmv_pipeline_result* r, double* out12); /* per pose: R row-major (9) then t (3) */
int pmv_pipeline_num_frames(const pmv_pipeline_result* r);
int pmv_pipeline_frame_feature_count(const pmv_pipeline_result* r, int k);
void pmv_pipeline_get_frame_features(const pmv_pipeline_result* r, int k, int* out3); /* (column,row,landmark id|-1) */
/* Run statistics: pmv_pipeline_stats_count() (= 25) doubles, in this order:
 *   [0] lk_calls [1] lk_points [2] detect_calls [3] pnp_calls [4] pnp_points [5] tri_calls [6] ba_calls [7] ba_obs [8] ba_points
 *   [9] heuristic_motion (frames whose pose came from motionHeuristics' fallback branch) [10] run seconds [11] init_offset
 *   [12] live landmarks at the end [13] scale; wall seconds per stage as seen by the calling host threads: [14] t_lk [15] t_detect
 *   [16] t_pnp [17] t_tri [18] t_ba [19] t_pnp_kernel [20] t_ba_kernel [21] t_tri_essential [22] t_tri_pose; [23] five-point
 *   RANSAC samples drawn [24] tri_ahead: two-view calls whose findEssentialMat + recoverPose had been computed ahead of the back-end
 *   by a helper thread (two-thread pipeline). The caller's buffer must hold pmv_pipeline_stats_count() doubles. */
int pmv_pipeline_stats_count(void);
void pmv_pipeline_get_stats(const pmv_pipeline_result* r, double* out25);

#ifdef __cplusplus
}
#endif
#endif
