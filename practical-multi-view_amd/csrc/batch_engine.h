// Batch engine interface (batch_engine.hip): request entry points used by the batched pipeline's plugin set.
#pragma once
#include <stdint.h>
struct pmv_ctx;
namespace pmv {
struct BatchEngine;
int batch_engine_get(pmv_ctx* ctx, int B, BatchEngine** out);   // creates (or grows) the context's engine for B concurrent sequences
void batch_engine_destroy(pmv_ctx* ctx);
// pyramids of the run's sequences built in the background while the sequences already track (see BatchEngine::slot_round)
int engine_build_begin(BatchEngine* E, int B, const int* first_slot, const int* n_frames, const int* build);
int engine_build_end(BatchEngine* E);
// per combiner (LK, detectors, PnP, BA, DLT): counts10 = {launch rounds, requests} x 5; times15 (may be null) = seconds spent
// {CPU time of the combiner thread, wall time processing batches, of that waiting for the GPU} x 5
void batch_engine_stats(BatchEngine* E, long long* counts10, double* times15);
// same contracts as pmv_lk_track / pmv_detect_* / pmv_pnp_ransac / pmv_ba_solve / pmv_triangulate_candidates; `seq` selects the
// sequence's back-end workspace set. Blocking; safe to call from many threads at once (one outstanding call per seq and stream role).
// predicted_iters (optional): how many LK iterations the caller expects each track to take (0..255) - the launch starts the expensive
// tracks first; iters_out (optional): what each track took. Neither changes a result.
int engine_lk(BatchEngine* E, int prev_slot, int next_slot, const float* prev_xy, int n, float* out_xy, uint8_t* status, float* err,
              const uint8_t* predicted_iters = nullptr, uint8_t* iters_out = nullptr);
int engine_detect(BatchEngine* E, int kind /* 1 GFTT, 2 ShiTomasi */, int slot, const int* cells, int n_cells, int max_per_cell, double quality,
                  double min_dist, int* out_xy, double* out_score, int* out_count);
int engine_pnp(BatchEngine* E, int seq, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec, int iterations,
               float reproj_err, double confidence, int* out_inliers, int* out_n_inliers);
int engine_ba(BatchEngine* E, int seq, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
              const double* K, double huber, int max_iterations);
int engine_dlt(BatchEngine* E, int seq, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
               uint8_t* out_mask, int* out_good);
int engine_fivepoint(BatchEngine* E, int seq, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models,
                     int* n_models, int* counts);
}  // namespace pmv
