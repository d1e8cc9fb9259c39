// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). Internal declarations shared by the oracle translation units.
#pragma once
#include "orc_common.h"
namespace orc {
struct LKParams { int win = 32; int max_level = 4; int max_iter = 30; double eps = 0.01; float min_eig = 1e-4f; };
void lk_track(const Image8& prev, const Image8& next, const float* prev_xy, int n, const LKParams& P, float* out_xy,
              uint8_t* out_status, float* out_err, int* levels_used, int nthreads = 1);
int gftt_cell(const uint8_t* img, int W, int H, int cx0, int cy0, int cw, int ch, int max_corners, double quality,
              double min_dist, int* out_xy, float* eig_out);
int shitomasi_cell(const uint8_t* img, int W, int cx0, int cy0, int cw, int ch, int max_feats, double quality, int* out_xy,
                   double* out_score, double* R_out);
int pnp_ransac(const float* obj, const float* img, int m, const double K[9], double rvec[3], double tvec[3], int max_iters,
               float reproj_err, double confidence, int* inliers, int* hyp_used);
void knn_match(const uint8_t* src, const uint8_t* cmp, int w, int h, const int* src_xy, int n, const int* cmp_xy, int m, int n_nn, int window,
               int* out_best, float* out_err);
int fast9_cell(const uint8_t* img, int W, int cx0, int cy0, int cw, int ch, int threshold, bool nonmax, int max_kp, int* out_xy, float* out_response);
struct BASummary { double initial_cost, final_cost; int iterations, successful_steps, termination; };
int ba_solve(double* cams, int nc, double* pts, int np, const double* obs, const int* cam_idx, const int* pt_idx, int nobs,
             const double* K, double huber, int max_iterations, BASummary* sum);
}
