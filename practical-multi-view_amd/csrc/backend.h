// Back-end (PnP / BA) launch interfaces shared by backend*.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pmv {

struct BAArgs {
    // problem (device)
    double* cams; double* pts; const double* obs; const int* cam_idx; const int* pt_idx; const double* K;
    const int* pobs_start; const int* pobs_list; const int* cobs_start; const int* cobs_list;
    const int4* erec;  // per entry e of pobs_list: (observation index, camera, point, dup flag) in one 16-byte record; dup flag: 0 = only
                       // observation of its (point, camera), 1 = first of several, 2 = a later one
    int nc, np, nobs, max_iterations;
    double huber;
    // workspaces (device)
    double *x, *cand, *scale, *diag, *D2, *step, *res, *J, *Einv, *gp, *Yd, *Wd, *S, *rhs, *Gpart, *summary;
    int ldw, krows, tiles_r, tiles_c, kslices, kper, gp_rows;
    unsigned long long* stamps;   // optional (diagnostic): 32 accumulated shader-clock phase timers
    double* out;                  // optional: host-mapped result block [summary 8 | cams 6*nc | pts 3*np] (multi-kernel LM)
};

hipError_t launch_ba_residuals(hipStream_t s, const double* cams, const double* pts, const double* obs, const int* cam_idx,
                               const int* pt_idx, int nobs, const double* K, double* out_r, double* out_J);
hipError_t launch_ba_lm(hipStream_t s, const BAArgs& A);
// multi-kernel LM (default); d_state: >= 512 + 4*BA_MAX_ITERATIONS B, d_part: >= (nobs/256 + 5*np/64 + 58*nc + 3*nobs + 8) doubles
constexpr int BA_MAX_ITERATIONS = 512;   // per-iteration flags of the multi-kernel LM
hipError_t launch_ba_multi(hipStream_t s, const BAArgs& A, void* d_state, double* d_part);
hipError_t launch_pnp(hipStream_t s, const float* d_obj, const float* d_img, int m, const double* d_K, const int* d_samples,
                      int n_hyp, float thr, double confidence, double* d_models, uint8_t* d_masks, int* d_counts,
                      double* d_rt_out, int* d_inliers, int* d_info, char* host_out /* mapped pinned [rt 48 B | info 16 B | inliers] */,
                      unsigned long long* d_stamps /* diagnostic, may be null */);

hipError_t launch_tri_dlt(hipStream_t s, const double* d_P1x4, const double* d_q1, const double* d_q2, const uint8_t* d_mask_in, int n,
                          double* d_Q, uint8_t* d_mask);

}  // namespace pmv
