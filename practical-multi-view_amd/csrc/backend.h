// Back-end (PnP / BA) launch interfaces shared by backend*.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pmv_hip.h"

namespace pmv {

// Device / pinned buffers of ONE back-end problem slot (PnP, BA, two-view DLT workspaces and the single-copy staging blocks). The
// context owns one (the plain C-ABI calls); the batch engine owns one per concurrent sequence.
constexpr int MAX_HYP = 1024;

struct BackendBuffers {
    unsigned done_seq = 0;   // sequence number of the last call that signals its completion through the result block
    // BA problem
    double *d_cams = nullptr, *d_pts = nullptr, *d_obs = nullptr, *d_K = nullptr;
    int *d_cam_idx = nullptr, *d_pt_idx = nullptr, *d_pobs_start = nullptr, *d_pobs_list = nullptr, *d_cobs_start = nullptr, *d_cobs_list = nullptr;
    // BA workspaces
    double *d_x = nullptr, *d_cand = nullptr, *d_scale = nullptr, *d_diag = nullptr, *d_D2 = nullptr, *d_step = nullptr, *d_res = nullptr,
           *d_J = nullptr, *d_Einv = nullptr, *d_gp = nullptr, *d_Yd = nullptr, *d_Wd = nullptr, *d_S = nullptr, *d_rhs = nullptr,
           *d_Gpart = nullptr, *d_summary = nullptr;
    size_t ydwd_elems = 0, gpart_elems = 0;
    unsigned long long* d_stamps = nullptr;
    void* d_bastate = nullptr;
    double* d_bapart = nullptr;
    char *d_tri_in = nullptr, *d_tri_out = nullptr;
    char* d_fp_work = nullptr;   // five-point round: [models FP_MAX_HYP x 90 doubles | n_models FP_MAX_HYP ints] (inputs share d_tri_in)
    char* d_h_stage = nullptr;   // device address of the pinned staging block: result blocks are written straight into it
    size_t tri_in_bytes = 0, tri_out_bytes = 0;
    // single-copy transfers: one pinned staging block and one device block per direction
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    char* d_ba_io = nullptr;   // [summary 8 | cams | pts | obs | K | cam_idx | pt_idx | pobs_start | pobs_list | cobs_start | cobs_list]
    size_t ba_io_bytes = 0;
    char* d_pnp_in = nullptr;  // [K 10 doubles | obj | img | samples]
    char* d_pnp_out = nullptr; // [rt 6 doubles | info 4 ints | inliers]
    size_t pnp_in_bytes = 0, pnp_out_bytes = 0;
    // PnP
    float *d_obj = nullptr, *d_img = nullptr;
    int *d_samples = nullptr, *d_counts = nullptr, *d_inliers = nullptr, *d_info = nullptr;
    double *d_models = nullptr, *d_rt = nullptr, *d_Kp = nullptr;
    uint8_t* d_masks = nullptr;
};

constexpr size_t PNP_HDR = 384;   // bytes: K (10 doubles) + 33 powers of ten + padding

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
    unsigned done_seq;            // != 0 (with `out`): the finish kernel stores it last into the low word of summary slot 7 (the host spins on it)
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
                      unsigned long long* d_stamps /* diagnostic, may be null */, unsigned done_seq = 0 /* see pnp_select_refit_body */);

// one problem of a batched multi-kernel LM launch chain: the arguments launch_ba_multi derives for a single solve, kept in device memory
struct BAProb {
    BAArgs A;
    void* st2[2];            // double-buffered BAGState
    int* chol_flags;
    double *part_cost, *part_gmax, *part4, *Ublk, *rhsblk, *candrot, *tmp3;
    int nbo, nbp, clear_blocks, tiles;
};
struct BABatchDims { int n_probs, max_eval_blocks, max_nc, max_nbp, max_tiles, max_m, max_iterations; };
// fills everything of `P` that launch_ba_multi computes from A / d_state / d_part (host side; the result is copied to the device)
void ba_fill_prob(BAProb& P, const BAArgs& A, void* d_state, double* d_part);
hipError_t launch_ba_multi_batch(hipStream_t s, const BAProb* d_probs, const BABatchDims& D);
// the one-workgroup-per-problem form (pmv_set_ba_mode(ctx, 1)): d_args[i] = problem i, max_m = largest 6 * nc of the batch
hipError_t launch_ba_lm_batch(hipStream_t s, const BAArgs* d_args, int n_probs, int max_m);
// one RANSAC problem of a batched PnP launch (device pointers)
struct PnPProblem {
    const float* obj; const float* img; const int* samples; const double* K;   // K: 9 doubles + 10^k table (see pmv_pnp_ransac)
    double* models; uint8_t* masks; int* counts; double* rt_out; int* inliers; int* info; char* host_out;
    int m, n_hyp; float thr; double confidence;
};
hipError_t launch_pnp_batch(hipStream_t s, const PnPProblem* d_probs, int n_probs, int max_hyp);
// one findEssentialMat RANSAC round of a batched five-point launch (backend_fivepoint.hip): n_hyp samples of 5 correspondences each
struct FivePointProblem {
    const int* samples; const double* q1; const double* q2;   // device inputs: 5 n_hyp indices, n normalised points each
    double* models_d; int* n_models_d;                        // device: n_hyp x 90 doubles, n_hyp ints (read by the scoring kernel)
    double* models_h; int* n_models_h; int* counts_h;         // mapped pinned results: models, model counts, n_hyp x 10 inlier counts
    int n, n_hyp; float thr;
};
constexpr int FP_MAX_HYP = 64;   // hypotheses per round
hipError_t launch_fivepoint_batch(hipStream_t s, const FivePointProblem* d_probs, int n_probs, int max_hyp);
// one two-view problem of a batched DLT launch
struct DltProblem { const double* P1x4; const double* q1; const double* q2; const uint8_t* mask_in; double* Q; uint8_t* mask; int n; };
hipError_t launch_tri_dlt_batch(hipStream_t s, const DltProblem* d_probs, int n_probs, int max_n);
hipError_t launch_tri_dlt(hipStream_t s, const double* d_P1x4, const double* d_q1, const double* d_q2, const uint8_t* d_mask_in, int n,
                          double* d_Q, uint8_t* d_mask);

// prepare / finish halves of the back-end entry points (backend.hip), shared by the single calls and the batch engine
struct PnPProblem; struct DltProblem;
}  // namespace pmv
struct pmv_ctx;
namespace pmv {
int backend_alloc(pmv_ctx* c, BackendBuffers** out);
void backend_free(BackendBuffers* b);
int pnp_check(pmv_ctx* ctx, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec, int iterations,
              double confidence, int* out_inliers, int* out_n_inliers);
void pnp_prepare(BackendBuffers* b, const float* obj_xyz, const float* img_xy, int m, const double* K, int iterations, float reproj_err,
                 double confidence, PnPProblem* P, size_t* in_bytes);
void pnp_finish(pmv_ctx* ctx, BackendBuffers* b, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec,
                int iterations, float reproj_err, double confidence, size_t in_bytes, int* out_inliers, int* out_n_inliers);
int ba_check(pmv_ctx* ctx, const double* cams, int nc, const double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx,
             int n_obs, const double* K, double huber_delta, int max_iterations);
int ba_prepare(pmv_ctx* ctx, BackendBuffers* b, const double* cams, int nc, const double* pts, int np, const double* obs_xy, const int* cam_idx,
               const int* pt_idx, int n_obs, const double* K, double huber_delta, int max_iterations, bool multi, BAArgs* A, size_t* io_bytes);
void ba_finish(pmv_ctx* ctx, BackendBuffers* b, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx,
               const int* pt_idx, int n_obs, const double* K, double huber_delta, int max_iterations, pmv_ba_summary* summary);
// five-point round: samples (5 n_hyp ints) of n normalised correspondences; thr = squared Sampson threshold as float
int fivepoint_prepare(pmv_ctx* ctx, BackendBuffers* b, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr,
                      FivePointProblem* P, size_t* in_bytes);
void fivepoint_finish(BackendBuffers* b, int n_hyp, size_t in_bytes, double* models, int* n_models, int* counts);
void dlt_prepare(BackendBuffers* b, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, DltProblem* P,
                 size_t* in_bytes);
void dlt_finish(pmv_ctx* ctx, BackendBuffers* b, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in,
                size_t in_bytes, double* out_Q, uint8_t* out_mask, int* out_good);
}  // namespace pmv
