// C-ABI entry points of the back-end stream: pmv_pnp_ransac, pmv_ba_residuals, pmv_ba_solve (include/pmv_hip.h).
#include "pmv_ctx.h"
#include "backend.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace pmv {

constexpr int MAX_HYP = 1024;

struct BackendBuffers {
    // BA problem
    double *d_cams = nullptr, *d_pts = nullptr, *d_obs = nullptr, *d_K = nullptr;
    int *d_cam_idx = nullptr, *d_pt_idx = nullptr, *d_pobs_start = nullptr, *d_pobs_list = nullptr, *d_cobs_start = nullptr, *d_cobs_list = nullptr;
    // BA workspaces
    double *d_x = nullptr, *d_cand = nullptr, *d_scale = nullptr, *d_diag = nullptr, *d_D2 = nullptr, *d_step = nullptr, *d_res = nullptr,
           *d_J = nullptr, *d_Einv = nullptr, *d_gp = nullptr, *d_Yd = nullptr, *d_Wd = nullptr, *d_S = nullptr, *d_rhs = nullptr,
           *d_Gpart = nullptr, *d_summary = nullptr;
    size_t ydwd_elems = 0, gpart_elems = 0;
    // pinned staging
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // PnP
    float *d_obj = nullptr, *d_img = nullptr;
    int *d_samples = nullptr, *d_counts = nullptr, *d_inliers = nullptr, *d_info = nullptr;
    double *d_models = nullptr, *d_rt = nullptr, *d_Kp = nullptr;
    uint8_t* d_masks = nullptr;
};

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

int backend_create(pmv_ctx* c) {
    BackendBuffers* b = new BackendBuffers();
    c->be = b;
    const size_t nc = (size_t)std::max(c->max_ba_cams, 1), np = (size_t)std::max(c->max_ba_points, 1), no = (size_t)std::max(c->max_ba_obs, 1);
    const size_t n = 6 * nc + 3 * np, m = 6 * nc;
    const size_t ldw = (size_t)round_up((int)m + 1, 16), krows = (size_t)round_up((int)(3 * np), 4);
#define CKB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(c, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
    CKB(hipMalloc(&b->d_cams, nc * 6 * 8)); CKB(hipMalloc(&b->d_pts, np * 3 * 8)); CKB(hipMalloc(&b->d_obs, no * 2 * 8)); CKB(hipMalloc(&b->d_K, 9 * 8));
    CKB(hipMalloc(&b->d_cam_idx, no * 4)); CKB(hipMalloc(&b->d_pt_idx, no * 4));
    CKB(hipMalloc(&b->d_pobs_start, (np + 1) * 4)); CKB(hipMalloc(&b->d_pobs_list, no * 4));
    CKB(hipMalloc(&b->d_cobs_start, (nc + 1) * 4)); CKB(hipMalloc(&b->d_cobs_list, no * 4));
    CKB(hipMalloc(&b->d_x, n * 8)); CKB(hipMalloc(&b->d_cand, n * 8)); CKB(hipMalloc(&b->d_scale, n * 8)); CKB(hipMalloc(&b->d_diag, n * 8));
    CKB(hipMalloc(&b->d_D2, n * 8)); CKB(hipMalloc(&b->d_step, n * 8));
    CKB(hipMalloc(&b->d_res, no * 2 * 8)); CKB(hipMalloc(&b->d_J, no * 18 * 8));
    CKB(hipMalloc(&b->d_Einv, np * 9 * 8)); CKB(hipMalloc(&b->d_gp, np * 3 * 8));
    b->ydwd_elems = krows * ldw;
    CKB(hipMalloc(&b->d_Yd, b->ydwd_elems * 8)); CKB(hipMalloc(&b->d_Wd, b->ydwd_elems * 8));
    CKB(hipMalloc(&b->d_S, m * m * 8)); CKB(hipMalloc(&b->d_rhs, m * 8));
    b->gpart_elems = (size_t)8 * ldw * ldw;   // up to 8 K-slices of an (ldw x ldw) tile grid
    CKB(hipMalloc(&b->d_Gpart, b->gpart_elems * 8));
    CKB(hipMalloc(&b->d_summary, 8 * 8));
    const size_t mt = (size_t)c->max_tracks;
    b->h_stage_bytes = std::max<size_t>(no * 18 * 8 + no * 2 * 8, std::max<size_t>(n * 8 + no * 32 + (np + nc + 2) * 4, mt * 32 + MAX_HYP * 20 + 4096));
    CKB(hipHostMalloc(&b->h_stage, b->h_stage_bytes));
    CKB(hipMalloc(&b->d_obj, mt * 12)); CKB(hipMalloc(&b->d_img, mt * 8));
    CKB(hipMalloc(&b->d_samples, MAX_HYP * 5 * 4)); CKB(hipMalloc(&b->d_counts, MAX_HYP * 4));
    CKB(hipMalloc(&b->d_inliers, mt * 4)); CKB(hipMalloc(&b->d_info, 16));
    CKB(hipMalloc(&b->d_models, MAX_HYP * 6 * 8)); CKB(hipMalloc(&b->d_rt, 6 * 8)); CKB(hipMalloc(&b->d_Kp, 9 * 8));
    CKB(hipMalloc(&b->d_masks, (size_t)MAX_HYP * mt));
#undef CKB
    return PMV_OK;
}

void backend_destroy(pmv_ctx* c) {
    BackendBuffers* b = c->be;
    if (!b) return;
    void* ptrs[] = {b->d_cams, b->d_pts, b->d_obs, b->d_K, b->d_cam_idx, b->d_pt_idx, b->d_pobs_start, b->d_pobs_list, b->d_cobs_start,
                    b->d_cobs_list, b->d_x, b->d_cand, b->d_scale, b->d_diag, b->d_D2, b->d_step, b->d_res, b->d_J, b->d_Einv, b->d_gp,
                    b->d_Yd, b->d_Wd, b->d_S, b->d_rhs, b->d_Gpart, b->d_summary, b->d_obj, b->d_img, b->d_samples, b->d_counts,
                    b->d_inliers, b->d_info, b->d_models, b->d_rt, b->d_Kp, b->d_masks};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    delete b;
    c->be = nullptr;
}

// cv::RNG (multiply-with-carry) and RANSACPointSetRegistrator::getSubset (5 distinct indices)
struct CvRNG {
    uint64_t state;
    explicit CvRNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

}  // namespace pmv

using namespace pmv;

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

extern "C" {

int pmv_pnp_ransac(pmv_ctx* ctx, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec,
                   int iterations, float reproj_err, double confidence, int* out_inliers, int* out_n_inliers) {
    REQ(ctx && obj_xyz && img_xy && K && rvec && tvec && out_inliers && out_n_inliers, PMV_ERR_INVALID, "pmv_pnp_ransac: null argument");
    *out_n_inliers = 0;
    REQ(m <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_pnp_ransac: m=%d exceeds max_tracks=%d", m, ctx->max_tracks);
    // cv::solvePnPRansac asserts npoints >= 4; with 4 points it switches to P3P, with 5 it runs the kernel once (not built: the
    // reference only calls it with >= tracked_features_tol points)
    REQ(m >= 6, PMV_ERR_DEGENERATE, "pmv_pnp_ransac: %d correspondences (need >= 6)", m);
    REQ(iterations >= 1 && iterations <= MAX_HYP, PMV_ERR_CAPACITY, "pmv_pnp_ransac: iterations=%d (1..%d)", iterations, MAX_HYP);
    REQ(confidence > 0 && confidence < 1, PMV_ERR_INVALID, "pmv_pnp_ransac: confidence must be in (0,1)");
    CKC(hipSetDevice(ctx->device));
    if (const char* dump = getenv("PMV_DUMP_PNP")) {   // debug: append the inputs of every call to a file
        if (FILE* f = fopen(dump, "ab")) {
            fwrite(&m, 4, 1, f); fwrite(obj_xyz, 12, m, f); fwrite(img_xy, 8, m, f); fwrite(K, 8, 9, f); fwrite(rvec, 8, 3, f); fwrite(tvec, 8, 3, f);
            fclose(f);
        }
    }
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    char* hs = (char*)b->h_stage;
    float* h_obj = (float*)hs; float* h_img = (float*)(hs + (size_t)m * 12); int* h_samples = (int*)(hs + (size_t)m * 20);
    double* h_K = (double*)(hs + (size_t)m * 20 + (size_t)iterations * 20 + 8 - (((size_t)m * 20 + (size_t)iterations * 20) % 8));
    memcpy(h_obj, obj_xyz, (size_t)m * 12);
    memcpy(h_img, img_xy, (size_t)m * 8);
    memcpy(h_K, K, 72);
    CvRNG rng((uint64_t)-1);
    for (int it = 0; it < iterations; it++) {
        int* idx = h_samples + it * 5;
        for (int i = 0; i < 5;) {
            int idx_i;
            for (;;) {
                idx_i = idx[i] = rng.uniform(0, m);
                int j = 0;
                for (; j < i; j++) if (idx_i == idx[j]) break;
                if (j == i) break;
            }
            i++;
        }
    }
    CKC(hipMemcpyAsync(b->d_obj, h_obj, (size_t)m * 12, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_img, h_img, (size_t)m * 8, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_samples, h_samples, (size_t)iterations * 20, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_Kp, h_K, 72, hipMemcpyHostToDevice, s));
    const float thr = (float)((double)reproj_err * (double)reproj_err);
    CKC(launch_pnp(s, b->d_obj, b->d_img, m, b->d_Kp, b->d_samples, iterations, thr, confidence, b->d_models, b->d_masks, b->d_counts,
                   b->d_rt, b->d_inliers, b->d_info));
    double* h_rt = (double*)hs;                      // staging reused for the results
    int* h_info = (int*)(hs + 64);
    int* h_inl = (int*)(hs + 128);
    CKC(hipMemcpyAsync(h_rt, b->d_rt, 48, hipMemcpyDeviceToHost, s));
    CKC(hipMemcpyAsync(h_info, b->d_info, 8, hipMemcpyDeviceToHost, s));
    CKC(hipMemcpyAsync(h_inl, b->d_inliers, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    CKC(hipStreamSynchronize(s));
    const int n = h_info[0];
    for (int i = 0; i < 3; i++) { rvec[i] = h_rt[i]; tvec[i] = h_rt[3 + i]; }
    *out_n_inliers = n;
    if (n > 0) memcpy(out_inliers, h_inl, (size_t)n * 4);
    return PMV_OK;
}

// debug/parity: models (n x 6) and inlier counts of the hypotheses evaluated by the last pmv_pnp_ransac call
int pmv_debug_pnp_hypotheses(pmv_ctx* ctx, int n, double* models, int* counts) {
    REQ(ctx && models && counts && n >= 1 && n <= MAX_HYP, PMV_ERR_INVALID, "pmv_debug_pnp_hypotheses: bad argument");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_back));
    CKC(hipMemcpy(models, ctx->be->d_models, (size_t)n * 48, hipMemcpyDeviceToHost));
    CKC(hipMemcpy(counts, ctx->be->d_counts, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PMV_OK;
}

int pmv_ba_residuals(pmv_ctx* ctx, const double* cams, int nc, const double* pts, int np, const double* obs_xy, const int* cam_idx,
                     const int* pt_idx, int n_obs, const double* K, double* out_r, double* out_J) {
    REQ(ctx && cams && pts && obs_xy && cam_idx && pt_idx && K && out_r && out_J, PMV_ERR_INVALID, "pmv_ba_residuals: null argument");
    REQ(nc >= 1 && nc <= ctx->max_ba_cams && np >= 1 && np <= ctx->max_ba_points && n_obs >= 0 && n_obs <= ctx->max_ba_obs, PMV_ERR_CAPACITY,
        "pmv_ba_residuals: nc=%d np=%d n_obs=%d exceed capacity %d/%d/%d", nc, np, n_obs, ctx->max_ba_cams, ctx->max_ba_points, ctx->max_ba_obs);
    for (int i = 0; i < n_obs; i++) REQ(cam_idx[i] >= 0 && cam_idx[i] < nc && pt_idx[i] >= 0 && pt_idx[i] < np, PMV_ERR_INVALID, "pmv_ba_residuals: index out of range at observation %d", i);
    if (n_obs == 0) return PMV_OK;
    CKC(hipSetDevice(ctx->device));
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    CKC(hipMemcpyAsync(b->d_cams, cams, (size_t)nc * 48, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_pts, pts, (size_t)np * 24, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_obs, obs_xy, (size_t)n_obs * 16, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_cam_idx, cam_idx, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_pt_idx, pt_idx, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_K, K, 72, hipMemcpyHostToDevice, s));
    CKC(launch_ba_residuals(s, b->d_cams, b->d_pts, b->d_obs, b->d_cam_idx, b->d_pt_idx, n_obs, b->d_K, b->d_res, b->d_J));
    CKC(hipMemcpyAsync(out_r, b->d_res, (size_t)n_obs * 16, hipMemcpyDeviceToHost, s));
    CKC(hipMemcpyAsync(out_J, b->d_J, (size_t)n_obs * 144, hipMemcpyDeviceToHost, s));
    CKC(hipStreamSynchronize(s));
    return PMV_OK;
}

int pmv_ba_solve(pmv_ctx* ctx, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx,
                 int n_obs, const double* K, double huber_delta, int max_iterations, pmv_ba_summary* summary) {
    REQ(ctx && cams && pts && obs_xy && cam_idx && pt_idx && K, PMV_ERR_INVALID, "pmv_ba_solve: null argument");
    REQ(nc >= 1 && nc <= ctx->max_ba_cams && np >= 1 && np <= ctx->max_ba_points && n_obs >= 1 && n_obs <= ctx->max_ba_obs, PMV_ERR_CAPACITY,
        "pmv_ba_solve: nc=%d np=%d n_obs=%d exceed capacity %d/%d/%d", nc, np, n_obs, ctx->max_ba_cams, ctx->max_ba_points, ctx->max_ba_obs);
    REQ(max_iterations >= 0 && huber_delta > 0, PMV_ERR_INVALID, "pmv_ba_solve: bad options");
    CKC(hipSetDevice(ctx->device));
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    // observation lists per point / per camera (counting sort, observation order preserved)
    char* hs = (char*)b->h_stage;
    int* pstart = (int*)hs; int* plist = pstart + (np + 1); int* cstart = plist + n_obs; int* clist = cstart + (nc + 1);
    std::fill(pstart, pstart + np + 1, 0);
    std::fill(cstart, cstart + nc + 1, 0);
    for (int i = 0; i < n_obs; i++) {
        REQ(cam_idx[i] >= 0 && cam_idx[i] < nc && pt_idx[i] >= 0 && pt_idx[i] < np, PMV_ERR_INVALID, "pmv_ba_solve: index out of range at observation %d", i);
        pstart[pt_idx[i] + 1]++; cstart[cam_idx[i] + 1]++;
    }
    for (int p = 0; p < np; p++) pstart[p + 1] += pstart[p];
    for (int c = 0; c < nc; c++) cstart[c + 1] += cstart[c];
    {
        std::vector<int> pf(pstart, pstart + np), cf(cstart, cstart + nc);
        for (int i = 0; i < n_obs; i++) { plist[pf[pt_idx[i]]++] = i; clist[cf[cam_idx[i]]++] = i; }
    }
    CKC(hipMemcpyAsync(b->d_pobs_start, pstart, (size_t)(np + 1) * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_pobs_list, plist, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_cobs_start, cstart, (size_t)(nc + 1) * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_cobs_list, clist, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_cams, cams, (size_t)nc * 48, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_pts, pts, (size_t)np * 24, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_obs, obs_xy, (size_t)n_obs * 16, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_cam_idx, cam_idx, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_pt_idx, pt_idx, (size_t)n_obs * 4, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_K, K, 72, hipMemcpyHostToDevice, s));
    BAArgs A;
    A.cams = b->d_cams; A.pts = b->d_pts; A.obs = b->d_obs; A.cam_idx = b->d_cam_idx; A.pt_idx = b->d_pt_idx; A.K = b->d_K;
    A.pobs_start = b->d_pobs_start; A.pobs_list = b->d_pobs_list; A.cobs_start = b->d_cobs_start; A.cobs_list = b->d_cobs_list;
    A.nc = nc; A.np = np; A.nobs = n_obs; A.max_iterations = max_iterations; A.huber = huber_delta;
    A.x = b->d_x; A.cand = b->d_cand; A.scale = b->d_scale; A.diag = b->d_diag; A.D2 = b->d_D2; A.step = b->d_step; A.res = b->d_res; A.J = b->d_J;
    A.Einv = b->d_Einv; A.gp = b->d_gp; A.Yd = b->d_Yd; A.Wd = b->d_Wd; A.S = b->d_S; A.rhs = b->d_rhs; A.Gpart = b->d_Gpart; A.summary = b->d_summary;
    const int m = 6 * nc;
    A.tiles_r = (m + 15) / 16; A.tiles_c = (m + 1 + 15) / 16;
    A.ldw = A.tiles_c * 16;
    A.krows = round_up(3 * np, 4);
    A.gp_rows = A.tiles_r * 16;
    int ks = 8 / (A.tiles_r * A.tiles_c);
    if (ks < 1) ks = 1;
    if (ks > 8) ks = 8;
    A.kper = round_up((A.krows + ks - 1) / ks, 4);
    A.kslices = (A.krows + A.kper - 1) / A.kper;
    REQ((size_t)A.krows * A.ldw <= b->ydwd_elems && (size_t)A.kslices * A.gp_rows * A.ldw <= b->gpart_elems, PMV_ERR_CAPACITY, "pmv_ba_solve: workspace too small");
    CKC(launch_ba_lm(s, A));
    double* h_out = (double*)hs;
    CKC(hipStreamSynchronize(s));   // staging (pstart...) is reused below
    CKC(hipMemcpyAsync(h_out, b->d_summary, 40, hipMemcpyDeviceToHost, s));
    CKC(hipMemcpyAsync(h_out + 8, b->d_cams, (size_t)nc * 48, hipMemcpyDeviceToHost, s));
    CKC(hipMemcpyAsync(h_out + 8 + (size_t)nc * 6, b->d_pts, (size_t)np * 24, hipMemcpyDeviceToHost, s));
    CKC(hipStreamSynchronize(s));
    memcpy(cams, h_out + 8, (size_t)nc * 48);
    memcpy(pts, h_out + 8 + (size_t)nc * 6, (size_t)np * 24);
    if (summary) {
        summary->initial_cost = h_out[0]; summary->final_cost = h_out[1]; summary->iterations = (int)h_out[2];
        summary->successful_steps = (int)h_out[3]; summary->termination = (int)h_out[4];
    }
    return PMV_OK;
}

}  // extern "C"
