// C-ABI entry points of the back-end stream: pmv_pnp_ransac, pmv_ba_residuals, pmv_ba_solve (include/pmv_hip.h).
#include "pmv_ctx.h"
#include "backend.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace pmv {

// Input block of a solve from mapped pinned host memory into HBM: a copy launch on the solve's own stream instead of a DMA-engine
// transfer (hipMemcpyAsync of ~50 KB goes through SDMA: 10-15 us until the first kernel of the chain may start; this is ~5).
__global__ __launch_bounds__(256) void k_stage_block(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned n16) { BACKEND_PRIO();
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}


static int round_up(int a, int b) { return (a + b - 1) / b * b; }

int backend_alloc(pmv_ctx* c, BackendBuffers** out) {
    BackendBuffers* b = new BackendBuffers();
    *out = b;
    const size_t nc = (size_t)std::max(c->max_ba_cams, 1), np = (size_t)std::max(c->max_ba_points, 1), no = (size_t)std::max(c->max_ba_obs, 1);
    const size_t n = 6 * nc + 3 * np, m = 6 * nc;
    const size_t ldw = (size_t)round_up((int)m + 1, 16), krows = (size_t)round_up((int)(3 * np), 16);
#define CKB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(c, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
    CKB(hipMalloc(&b->d_cams, nc * 6 * 8)); CKB(hipMalloc(&b->d_pts, np * 3 * 8)); CKB(hipMalloc(&b->d_obs, no * 2 * 8)); CKB(hipMalloc(&b->d_K, 9 * 8));
    CKB(hipMalloc(&b->d_cam_idx, no * 4)); CKB(hipMalloc(&b->d_pt_idx, no * 4));
    CKB(hipMalloc(&b->d_pobs_start, (np + 1) * 4)); CKB(hipMalloc(&b->d_pobs_list, no * 4));
    CKB(hipMalloc(&b->d_cobs_start, (nc + 1) * 4)); CKB(hipMalloc(&b->d_cobs_list, no * 4));
    CKB(hipMalloc(&b->d_x, 2 * n * 8)); CKB(hipMalloc(&b->d_cand, n * 8)); CKB(hipMalloc(&b->d_scale, n * 8)); CKB(hipMalloc(&b->d_diag, n * 8));
    CKB(hipMalloc(&b->d_D2, n * 8)); CKB(hipMalloc(&b->d_step, n * 8));
    CKB(hipMalloc(&b->d_res, 2 * no * 2 * 8)); CKB(hipMalloc(&b->d_J, 2 * no * 18 * 8));   // two buffers: current point / candidate
    CKB(hipMalloc(&b->d_Einv, np * 9 * 8)); CKB(hipMalloc(&b->d_gp, np * 3 * 8));
    b->ydwd_elems = krows * ldw;
    CKB(hipMalloc(&b->d_Yd, 2 * b->ydwd_elems * 8));   // Yt | [Wt | g] adjacent: one clear per solve (multi-kernel LM)
    CKB(hipMalloc(&b->d_Wd, b->ydwd_elems * 8));       // [Wt | g] of the single-workgroup LM
    CKB(hipMalloc(&b->d_S, m * m * 8)); CKB(hipMalloc(&b->d_rhs, m * 8));
    b->gpart_elems = (size_t)8 * ldw * ldw;   // up to 8 K-slices of an (ldw x ldw) tile grid
    CKB(hipMalloc(&b->d_Gpart, b->gpart_elems * 8));
    CKB(hipMalloc(&b->d_summary, 8 * 8));
    CKB(hipMalloc(&b->d_stamps, 32 * 8));
    CKB(hipMalloc(&b->d_bastate, 512 + 4 * BA_MAX_ITERATIONS));
    CKB(hipMalloc(&b->d_bapart, ((size_t)(no + 255) / 256 + 5 * ((size_t)(np + 63) / 64) + 64 * (size_t)nc + 3 * (size_t)no + 64) * 8));
    CKB(hipMemset(b->d_stamps, 0, 32 * 8));
    const size_t mt = (size_t)c->max_tracks;
    b->h_stage_bytes = std::max<size_t>(no * 18 * 8 + no * 2 * 8, std::max<size_t>(n * 8 + no * 32 + (np + nc + 2) * 4, mt * 32 + MAX_HYP * 20 + 4096));
    CKB(hipHostMalloc(&b->h_stage, b->h_stage_bytes));
    CKB(hipMalloc(&b->d_obj, mt * 12)); CKB(hipMalloc(&b->d_img, mt * 8));
    CKB(hipMalloc(&b->d_samples, MAX_HYP * 5 * 4)); CKB(hipMalloc(&b->d_counts, MAX_HYP * 4));
    CKB(hipMalloc(&b->d_inliers, mt * 4)); CKB(hipMalloc(&b->d_info, 16));
    CKB(hipMalloc(&b->d_models, MAX_HYP * 6 * 8)); CKB(hipMalloc(&b->d_rt, 6 * 8)); CKB(hipMalloc(&b->d_Kp, 9 * 8));
    CKB(hipMalloc(&b->d_masks, (size_t)MAX_HYP * mt));
    b->ba_io_bytes = (8 + nc * 6 + np * 3 + no * 2 + 10) * 8 + (no * 8 + np + nc + 8) * 4 + 128;
    CKB(hipMalloc(&b->d_ba_io, b->ba_io_bytes));
    b->pnp_in_bytes = PNP_HDR + mt * 20 + (size_t)MAX_HYP * 20 + 64;
    b->pnp_out_bytes = 48 + 16 + mt * 4 + 64;
    CKB(hipMalloc(&b->d_pnp_in, b->pnp_in_bytes));
    CKB(hipMalloc(&b->d_pnp_out, b->pnp_out_bytes));
    b->tri_in_bytes = 48 * 8 + mt * 32 + mt + 64;
    b->tri_out_bytes = mt * 16 * 8 + mt * 4 + 64;
    CKB(hipMalloc(&b->d_tri_in, b->tri_in_bytes));
    CKB(hipMalloc(&b->d_tri_out, b->tri_out_bytes));
    CKB(hipMalloc(&b->d_fp_work, (size_t)FP_MAX_HYP * (90 * 8 + 4) + 64));
    b->h_stage_bytes = std::max(b->h_stage_bytes, std::max(b->ba_io_bytes, b->pnp_in_bytes + b->pnp_out_bytes));
    b->h_stage_bytes = std::max(b->h_stage_bytes, b->tri_in_bytes + b->tri_out_bytes);
    b->h_stage_bytes = std::max(b->h_stage_bytes, b->tri_in_bytes + (size_t)FP_MAX_HYP * (90 * 8 + 44) + 256);   // five-point round in + out
    (void)hipHostFree(b->h_stage);
    CKB(hipHostMalloc(&b->h_stage, b->h_stage_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    CKB(hipHostGetDevicePointer((void**)&b->d_h_stage, b->h_stage, 0));
    {   // every buffer a kernel may touch exists (a missed allocation must fail here, not as a GPU fault later)
        const void* must[] = {b->d_x, b->d_cand, b->d_scale, b->d_diag, b->d_D2, b->d_step, b->d_res, b->d_J, b->d_Einv, b->d_gp, b->d_Yd, b->d_Wd,
                              b->d_S, b->d_rhs, b->d_Gpart, b->d_summary, b->d_stamps, b->d_bastate, b->d_bapart, b->d_counts, b->d_inliers,
                              b->d_info, b->d_models, b->d_rt, b->d_masks, b->d_ba_io, b->d_pnp_in, b->d_pnp_out, b->d_tri_in, b->d_tri_out,
                              b->h_stage, b->d_h_stage};
        for (const void* p : must)
            if (!p) { set_err(c, "backend_create: internal error, a back-end buffer was not allocated"); return PMV_ERR_HIP; }
    }
#undef CKB
    return PMV_OK;
}
int backend_create(pmv_ctx* c) { return backend_alloc(c, &c->be); }

void backend_free(BackendBuffers* b) {
    if (!b) return;
    void* ptrs[] = {b->d_cams, b->d_pts, b->d_obs, b->d_K, b->d_cam_idx, b->d_pt_idx, b->d_pobs_start, b->d_pobs_list, b->d_cobs_start,
                    b->d_cobs_list, b->d_x, b->d_cand, b->d_scale, b->d_diag, b->d_D2, b->d_step, b->d_res, b->d_J, b->d_Einv, b->d_gp,
                    b->d_Yd, b->d_Wd, b->d_S, b->d_rhs, b->d_Gpart, b->d_summary, b->d_obj, b->d_img, b->d_samples, b->d_counts,
                    b->d_inliers, b->d_info, b->d_models, b->d_rt, b->d_Kp, b->d_masks, b->d_ba_io, b->d_pnp_in, b->d_pnp_out, b->d_bastate, b->d_bapart, b->d_tri_in, b->d_tri_out,
                    b->d_fp_work};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (b->d_stamps) (void)hipFree(b->d_stamps);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    delete b;
}
void backend_destroy(pmv_ctx* c) {
    backend_free(c->be); c->be = nullptr;
    if (c->be_ahead) { backend_free(c->be_ahead); c->be_ahead = nullptr; }
    if (c->s_ahead) { (void)hipStreamSynchronize(c->s_ahead); (void)hipStreamDestroy(c->s_ahead); c->s_ahead = nullptr; }
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

}  // extern "C"

// ---- PnP: prepare (host slab + device problem record) / finish (results out of the pinned block), shared by the single call
// and by the batch engine (several sequences' calls in one launch) ------------------------------------------------------------
int pmv::pnp_check(pmv_ctx* ctx, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec, int iterations,
                   double confidence, int* out_inliers, int* out_n_inliers) {
    REQ(ctx && obj_xyz && img_xy && K && rvec && tvec && out_inliers && out_n_inliers, PMV_ERR_INVALID, "pmv_pnp_ransac: null argument");
    *out_n_inliers = 0;
    REQ(m <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_pnp_ransac: m=%d exceeds max_tracks=%d", m, ctx->max_tracks);
    // cv::solvePnPRansac asserts npoints >= 4; with 4 points it switches to P3P, with 5 it runs the kernel once (not built: the
    // reference only calls it with >= tracked_features_tol points)
    REQ(m >= 6, PMV_ERR_DEGENERATE, "pmv_pnp_ransac: %d correspondences (need >= 6)", m);
    REQ(iterations >= 1 && iterations <= MAX_HYP, PMV_ERR_CAPACITY, "pmv_pnp_ransac: iterations=%d (1..%d)", iterations, MAX_HYP);
    REQ(confidence > 0 && confidence < 1, PMV_ERR_INVALID, "pmv_pnp_ransac: confidence must be in (0,1)");
    return PMV_OK;
}
// packs [K 10 doubles, 10^k for k = -16..16 (CvLevMarq's lambda) | obj 3m floats | img 2m floats | samples 5*iterations ints] into b's
// pinned block and describes the problem with b's device buffers; *in_bytes = bytes to copy h_stage -> d_pnp_in
void pmv::pnp_prepare(BackendBuffers* b, const float* obj_xyz, const float* img_xy, int m, const double* K, int iterations, float reproj_err,
                      double confidence, PnPProblem* P, size_t* in_bytes_out) {
    char* hs = (char*)b->h_stage;
    double* h_K = (double*)hs;
    float* h_obj = (float*)(hs + PNP_HDR);
    float* h_img = h_obj + (size_t)3 * m;
    int* h_samples = (int*)(h_img + (size_t)2 * m);
    const size_t in_bytes = PNP_HDR + (size_t)m * 20 + (size_t)iterations * 20;
    memcpy(h_K, K, 72);
    for (int k = -16; k <= 16; k++) h_K[10 + 16 + k] = std::exp(k * std::log(10.));   // as CvLevMarq::step evaluates it (host libm)
    memcpy(h_obj, obj_xyz, (size_t)m * 12);
    memcpy(h_img, img_xy, (size_t)m * 8);
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
    char* ho = hs + ((in_bytes + 63) & ~(size_t)63);
    P->K = (const double*)b->d_pnp_in;
    P->obj = (const float*)(b->d_pnp_in + PNP_HDR);
    P->img = P->obj + (size_t)3 * m;
    P->samples = (const int*)(P->img + (size_t)2 * m);
    P->models = b->d_models; P->masks = b->d_masks; P->counts = b->d_counts;
    P->rt_out = (double*)b->d_pnp_out;
    P->info = (int*)(b->d_pnp_out + 48);
    P->inliers = (int*)(b->d_pnp_out + 64);
    P->host_out = b->d_h_stage + (ho - hs);   // the refit kernel writes [rt 48 B | info 16 B | inliers] straight into the pinned block
    P->m = m; P->n_hyp = iterations;
    P->thr = (float)((double)reproj_err * (double)reproj_err);
    P->confidence = confidence;
    *in_bytes_out = in_bytes;
}
void pmv::pnp_finish(pmv_ctx* ctx, BackendBuffers* b, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec,
                     int iterations, float reproj_err, double confidence, size_t in_bytes, int* out_inliers, int* out_n_inliers) {
    const char* ho = (const char*)b->h_stage + ((in_bytes + 63) & ~(size_t)63);
    const double* h_rt = (const double*)ho;
    const int* h_info = (const int*)(ho + 48);
    const int* h_inl = (const int*)(ho + 64);
    const int n = h_info[0];
    if (ctx->log.on) {   // [kind 0, m, iterations, n_inliers | obj 3m f32 | img 2m f32 | K 9, rvec_in 3, tvec_in 3, reproj_err, confidence f64 | rvec_out 3, tvec_out 3 f64 | inliers]
        std::lock_guard<std::mutex> lk(ctx->log.mu);
        ctx->log.blobs.emplace_back();
        std::vector<char>& bl = ctx->log.blobs.back();
        const int hdr[4] = {0, m, iterations, n};
        const double opt[2] = {(double)reproj_err, confidence};
        pmv_call_log::put(bl, hdr, 16); pmv_call_log::put(bl, obj_xyz, (size_t)m * 12); pmv_call_log::put(bl, img_xy, (size_t)m * 8);
        pmv_call_log::put(bl, K, 72); pmv_call_log::put(bl, rvec, 24); pmv_call_log::put(bl, tvec, 24); pmv_call_log::put(bl, opt, 16);
        pmv_call_log::put(bl, h_rt, 48); pmv_call_log::put(bl, h_inl, (size_t)(n > 0 ? n : 0) * 4);
    }
    for (int i = 0; i < 3; i++) { rvec[i] = h_rt[i]; tvec[i] = h_rt[3 + i]; }
    *out_n_inliers = n;
    if (n > 0) memcpy(out_inliers, h_inl, (size_t)n * 4);
}

extern "C" {

static bool stamps_on() { static const bool on = getenv("PMV_BA_STAMPS") != nullptr; return on; }

int pmv_pnp_ransac(pmv_ctx* ctx, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec,
                   int iterations, float reproj_err, double confidence, int* out_inliers, int* out_n_inliers) {
    int rc = pnp_check(ctx, obj_xyz, img_xy, m, K, rvec, tvec, iterations, confidence, out_inliers, out_n_inliers);
    if (rc) return rc;
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    static const char* const dump = getenv("PMV_DUMP_PNP");   // (the diagnostic switches are read once per process: getenv walks the whole environment)
    if (dump) {   // debug: append the inputs of every call to a file
        if (FILE* f = fopen(dump, "ab")) {
            fwrite(&m, 4, 1, f); fwrite(obj_xyz, 12, m, f); fwrite(img_xy, 8, m, f); fwrite(K, 8, 9, f); fwrite(rvec, 8, 3, f); fwrite(tvec, 8, 3, f);
            fclose(f);
        }
    }
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    PnPProblem P;
    size_t in_bytes = 0;
    pnp_prepare(b, obj_xyz, img_xy, m, K, iterations, reproj_err, confidence, &P, &in_bytes);
    static const bool stage_by_kernel = !(getenv("PMV_BA_STAGE") && !strcmp(getenv("PMV_BA_STAGE"), "dma"));
    if (stage_by_kernel) {
        const unsigned n16 = (unsigned)((in_bytes + 15) >> 4);
        hipLaunchKernelGGL(k_stage_block, dim3(std::min(8u, (n16 + 255u) / 256u)), dim3(256), 0, s, (const uint4*)b->d_h_stage, (uint4*)b->d_pnp_in, n16);
        CKC(hipGetLastError());
    } else CKC(hipMemcpyAsync(b->d_pnp_in, b->h_stage, in_bytes, hipMemcpyHostToDevice, s));
    // the refit kernel writes [rt | info | inliers] straight into the pinned block and, last, this call's sequence number into the
    // block's fourth info word: the result is read as soon as that store lands (PMV_BACK_WAIT=sync: hipStreamSynchronize instead)
    static const bool flag_wait = !(getenv("PMV_BACK_WAIT") && !strcmp(getenv("PMV_BACK_WAIT"), "sync"));
    volatile unsigned* done_word = (volatile unsigned*)((char*)b->h_stage + ((in_bytes + 63) & ~(size_t)63) + 60);
    if (++b->done_seq == 0) b->done_seq = 1;
    *done_word = 0;
    CKC(launch_pnp(s, P.obj, P.img, m, P.K, P.samples, iterations, P.thr, confidence, P.models, P.masks, P.counts,
                   P.rt_out, P.inliers, P.info, P.host_out, stamps_on() ? b->d_stamps : nullptr, flag_wait ? b->done_seq : 0u));
    if (flag_wait) {
        (void)hipStreamQuery(s);   // lets the runtime retire the commands of earlier calls now, while the GPU works on this one
        for (unsigned spins = 1;; spins++) {
            if (__atomic_load_n(done_word, __ATOMIC_ACQUIRE) == b->done_seq) break;
            if ((spins & 0xffffu) == 0) {   // a faulted launch never signals: ask the runtime now and then
                const hipError_t e = hipStreamQuery(s);
                if (e != hipSuccess && e != hipErrorNotReady) CKC(e);
            }
            __builtin_ia32_pause();
        }
    } else CKC(hipStreamSynchronize(s));
    pnp_finish(ctx, b, obj_xyz, img_xy, m, K, rvec, tvec, iterations, reproj_err, confidence, in_bytes, out_inliers, out_n_inliers);
    return PMV_OK;
}

// ---- call log (teacher-forced replay of a pipeline run's back-end calls through another implementation) ----------------
int pmv_record_enable(pmv_ctx* ctx, int on) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    if (on) ctx->log.blobs.clear();
    ctx->log.on = on != 0;
    return PMV_OK;
}
int pmv_record_count(pmv_ctx* ctx) { return ctx ? (int)ctx->log.blobs.size() : PMV_ERR_INVALID; }
long long pmv_record_size(pmv_ctx* ctx, int i) {
    if (!ctx || i < 0 || i >= (int)ctx->log.blobs.size()) return PMV_ERR_INVALID;
    return (long long)ctx->log.blobs[i].size();
}
int pmv_record_get(pmv_ctx* ctx, int i, void* out, long long capacity) {
    REQ(ctx && out && i >= 0 && i < (int)ctx->log.blobs.size(), PMV_ERR_INVALID, "pmv_record_get: bad argument");
    REQ(capacity >= (long long)ctx->log.blobs[i].size(), PMV_ERR_CAPACITY, "pmv_record_get: buffer too small");
    memcpy(out, ctx->log.blobs[i].data(), ctx->log.blobs[i].size());
    return PMV_OK;
}

// diagnostic: accumulated per-phase shader-clock counters of k_ba_lm (enabled by PMV_BA_STAMPS=1), 32 values
int pmv_debug_ba_stamps(pmv_ctx* ctx, unsigned long long* out32) {
    REQ(ctx && out32, PMV_ERR_INVALID, "null argument");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_back));
    CKC(hipMemcpy(out32, ctx->be->d_stamps, 256, hipMemcpyDeviceToHost));
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
    tl_prof = &ctx->prof;
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

}  // extern "C"

// ---- BA: check / prepare (pinned io block, observation lists, BAArgs over b's workspaces) / finish -------------------------------
int pmv::ba_check(pmv_ctx* ctx, const double* cams, int nc, const double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx,
                  int n_obs, const double* K, double huber_delta, int max_iterations) {
    REQ(ctx && cams && pts && obs_xy && cam_idx && pt_idx && K, PMV_ERR_INVALID, "pmv_ba_solve: null argument");
    REQ(nc >= 1 && nc <= ctx->max_ba_cams && np >= 1 && np <= ctx->max_ba_points && n_obs >= 1 && n_obs <= ctx->max_ba_obs, PMV_ERR_CAPACITY,
        "pmv_ba_solve: nc=%d np=%d n_obs=%d exceed capacity %d/%d/%d", nc, np, n_obs, ctx->max_ba_cams, ctx->max_ba_points, ctx->max_ba_obs);
    REQ(max_iterations >= 0 && max_iterations <= BA_MAX_ITERATIONS && huber_delta > 0, PMV_ERR_INVALID, "pmv_ba_solve: bad options");
    for (int i = 0; i < n_obs; i++)
        REQ(cam_idx[i] >= 0 && cam_idx[i] < nc && pt_idx[i] >= 0 && pt_idx[i] < np, PMV_ERR_INVALID, "pmv_ba_solve: index out of range at observation %d", i);
    const int m = 6 * nc;
    REQ(((size_t)(m + 1) * m + (size_t)m) * 8 <= 150 * 1024, PMV_ERR_CAPACITY, "pmv_ba_solve: %d cameras exceed the LDS-resident reduced system (max 22)", nc);
    return PMV_OK;
}
int pmv::ba_prepare(pmv_ctx* ctx, BackendBuffers* b, const double* cams, int nc, const double* pts, int np, const double* obs_xy, const int* cam_idx,
                    const int* pt_idx, int n_obs, const double* K, double huber_delta, int max_iterations, bool multi, BAArgs* Aout, size_t* io_bytes_out) {
    // one pinned block mirrors the device block: [summary 8 | cams | pts | obs | K 10 | cam_idx | pt_idx | pstart | plist | cstart | clist]
    char* hs = (char*)b->h_stage;
    double* h_sum = (double*)hs;
    double* h_cams = h_sum + 8;
    double* h_pts = h_cams + (size_t)nc * 6;
    double* h_obs = h_pts + (size_t)np * 3;
    double* h_K = h_obs + (size_t)n_obs * 2;
    int* h_ci = (int*)(h_K + 10);
    int* h_pi = h_ci + n_obs;
    int* pstart = h_pi + n_obs; int* plist = pstart + (np + 1); int* cstart = plist + n_obs; int* clist = cstart + (nc + 1);
    int* odup = (int*)(((uintptr_t)(clist + n_obs) + 15) & ~(uintptr_t)15);   // 16-byte records (observation, camera, point, dup flag)
    const size_t io_bytes = (size_t)((char*)(odup + (size_t)4 * n_obs) - hs);
    REQ(io_bytes <= b->ba_io_bytes, PMV_ERR_CAPACITY, "pmv_ba_solve: io block too small");
    memcpy(h_cams, cams, (size_t)nc * 48); memcpy(h_pts, pts, (size_t)np * 24); memcpy(h_obs, obs_xy, (size_t)n_obs * 16);
    memcpy(h_K, K, 72); memcpy(h_ci, cam_idx, (size_t)n_obs * 4); memcpy(h_pi, pt_idx, (size_t)n_obs * 4);
    // observation lists per point / per camera (counting sort, observation order preserved)
    std::fill(pstart, pstart + np + 1, 0);
    std::fill(cstart, cstart + nc + 1, 0);
    for (int i = 0; i < n_obs; i++) { pstart[pt_idx[i] + 1]++; cstart[cam_idx[i] + 1]++; }
    for (int p = 0; p < np; p++) pstart[p + 1] += pstart[p];
    for (int c = 0; c < nc; c++) cstart[c + 1] += cstart[c];
    {
        std::vector<int> pf(pstart, pstart + np), cf(cstart, cstart + nc);
        for (int i = 0; i < n_obs; i++) { plist[pf[pt_idx[i]]++] = i; clist[cf[cam_idx[i]]++] = i; }
    }
    // a point seen twice by the same camera (the reference's feat_corr duplicates): the first entry carries the group
    for (int p = 0; p < np; p++) {
        for (int e = pstart[p]; e < pstart[p + 1]; e++) {
            const int c = cam_idx[plist[e]];
            bool earlier = false, later = false;
            for (int e2 = pstart[p]; e2 < e; e2++) earlier = earlier || cam_idx[plist[e2]] == c;
            for (int e2 = e + 1; e2 < pstart[p + 1]; e2++) later = later || cam_idx[plist[e2]] == c;
            odup[4 * e] = plist[e]; odup[4 * e + 1] = c; odup[4 * e + 2] = p; odup[4 * e + 3] = earlier ? 2 : (later ? 1 : 0);
        }
    }
    const int m = 6 * nc;
    const int tiles_r = (m + 15) / 16, tiles_c = (m + 1 + 15) / 16;
    char* dio = b->d_ba_io;
    double* d_sum = (double*)dio;
    double* d_cams = d_sum + 8;
    double* d_pts = d_cams + (size_t)nc * 6;
    double* d_obs = d_pts + (size_t)np * 3;
    double* d_K = d_obs + (size_t)n_obs * 2;
    int* d_ci = (int*)(d_K + 10);
    int* d_pi = d_ci + n_obs;
    int* d_pstart = d_pi + n_obs; int* d_plist = d_pstart + (np + 1); int* d_cstart = d_plist + n_obs; int* d_clist = d_cstart + (nc + 1);
    const int4* d_erec = (const int4*)(dio + ((char*)odup - hs));
    BAArgs A;
    A.cams = d_cams; A.pts = d_pts; A.obs = d_obs; A.cam_idx = d_ci; A.pt_idx = d_pi; A.K = d_K;
    A.pobs_start = d_pstart; A.pobs_list = d_plist; A.cobs_start = d_cstart; A.cobs_list = d_clist; A.erec = d_erec;
    A.nc = nc; A.np = np; A.nobs = n_obs; A.max_iterations = max_iterations; A.huber = huber_delta;
    A.x = b->d_x; A.cand = b->d_cand; A.scale = b->d_scale; A.diag = b->d_diag; A.D2 = b->d_D2; A.step = b->d_step; A.res = b->d_res; A.J = b->d_J;
    A.Einv = b->d_Einv; A.gp = b->d_gp; A.Yd = b->d_Yd; A.Wd = b->d_Wd; A.S = b->d_S; A.rhs = b->d_rhs; A.Gpart = b->d_Gpart; A.summary = d_sum;
    A.stamps = stamps_on() ? b->d_stamps : nullptr;
    A.out = nullptr;
    A.done_seq = 0;
    A.tiles_r = tiles_r; A.tiles_c = tiles_c;
    A.ldw = A.tiles_c * 16;
    A.krows = round_up(3 * np, 16);
    A.gp_rows = A.tiles_r * 16;
    int ks = 8 / (A.tiles_r * A.tiles_c);
    if (ks < 1) ks = 1;
    if (ks > 8) ks = 8;
    if (multi) ks = 8;   // multi-kernel paths: one wavefront per (tile, K-slice) anywhere on the chip
    A.kper = round_up((A.krows + ks - 1) / ks, 16);
    A.kslices = (A.krows + A.kper - 1) / A.kper;
    REQ((size_t)A.krows * A.ldw <= b->ydwd_elems && (size_t)A.kslices * A.gp_rows * A.ldw <= b->gpart_elems, PMV_ERR_CAPACITY, "pmv_ba_solve: workspace too small");
    if (multi) {
        A.Wd = A.Yd + (size_t)A.krows * A.ldw;
        A.out = (double*)b->d_h_stage;   // [summary 8 | cams | pts] of the result, straight into the pinned block
    }
    *Aout = A;
    *io_bytes_out = io_bytes;
    return PMV_OK;
}
void pmv::ba_finish(pmv_ctx* ctx, BackendBuffers* b, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx,
                    const int* pt_idx, int n_obs, const double* K, double huber_delta, int max_iterations, pmv_ba_summary* summary) {
    const double* h_out = (const double*)b->h_stage;
    static const bool ba_trace = getenv("PMV_BA_TRACE") != nullptr;
    if (ba_trace) {   // diagnostic: first / last camera before and after the solve
        fprintf(stderr, "[ba-trace] nc=%d np=%d nobs=%d cost %.15g -> %.15g it %d ok %d\n", nc, np, n_obs, h_out[0], h_out[1], (int)h_out[2], (int)h_out[3]);
        for (int c : {0, nc - 1}) {
            fprintf(stderr, "[ba-trace]   cam %d in ", c);
            for (int k = 0; k < 6; k++) fprintf(stderr, " %.15g", cams[6 * c + k]);
            fprintf(stderr, "\n[ba-trace]   cam %d out", c);
            for (int k = 0; k < 6; k++) fprintf(stderr, " %.15g", h_out[8 + 6 * c + k]);
            fprintf(stderr, "\n");
        }
    }
    if (ctx->log.on) {   // [kind 1, nc, np, n_obs, max_iterations | cams_in 6nc, pts_in 3np, obs 2n_obs, K 9, huber f64 | cam_idx, pt_idx i32 | cams_out, pts_out, summary 5 f64]
        std::lock_guard<std::mutex> lk(ctx->log.mu);
        ctx->log.blobs.emplace_back();
        std::vector<char>& bl = ctx->log.blobs.back();
        const int hdr[5] = {1, nc, np, n_obs, max_iterations};
        pmv_call_log::put(bl, hdr, 20); pmv_call_log::put(bl, cams, (size_t)nc * 48); pmv_call_log::put(bl, pts, (size_t)np * 24);
        pmv_call_log::put(bl, obs_xy, (size_t)n_obs * 16); pmv_call_log::put(bl, K, 72); pmv_call_log::put(bl, &huber_delta, 8);
        pmv_call_log::put(bl, cam_idx, (size_t)n_obs * 4); pmv_call_log::put(bl, pt_idx, (size_t)n_obs * 4);
        pmv_call_log::put(bl, h_out + 8, ((size_t)nc * 6 + (size_t)np * 3) * 8); pmv_call_log::put(bl, h_out, 40);
    }
    memcpy(cams, h_out + 8, (size_t)nc * 48);
    memcpy(pts, h_out + 8 + (size_t)nc * 6, (size_t)np * 24);
    if (summary) {
        summary->initial_cost = h_out[0]; summary->final_cost = h_out[1]; summary->iterations = (int)h_out[2];
        summary->successful_steps = (int)h_out[3]; summary->termination = (int)h_out[4];
    }
}

extern "C" {

int pmv_set_ba_mode(pmv_ctx* ctx, int mode) {
    if (!ctx || (mode != 0 && mode != 1)) { set_err(ctx, "pmv_set_ba_mode: mode must be 0 (launch chain) or 1 (one workgroup per solve)"); return PMV_ERR_INVALID; }
    ctx->ba_mode = mode;
    return PMV_OK;
}

int pmv_ba_solve(pmv_ctx* ctx, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx,
                 int n_obs, const double* K, double huber_delta, int max_iterations, pmv_ba_summary* summary) {
    int rc = ba_check(ctx, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber_delta, max_iterations);
    if (rc) return rc;
    if (max_iterations == 0) {   // ceres::Solve with max_num_iterations = 0 leaves the parameters untouched (costs are not evaluated here)
        if (summary) { summary->initial_cost = summary->final_cost = 0.0; summary->iterations = 0; summary->successful_steps = 0; summary->termination = 0; }
        return PMV_OK;
    }
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    // launch mode: multi (one launch per LM phase, default) | single (one persistent workgroup); PMV_BA_MODE overrides
    static const int mode = [] { const char* e = getenv("PMV_BA_MODE"); if (getenv("PMV_BA_SINGLE")) return 0;
                                 return (e && !strcmp(e, "single")) ? 0 : 1; }();
    const bool single = mode == 0 || ctx->ba_mode == 1;
    BAArgs A;
    size_t io_bytes = 0;
    rc = ba_prepare(ctx, b, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber_delta, max_iterations, !single, &A, &io_bytes);
    if (rc) return rc;
    char* hs = (char*)b->h_stage;
    static const bool check = getenv("PMV_BA_CHECK") != nullptr;
    // multi-kernel chain: the finish kernel writes the result into hs and, last, this call's sequence number into the low word of summary
    // slot 7; the result is read as soon as that store lands (PMV_BACK_WAIT=sync: hipStreamSynchronize instead; see pmv_pnp_ransac)
    static const bool flag_wait_env = !(getenv("PMV_BACK_WAIT") && !strcmp(getenv("PMV_BACK_WAIT"), "sync"));
    const bool flag_wait = flag_wait_env && !single && !check;
    volatile unsigned* done_word = (volatile unsigned*)((double*)hs + 7);
    if (flag_wait) {
        if (++b->done_seq == 0) b->done_seq = 1;
        ((double*)hs)[7] = 0.0;
        A.done_seq = b->done_seq;
    }
    static const bool stage_by_kernel = !(getenv("PMV_BA_STAGE") && !strcmp(getenv("PMV_BA_STAGE"), "dma"));
    if (stage_by_kernel && !check) {   // (h_stage and d_ba_io are 16-byte aligned and have >= 16 B of slack)
        const unsigned n16 = (unsigned)((io_bytes + 15) >> 4);
        hipLaunchKernelGGL(k_stage_block, dim3(std::min(64u, (n16 + 255u) / 256u)), dim3(256), 0, s, (const uint4*)b->d_h_stage, (uint4*)b->d_ba_io, n16);
        CKC(hipGetLastError());
    } else CKC(hipMemcpyAsync(b->d_ba_io, hs, io_bytes, hipMemcpyHostToDevice, s));
    std::vector<char> saved;
    if (check) saved.assign(hs, hs + io_bytes);
    if (single) CKC(launch_ba_lm(s, A));
    else CKC(launch_ba_multi(s, A, b->d_bastate, b->d_bapart));
    const size_t out_bytes = (8 + (size_t)nc * 6 + (size_t)np * 3) * 8;
    if (single) CKC(hipMemcpyAsync(hs, b->d_ba_io, out_bytes, hipMemcpyDeviceToHost, s));   // (multi: the finish kernel wrote into hs)
    if (flag_wait) {
        (void)hipStreamQuery(s);   // lets the runtime retire the commands of earlier calls while the GPU works on this one
        for (unsigned spins = 1;; spins++) {
            if (__atomic_load_n(done_word, __ATOMIC_ACQUIRE) == b->done_seq) break;
            if ((spins & 0xffffu) == 0) {   // a faulted launch never signals: ask the runtime now and then
                const hipError_t e = hipStreamQuery(s);
                if (e != hipSuccess && e != hipErrorNotReady) CKC(e);
            }
            __builtin_ia32_pause();
        }
    } else CKC(hipStreamSynchronize(s));
    const double* h_out = (const double*)hs;
    if (!single && check) {   // diagnostic: the same problem through the one-workgroup kernel, differences to stderr
        std::vector<double> got(h_out, h_out + out_bytes / 8);
        CKC(hipMemcpyAsync(b->d_ba_io, saved.data(), io_bytes, hipMemcpyHostToDevice, s));
        A.Wd = b->d_Wd;
        CKC(launch_ba_lm(s, A));
        CKC(hipMemcpyAsync(hs, b->d_ba_io, out_bytes, hipMemcpyDeviceToHost, s));
        CKC(hipStreamSynchronize(s));
        double dmax = 0;
        for (size_t i = 8; i < out_bytes / 8; i++) dmax = std::max(dmax, fabs(got[i] - h_out[i]));
        fprintf(stderr, "[ba-check] nc=%d np=%d nobs=%d multi: cost %.12g -> %.12g it %d ok %d term %d | single: %.12g -> %.12g it %d ok %d term %d | max|dx| %.3e\n",
                nc, np, n_obs, got[0], got[1], (int)got[2], (int)got[3], (int)got[4], h_out[0], h_out[1], (int)h_out[2], (int)h_out[3],
                (int)h_out[4], dmax);
        {   // sensitivity: the same problem with the cameras perturbed by 1e-9 (how much does one solve amplify?)
            std::vector<double> base(h_out, h_out + out_bytes / 8);
            std::vector<char> pert(saved);
            double* pc = (double*)pert.data() + 8;
            for (int i = 0; i < 6 * nc; i++) pc[i] += 1e-9 * ((i * 2654435761u >> 7) % 3 - 1.0);
            CKC(hipMemcpyAsync(b->d_ba_io, pert.data(), io_bytes, hipMemcpyHostToDevice, s));
            CKC(launch_ba_lm(s, A));
            CKC(hipMemcpyAsync(hs, b->d_ba_io, out_bytes, hipMemcpyDeviceToHost, s));
            CKC(hipStreamSynchronize(s));
            double d2 = 0;
            for (size_t i = 8; i < out_bytes / 8; i++) d2 = std::max(d2, fabs(base[i] - h_out[i]));
            fprintf(stderr, "[ba-check]    input perturbed by 1e-9 -> single-kernel output moves by %.3e\n", d2);
        }
        memcpy(hs, got.data(), out_bytes);
    }
    ba_finish(ctx, b, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber_delta, max_iterations, summary);
    return PMV_OK;
}

}  // extern "C"

// ---- five-point RANSAC round: prepare / finish ----------------------------------------------------------------------------------
int pmv::fivepoint_prepare(pmv_ctx* ctx, BackendBuffers* b, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr,
                           FivePointProblem* P, size_t* in_bytes_out) {
    REQ(q1 && q2 && samples && n >= 5 && n <= ctx->max_tracks && n_hyp >= 1 && n_hyp <= FP_MAX_HYP, PMV_ERR_INVALID,
        "five-point round: n=%d (5..max_tracks=%d), n_hyp=%d (1..%d)", n, ctx->max_tracks, n_hyp, FP_MAX_HYP);
    char* hs = (char*)b->h_stage;
    // pinned in-block [q1 2n | q2 2n doubles | samples 5 n_hyp ints], out-block [models 90 n_hyp doubles | n_models n_hyp | counts 10 n_hyp ints]
    double* h_q1 = (double*)hs;
    double* h_q2 = h_q1 + (size_t)2 * n;
    int* h_s = (int*)(h_q2 + (size_t)2 * n);
    const size_t in_bytes = (size_t)4 * n * 8 + (size_t)5 * n_hyp * 4;
    REQ(in_bytes <= b->tri_in_bytes, PMV_ERR_CAPACITY, "five-point round: input block too small");
    memcpy(h_q1, q1, (size_t)n * 16); memcpy(h_q2, q2, (size_t)n * 16); memcpy(h_s, samples, (size_t)5 * n_hyp * 4);
    P->q1 = (const double*)b->d_tri_in;
    P->q2 = P->q1 + (size_t)2 * n;
    P->samples = (const int*)(P->q2 + (size_t)2 * n);
    P->models_d = (double*)b->d_fp_work;
    P->n_models_d = (int*)(b->d_fp_work + (size_t)FP_MAX_HYP * 90 * 8);
    char* ho = hs + ((in_bytes + 63) & ~(size_t)63);
    char* dho = b->d_h_stage + (ho - hs);
    P->models_h = (double*)dho;
    P->n_models_h = (int*)(dho + (size_t)n_hyp * 90 * 8);
    P->counts_h = P->n_models_h + n_hyp;
    P->n = n; P->n_hyp = n_hyp; P->thr = thr;
    // hypotheses whose sample is degenerate write no models: clear the counts the host will read
    memset(ho + (size_t)n_hyp * 90 * 8, 0, (size_t)n_hyp * 44);
    *in_bytes_out = in_bytes;
    return PMV_OK;
}
void pmv::fivepoint_finish(BackendBuffers* b, int n_hyp, size_t in_bytes, double* models, int* n_models, int* counts) {
    const char* ho = (const char*)b->h_stage + ((in_bytes + 63) & ~(size_t)63);
    memcpy(models, ho, (size_t)n_hyp * 90 * 8);
    memcpy(n_models, ho + (size_t)n_hyp * 90 * 8, (size_t)n_hyp * 4);
    memcpy(counts, ho + (size_t)n_hyp * 90 * 8 + (size_t)n_hyp * 4, (size_t)n_hyp * 40);
}

// ---- two-view DLT: prepare / finish -------------------------------------------------------------------------------------------
void pmv::dlt_prepare(BackendBuffers* b, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, DltProblem* P,
                      size_t* in_bytes_out) {
    char* hs = (char*)b->h_stage;
    // pinned in-block [P1x4 48 | q1 2n | q2 2n | mask n bytes], out-block [Q 16n doubles | mask 4n bytes]
    double* h_P = (double*)hs;
    double* h_q1 = h_P + 48;
    double* h_q2 = h_q1 + (size_t)2 * n;
    uint8_t* h_m = (uint8_t*)(h_q2 + (size_t)2 * n);
    const size_t in_bytes = (48 + (size_t)4 * n) * 8 + (size_t)n;
    memcpy(h_P, P1x4, 48 * 8); memcpy(h_q1, q1, (size_t)n * 16); memcpy(h_q2, q2, (size_t)n * 16); memcpy(h_m, mask_in, (size_t)n);
    P->P1x4 = (const double*)b->d_tri_in;
    P->q1 = P->P1x4 + 48;
    P->q2 = P->q1 + (size_t)2 * n;
    P->mask_in = (const uint8_t*)(P->q2 + (size_t)2 * n);
    char* ho = hs + ((in_bytes + 63) & ~(size_t)63);
    P->Q = (double*)(b->d_h_stage + (ho - hs));   // results go straight into the pinned block (coalesced 8-byte stores)
    P->mask = (uint8_t*)(P->Q + (size_t)16 * n);
    P->n = n;
    *in_bytes_out = in_bytes;
}
void pmv::dlt_finish(pmv_ctx* ctx, BackendBuffers* b, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in,
                     size_t in_bytes, double* out_Q, uint8_t* out_mask, int* out_good) {
    const char* ho = (const char*)b->h_stage + ((in_bytes + 63) & ~(size_t)63);
    memcpy(out_Q, ho, (size_t)16 * n * 8);
    memcpy(out_mask, ho + (size_t)16 * n * 8, (size_t)4 * n);
    for (int c = 0; c < 4; c++) {
        int g = 0;
        for (int i = 0; i < n; i++) g += out_mask[(size_t)c * n + i];
        out_good[c] = g;
    }
    if (ctx->log.on) {   // [kind 2, n | q1 2n, q2 2n, P 48 f64 | mask_in n u8 | Q 16n f64 | mask 4n u8 | good 4 i32]
        std::lock_guard<std::mutex> lk(ctx->log.mu);
        ctx->log.blobs.emplace_back();
        std::vector<char>& bl = ctx->log.blobs.back();
        const int hdr[2] = {2, n};
        pmv_call_log::put(bl, hdr, 8); pmv_call_log::put(bl, q1, (size_t)n * 16); pmv_call_log::put(bl, q2, (size_t)n * 16); pmv_call_log::put(bl, P1x4, 384);
        pmv_call_log::put(bl, mask_in, (size_t)n); pmv_call_log::put(bl, out_Q, (size_t)n * 128); pmv_call_log::put(bl, out_mask, (size_t)n * 4);
        pmv_call_log::put(bl, out_good, 16);
    }
}

extern "C" {

// one round of cv::findEssentialMat's RANSAC (OpenCVFivePointTri.cpp:24) on the device: see include/pmv_hip.h
int pmv_fivepoint_hypotheses(pmv_ctx* ctx, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models,
                             int* n_models, int* counts) {
    REQ(ctx && models && n_models && counts, PMV_ERR_INVALID, "pmv_fivepoint_hypotheses: null argument");
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    BackendBuffers* b = ctx->be;
    hipStream_t s = ctx->s_back;
    FivePointProblem P;
    size_t in_bytes = 0;
    int rc = fivepoint_prepare(ctx, b, q1, q2, n, samples, n_hyp, thr, &P, &in_bytes);
    if (rc) return rc;
    CKC(hipMemcpyAsync(b->d_tri_in, b->h_stage, in_bytes, hipMemcpyHostToDevice, s));
    CKC(hipMemcpyAsync(b->d_tri_out, &P, sizeof(P), hipMemcpyHostToDevice, s));   // the problem record (d_tri_out is free during a five-point round)
    CKC(launch_fivepoint_batch(s, (const FivePointProblem*)b->d_tri_out, 1, n_hyp));
    CKC(hipStreamSynchronize(s));
    fivepoint_finish(b, n_hyp, in_bytes, models, n_models, counts);
    return PMV_OK;
}

// the per-point part of cv::recoverPose (OpenCVFivePointTri.cpp:27): DLT triangulation + cheirality for the four candidates
static int triangulate_on(pmv_ctx* ctx, BackendBuffers* b, hipStream_t s, const double* q1, const double* q2, int n, const double* P1x4,
                          const uint8_t* mask_in, double* out_Q, uint8_t* out_mask, int* out_good) {
    DltProblem P;
    size_t in_bytes = 0;
    dlt_prepare(b, q1, q2, n, P1x4, mask_in, &P, &in_bytes);
    CKC(hipMemcpyAsync(b->d_tri_in, b->h_stage, in_bytes, hipMemcpyHostToDevice, s));
    CKC(launch_tri_dlt(s, P.P1x4, P.q1, P.q2, P.mask_in, n, P.Q, P.mask));
    CKC(hipStreamSynchronize(s));
    dlt_finish(ctx, b, q1, q2, n, P1x4, mask_in, in_bytes, out_Q, out_mask, out_good);
    return PMV_OK;
}

int pmv_triangulate_candidates(pmv_ctx* ctx, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in,
                               double* out_Q, uint8_t* out_mask, int* out_good) {
    REQ(ctx && q1 && q2 && P1x4 && mask_in && out_Q && out_mask && out_good, PMV_ERR_INVALID, "pmv_triangulate_candidates: null argument");
    REQ(n >= 1 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_triangulate_candidates: n=%d (1..max_tracks=%d)", n, ctx->max_tracks);
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    return triangulate_on(ctx, ctx->be, ctx->s_back, q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good);
}

int pmv_triangulate_candidates_ahead(pmv_ctx* ctx, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in,
                                     double* out_Q, uint8_t* out_mask, int* out_good) {
    REQ(ctx && q1 && q2 && P1x4 && mask_in && out_Q && out_mask && out_good, PMV_ERR_INVALID, "pmv_triangulate_candidates_ahead: null argument");
    REQ(n >= 1 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_triangulate_candidates_ahead: n=%d (1..max_tracks=%d)", n, ctx->max_tracks);
    std::lock_guard<std::mutex> lk(ctx->ahead_mu);   // one call at a time on the auxiliary lane (its callers are helper threads)
    CKC(hipSetDevice(ctx->device));
    if (!ctx->be_ahead) {
        BackendBuffers* b = nullptr;
        const int rc = backend_alloc(ctx, &b);
        if (rc != PMV_OK) { if (b) backend_free(b); return rc; }
        ctx->be_ahead = b;
        CKC(hipStreamCreateWithFlags(&ctx->s_ahead, hipStreamNonBlocking));
    }
    return triangulate_on(ctx, ctx->be_ahead, ctx->s_ahead, q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good);
}

}  // extern "C"
