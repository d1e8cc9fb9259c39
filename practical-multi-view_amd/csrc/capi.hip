// C-ABI implementation (include/pmv_hip.h): context, frame slots, launches, D2H staging.
#include "pmv_ctx.h"
#include <algorithm>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

using namespace pmv;

static thread_local char g_create_err[512] = "";
thread_local pmv::Profiler* pmv::tl_prof = nullptr;

// The message goes to the calling thread's own buffer first (what ck() of hip_pipeline.hip and the batch engine's sequence threads
// report: a failing sequence must not pick up another sequence's text) and then, under a mutex, to the context's buffer that
// pmv_last_error() hands to the caller of the entry point.
static thread_local char tl_err[512] = "";
const char* pmv::thread_error() { return tl_err; }
void pmv::set_err(pmv_ctx* c, const char* fmt, ...) {
    char tmp[512];   // (the arguments may point into tl_err or c->err themselves)
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    memcpy(tl_err, tmp, sizeof(tmp));
    if (!c) { memcpy(g_create_err, tmp, sizeof(tmp)); return; }
    std::lock_guard<std::mutex> lk(c->err_mu);
    memcpy(c->err, tmp, sizeof(tmp));
}

PyrLayout pmv::make_layout(int w, int h) {
    PyrLayout L;
    memset(&L, 0, sizeof(L));
    // cv::buildOpticalFlowPyramid(img, pyr, Size(32,32), maxLevel=4): stop when the next level would be <= winSize
    int lw = w, lh = h, n = 0;
    uint32_t off = 0;
    for (int level = 0; level <= 4; level++) {
        L.w[level] = lw; L.h[level] = lh;
        L.stride[level] = (lw + 2 * PAD + 63) & ~63;
        L.off[level] = off;
        off += (uint32_t)L.stride[level] * (uint32_t)(lh + 2 * PAD);
        n = level + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= LK_WIN || lh <= LK_WIN) break;
    }
    L.n_levels = n;
    // No separate copy of the gray frame: staging writes its rows into the interior of the padded level 0 and k_pad_level0 adds the
    // REFLECT_101 frame around them in place (1.11 MB per 1241x376 slot instead of 1.58 MB: 192 sequences of the metric configuration
    // fit the 288 GB instead of 128)
    L.gray_off = L.off[0] + (uint32_t)PAD * (uint32_t)L.stride[0] + (uint32_t)PAD;
    L.slot_bytes = (off + 4095) & ~4095u;
    return L;
}

extern "C" {

const char* pmv_last_error(pmv_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int pmv_ctx_create(pmv_ctx** out, int device, int max_w, int max_h, int n_slots, int max_tracks, int max_ba_cams,
                   int max_ba_points, int max_ba_obs) {
    if (!out || max_w < 40 || max_h < 40 || n_slots < 1 || max_tracks < 1) {
        set_err(nullptr, "pmv_ctx_create: invalid argument");
        return PMV_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        set_err(nullptr, "pmv_ctx_create: no HIP device %d (count %d) — this library has no CPU fallback", device, ndev);
        return PMV_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_err(nullptr, "hipGetDeviceProperties failed"); return PMV_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "pmv_ctx_create: device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
        return PMV_ERR_NO_DEVICE;
    }
    pmv_ctx* c = new pmv_ctx();
    c->device = device;
    c->max_w = max_w; c->max_h = max_h; c->n_slots = n_slots; c->max_tracks = max_tracks;
    c->max_ba_cams = max_ba_cams; c->max_ba_points = max_ba_points; c->max_ba_obs = max_ba_obs;
    c->cap = make_layout(max_w, max_h);
    c->slot_layout.assign(n_slots, PyrLayout());
    for (auto& l : c->slot_layout) l.n_levels = 0;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(nullptr, "%s: %s", #x, hipGetErrorString(e_)); pmv_ctx_destroy(c); return PMV_ERR_HIP; } } while (0)
    CK(hipSetDevice(device));
    CK(frontend_prepare_device());
    CK(backend_prepare_device());
    CK(hipStreamCreateWithFlags(&c->s_front, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c->s_back, hipStreamNonBlocking));
    CK(hipMalloc(&c->d_slots, (size_t)c->cap.slot_bytes * n_slots));
    CK(hipMalloc(&c->d_tight, (size_t)pmv_ctx::TIGHT_FRAMES * max_w * max_h + 256));
    const size_t nt = (size_t)max_tracks;
    CK(hipMalloc(&c->d_prev_xy, nt * 12 + 64));   // track coordinates followed by the block -> track order
    CK(hipMalloc(&c->d_out_xy, nt * 8));
    CK(hipMalloc(&c->d_status, nt)); CK(hipMalloc(&c->d_err, nt * 4));
    CK(hipHostMalloc(&c->h_prev_xy, nt * 12 + 64));
    // LK results: 13 bytes per track, written by the kernel through the device aliases of these mapped pinned buffers (no D2H copies)
    CK(hipHostMalloc(&c->h_out_xy, nt * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&c->h_status, nt, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&c->h_err, nt * 4, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&c->dm_out_xy, c->h_out_xy, 0));
    CK(hipHostGetDevicePointer((void**)&c->dm_status, c->h_status, 0));
    CK(hipHostGetDevicePointer((void**)&c->dm_err, c->h_err, 0));
    CK(hipMalloc(&c->d_knn, nt * 16 + 64));
    CK(hipHostMalloc(&c->h_knn, nt * 16 + 64));
    CK(hipHostMalloc(&c->h_work, (size_t)nt * 2, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&c->dm_work, c->h_work, 0));
    for (auto& a : c->lk_work) a.store(0);
    CK(hipMalloc(&c->d_cells, MAX_CELLS * CELL_STRIDE * 4));
    CK(hipHostMalloc(&c->h_cells, MAX_CELLS * CELL_STRIDE * 4));
    CK(hipMalloc(&c->d_eig, (size_t)MAX_CELLS * CELL_PIX * sizeof(double)));   // shared by GFTT (f32) and ShiTomasi (f64)
    CK(hipMalloc(&c->d_cellmax, MAX_CELLS * 8));
    CK(hipMalloc(&c->d_spill, (size_t)MAX_CELLS * CELL_PIX * 4));
    CK(hipMalloc(&c->d_det_xy, (size_t)MAX_CELLS * MAX_PER_CELL * 8));
    CK(hipMalloc(&c->d_det_score, (size_t)MAX_CELLS * MAX_PER_CELL * 8));
    CK(hipMalloc(&c->d_det_count, MAX_CELLS * 4));
    CK(hipMalloc(&c->d_flags, 16));
    CK(hipMemset(c->d_flags, 0, 16));
    CK(hipHostMalloc(&c->h_det_xy, (size_t)MAX_CELLS * MAX_PER_CELL * 8));
    CK(hipHostMalloc(&c->h_det_score, (size_t)MAX_CELLS * MAX_PER_CELL * 8));
    CK(hipHostMalloc(&c->h_det_count, MAX_CELLS * 4 + 16));
    int rc = backend_create(c);
    if (rc != PMV_OK) { snprintf(g_create_err, sizeof(g_create_err), "%s", c->err); pmv_ctx_destroy(c); return rc; }
    {   // every buffer a kernel may touch exists (a missed allocation must fail here, not as a GPU fault later)
        const void* must[] = {c->dm_out_xy, c->dm_status, c->dm_err, c->d_slots, c->d_tight, c->d_prev_xy, c->d_out_xy, c->d_status, c->d_err, c->h_prev_xy, c->h_out_xy, c->h_status, c->h_err,
                              c->d_cells, c->d_eig, c->d_cellmax, c->d_det_xy, c->d_det_score, c->d_det_count, c->d_flags, c->h_det_xy,
                              c->h_det_score, c->h_det_count};
        for (const void* p : must)
            if (!p) { snprintf(g_create_err, sizeof(g_create_err), "pmv_ctx_create: internal error, a front-end buffer was not allocated"); pmv_ctx_destroy(c); return PMV_ERR_HIP; }
    }
#undef CK
    *out = c;
    return PMV_OK;
}

void pmv_ctx_destroy(pmv_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    batch_engine_destroy(c);
    ingest_destroy(c);
    if (c->s_front) hipStreamSynchronize(c->s_front);
    if (c->s_back) hipStreamSynchronize(c->s_back);
    backend_destroy(c);
    c->prof.destroy();
    if (c->d_lk_stamps) hipFree(c->d_lk_stamps);
    if (c->h_work) hipHostFree(c->h_work);
    if (c->d_knn) hipFree(c->d_knn);
    if (c->h_knn) hipHostFree(c->h_knn);
    hipFree(c->d_slots); hipFree(c->d_prev_xy); hipFree(c->d_out_xy); hipFree(c->d_status); hipFree(c->d_err);
    hipHostFree(c->h_prev_xy); hipHostFree(c->h_out_xy); hipHostFree(c->h_status); hipHostFree(c->h_err);
    hipFree(c->d_tight);
    hipFree(c->d_cells); hipFree(c->d_eig); hipFree(c->d_cellmax); hipFree(c->d_det_xy); hipFree(c->d_det_score);
    hipFree(c->d_det_count); hipFree(c->d_flags); hipFree(c->d_spill);
    hipHostFree(c->h_det_xy); hipHostFree(c->h_det_score); hipHostFree(c->h_det_count); hipHostFree(c->h_cells);
    if (c->s_front) hipStreamDestroy(c->s_front);
    if (c->s_back) hipStreamDestroy(c->s_back);
    delete c;
}

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

int pmv_sync(pmv_ctx* ctx) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    CKC(hipStreamSynchronize(ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_back));
    return PMV_OK;
}

}  // extern "C"
// tight != nullptr: level 0 comes from n tight gray frames at `tight` (device) instead of from the levels' own interior
int pmv::build_levels_on(pmv_ctx* ctx, hipStream_t stream, int first_slot, int n, const PyrLayout& L, const uint8_t* tight) {
    CKC(launch_pad_level0(stream, ctx->d_slots, L, first_slot, n, tight));
    for (int l = 1; l < L.n_levels; l++) CKC(launch_pyrdown(stream, ctx->d_slots, L, l, first_slot, n));
    return PMV_OK;
}

// NOTE: slots are addressed with the CAPACITY slot size (ctx->cap.slot_bytes); a frame smaller than max_w x max_h
// uses its own level geometry inside the slot but the same slot pitch.
PyrLayout pmv::layout_for(pmv_ctx* ctx, int w, int h) {
    PyrLayout L = make_layout(w, h);
    L.slot_bytes = ctx->cap.slot_bytes;
    return L;
}
extern "C" {

int pmv_frames_stage(pmv_ctx* ctx, int first_slot, int n, const uint8_t* gray, int w, int h) {
    REQ(ctx && gray, PMV_ERR_INVALID, "pmv_frames_stage: null argument");
    REQ(first_slot >= 0 && n >= 1 && first_slot + n <= ctx->n_slots, PMV_ERR_CAPACITY, "pmv_frames_stage: slots [%d,%d) out of range (n_slots %d)", first_slot, first_slot + n, ctx->n_slots);
    REQ(w >= 40 && h >= 40 && w <= ctx->max_w && h <= ctx->max_h, PMV_ERR_CAPACITY, "pmv_frames_stage: frame %dx%d outside capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
    CKC(hipSetDevice(ctx->device));
    PyrLayout L = layout_for(ctx, w, h);
    // contiguous copies into the landing area, then level 0 (interior + REFLECT_101 frame) written by k_pad_level0 from there
    const size_t fb = (size_t)w * h;
    for (int i0 = 0; i0 < n; i0 += pmv_ctx::TIGHT_FRAMES) {
        const int nb = std::min((int)pmv_ctx::TIGHT_FRAMES, n - i0);
        CKC(hipMemcpyAsync(ctx->d_tight, gray + (size_t)i0 * fb, (size_t)nb * fb, hipMemcpyHostToDevice, ctx->s_front));
        CKC(launch_pad_level0(ctx->s_front, ctx->d_slots, L, first_slot + i0, nb, ctx->d_tight));
    }
    CKC(hipStreamSynchronize(ctx->s_front));
    for (int i = 0; i < n; i++) { ctx->slot_layout[first_slot + i] = L; ctx->slot_layout[first_slot + i].n_levels = -L.n_levels; }
    return PMV_OK;
}

int pmv_frames_build(pmv_ctx* ctx, int first_slot, int n) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    tl_prof = &ctx->prof;
    return pmv::pmv_frames_build_on(ctx, ctx->s_front, first_slot, n);
}
}  // extern "C"
int pmv::pmv_frames_build_on(pmv_ctx* ctx, hipStream_t stream, int first_slot, int n) {
    REQ(first_slot >= 0 && n >= 1 && first_slot + n <= ctx->n_slots, PMV_ERR_CAPACITY, "pmv_frames_build: slot range");
    CKC(hipSetDevice(ctx->device));
    // consecutive slots with identical geometry are built in one batched launch per level
    int i = 0;
    while (i < n) {
        PyrLayout L = ctx->slot_layout[first_slot + i];
        REQ(L.n_levels != 0, PMV_ERR_INVALID, "pmv_frames_build: slot %d was never staged", first_slot + i);
        int j = i + 1;
        while (j < n && ctx->slot_layout[first_slot + j].w[0] == L.w[0] && ctx->slot_layout[first_slot + j].h[0] == L.h[0]) j++;
        if (L.n_levels < 0) L.n_levels = -L.n_levels;
        int rc = build_levels_on(ctx, stream, first_slot + i, j - i, L);
        if (rc) return rc;
        for (int k = i; k < j; k++) ctx->slot_layout[first_slot + k].n_levels = L.n_levels;
        i = j;
    }
    return PMV_OK;
}
extern "C" {

int pmv_frame_upload(pmv_ctx* ctx, int slot, const uint8_t* gray, int w, int h, int stride) {
    REQ(ctx && gray, PMV_ERR_INVALID, "pmv_frame_upload: null argument");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_frame_upload: slot %d out of range", slot);
    REQ(w >= 40 && h >= 40 && w <= ctx->max_w && h <= ctx->max_h && stride >= w, PMV_ERR_CAPACITY, "pmv_frame_upload: frame %dx%d outside capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    PyrLayout L = layout_for(ctx, w, h);
    CKC(hipMemcpy2DAsync(ctx->d_tight, (size_t)w, gray, stride, w, h, hipMemcpyHostToDevice, ctx->s_front));
    int rc = build_levels_on(ctx, ctx->s_front, slot, 1, L, ctx->d_tight);
    if (rc) return rc;
    CKC(hipStreamSynchronize(ctx->s_front));   // the host buffer may be reused by the caller
    ctx->slot_layout[slot] = L;
    return PMV_OK;
}

// Frame::Frame(file) + Frame::init for a COLOUR image (Frame.cpp:33,40-41: imread(IMREAD_COLOR) gives BGR, cvtColor(BGR2GRAY) the u8 `bw`
// everything else works on): the conversion runs on the device, in front of the landing area.
int pmv_frame_upload_bgr(pmv_ctx* ctx, int slot, const uint8_t* bgr, int w, int h, int stride) {
    REQ(ctx && bgr, PMV_ERR_INVALID, "pmv_frame_upload_bgr: null argument");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_frame_upload_bgr: slot %d out of range", slot);
    REQ(w >= 40 && h >= 40 && w <= ctx->max_w && h <= ctx->max_h && stride >= 3 * w, PMV_ERR_CAPACITY, "pmv_frame_upload_bgr: frame %dx%d outside capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    PyrLayout L = layout_for(ctx, w, h);
    // landing area: frame 0 receives the gray image, frames 1..3 hold the three bytes per pixel on their way in (TIGHT_FRAMES >= 4)
    uint8_t* d_bgr = ctx->d_tight + (size_t)ctx->max_w * ctx->max_h;
    CKC(hipMemcpy2DAsync(d_bgr, (size_t)3 * w, bgr, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, ctx->s_front));
    CKC(launch_bgr2gray(ctx->s_front, d_bgr, w, h, 3 * w, ctx->d_tight));
    int rc = build_levels_on(ctx, ctx->s_front, slot, 1, L, ctx->d_tight);
    if (rc) return rc;
    CKC(hipStreamSynchronize(ctx->s_front));
    ctx->slot_layout[slot] = L;
    return PMV_OK;
}

int pmv_frame_num_levels(pmv_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= ctx->n_slots) return PMV_ERR_INVALID;
    const int n = ctx->slot_layout[slot].n_levels;
    return n > 0 ? n - 1 : PMV_ERR_INVALID;
}

int pmv_frame_get_level(pmv_ctx* ctx, int slot, int level, uint8_t* out, int* w, int* h) {
    REQ(ctx && out, PMV_ERR_INVALID, "null argument");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "slot out of range");
    const PyrLayout& L = ctx->slot_layout[slot];
    REQ(L.n_levels > 0 && level >= 0 && level < L.n_levels, PMV_ERR_INVALID, "level %d not built for slot %d", level, slot);
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    const uint8_t* org = level_origin((const uint8_t*)ctx->d_slots + (size_t)slot * L.slot_bytes, L, level);
    CKC(hipMemcpy2D(out, L.w[level], org, L.stride[level], L.w[level], L.h[level], hipMemcpyDeviceToHost));
    if (w) *w = L.w[level];
    if (h) *h = L.h[level];
    return PMV_OK;
}

int pmv_frame_get_level_padded(pmv_ctx* ctx, int slot, int level, uint8_t* out, int* pw, int* ph) {
    REQ(ctx && out, PMV_ERR_INVALID, "null argument");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "slot out of range");
    const PyrLayout& L = ctx->slot_layout[slot];
    REQ(L.n_levels > 0 && level >= 0 && level < L.n_levels, PMV_ERR_INVALID, "level %d not built for slot %d", level, slot);
    static_assert(PAD == PMV_PYR_PAD, "header constant");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    const uint8_t* base = (const uint8_t*)ctx->d_slots + (size_t)slot * L.slot_bytes + L.off[level];
    const int w = L.w[level] + 2 * PAD, h = L.h[level] + 2 * PAD;
    CKC(hipMemcpy2D(out, w, base, L.stride[level], w, h, hipMemcpyDeviceToHost));
    if (pw) *pw = w;
    if (ph) *ph = h;
    return PMV_OK;
}

int pmv_lk_track(pmv_ctx* ctx, int prev_slot, int next_slot, const float* prev_xy, int n, float* out_xy,
                 uint8_t* out_status, float* out_err) {
    REQ(ctx && (n == 0 || (prev_xy && out_xy && out_status && out_err)), PMV_ERR_INVALID, "pmv_lk_track: null argument");
    REQ(n >= 0 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_lk_track: n=%d exceeds max_tracks=%d", n, ctx->max_tracks);
    REQ(prev_slot >= 0 && prev_slot < ctx->n_slots && next_slot >= 0 && next_slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_lk_track: slot out of range");
    if (ctx->ingest) { int rc_ = ingest_require(ctx, prev_slot > next_slot ? prev_slot : next_slot); if (rc_) return rc_; }
    const PyrLayout& L = ctx->slot_layout[prev_slot];
    const PyrLayout& L2 = ctx->slot_layout[next_slot];
    REQ(L.n_levels > 0 && L2.n_levels > 0, PMV_ERR_INVALID, "pmv_lk_track: slot has no pyramid");
    REQ(L.w[0] == L2.w[0] && L.h[0] == L2.h[0], PMV_ERR_INVALID, "pmv_lk_track: frame sizes differ");
    if (n == 0) return PMV_OK;
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    memcpy(ctx->h_prev_xy, prev_xy, (size_t)n * 8);
    // XCD-aware block order: workgroup b runs on XCD b % 8 and every XCD has its own L2, so the tracks (which arrive in hash
    // order, i.e. spatially random) are dealt out by x position: the k-th track of equal-count stripe s goes to block 8k + s.
    // Each L2 then fetches one vertical stripe of the two pyramids instead of all of them (measured: 5x less HBM traffic).
    int* order = (int*)(ctx->h_prev_xy + (size_t)2 * n);
    const int nb = (n + 7) / 8 * 8;
    {
        static thread_local std::vector<std::pair<float, int>> byx;
        byx.resize(n);
        for (int i = 0; i < n; i++) byx[i] = {prev_xy[2 * i], i};
        std::sort(byx.begin(), byx.end());
        for (int b = 0; b < nb; b++) order[b] = -1;
        for (int i = 0; i < n; i++) {
            const int s8 = (int)((long)i * 8 / n), first = (int)(((long)s8 * n + 7) / 8);   // stripe and its first sorted index
            order[(i - first) * 8 + s8] = byx[i].second;
        }
    }
    CKC(hipMemcpyAsync(ctx->d_prev_xy, ctx->h_prev_xy, (size_t)n * 8 + (size_t)nb * 4, hipMemcpyHostToDevice, ctx->s_front));
    LKParams P;
    P.max_iter = 30; P.eps2 = 1e-4f; P.eps2d = 0.01 * 0.01; P.min_eig = 1e-4f;
    static const bool lk_stamps = getenv("PMV_LK_STAMPS") != nullptr;   // read once per process
    if (!ctx->d_lk_stamps && lk_stamps) { CKC(hipMalloc(&ctx->d_lk_stamps, 16 * 8)); CKC(hipMemset(ctx->d_lk_stamps, 0, 16 * 8)); }
    P.stamps = ctx->d_lk_stamps;
    CKC(launch_lk(ctx->s_front, ctx->d_slots + (size_t)prev_slot * L.slot_bytes, ctx->d_slots + (size_t)next_slot * L.slot_bytes,
                  L, ctx->d_prev_xy, (const int*)(ctx->d_prev_xy + (size_t)2 * n), nb, n, P, ctx->dm_out_xy, ctx->dm_status, ctx->dm_err, ctx->dm_work));
    CKC(hipStreamSynchronize(ctx->s_front));   // the kernel wrote positions / status / err straight into mapped pinned memory
    memcpy(out_xy, ctx->h_out_xy, (size_t)n * 8);
    memcpy(out_status, ctx->h_status, (size_t)n);
    memcpy(out_err, ctx->h_err, (size_t)n * 4);
    ctx->add_lk_work(ctx->h_work, (size_t)n);
    return PMV_OK;
}

static void pack_cells(int* dst, const int* cells, int n_cells, int slot) {   // (x0, y0, w, h) -> device cell records of that frame slot
    for (int i = 0; i < n_cells; i++) {
        int* d = dst + (size_t)i * CELL_STRIDE;
        d[0] = cells[4 * i]; d[1] = cells[4 * i + 1]; d[2] = cells[4 * i + 2]; d[3] = cells[4 * i + 3]; d[4] = slot; d[5] = d[6] = d[7] = 0;
    }
}
static int check_cells(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell) {   // max_per_cell: already >= 1
    REQ(ctx && cells, PMV_ERR_INVALID, "detect: null argument");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "detect: slot out of range");
    REQ(n_cells >= 1 && n_cells <= MAX_CELLS, PMV_ERR_CAPACITY, "detect: n_cells=%d (max %d)", n_cells, MAX_CELLS);
    REQ(max_per_cell >= 1 && max_per_cell <= MAX_PER_CELL, PMV_ERR_CAPACITY, "detect: max_per_cell=%d (max %d)", max_per_cell, MAX_PER_CELL);
    if (ctx->ingest) { int rc_ = ingest_require(ctx, slot); if (rc_) return rc_; }
    const PyrLayout& L = ctx->slot_layout[slot];
    REQ(L.n_levels > 0, PMV_ERR_INVALID, "detect: slot %d has no frame", slot);
    for (int i = 0; i < n_cells; i++) {
        const int* c = cells + 4 * i;
        REQ(c[2] >= 3 && c[3] >= 3 && c[2] <= CELL_MAX && c[3] <= CELL_MAX && c[0] >= 0 && c[1] >= 0 && c[0] + c[2] <= L.w[0] && c[1] + c[3] <= L.h[0],
            PMV_ERR_INVALID, "detect: cell %d (%d,%d,%d,%d) invalid for %dx%d frame", i, c[0], c[1], c[2], c[3], L.w[0], L.h[0]);
    }
    return PMV_OK;
}

int pmv_detect_gftt(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, double quality,
                    double min_dist, int* out_xy, int* out_count) {
    // cv::goodFeaturesToTrack: maxCorners <= 0 means "no limit" (the reference gets there when min_tracked_features < number of
    // grid cells, OdometryPipeline.cpp:438 integer division). The caller's out_xy then holds PMV_GFTT_UNLIMITED_CAP corners per cell.
    const int unlimited = max_per_cell <= 0;
    if (unlimited) max_per_cell = MAX_PER_CELL;
    int rc = check_cells(ctx, slot, cells, n_cells, max_per_cell);
    if (rc) return rc;
    REQ(out_xy && out_count, PMV_ERR_INVALID, "pmv_detect_gftt: null output");
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    const PyrLayout& L = ctx->slot_layout[slot];
    pack_cells(ctx->h_cells, cells, n_cells, slot);
    CKC(hipMemcpyAsync(ctx->d_cells, ctx->h_cells, (size_t)n_cells * CELL_STRIDE * 4, hipMemcpyHostToDevice, ctx->s_front));
    CKC(hipMemsetAsync(ctx->d_flags, 0, 16, ctx->s_front));   // overflow bits are per call: one overflow must not poison later calls
    static const bool gftt_dbg = getenv("PMV_GFTT_DBG") != nullptr;
    if (gftt_dbg) { const int on = 0x40000000; CKC(hipMemcpyAsync(ctx->d_flags, &on, 4, hipMemcpyHostToDevice, ctx->s_front)); }
    CKC(launch_gftt(ctx->s_front, ctx->d_slots, L, ctx->d_cells, n_cells, max_per_cell, quality,
                    min_dist, unlimited, (float*)ctx->d_eig, (unsigned*)ctx->d_cellmax, ctx->d_det_xy, ctx->d_det_count, ctx->d_flags, ctx->d_spill));
    const size_t nxy = (size_t)n_cells * max_per_cell * 8;
    CKC(hipMemcpyAsync(ctx->h_det_xy, ctx->d_det_xy, nxy, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_count, ctx->d_det_count, (size_t)n_cells * 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_count + MAX_CELLS, ctx->d_flags, 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_front));
    if (gftt_dbg) {
        int f[4];
        CKC(hipMemcpy(f, ctx->d_flags, 16, hipMemcpyDeviceToHost));
        fprintf(stderr, "[gftt-dbg] cell 0: %d raw records, %d above the threshold; cycles: compaction %d, key set-up %d, rounds %d\n", f[0] & 0xffff, (f[0] >> 16) & 0x3fff, f[1], f[2], f[3]);
        ctx->h_det_count[MAX_CELLS] &= ~0x40000000;
    }
    REQ((ctx->h_det_count[MAX_CELLS] & 4) == 0, PMV_ERR_OVERFLOW, "pmv_detect_gftt: more than %d corners in a cell with max_per_cell <= 0 (no limit)", MAX_PER_CELL);
    memcpy(out_xy, ctx->h_det_xy, nxy);
    memcpy(out_count, ctx->h_det_count, (size_t)n_cells * 4);
    return PMV_OK;
}

int pmv_detect_shitomasi(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, double quality,
                         int* out_xy, double* out_score, int* out_count) {
    if (max_per_cell <= 0) {   // ShiTomasiFeatureExtractor.cpp:37-44: `if (i >= max) break` before the first feature -> nothing
        REQ(ctx && cells && out_count && n_cells >= 1 && n_cells <= MAX_CELLS, PMV_ERR_INVALID, "pmv_detect_shitomasi: bad argument");
        for (int i = 0; i < n_cells; i++) out_count[i] = 0;
        return PMV_OK;
    }
    int rc = check_cells(ctx, slot, cells, n_cells, max_per_cell);
    if (rc) return rc;
    REQ(out_xy && out_score && out_count, PMV_ERR_INVALID, "pmv_detect_shitomasi: null output");
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    const PyrLayout& L = ctx->slot_layout[slot];
    pack_cells(ctx->h_cells, cells, n_cells, slot);
    CKC(hipMemcpyAsync(ctx->d_cells, ctx->h_cells, (size_t)n_cells * CELL_STRIDE * 4, hipMemcpyHostToDevice, ctx->s_front));
    CKC(hipMemsetAsync(ctx->d_flags, 0, 16, ctx->s_front));
    CKC(launch_shitomasi(ctx->s_front, ctx->d_slots, L, ctx->d_cells, n_cells, max_per_cell, quality,
                         ctx->d_eig, (unsigned long long*)ctx->d_cellmax, ctx->d_det_xy, ctx->d_det_score, ctx->d_det_count, ctx->d_flags, ctx->d_spill));
    const size_t nxy = (size_t)n_cells * max_per_cell * 8;
    CKC(hipMemcpyAsync(ctx->h_det_xy, ctx->d_det_xy, nxy, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_score, ctx->d_det_score, nxy, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_count, ctx->d_det_count, (size_t)n_cells * 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_count + MAX_CELLS, ctx->d_flags, 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_front));
    memcpy(out_xy, ctx->h_det_xy, nxy);
    memcpy(out_score, ctx->h_det_score, nxy);
    memcpy(out_count, ctx->h_det_count, (size_t)n_cells * 4);
    return PMV_OK;
}

// ---- per-kernel HIP-event timing -------------------------------------------------------------------------------------
int pmv_prof_enable(pmv_ctx* ctx, int on) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_back));
    if (on) for (int i = 0; i < K_COUNT; i++) { ctx->prof.used[i] = 0; ctx->prof.dropped[i] = 0; }
    ctx->prof.enabled = on != 0;
    ctx->prof.mask = ~0u;
    ctx->prof.chain_detail = false;
    return PMV_OK;
}
// restrict recording to the classes whose bit is set (events cost host time and a queue barrier each: the timed region of the
// bench records only the kernel its roofline object reports)
int pmv_prof_select(pmv_ctx* ctx, unsigned mask) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    ctx->prof.mask = mask;
    ctx->prof.chain_detail = ((mask >> K_BAM_EVAL0) & 0x3fu) != 0 && mask != ~0u;
    return PMV_OK;
}
// diagnostic: phase timers of k_lk for track 0 (PMV_LK_STAMPS=1): [0] level entry, [1] I tile, [2] Scharr, [3] samples + A, [4] iterations
// (+ J tiles), [5] error pass; [8] iteration count
int pmv_debug_lk_stamps(pmv_ctx* ctx, unsigned long long* out16) {
    REQ(ctx && out16, PMV_ERR_INVALID, "null argument");
    memset(out16, 0, 16 * 8);
    if (!ctx->d_lk_stamps) return PMV_OK;
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    CKC(hipMemcpy(out16, ctx->d_lk_stamps, 16 * 8, hipMemcpyDeviceToHost));
    return PMV_OK;
}
// work counters of k_lk accumulated since the context was created or since the last reset: [0] LK iterations executed, [1] (track,
// level) pairs that entered the iteration loop, [2] tracks. bench.py turns them into SURVEY.md §8(d)'s OPS_lk = sum over tracks and
// levels of 1024 * (40 + 14 * iterations).
int pmv_lk_counters(pmv_ctx* ctx, unsigned long long* out3, int reset) {
    REQ(ctx && out3, PMV_ERR_INVALID, "null argument");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    for (int i = 0; i < 3; i++) out3[i] = reset ? ctx->lk_work[i].exchange(0) : ctx->lk_work[i].load();
    return PMV_OK;
}
int pmv_prof_kernel_count(void) { return K_COUNT; }
const char* pmv_prof_kernel_name(int id) { return kernel_name(id); }
int pmv_prof_read(pmv_ctx* ctx, int id, int* launches, double* total_ms, double* max_ms) {
    REQ(ctx && id >= 0 && id < K_COUNT && launches && total_ms && max_ms, PMV_ERR_INVALID, "pmv_prof_read: bad argument");
    CKC(hipSetDevice(ctx->device));
    CKC(hipStreamSynchronize(ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_back));
    double tot = 0, mx = 0;
    const int n = ctx->prof.used[id] / 2;
    for (int i = 0; i < n; i++) {
        float ms = 0;
        CKC(hipEventElapsedTime(&ms, ctx->prof.ev[id][2 * i], ctx->prof.ev[id][2 * i + 1]));
        tot += ms;
        if (ms > mx) mx = ms;
    }
    *launches = n; *total_ms = tot; *max_ms = mx;
    return PMV_OK;
}

int pmv_debug_gftt_response(pmv_ctx* ctx, int slot, const int* cell, float* out) {
    int rc = check_cells(ctx, slot, cell, 1, 1);
    if (rc) return rc;
    REQ(out, PMV_ERR_INVALID, "null output");
    CKC(hipSetDevice(ctx->device));
    pack_cells(ctx->h_cells, cell, 1, slot);
    CKC(hipMemcpyAsync(ctx->d_cells, ctx->h_cells, (size_t)CELL_STRIDE * 4, hipMemcpyHostToDevice, ctx->s_front));
    CKC(launch_gftt_response(ctx->s_front, ctx->d_slots, ctx->slot_layout[slot], ctx->d_cells, 1, (float*)ctx->d_eig, (unsigned*)ctx->d_cellmax));
    CKC(hipStreamSynchronize(ctx->s_front));
    CKC(hipMemcpy(out, ctx->d_eig, (size_t)cell[2] * cell[3] * sizeof(float), hipMemcpyDeviceToHost));
    return PMV_OK;
}

int pmv_debug_shitomasi_response(pmv_ctx* ctx, int slot, const int* cell, double* out) {
    int rc = check_cells(ctx, slot, cell, 1, 1);
    if (rc) return rc;
    REQ(out, PMV_ERR_INVALID, "null output");
    int xy[2], cnt; double sc;
    rc = pmv_detect_shitomasi(ctx, slot, cell, 1, 1, 0.4, xy, &sc, &cnt);
    if (rc) return rc;
    CKC(hipMemcpy(out, ctx->d_eig, (size_t)cell[2] * cell[3] * sizeof(double), hipMemcpyDeviceToHost));
    return PMV_OK;
}

}  // extern "C"
