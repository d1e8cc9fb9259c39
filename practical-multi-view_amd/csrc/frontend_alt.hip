// The reference's alternative front-end plugins: kNNFeatureMatcher (kNNFeatureMatcher.cpp:3-122, arithmetic fully in
// the reference's own source) and OpenCVFASTFeatureExtractor (OpenCVFASTFeatureExtractor.cpp:4-21 -> cv::FAST 9_16). Neither is
// instantiated by the reference's pipeline (OdometryPipeline.cpp:68-69 hard-wires GFTT + LK); they complete the plugin matrix.
// Integer / index results are bit-exact with oracle/orc_knn.cpp and oracle/orc_fast9.cpp.
#include "pmv_ctx.h"
#include <cstring>
#include <algorithm>

namespace pmv {

// =========================================================================================================
// kNN matcher: one wavefront per source feature
// =========================================================================================================
constexpr int KNN_MAX_NN = 8;

__device__ inline float knn_compare(const uint8_t* __restrict__ I, const uint8_t* __restrict__ J, int st, int w, int h, int sx, int sy, int cx, int cy, int win) {
    // kNNFeatureMatcher.cpp:103-122: x outer, y inner, pixels outside either image skipped; float accumulator fed through a double addition
    float err = 0.f;
    for (int x = -win; x < win + 1; x++)
        for (int y = -win; y < win + 1; y++) {
            if (sx + x < 0 || sy + y < 0 || cx + x < 0 || cy + y < 0 || sx + x >= w || sy + y >= h || cx + x >= w || cy + y >= h) continue;
            const float d = (float)I[(ptrdiff_t)(sy + y) * st + sx + x] - (float)J[(ptrdiff_t)(cy + y) * st + cx + x];
            err = (float)((double)err + (double)d * (double)d);
        }
    return err;
}

__global__ __launch_bounds__(64) void k_knn_match(const uint8_t* __restrict__ slots, PyrLayout L, unsigned long long src_off, unsigned long long cmp_off,
                                                  const int* __restrict__ src_xy, int n, const int* __restrict__ cmp_xy, int m, int n_nn, int window,
                                                  int* __restrict__ out_best, float* __restrict__ out_err) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const int fx = src_xy[2 * i], fy = src_xy[2 * i + 1];
    int nnv[KNN_MAX_NN];
    int nearest = -1;   // the default Feature: column 0, row 0
    for (int k = 0; k < n_nn; k++) {
        unsigned long long best = ~0ull;   // (distance << 32) | index: the first candidate with the smallest distance wins
        for (int j = lane; j < m; j += 64) {
            const int cx = cmp_xy[2 * j], cy = cmp_xy[2 * j + 1];
            if (cx == fx && cy == fy) continue;
            bool fresh = true;
            for (int q = 0; q < k; q++) {
                const int qx = nnv[q] < 0 ? 0 : cmp_xy[2 * nnv[q]], qy = nnv[q] < 0 ? 0 : cmp_xy[2 * nnv[q] + 1];
                if (cx == qx && cy == qy) fresh = false;
            }
            if (!fresh) continue;
            const int dx = abs(fx - cx), dy = abs(fy - cy);
            const unsigned long long key = ((unsigned long long)(unsigned)(dx > dy ? dx : dy) << 32) | (unsigned)j;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
        if (best != ~0ull) nearest = (int)(best & 0xffffffffu);
        nnv[k] = nearest;
    }
    const uint8_t* I = level_origin(slots + src_off, L, 0);
    const uint8_t* J = level_origin(slots + cmp_off, L, 0);
    const int win = (int)ceilf((float)window / 2.f);
    float e = 0.f;
    int cand = -1;
    if (lane < n_nn) {
        cand = nnv[lane];
        const int cx = cand < 0 ? 0 : cmp_xy[2 * cand], cy = cand < 0 ? 0 : cmp_xy[2 * cand + 1];
        const float acc = knn_compare(I, J, L.stride[0], L.w[0], L.h[0], fx, fy, cx, cy, win);
        e = (float)(sqrt((double)acc) / ((double)window * (double)window));
    }
    // :19-31 sequential `_err < err || err == 0`
    float err = 0.f;
    int best_idx = -1;
    for (int k = 0; k < n_nn; k++) {
        const float ek = __shfl(e, k, 64);
        const int ck = __shfl(cand, k, 64);
        if (ek < err || err == 0.f) { err = ek; best_idx = ck; }
    }
    if (lane == 0) { out_best[i] = best_idx; out_err[i] = err; }
}

// =========================================================================================================
// FAST 9_16: score map (0 = not a corner) + ordered selection
// =========================================================================================================
__global__ __launch_bounds__(256) void k_fast_score(const uint8_t* __restrict__ slots, PyrLayout L, const int* __restrict__ cells, int threshold, int nonmax,
                                                    uint8_t* __restrict__ score) {
    const int cell = blockIdx.z;
    const int* c = cells + CELL_STRIDE * cell;
    const int cx0 = c[0], cy0 = c[1], cw = c[2], ch = c[3];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cw * ch) return;
    const int y = idx / cw, x = idx - y * cw;
    uint8_t out = 0;
    if (x >= 3 && x < cw - 3 && y >= 3 && y < ch - 3) {
        const int st = L.stride[0];
        const uint8_t* p = level_origin(slots + (size_t)c[4] * L.slot_bytes, L, 0) + (ptrdiff_t)(cy0 + y) * st + cx0 + x;
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * st]; d[1] = v - p[3 * st + 1]; d[2] = v - p[2 * st + 2]; d[3] = v - p[st + 3];
        d[4] = v - p[3]; d[5] = v - p[-st + 3]; d[6] = v - p[-2 * st + 2]; d[7] = v - p[-3 * st + 1];
        d[8] = v - p[-3 * st]; d[9] = v - p[-3 * st - 1]; d[10] = v - p[-2 * st - 2]; d[11] = v - p[-st - 3];
        d[12] = v - p[-3]; d[13] = v - p[st - 3]; d[14] = v - p[2 * st - 2]; d[15] = v - p[3 * st - 1];
        unsigned dark = 0, bright = 0;   // circle pixel darker than v - t  <=>  d > t ; brighter than v + t  <=>  d < -t
#pragma unroll
        for (int k = 0; k < 16; k++) { dark |= (unsigned)(d[k] > threshold) << k; bright |= (unsigned)(d[k] < -threshold) << k; }
        dark |= dark << 16; bright |= bright << 16;
        unsigned rd = dark, rb = bright;
#pragma unroll
        for (int s = 1; s <= 8; s++) { rd &= dark >> s; rb &= bright >> s; }
        if (((rd | rb) & 0x1ffffu) != 0) {
            if (!nonmax) out = 1;
            else {
                // cornerScore<16>: a0 = max over the 16 arcs of nine of min(d), b0 = min over arcs of max(d), score = -b0 - 1
                int a0 = threshold;
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    int a = d[s];
#pragma unroll
                    for (int q = 1; q < 9; q++) { const int t = d[(s + q) & 15]; a = t < a ? t : a; }
                    a0 = a > a0 ? a : a0;
                }
                int b0 = -a0;
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    int b = d[s];
#pragma unroll
                    for (int q = 1; q < 9; q++) { const int t = d[(s + q) & 15]; b = t > b ? t : b; }
                    b0 = b < b0 ? b : b0;
                }
                out = (uint8_t)(-b0 - 1);
            }
        }
    }
    score[(size_t)(unsigned)c[5] + idx] = out;
}

// one 256-thread workgroup per cell: rows in order, columns in order -> keypoints in cv::FAST's raster order, first `max_kp` kept
__global__ __launch_bounds__(256) void k_fast_select(const int* __restrict__ cells, const uint8_t* __restrict__ score, int nonmax, int max_kp,
                                                     int* __restrict__ out_xy, float* __restrict__ out_resp, int* __restrict__ out_count) {
    __shared__ int wsum[4];
    __shared__ int sbase;
    const int cell = blockIdx.x, tid = threadIdx.x;
    const int* c = cells + CELL_STRIDE * cell;
    const int cw = c[2], ch = c[3];
    const uint8_t* S = score + (size_t)(unsigned)c[5];
    if (tid == 0) sbase = 0;
    __syncthreads();
    for (int y = 3; y < ch - 3; y++) {
        for (int x0 = 0; x0 < cw; x0 += 256) {
            const int x = x0 + tid;
            bool kp = false;
            int s = 0;
            if (x >= 3 && x < cw - 3) {
                const uint8_t* r = S + (size_t)y * cw + x;
                s = r[0];
                if (s > 0)
                    kp = !nonmax || (s > r[1] && s > r[-1] && s > r[-cw - 1] && s > r[-cw] && s > r[-cw + 1] && s > r[cw - 1] && s > r[cw] && s > r[cw + 1]);
            }
            const unsigned long long bal = __ballot(kp);
            const int within = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) wsum[tid >> 6] = __popcll(bal);
            __syncthreads();
            int before = 0;
            for (int w = 0; w < (tid >> 6); w++) before += wsum[w];
            const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            const int base = sbase;
            const int pos = base + before + within;
            if (kp && pos < max_kp) {
                out_xy[((size_t)cell * max_kp + pos) * 2] = x;
                out_xy[((size_t)cell * max_kp + pos) * 2 + 1] = y;
                out_resp[(size_t)cell * max_kp + pos] = nonmax ? (float)s : 0.f;
            }
            __syncthreads();
            if (tid == 0) sbase = base + total;
            __syncthreads();
            if (sbase >= max_kp) { y = ch; break; }   // block-uniform
        }
    }
    if (tid == 0) out_count[cell] = sbase < max_kp ? sbase : max_kp;
}

}  // namespace pmv

using namespace pmv;

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

extern "C" {

int pmv_knn_match(pmv_ctx* ctx, int src_slot, int cmp_slot, const int* src_xy, int n, const int* cmp_xy, int m, int n_neighbours, int window,
                  int* out_best, float* out_err) {
    REQ(ctx && (n == 0 || (src_xy && out_best && out_err)) && (m == 0 || cmp_xy), PMV_ERR_INVALID, "pmv_knn_match: null argument");
    REQ(n >= 0 && n <= ctx->max_tracks && m >= 0 && m <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_knn_match: n=%d / m=%d exceed max_tracks=%d", n, m, ctx->max_tracks);
    REQ(n_neighbours >= 1 && n_neighbours <= KNN_MAX_NN && window >= 1 && window <= 63, PMV_ERR_INVALID, "pmv_knn_match: n_neighbours 1..%d, window 1..63", KNN_MAX_NN);
    REQ(src_slot >= 0 && src_slot < ctx->n_slots && cmp_slot >= 0 && cmp_slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_knn_match: slot out of range");
    if (ctx->ingest) { int rc_ = ingest_require(ctx, src_slot > cmp_slot ? src_slot : cmp_slot); if (rc_) return rc_; }
    const PyrLayout& L = ctx->slot_layout[src_slot];
    const PyrLayout& L2 = ctx->slot_layout[cmp_slot];
    REQ(L.n_levels > 0 && L2.n_levels > 0 && L.w[0] == L2.w[0] && L.h[0] == L2.h[0], PMV_ERR_INVALID, "pmv_knn_match: slots have no frame / sizes differ");
    if (n == 0) return PMV_OK;
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    // staging: [src_xy 2n | cmp_xy 2m] ints in the LK coordinate buffers (sized 12 B per track + 64), results through the mapped LK result blocks
    int* h = (int*)ctx->h_knn;
    memcpy(h, src_xy, (size_t)n * 8);
    if (m) memcpy(h + 2 * n, cmp_xy, (size_t)m * 8);
    CKC(hipMemcpyAsync(ctx->d_knn, h, ((size_t)n + m) * 8, hipMemcpyHostToDevice, ctx->s_front));
    const int* d_src = (const int*)ctx->d_knn;
    const int* d_cmp = d_src + 2 * n;
    hipLaunchKernelGGL(k_knn_match, dim3(n), dim3(64), 0, ctx->s_front, (const uint8_t*)ctx->d_slots, L, (unsigned long long)src_slot * L.slot_bytes,
                       (unsigned long long)cmp_slot * L.slot_bytes, d_src, n, d_cmp, m, n_neighbours, window, (int*)ctx->dm_out_xy, ctx->dm_err);
    CKC(hipGetLastError());
    CKC(hipStreamSynchronize(ctx->s_front));
    memcpy(out_best, ctx->h_out_xy, (size_t)n * 4);
    memcpy(out_err, ctx->h_err, (size_t)n * 4);
    return PMV_OK;
}

int pmv_detect_fast(pmv_ctx* ctx, int slot, const int* cells, int n_cells, int max_per_cell, int threshold, int nonmax, int* out_xy, float* out_response,
                    int* out_count) {
    REQ(ctx && cells && out_count && n_cells >= 1 && n_cells <= MAX_CELLS, PMV_ERR_INVALID, "pmv_detect_fast: bad argument");
    if (max_per_cell <= 0) { for (int i = 0; i < n_cells; i++) out_count[i] = 0; return PMV_OK; }   // OpenCVFASTFeatureExtractor.cpp:12 `if (i >= max) break`
    REQ(out_xy && out_response, PMV_ERR_INVALID, "pmv_detect_fast: null output");
    REQ(slot >= 0 && slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_detect_fast: slot out of range");
    if (ctx->ingest) { int rc_ = ingest_require(ctx, slot); if (rc_) return rc_; }
    const PyrLayout& L = ctx->slot_layout[slot];
    REQ(L.n_levels > 0, PMV_ERR_INVALID, "pmv_detect_fast: slot %d has no frame", slot);
    REQ((size_t)n_cells * max_per_cell <= (size_t)MAX_CELLS * MAX_PER_CELL, PMV_ERR_CAPACITY, "pmv_detect_fast: n_cells * max_per_cell = %zu exceeds %d",
        (size_t)n_cells * max_per_cell, MAX_CELLS * MAX_PER_CELL);
    // a FAST "cell" may be as large as the frame (kNNFeatureMatcher calls the extractor on the whole next frame, :11)
    size_t tot = 0;
    int maxpix = 0;
    for (int i = 0; i < n_cells; i++) {
        const int* c = cells + 4 * i;
        REQ(c[2] >= 1 && c[3] >= 1 && c[0] >= 0 && c[1] >= 0 && c[0] + c[2] <= L.w[0] && c[1] + c[3] <= L.h[0], PMV_ERR_INVALID,
            "pmv_detect_fast: cell %d (%d,%d,%d,%d) invalid for %dx%d frame", i, c[0], c[1], c[2], c[3], L.w[0], L.h[0]);
        int* d = ctx->h_cells + (size_t)i * CELL_STRIDE;
        d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = c[3]; d[4] = slot; d[5] = (int)tot; d[6] = d[7] = 0;
        tot += (size_t)c[2] * c[3];
        maxpix = std::max(maxpix, c[2] * c[3]);
    }
    REQ(tot <= (size_t)MAX_CELLS * CELL_PIX * sizeof(double), PMV_ERR_CAPACITY, "pmv_detect_fast: cells cover %zu pixels (max %zu)", tot, (size_t)MAX_CELLS * CELL_PIX * 8);
    tl_prof = &ctx->prof;
    CKC(hipSetDevice(ctx->device));
    CKC(hipMemcpyAsync(ctx->d_cells, ctx->h_cells, (size_t)n_cells * CELL_STRIDE * 4, hipMemcpyHostToDevice, ctx->s_front));
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    hipLaunchKernelGGL(k_fast_score, dim3((maxpix + 255) / 256, 1, n_cells), dim3(256), 0, ctx->s_front, (const uint8_t*)ctx->d_slots, L, ctx->d_cells, threshold,
                       nonmax ? 1 : 0, (uint8_t*)ctx->d_eig);
    hipLaunchKernelGGL(k_fast_select, dim3(n_cells), dim3(256), 0, ctx->s_front, ctx->d_cells, (const uint8_t*)ctx->d_eig, nonmax ? 1 : 0, max_per_cell,
                       ctx->d_det_xy, (float*)ctx->d_det_score, ctx->d_det_count);
    CKC(hipGetLastError());
    const size_t nk = (size_t)n_cells * max_per_cell;
    CKC(hipMemcpyAsync(ctx->h_det_xy, ctx->d_det_xy, nk * 8, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_score, ctx->d_det_score, nk * 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipMemcpyAsync(ctx->h_det_count, ctx->d_det_count, (size_t)n_cells * 4, hipMemcpyDeviceToHost, ctx->s_front));
    CKC(hipStreamSynchronize(ctx->s_front));
    memcpy(out_xy, ctx->h_det_xy, nk * 8);
    memcpy(out_response, ctx->h_det_score, nk * 4);
    memcpy(out_count, ctx->h_det_count, (size_t)n_cells * 4);
    return PMV_OK;
}

}  // extern "C"
