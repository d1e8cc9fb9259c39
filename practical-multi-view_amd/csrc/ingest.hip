// Streamed frame ingest (SURVEY.md §8f "next #2"): the reference's front-end thread loads one image per iteration
// (Frame::Frame / Frame::init, Frame.cpp:31-42; featureExtractionThread, OdometryPipeline.cpp:212-220) and only then tracks it.
// Here decoded gray frames in HOST memory are moved to HBM by an ingest thread on a third HIP stream: pageable memory goes
// through a ring of pinned chunks (a registered / hipHostMalloc'ed source is DMA'ed in place), each chunk's pyramids are built on
// the same stream as soon as its copy lands, and an event per chunk lets the front-end stream wait for exactly the frames it is
// about to touch. Copies and pyramid builds overlap the front-end and back-end kernels, so a sequence handed over in host memory
// costs (almost) nothing over one that is already resident (bench.py: `pcie_inclusive` vs `value`).
#include "pmv_ctx.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace pmv {

struct Ingest {
    static constexpr int CHUNK = 16;   // frames per chunk: 7.5 MB at 1241x376 — large enough for DMA efficiency, small enough to start early
    static constexpr int NBUF = 4;     // pinned ring depth (NBUF * CHUNK == pmv_ctx::TIGHT_FRAMES: the device landing area is the same ring)
    hipStream_t stream = nullptr;
    uint8_t* h_ring = nullptr;         // NBUF * CHUNK * frame bytes, pinned (kept across runs)
    size_t ring_frame_bytes = 0;
    hipEvent_t buf_free[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ready;     // per chunk: copy + pyramids done
    // one open stream
    std::thread th;
    bool open = false;
    int first_slot = 0, n = 0, w = 0, h = 0;
    const uint8_t* src = nullptr;
    bool src_pinned = false;
    std::mutex mu;
    std::condition_variable cv;
    int chunks_enqueued = 0;           // guarded by mu
    int error = 0;                     // guarded by mu
    char err[256] = "";
    int last_waited = -1;              // front-end thread only
};

static void ingest_fail(Ingest* g, int code, const char* what, hipError_t e) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->error = code;
    snprintf(g->err, sizeof(g->err), "ingest: %s: %s", what, hipGetErrorString(e));
    g->cv.notify_all();
}

static void ingest_thread(pmv_ctx* ctx) {
    Ingest* g = ctx->ingest;
    tl_prof = &ctx->prof;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { ingest_fail(g, PMV_ERR_HIP, "hipSetDevice", e); return; }
    const size_t fb = (size_t)g->w * g->h;
    const PyrLayout L = layout_for(ctx, g->w, g->h);
    const int nchunks = (g->n + Ingest::CHUNK - 1) / Ingest::CHUNK;
    for (int c = 0; c < nchunks; c++) {
        const int f0 = c * Ingest::CHUNK, nb = std::min(Ingest::CHUNK, g->n - f0), buf = c % Ingest::NBUF;
        const uint8_t* hsrc = g->src + (size_t)f0 * fb;
        if (!g->src_pinned) {
            if (c >= Ingest::NBUF && (e = hipEventSynchronize(g->buf_free[buf])) != hipSuccess) { ingest_fail(g, PMV_ERR_HIP, "hipEventSynchronize", e); return; }
            uint8_t* stage = g->h_ring + (size_t)buf * Ingest::CHUNK * g->ring_frame_bytes;
            memcpy(stage, hsrc, (size_t)nb * fb);
            hsrc = stage;
        }
        // one contiguous copy per chunk into the landing area (a ring of NBUF chunk-sized regions, reused in stream order), level 0 from there
        uint8_t* land = ctx->d_tight + (size_t)buf * Ingest::CHUNK * fb;
        e = hipMemcpyAsync(land, hsrc, (size_t)nb * fb, hipMemcpyHostToDevice, g->stream);
        if (e != hipSuccess) { ingest_fail(g, PMV_ERR_HIP, "hipMemcpyAsync", e); return; }
        if ((e = hipEventRecord(g->buf_free[buf], g->stream)) != hipSuccess) { ingest_fail(g, PMV_ERR_HIP, "hipEventRecord", e); return; }
        if (build_levels_on(ctx, g->stream, g->first_slot + f0, nb, L, land) != PMV_OK) { ingest_fail(g, PMV_ERR_HIP, "pyramid launch", hipGetLastError()); return; }
        if ((e = hipEventRecord(g->ready[c], g->stream)) != hipSuccess) { ingest_fail(g, PMV_ERR_HIP, "hipEventRecord", e); return; }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            for (int i = 0; i < nb; i++) ctx->slot_layout[g->first_slot + f0 + i] = L;   // published with the counter below
            g->chunks_enqueued = c + 1;
        }
        g->cv.notify_all();
    }
}

int ingest_require(pmv_ctx* ctx, int slot) {
    Ingest* g = ctx->ingest;
    if (!g || !g->open || slot < g->first_slot || slot >= g->first_slot + g->n) return PMV_OK;
    const int c = (slot - g->first_slot) / Ingest::CHUNK;
    if (c <= g->last_waited) return PMV_OK;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return g->chunks_enqueued > c || g->error; });
        if (g->error) { set_err(ctx, "%s", g->err); return g->error; }
    }
    const hipError_t e = hipStreamWaitEvent(ctx->s_front, g->ready[c], 0);   // in-order stream: chunk c done implies every earlier one
    if (e != hipSuccess) { set_err(ctx, "hipStreamWaitEvent: %s", hipGetErrorString(e)); return PMV_ERR_HIP; }
    g->last_waited = c;
    return PMV_OK;
}

static void ingest_close(pmv_ctx* ctx) {
    Ingest* g = ctx->ingest;
    if (!g || !g->open) return;
    if (g->th.joinable()) g->th.join();
    (void)hipStreamSynchronize(g->stream);
    g->open = false;
}

void ingest_destroy(pmv_ctx* ctx) {
    Ingest* g = ctx->ingest;
    if (!g) return;
    ingest_close(ctx);
    for (auto& ev : g->ready) (void)hipEventDestroy(ev);
    for (auto& ev : g->buf_free) if (ev) (void)hipEventDestroy(ev);
    if (g->h_ring) (void)hipHostFree(g->h_ring);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    ctx->ingest = nullptr;
}

}  // namespace pmv

using namespace pmv;

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

extern "C" {

int pmv_frames_stream_begin(pmv_ctx* ctx, int first_slot, int n, const uint8_t* gray, int w, int h) {
    REQ(ctx && gray, PMV_ERR_INVALID, "pmv_frames_stream_begin: null argument");
    REQ(first_slot >= 0 && n >= 1 && first_slot + n <= ctx->n_slots, PMV_ERR_CAPACITY, "pmv_frames_stream_begin: slots [%d,%d) out of range (n_slots %d)", first_slot, first_slot + n, ctx->n_slots);
    REQ(w >= 40 && h >= 40 && w <= ctx->max_w && h <= ctx->max_h, PMV_ERR_CAPACITY, "pmv_frames_stream_begin: frame %dx%d outside capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
    REQ(!ctx->ingest || !ctx->ingest->open, PMV_ERR_INVALID, "pmv_frames_stream_begin: a stream is already open (pmv_frames_stream_end first)");
    CKC(hipSetDevice(ctx->device));
    if (!ctx->ingest) {
        ctx->ingest = new Ingest();
        CKC(hipStreamCreateWithFlags(&ctx->ingest->stream, hipStreamNonBlocking));
        for (auto& ev : ctx->ingest->buf_free) CKC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    Ingest* g = ctx->ingest;
    // frames in the slots about to be overwritten may still be read by work in flight on the front-end stream
    CKC(hipStreamSynchronize(ctx->s_front));
    const size_t fb = (size_t)w * h;
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, gray) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();   // a plain malloc'ed pointer makes hipPointerGetAttributes fail: that is the "pageable" answer
    if (!pinned && g->ring_frame_bytes < fb) {
        if (g->h_ring) { CKC(hipHostFree(g->h_ring)); g->h_ring = nullptr; }
        CKC(hipHostMalloc(&g->h_ring, (size_t)Ingest::NBUF * Ingest::CHUNK * fb));
        g->ring_frame_bytes = fb;
    }
    const int nchunks = (n + Ingest::CHUNK - 1) / Ingest::CHUNK;
    while ((int)g->ready.size() < nchunks) {
        hipEvent_t ev;
        CKC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->ready.push_back(ev);
    }
    g->first_slot = first_slot; g->n = n; g->w = w; g->h = h; g->src = gray; g->src_pinned = pinned;
    g->chunks_enqueued = 0; g->error = 0; g->last_waited = -1;
    for (int i = 0; i < n; i++) ctx->slot_layout[first_slot + i].n_levels = 0;   // not there yet
    g->open = true;
    g->th = std::thread(ingest_thread, ctx);
    return PMV_OK;
}

int pmv_frames_stream_end(pmv_ctx* ctx) {
    REQ(ctx, PMV_ERR_INVALID, "null ctx");
    if (!ctx->ingest || !ctx->ingest->open) return PMV_OK;
    CKC(hipSetDevice(ctx->device));
    ingest_close(ctx);
    REQ(ctx->ingest->error == 0, ctx->ingest->error, "%s", ctx->ingest->err);
    return PMV_OK;
}

}  // extern "C"
