// Batch engine (several independent sequences per GPU through "the same kernels with a leading batch dimension").
//
// A single sequence can never fill 256 CUs: frame k+1's tracks depend on frame k's result (~300 workgroups in flight) and the
// back-end is a serial chain of small solves. Sequences, however, are independent (the path shards by sequence). Here every
// sequence keeps the reference's own host structure — a front-end and a back-end thread running the unchanged adapters of
// host/vo_pipeline.cpp — but its plugin calls do not launch anything themselves: they hand a request to one of two COMBINER
// threads (front-end stream: LK + detectors; back-end stream: PnP + BA + two-view DLT) and sleep. A combiner takes whatever
// requests have accumulated while the previous launch was running, issues ONE batched launch per kernel class for all of them
// (k_lk_batch, k_gftt_* over the cells of several frames, k_pnp_*_batch, the k_bamB_* chain with the problem index in
// blockIdx.y, k_tri_dlt_batch), synchronises once and wakes the callers. Only the two combiner threads talk to the HIP runtime,
// so there is no runtime-lock contention and no per-sequence stream; the batch size adapts to the load by itself. Every
// block of a batched launch executes exactly the code and the block index of the single-sequence launch, so each sequence's
// results are bit-identical to its own single run (tests/test_batch_gpu.py).
#include "pmv_ctx.h"
#include "backend.h"
#include "batch_engine.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>

namespace pmv {

namespace {

struct Req {
    int kind = 0;          // 0 LK, 1 GFTT, 2 ShiTomasi | 10 PnP, 11 BA, 12 DLT
    int rc = PMV_OK;
    bool done = false;
    char err[200] = "";
    virtual ~Req() {}
};
struct LKReq : Req {
    int prev_slot, next_slot, n;
    const float* prev_xy; float* out_xy; uint8_t* status; float* err_out;
    std::vector<int> order;   // XCD-aware block -> track order of THIS request (local indices, -1 = padding), built by the caller
    int base = 0;             // filled by the combiner: first index in the concatenated arrays
};
struct DetReq : Req {
    int slot, n_cells, max_per_cell, unlimited;
    const int* cells; double quality, min_dist;
    int* out_xy; double* out_score; int* out_count;
    int cell_base = 0;
};
struct PnPReq : Req { BackendBuffers* b; PnPProblem P; size_t in_bytes; };
struct BAReq : Req { BackendBuffers* b; BAArgs A; size_t io_bytes; int max_iterations; };
struct DltReq : Req { BackendBuffers* b; DltProblem P; size_t in_bytes; };

struct Growable {   // device (or pinned host) buffer that only grows
    void* p = nullptr; size_t cap = 0; bool host = false;
    hipError_t ensure(size_t need) {
        if (need <= cap) return hipSuccess;
        if (p) { hipError_t e = host ? hipHostFree(p) : hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        need = (need * 5 / 4 + 4095) & ~(size_t)4095;
        hipError_t e = host ? hipHostMalloc(&p, need) : hipMalloc(&p, need);
        if (e == hipSuccess) cap = need;
        return e;
    }
    void release() { if (p) { (void)(host ? hipHostFree(p) : hipFree(p)); p = nullptr; cap = 0; } }
};

struct Combiner {
    std::mutex mu;
    std::condition_variable cv_new, cv_done;
    std::vector<Req*> pending;
    bool stop = false;
    std::thread th;
    long batches = 0, requests = 0;
};

}  // namespace

struct BatchEngine {
    pmv_ctx* ctx = nullptr;
    int B = 0;
    int linger_us = 0;
    std::vector<BackendBuffers*> slots;   // one back-end workspace set per concurrent sequence
    Combiner front, back;
    // front staging
    Growable h_front{nullptr, 0, true}, d_front, h_cells{nullptr, 0, true}, d_cells, d_eig, d_cellmax, d_det_xy, d_det_score, d_det_count, h_det{nullptr, 0, true};
    float* h_out_xy = nullptr; float* h_err = nullptr; uint8_t* h_status = nullptr;    // mapped pinned LK results
    float* dm_out_xy = nullptr; float* dm_err = nullptr; uint8_t* dm_status = nullptr;
    size_t cap_tracks = 0;
    int* d_flags = nullptr;
    // back descriptors
    Growable h_desc{nullptr, 0, true}, d_desc;
};

namespace {

void fail_all(std::vector<Req*>& batch, int code, const char* what, hipError_t e) {
    for (Req* r : batch) { r->rc = code; snprintf(r->err, sizeof(r->err), "batch engine: %s: %s", what, hipGetErrorString(e)); }
}
#define EK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail_all(batch, PMV_ERR_HIP, #x, e_); return; } } while (0)

// ---- front-end stream ---------------------------------------------------------------------------------------------------------
void process_front(BatchEngine* E, std::vector<Req*>& batch) {
    pmv_ctx* ctx = E->ctx;
    hipStream_t s = ctx->s_front;
    std::vector<LKReq*> lk;
    std::vector<DetReq*> det;
    for (Req* r : batch) { if (r->kind == 0) lk.push_back((LKReq*)r); else det.push_back((DetReq*)r); }
    // ---- LK: one launch for the tracks of every requesting sequence
    int total_tracks = 0, total_blocks = 0;
    PyrLayout L{};
    bool have_L = false;
    for (LKReq* r : lk) {
        const PyrLayout& a = ctx->slot_layout[r->prev_slot];
        const PyrLayout& b2 = ctx->slot_layout[r->next_slot];
        if (a.n_levels <= 0 || b2.n_levels <= 0 || a.w[0] != b2.w[0] || a.h[0] != b2.h[0]) { r->rc = PMV_ERR_INVALID; snprintf(r->err, sizeof(r->err), "batch LK: slot has no pyramid / sizes differ"); continue; }
        if (!have_L) { L = a; have_L = true; }
        else if (a.w[0] != L.w[0] || a.h[0] != L.h[0]) { r->rc = PMV_ERR_INVALID; snprintf(r->err, sizeof(r->err), "batch LK: all sequences of a batch must share the frame size"); continue; }
        r->base = total_tracks;
        total_tracks += r->n;
        total_blocks += (int)r->order.size();
    }
    if (total_tracks > 0) {
        if ((size_t)total_tracks > E->cap_tracks) { fail_all(batch, PMV_ERR_CAPACITY, "more tracks than B * max_tracks", hipSuccess); return; }
        const size_t off_blocks = (sizeof(LKSeq) * lk.size() + 63) & ~(size_t)63;
        const size_t off_xy = (off_blocks + sizeof(int2) * (size_t)total_blocks + 63) & ~(size_t)63;
        const size_t bytes = off_xy + (size_t)total_tracks * 8;
        EK(E->h_front.ensure(bytes)); EK(E->d_front.ensure(bytes));
        char* hb = (char*)E->h_front.p;
        LKSeq* hseq = (LKSeq*)hb;
        int2* hblk = (int2*)(hb + off_blocks);
        float* hxy = (float*)(hb + off_xy);
        int q = 0, bpos = 0;
        for (LKReq* r : lk) {
            if (r->rc != PMV_OK) continue;
            hseq[q].prev_off = (unsigned long long)r->prev_slot * L.slot_bytes;
            hseq[q].next_off = (unsigned long long)r->next_slot * L.slot_bytes;
            for (int o : r->order) { hblk[bpos].x = q; hblk[bpos].y = o < 0 ? -1 : r->base + o; bpos++; }
            memcpy(hxy + (size_t)2 * r->base, r->prev_xy, (size_t)r->n * 8);
            q++;
        }
        EK(hipMemcpyAsync(E->d_front.p, hb, bytes, hipMemcpyHostToDevice, s));
        LKParams P;
        P.max_iter = 30; P.eps2 = 1e-4f; P.eps2d = 0.01 * 0.01; P.min_eig = 1e-4f; P.stamps = nullptr; P.counters = ctx->d_lk_counters;
        char* db = (char*)E->d_front.p;
        EK(launch_lk_batch(s, ctx->d_slots, (const LKSeq*)db, (const int2*)(db + off_blocks), total_blocks, L, (const float*)(db + off_xy), P,
                           E->dm_out_xy, E->dm_status, E->dm_err));
    }
    // ---- detectors: requests with the same parameters share a launch (cells of several frames)
    struct Group { int kind, max_per_cell, unlimited; double quality, min_dist; std::vector<DetReq*> reqs; int n_cells = 0; size_t out_off = 0; };
    std::vector<Group> groups;
    for (DetReq* r : det) {
        Group* g = nullptr;
        for (Group& x : groups)
            if (x.kind == r->kind && x.max_per_cell == r->max_per_cell && x.unlimited == r->unlimited && x.quality == r->quality && x.min_dist == r->min_dist) { g = &x; break; }
        if (!g) { groups.push_back(Group{r->kind, r->max_per_cell, r->unlimited, r->quality, r->min_dist, {}, 0, 0}); g = &groups.back(); }
        r->cell_base = g->n_cells;
        g->n_cells += r->n_cells;
        g->reqs.push_back(r);
    }
    if (!groups.empty()) {
        if (!have_L) L = ctx->slot_layout[det[0]->slot];
        size_t tot_cells = 0, tot_out = 0;
        for (Group& g : groups) { g.out_off = tot_out; tot_cells += g.n_cells; tot_out += (size_t)g.n_cells * g.max_per_cell; }
        EK(E->h_cells.ensure(tot_cells * CELL_STRIDE * 4)); EK(E->d_cells.ensure(tot_cells * CELL_STRIDE * 4));
        EK(E->d_eig.ensure(tot_cells * CELL_PIX * sizeof(double))); EK(E->d_cellmax.ensure(tot_cells * 8));
        EK(E->d_det_xy.ensure(tot_out * 8)); EK(E->d_det_score.ensure(tot_out * 8)); EK(E->d_det_count.ensure(tot_cells * 4));
        EK(E->h_det.ensure(tot_out * 16 + tot_cells * 4 + 64));
        int* hc = (int*)E->h_cells.p;
        size_t cpos = 0;
        for (Group& g : groups)
            for (DetReq* r : g.reqs)
                for (int i = 0; i < r->n_cells; i++, cpos++) {
                    int* d = hc + cpos * CELL_STRIDE;
                    d[0] = r->cells[4 * i]; d[1] = r->cells[4 * i + 1]; d[2] = r->cells[4 * i + 2]; d[3] = r->cells[4 * i + 3]; d[4] = r->slot; d[5] = d[6] = d[7] = 0;
                }
        EK(hipMemcpyAsync(E->d_cells.p, hc, tot_cells * CELL_STRIDE * 4, hipMemcpyHostToDevice, s));
        EK(hipMemsetAsync(E->d_flags, 0, 16, s));
        size_t c0 = 0;
        char* hd = (char*)E->h_det.p;
        for (Group& g : groups) {
            const int* dc = (const int*)E->d_cells.p + c0 * CELL_STRIDE;
            int* dxy = (int*)E->d_det_xy.p + g.out_off * 2;
            double* dsc = (double*)E->d_det_score.p + g.out_off;
            int* dcnt = (int*)E->d_det_count.p + c0;
            if (g.kind == 1)
                EK(launch_gftt(s, ctx->d_slots, L, dc, g.n_cells, g.max_per_cell, g.quality, g.min_dist, g.unlimited, (float*)E->d_eig.p + c0 * CELL_PIX,
                               (unsigned*)E->d_cellmax.p + c0, dxy, dcnt, E->d_flags));
            else
                EK(launch_shitomasi(s, ctx->d_slots, L, dc, g.n_cells, g.max_per_cell, g.quality, (double*)E->d_eig.p + c0 * CELL_PIX,
                                    (unsigned long long*)E->d_cellmax.p + c0, dxy, dsc, dcnt, E->d_flags));
            c0 += g.n_cells;
        }
        EK(hipMemcpyAsync(hd, E->d_det_xy.p, tot_out * 8, hipMemcpyDeviceToHost, s));
        EK(hipMemcpyAsync(hd + tot_out * 8, E->d_det_score.p, tot_out * 8, hipMemcpyDeviceToHost, s));
        EK(hipMemcpyAsync(hd + tot_out * 16, E->d_det_count.p, tot_cells * 4, hipMemcpyDeviceToHost, s));
        EK(hipMemcpyAsync(hd + tot_out * 16 + tot_cells * 4, E->d_flags, 4, hipMemcpyDeviceToHost, s));
    }
    EK(hipStreamSynchronize(s));
    for (LKReq* r : lk) {
        if (r->rc != PMV_OK) continue;
        memcpy(r->out_xy, E->h_out_xy + (size_t)2 * r->base, (size_t)r->n * 8);
        memcpy(r->status, E->h_status + r->base, (size_t)r->n);
        memcpy(r->err_out, E->h_err + r->base, (size_t)r->n * 4);
    }
    if (!groups.empty()) {
        size_t tot_cells = 0, tot_out = 0;
        for (Group& g : groups) { tot_cells += g.n_cells; tot_out += (size_t)g.n_cells * g.max_per_cell; }
        const char* hd = (const char*)E->h_det.p;
        const int flags = *(const int*)(hd + tot_out * 16 + tot_cells * 4);
        const int* hxy = (const int*)hd;
        const double* hsc = (const double*)(hd + tot_out * 8);
        const int* hcnt = (const int*)(hd + tot_out * 16);
        size_t c0 = 0;
        for (Group& g : groups) {
            for (DetReq* r : g.reqs) {
                if ((g.kind == 1 && (flags & 5)) || (g.kind == 2 && (flags & 2))) { r->rc = PMV_ERR_OVERFLOW; snprintf(r->err, sizeof(r->err), "detector candidate list overflow (batched launch)"); continue; }
                const size_t o = g.out_off + (size_t)r->cell_base * g.max_per_cell;
                memcpy(r->out_xy, hxy + o * 2, (size_t)r->n_cells * g.max_per_cell * 8);
                if (r->out_score) memcpy(r->out_score, hsc + o, (size_t)r->n_cells * g.max_per_cell * 8);
                memcpy(r->out_count, hcnt + c0 + r->cell_base, (size_t)r->n_cells * 4);
            }
            c0 += g.n_cells;
        }
    }
}

// ---- back-end stream ----------------------------------------------------------------------------------------------------------
void process_back(BatchEngine* E, std::vector<Req*>& batch) {
    pmv_ctx* ctx = E->ctx;
    hipStream_t s = ctx->s_back;
    std::vector<PnPReq*> pnp; std::vector<BAReq*> ba; std::vector<DltReq*> dlt;
    for (Req* r : batch) { if (r->kind == 10) pnp.push_back((PnPReq*)r); else if (r->kind == 11) ba.push_back((BAReq*)r); else dlt.push_back((DltReq*)r); }
    const size_t off_ba = (sizeof(PnPProblem) * pnp.size() + 255) & ~(size_t)255;
    const size_t off_dlt = (off_ba + sizeof(BAProb) * ba.size() + 255) & ~(size_t)255;
    const size_t desc_bytes = off_dlt + sizeof(DltProblem) * dlt.size();
    EK(E->h_desc.ensure(desc_bytes + 256)); EK(E->d_desc.ensure(desc_bytes + 256));
    char* hd = (char*)E->h_desc.p;
    char* dd = (char*)E->d_desc.p;
    int max_hyp = 0;
    for (size_t i = 0; i < pnp.size(); i++) {
        PnPReq* r = pnp[i];
        EK(hipMemcpyAsync(r->b->d_pnp_in, r->b->h_stage, r->in_bytes, hipMemcpyHostToDevice, s));
        ((PnPProblem*)hd)[i] = r->P;
        max_hyp = std::max(max_hyp, r->P.n_hyp);
    }
    // BA: one chain per distinct iteration cap (in practice one)
    std::vector<int> ba_iters;
    for (BAReq* r : ba) if (std::find(ba_iters.begin(), ba_iters.end(), r->max_iterations) == ba_iters.end()) ba_iters.push_back(r->max_iterations);
    std::vector<BAReq*> ba_sorted;
    for (int it : ba_iters) for (BAReq* r : ba) if (r->max_iterations == it) ba_sorted.push_back(r);
    for (size_t i = 0; i < ba_sorted.size(); i++) {
        BAReq* r = ba_sorted[i];
        EK(hipMemcpyAsync(r->b->d_ba_io, r->b->h_stage, r->io_bytes, hipMemcpyHostToDevice, s));
        ba_fill_prob(((BAProb*)(hd + off_ba))[i], r->A, r->b->d_bastate, r->b->d_bapart);
    }
    int max_n = 0;
    for (size_t i = 0; i < dlt.size(); i++) {
        DltReq* r = dlt[i];
        EK(hipMemcpyAsync(r->b->d_tri_in, r->b->h_stage, r->in_bytes, hipMemcpyHostToDevice, s));
        ((DltProblem*)(hd + off_dlt))[i] = r->P;
        max_n = std::max(max_n, r->P.n);
    }
    EK(hipMemcpyAsync(dd, hd, desc_bytes, hipMemcpyHostToDevice, s));
    if (!pnp.empty()) EK(launch_pnp_batch(s, (const PnPProblem*)dd, (int)pnp.size(), max_hyp));
    size_t i0 = 0;
    for (int it : ba_iters) {
        BABatchDims D{0, 0, 0, 0, 0, 0, it};
        size_t i1 = i0;
        while (i1 < ba_sorted.size() && ba_sorted[i1]->max_iterations == it) {
            const BAProb& P = ((const BAProb*)(hd + off_ba))[i1];
            D.max_eval_blocks = std::max(D.max_eval_blocks, P.nbo + P.clear_blocks);
            D.max_nc = std::max(D.max_nc, P.A.nc); D.max_nbp = std::max(D.max_nbp, P.nbp); D.max_tiles = std::max(D.max_tiles, P.tiles);
            D.max_m = std::max(D.max_m, 6 * P.A.nc);
            i1++;
        }
        D.n_probs = (int)(i1 - i0);
        EK(launch_ba_multi_batch(s, (const BAProb*)(dd + off_ba) + i0, D));
        i0 = i1;
    }
    if (!dlt.empty()) EK(launch_tri_dlt_batch(s, (const DltProblem*)(dd + off_dlt), (int)dlt.size(), max_n));
    EK(hipStreamSynchronize(s));   // every kernel wrote its results straight into the requests' pinned blocks
}

void combiner_loop(BatchEngine* E, Combiner* C, bool is_front) {
    (void)hipSetDevice(E->ctx->device);
    tl_prof = &E->ctx->prof;
    for (;;) {
        std::vector<Req*> batch;
        {
            std::unique_lock<std::mutex> lk(C->mu);
            C->cv_new.wait(lk, [&] { return !C->pending.empty() || C->stop; });
            if (C->pending.empty() && C->stop) return;
            if (E->linger_us > 0 && (int)C->pending.size() < E->B) {   // optional: give stragglers a moment to join the batch
                lk.unlock();
                std::this_thread::sleep_for(std::chrono::microseconds(E->linger_us));
                lk.lock();
            }
            batch.swap(C->pending);
        }
        if (is_front) process_front(E, batch); else process_back(E, batch);
        {
            std::lock_guard<std::mutex> lk(C->mu);
            for (Req* r : batch) r->done = true;
            C->batches++; C->requests += (long)batch.size();
        }
        C->cv_done.notify_all();
    }
}

int submit(pmv_ctx* ctx, Combiner& C, Req* r) {
    {
        std::unique_lock<std::mutex> lk(C.mu);
        C.pending.push_back(r);
        C.cv_new.notify_one();
        C.cv_done.wait(lk, [&] { return r->done; });
    }
    if (r->rc != PMV_OK) set_err(ctx, "%s", r->err);
    return r->rc;
}

}  // namespace

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

void batch_engine_destroy(pmv_ctx* ctx) {
    BatchEngine* E = ctx->engine;
    if (!E) return;
    for (Combiner* C : {&E->front, &E->back}) {
        { std::lock_guard<std::mutex> lk(C->mu); C->stop = true; }
        C->cv_new.notify_all();
        if (C->th.joinable()) C->th.join();
    }
    for (BackendBuffers* b : E->slots) backend_free(b);
    for (Growable* g : {&E->h_front, &E->d_front, &E->h_cells, &E->d_cells, &E->d_eig, &E->d_cellmax, &E->d_det_xy, &E->d_det_score, &E->d_det_count, &E->h_det,
                        &E->h_desc, &E->d_desc}) g->release();
    if (E->h_out_xy) (void)hipHostFree(E->h_out_xy);
    if (E->h_err) (void)hipHostFree(E->h_err);
    if (E->h_status) (void)hipHostFree(E->h_status);
    if (E->d_flags) (void)hipFree(E->d_flags);
    delete E;
    ctx->engine = nullptr;
}

int batch_engine_get(pmv_ctx* ctx, int B, BatchEngine** out) {
    REQ(B >= 1 && B <= 256, PMV_ERR_CAPACITY, "batch size %d (1..256)", B);
    if (ctx->engine && ctx->engine->B >= B) { *out = ctx->engine; return PMV_OK; }
    batch_engine_destroy(ctx);
    CKC(hipSetDevice(ctx->device));
    BatchEngine* E = new BatchEngine();
    ctx->engine = E;
    E->ctx = ctx; E->B = B;
    if (const char* e = getenv("PMV_BATCH_LINGER_US")) E->linger_us = atoi(e);
    for (int i = 0; i < B; i++) {
        BackendBuffers* b = nullptr;
        const int rc = backend_alloc(ctx, &b);
        if (b) E->slots.push_back(b);
        if (rc != PMV_OK) { batch_engine_destroy(ctx); return rc; }
    }
    E->cap_tracks = (size_t)B * ctx->max_tracks;
    CKC(hipHostMalloc(&E->h_out_xy, E->cap_tracks * 8, hipHostMallocMapped));
    CKC(hipHostMalloc(&E->h_status, E->cap_tracks, hipHostMallocMapped));
    CKC(hipHostMalloc(&E->h_err, E->cap_tracks * 4, hipHostMallocMapped));
    CKC(hipHostGetDevicePointer((void**)&E->dm_out_xy, E->h_out_xy, 0));
    CKC(hipHostGetDevicePointer((void**)&E->dm_status, E->h_status, 0));
    CKC(hipHostGetDevicePointer((void**)&E->dm_err, E->h_err, 0));
    CKC(hipMalloc(&E->d_flags, 16));
    E->front.th = std::thread(combiner_loop, E, &E->front, true);
    E->back.th = std::thread(combiner_loop, E, &E->back, false);
    *out = E;
    return PMV_OK;
}

void batch_engine_stats(BatchEngine* E, long* out4) {
    out4[0] = E->front.batches; out4[1] = E->front.requests; out4[2] = E->back.batches; out4[3] = E->back.requests;
}

// ---- request entry points (called from the sequences' own host threads) ---------------------------------------------------------
int engine_lk(BatchEngine* E, int prev_slot, int next_slot, const float* prev_xy, int n, float* out_xy, uint8_t* status, float* err) {
    pmv_ctx* ctx = E->ctx;
    REQ(n >= 0 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_lk_track: n=%d exceeds max_tracks=%d", n, ctx->max_tracks);
    REQ(prev_slot >= 0 && prev_slot < ctx->n_slots && next_slot >= 0 && next_slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_lk_track: slot out of range");
    if (n == 0) return PMV_OK;
    LKReq r;
    r.kind = 0; r.prev_slot = prev_slot; r.next_slot = next_slot; r.n = n; r.prev_xy = prev_xy; r.out_xy = out_xy; r.status = status; r.err_out = err;
    // the same XCD-aware dealing as pmv_lk_track (stripe s of the x-sorted tracks -> blocks 8k + s); a request's block range starts at
    // a multiple of 8 in the concatenated launch, so block % 8 (= XCD) is preserved
    const int nb = (n + 7) / 8 * 8;
    r.order.assign(nb, -1);
    std::vector<std::pair<float, int>> byx(n);
    for (int i = 0; i < n; i++) byx[i] = {prev_xy[2 * i], i};
    std::sort(byx.begin(), byx.end());
    for (int i = 0; i < n; i++) {
        const int s8 = (int)((long)i * 8 / n), first = (int)(((long)s8 * n + 7) / 8);
        r.order[(i - first) * 8 + s8] = byx[i].second;
    }
    return submit(ctx, E->front, &r);
}

int engine_detect(BatchEngine* E, int kind, int slot, const int* cells, int n_cells, int max_per_cell, double quality, double min_dist, int* out_xy,
                  double* out_score, int* out_count) {
    pmv_ctx* ctx = E->ctx;
    REQ(cells && out_xy && out_count && n_cells >= 1 && n_cells <= MAX_CELLS, PMV_ERR_INVALID, "detect: bad argument");
    if (kind == 2 && max_per_cell <= 0) { for (int i = 0; i < n_cells; i++) out_count[i] = 0; return PMV_OK; }
    DetReq r;
    r.kind = kind; r.slot = slot; r.cells = cells; r.n_cells = n_cells; r.unlimited = max_per_cell <= 0;
    r.max_per_cell = r.unlimited ? MAX_PER_CELL : max_per_cell;
    REQ(r.max_per_cell <= MAX_PER_CELL, PMV_ERR_CAPACITY, "detect: max_per_cell=%d (max %d)", max_per_cell, MAX_PER_CELL);
    REQ(slot >= 0 && slot < ctx->n_slots && ctx->slot_layout[slot].n_levels > 0, PMV_ERR_INVALID, "detect: slot %d has no frame", slot);
    const PyrLayout& L = ctx->slot_layout[slot];
    for (int i = 0; i < n_cells; i++) {
        const int* c = cells + 4 * i;
        REQ(c[2] >= 3 && c[3] >= 3 && c[2] <= CELL_MAX && c[3] <= CELL_MAX && c[0] >= 0 && c[1] >= 0 && c[0] + c[2] <= L.w[0] && c[1] + c[3] <= L.h[0],
            PMV_ERR_INVALID, "detect: cell %d invalid", i);
    }
    r.quality = quality; r.min_dist = min_dist; r.out_xy = out_xy; r.out_score = out_score; r.out_count = out_count;
    return submit(ctx, E->front, &r);
}

int engine_pnp(BatchEngine* E, int seq, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec, int iterations,
               float reproj_err, double confidence, int* out_inliers, int* out_n_inliers) {
    pmv_ctx* ctx = E->ctx;
    int rc = pnp_check(ctx, obj_xyz, img_xy, m, K, rvec, tvec, iterations, confidence, out_inliers, out_n_inliers);
    if (rc) return rc;
    PnPReq r;
    r.kind = 10; r.b = E->slots[seq];
    pnp_prepare(r.b, obj_xyz, img_xy, m, K, iterations, reproj_err, confidence, &r.P, &r.in_bytes);
    rc = submit(ctx, E->back, &r);
    if (rc) return rc;
    pnp_finish(ctx, r.b, obj_xyz, img_xy, m, K, rvec, tvec, iterations, reproj_err, confidence, r.in_bytes, out_inliers, out_n_inliers);
    return PMV_OK;
}

int engine_ba(BatchEngine* E, int seq, double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx, const int* pt_idx, int n_obs,
              const double* K, double huber, int max_iterations) {
    pmv_ctx* ctx = E->ctx;
    int rc = ba_check(ctx, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations);
    if (rc) return rc;
    if (max_iterations == 0) return PMV_OK;
    BAReq r;
    r.kind = 11; r.b = E->slots[seq]; r.max_iterations = max_iterations;
    rc = ba_prepare(ctx, r.b, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, true, &r.A, &r.io_bytes);
    if (rc) return rc;
    rc = submit(ctx, E->back, &r);
    if (rc) return rc;
    ba_finish(ctx, r.b, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, nullptr);
    return PMV_OK;
}

int engine_dlt(BatchEngine* E, int seq, const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
               uint8_t* out_mask, int* out_good) {
    pmv_ctx* ctx = E->ctx;
    REQ(n >= 1 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_triangulate_candidates: n=%d (1..max_tracks=%d)", n, ctx->max_tracks);
    DltReq r;
    r.kind = 12; r.b = E->slots[seq];
    dlt_prepare(r.b, q1, q2, n, P1x4, mask_in, &r.P, &r.in_bytes);
    const int rc = submit(ctx, E->back, &r);
    if (rc) return rc;
    dlt_finish(ctx, r.b, q1, q2, n, P1x4, mask_in, r.in_bytes, out_Q, out_mask, out_good);
    return PMV_OK;
}

}  // namespace pmv
