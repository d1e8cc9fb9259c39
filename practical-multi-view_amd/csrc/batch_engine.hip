// Batch engine (several independent sequences per GPU through "the same kernels with a leading batch dimension").
//
// A single sequence can never fill 256 CUs: frame k+1's tracks depend on frame k's result (~300 workgroups in flight) and the
// back-end is a serial chain of small solves. Sequences, however, are independent (the path shards by sequence). Here every
// sequence keeps the reference's own host structure — a front-end and a back-end thread running the unchanged adapters of
// host/vo_pipeline.cpp — but its plugin calls do not launch anything themselves: they hand a request to the COMBINER thread of
// their kernel class (LK, detectors, PnP, BA, two-view DLT; one HIP stream each, so the classes overlap on the GPU) and sleep. A
// combiner takes whatever requests have accumulated while its previous launch was running, issues ONE batched launch for all of
// them (k_lk_batch, k_gftt_* / k_st_* over the cells of several frames, k_pnp_*_batch, the k_bamB_* chain with the problem index
// in blockIdx.y, k_tri_dlt_batch), synchronises once and wakes exactly the callers it served. Only the five combiner threads talk
// to the HIP runtime: no runtime-lock contention, no per-sequence stream; the batch size adapts to the load by itself. Inputs that
// the callers prepared in their pinned blocks are pulled into HBM by one gather kernel per batch (k_stage_in) instead of one DMA
// per request. Every block of a batched launch executes exactly the code and the block index of the single-sequence launch, so
// each sequence's results are bit-identical to its own single run (tests/test_batch_gpu.py).
#include "pmv_ctx.h"
#include "backend.h"
#include "batch_engine.h"
#include <sys/prctl.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>
#include <time.h>

namespace pmv {

namespace {

struct Req {
    int kind = 0;          // 0 LK, 1 GFTT, 2 ShiTomasi | 10 PnP, 11 BA, 12 DLT
    int rc = PMV_OK;
    // completion word: the owner sleeps on it (futex), the combiner stores 1 and wakes that one sleeper. No lock is involved: with a
    // condition variable under the queue's mutex the 50-70 owners of a round woke up one by one into a fight for that mutex, while the
    // next requests were waiting to get in through the same mutex.
    std::atomic<int> done{0};
    char err[200] = "";
    virtual ~Req() {}
};
struct LKReq : Req {
    int prev_slot, next_slot, n;
    const float* prev_xy; float* out_xy; uint8_t* status; float* err_out;
    uint8_t* iters_out = nullptr;   // optional: LK iterations each track took (the caller's ordering hint for its next request)
    std::vector<int> order;   // block -> track order of THIS request (local indices, -1 = padding), built by the caller
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
struct FPReq : Req { BackendBuffers* b; FivePointProblem P; size_t in_bytes; };

struct Growable {   // device (or mapped pinned host) buffer that only grows; dev = the address kernels use (alias of a host buffer)
    void* p = nullptr; size_t cap = 0; bool host = false; char* dev = nullptr;
    hipError_t ensure(size_t need) {
        if (need <= cap) return hipSuccess;
        if (p) { hipError_t e = host ? hipHostFree(p) : hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        need = (need * 5 / 4 + 4095) & ~(size_t)4095;
        hipError_t e = host ? hipHostMalloc(&p, need, hipHostMallocMapped | hipHostMallocCoherent) : hipMalloc(&p, need);
        if (e != hipSuccess) return e;
        cap = need;
        dev = (char*)p;
        if (host) e = hipHostGetDevicePointer((void**)&dev, p, 0);
        return e;
    }
    void release() { if (p) { (void)(host ? hipHostFree(p) : hipFree(p)); p = nullptr; cap = 0; } }
};

enum Role { R_LK = 0, R_DET, R_PNP, R_BA, R_DLT, R_FP, R_COUNT };
struct Queue {   // pending requests of one kernel class
    std::mutex mu;
    std::condition_variable cv_new;
    std::vector<Req*> pending;
    std::chrono::steady_clock::time_point first_arrival;   // when `pending` last went from empty to non-empty
    int min_batch = 0;     // copy of BatchEngine::min_batch of the class: submit() wakes a combiner at the first and at the min_batch-th request only
    bool stop = false;
};
constexpr int MAX_LANES = 4;
// One combiner = one thread + one HIP stream + its staging buffers. A class may have several (PMV_BATCH_LANES, default 1): while
// one waits for its launch, the next takes the requests that have arrived meanwhile instead of letting them sit for a whole round.
struct Combiner {
    std::thread th;
    hipStream_t s = nullptr;           // this combiner's stream (lanes of a class may share one: BatchEngine::streams)
    bool owns_stream = true;
    hipEvent_t ev = nullptr;           // blocking-sync event for the interrupt-driven wait
    Growable h_desc{nullptr, 0, true}, d_desc;   // per-batch descriptors (+ stage-in jobs), pinned mirror and device copy
    long batches = 0, requests = 0;
    double t_idle = 0, t_work = 0, t_sync = 0;   // seconds: waiting for requests / processing a batch / inside hipStreamSynchronize
    double t_cpu = 0;                            // CPU seconds of the combiner thread itself
    // LK staging + mapped pinned result blocks; detector buffers (only used by combiners of those classes)
    Growable h_front{nullptr, 0, true}, d_front, h_cells{nullptr, 0, true}, d_cells, d_eig, d_cellmax, d_spill, d_det_xy, d_det_score, d_det_count, h_det{nullptr, 0, true};
    float* h_out_xy = nullptr; float* h_err = nullptr; uint8_t* h_status = nullptr; uint16_t* h_work = nullptr;
    float* dm_out_xy = nullptr; float* dm_err = nullptr; uint8_t* dm_status = nullptr; uint16_t* dm_work = nullptr;
    int* d_flags = nullptr;
    // completion word of the "flag" wait: the last launch of a round is k_signal, which stores the round number into mapped pinned memory
    unsigned* h_done = nullptr; unsigned* dm_done = nullptr; unsigned done_seq = 0;
    double ema_wait_us = 0;            // smoothed duration of the wait of a round (how long to sleep before the first look)
    long n_polls = 0; double t_first_sleep = 0, t_prep = 0, t_post = 0;   // diagnostic (PMV_BATCH_TIMING=1, printed when the engine goes)
};

__global__ void k_signal(unsigned* done, unsigned seq) { BACKEND_PRIO(); __threadfence_system(); __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// one input block to pull from mapped pinned host memory into HBM (16-byte granules; both buffers have >= 16 B of slack)
struct StageJob { const char* src; char* dst; unsigned bytes, pad; };
__global__ __launch_bounds__(256) void k_stage_in(const StageJob* __restrict__ jobs) { BACKEND_PRIO();
    const StageJob j = jobs[blockIdx.y];
    const unsigned n16 = (j.bytes + 15u) >> 4;
    const uint4* __restrict__ src = (const uint4*)j.src;
    uint4* __restrict__ dst = (uint4*)j.dst;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

struct BatchEngine {
    pmv_ctx* ctx = nullptr;
    int B = 0;
    int linger_us = 0;
    // Round-forming policy: a combiner that finds fewer than min_batch[class] requests waiting gives the others up to max_linger_us
    // (counted from the arrival of the oldest one) to join. Without it the classes have a second, stable operating point - lanes firing as
    // soon as anything is pending, rounds of a third of the size, the fixed cost of a round (launches, completion signal, wake-ups) paid
    // three times as often: the same binary then ran at 53 k instead of 65-69 k frames/s in one run out of five (profiles/r03_batch_exp_af.log).
    int min_batch[R_COUNT] = {0, 0, 0, 0, 0, 0};
    int max_linger_us = 300;
    int wait_mode = 3;   // 0 spin (hipStreamSynchronize), 1 query + yield, 2 blocking event, 3 completion word + timed sleeps
    std::vector<BackendBuffers*> slots;   // one back-end workspace set per concurrent sequence
    // combiners (thread + stream) per class. Round 2, host-bound: 2 and 3 per class cost more host CPU (smaller batches) than they won in
    // latency (25.7k -> 23.3k -> 19.5k frames/s at B = 64). Round 3, with the track tables the host has headroom and the LK class is the one
    // that is busy all the time - a launch ends with its slowest track, so a single LK stream idles most SIMDs during every tail: PMV_BATCH_LANES
    // sets all classes, PMV_BATCH_LANES_LK / _PNP / _BA one class.
    int lanes[R_COUNT] = {1, 1, 1, 1, 1, 1};
    // HIP streams per class (<= lanes): lane l launches on stream l % streams. Lanes that share a stream overlap their HOST halves (forming a
    // round, scattering its results, waking the owners) with each other's kernels while the kernels themselves run one after the other
    // with every wave slot of the class to themselves; lanes on different streams also overlap their kernels.
    int streams[R_COUNT] = {0, 0, 0, 0, 0, 0};   // 0 = one per lane
    Queue queue[R_COUNT];
    Combiner comb[R_COUNT][MAX_LANES];
    size_t cap_tracks = 0;
    bool lk_lpt = true;       // PMV_LK_LPT=0: the round-2 block order (x-sorted stripes per XCD, request after request)
    bool exclusive = false;
    std::mutex exclusive_mu;
    // Pyramids of a batched run are built WHILE the sequences already track (engine_build_begin): round r = frames [r * BUILD_CHUNK,
    // (r + 1) * BUILD_CHUNK) of every sequence, enqueued round by round on the context's front-end stream with an event after each.
    // A front-end launch that touches slot s first makes its stream wait for the event of slot_round[s] (a GPU-side dependency; the host
    // only waits until that round has been ENQUEUED, which is milliseconds after the start). Before this the 1.1 ms of pyramid kernels
    // per sequence ran back to back in front of everything: 141 ms of a 3 s pass at B = 128 with nothing else on the GPU.
    static constexpr int BUILD_CHUNK = 32;
    std::vector<int> slot_round;            // per frame slot: its build round, -1 = nothing to wait for
    std::vector<hipEvent_t> build_ev;       // per round
    std::atomic<int> build_enqueued{0};     // rounds whose launches and event are in the stream
    std::atomic<int> build_error{0};
    std::thread build_thread;
};

namespace {

void fail_all(std::vector<Req*>& batch, int code, const char* what, hipError_t e) {
    for (Req* r : batch) { r->rc = code; snprintf(r->err, sizeof(r->err), "batch engine: %s: %s", what, hipGetErrorString(e)); }
}
#define EK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail_all(batch, PMV_ERR_HIP, #x, e_); return; } } while (0)

// Waiting for a batch: hipStreamSynchronize spins on a host core; with five combiners and 2 B sequence threads on a 16-core share
// the cores are better spent on the sequences' host work, so the default is an interrupt-driven wait on a blocking event
// (PMV_BATCH_WAIT=spin | yield | block).
hipError_t wait_stream(BatchEngine* E, Combiner& C);
#define SYNC_TIMED(C) do { const auto t0_ = std::chrono::steady_clock::now(); EK(wait_stream(E, (C))); (C).t_sync += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); } while (0)

hipError_t wait_stream(BatchEngine* E, Combiner& C) {
    if (E->wait_mode == 0) return hipStreamSynchronize(C.s);
    if (E->wait_mode == 1) {
        for (;;) {
            const hipError_t e = hipStreamQuery(C.s);
            if (e != hipErrorNotReady) return e;
            std::this_thread::yield();
        }
    }
    if (E->wait_mode == 2) {
        // (measured: with a blocking-sync event the runtime still spins 100 + 200 us before it sleeps in the driver - for rounds of
        // 0.15-1.4 ms the five combiner threads burned 17 of the 59 CPU-seconds of a B = 128 run)
        hipError_t e = hipEventRecord(C.ev, C.s);
        if (e != hipSuccess) return e;
        return hipEventSynchronize(C.ev);
    }
    // completion word: sleep through most of the expected duration, then look every ~10 us (timer slack of the thread is 1 us)
    const unsigned seq = ++C.done_seq;
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, C.s, C.dm_done, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    auto done = [&] { return __atomic_load_n(C.h_done, __ATOMIC_ACQUIRE) == seq; };
    if (!done()) {
        const double first = 0.7 * C.ema_wait_us;
        if (first > 25) { std::this_thread::sleep_for(std::chrono::nanoseconds((long)(first * 1e3))); C.t_first_sleep += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
        int looks = 0;
        while (!done()) {
            C.n_polls++;
            std::this_thread::sleep_for(std::chrono::microseconds(10));
            if ((++looks & 255) == 0) {   // a faulted launch never signals: ask the runtime now and then
                e = hipStreamQuery(C.s);
                if (e != hipSuccess && e != hipErrorNotReady) return e;
            }
        }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    C.ema_wait_us = C.ema_wait_us == 0 ? us : 0.8 * C.ema_wait_us + 0.2 * us;
    // The completion word tells US that the round is over; the runtime has not been asked anything about this stream since the launches.
    // One query per round lets it retire the round's commands now instead of finding them all still on its books at some later launch.
    static const bool query_each_round = !(getenv("PMV_BATCH_QUERY") && atoi(getenv("PMV_BATCH_QUERY")) == 0);
    if (query_each_round) { e = hipStreamQuery(C.s); if (e != hipSuccess && e != hipErrorNotReady) return e; }
    return hipSuccess;
}

// make stream `s` wait for the pyramid build round `need` (see BatchEngine::slot_round)
hipError_t wait_built(BatchEngine* E, hipStream_t s, int need) {
    if (need < 0 || need >= (int)E->build_ev.size()) return hipSuccess;
    while (E->build_enqueued.load(std::memory_order_acquire) <= need) {
        if (E->build_error.load()) return hipErrorUnknown;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return hipStreamWaitEvent(s, E->build_ev[(size_t)need], 0);
}

// ---- LK -------------------------------------------------------------------------------------------------------------------------
void process_lk(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    pmv_ctx* ctx = E->ctx;
    hipStream_t s = C.s;
    std::vector<LKReq*> lk;
    for (Req* r : batch) lk.push_back((LKReq*)r);
    // ---- LK: one launch for the tracks of every requesting sequence
    int total_tracks = 0, total_blocks = 0, need_round = -1;
    PyrLayout L{};
    bool have_L = false;
    // (a slot whose pyramid the background build has not reached yet is "staged": n_levels < 0; the launch below waits for its round)
    for (LKReq* r : lk) {
        PyrLayout a = ctx->slot_layout[r->prev_slot];
        const PyrLayout& b2 = ctx->slot_layout[r->next_slot];
        // staged but not built, and not on the background build's list either: there is no pyramid to track on
        const bool unbuilt = (a.n_levels < 0 && (E->slot_round.empty() || E->slot_round[(size_t)r->prev_slot] < 0)) ||
                             (b2.n_levels < 0 && (E->slot_round.empty() || E->slot_round[(size_t)r->next_slot] < 0));
        if (a.n_levels < 0) a.n_levels = -a.n_levels;
        if (unbuilt || a.n_levels == 0 || b2.n_levels == 0 || a.w[0] != b2.w[0] || a.h[0] != b2.h[0]) { r->rc = PMV_ERR_INVALID; snprintf(r->err, sizeof(r->err), "batch LK: slot has no pyramid / sizes differ"); continue; }
        if (!have_L) { L = a; have_L = true; }
        else if (a.w[0] != L.w[0] || a.h[0] != L.h[0]) { r->rc = PMV_ERR_INVALID; snprintf(r->err, sizeof(r->err), "batch LK: all sequences of a batch must share the frame size"); continue; }
        r->base = total_tracks;
        total_tracks += r->n;
        total_blocks += (int)r->order.size();
        if (!E->slot_round.empty()) need_round = std::max(need_round, std::max(E->slot_round[(size_t)r->prev_slot], E->slot_round[(size_t)r->next_slot]));
    }
    if (total_tracks > 0) {
        EK(wait_built(E, s, need_round));
        if ((size_t)total_tracks > E->cap_tracks) { fail_all(batch, PMV_ERR_CAPACITY, "more tracks than B * max_tracks", hipSuccess); return; }
        const size_t bytes = sizeof(LKBlock) * (size_t)total_blocks;
        EK(C.h_front.ensure(bytes + 64));
        LKBlock* hblk = (LKBlock*)C.h_front.p;
        int bpos = 0;
        std::vector<LKReq*> live;
        auto put = [&](const LKReq* r, int o) {
            if (o < 0) return;   // (padding entries of the striped order: the launch carries real tracks only)
            LKBlock& k = hblk[bpos++];
            k.prev_off = (unsigned long long)r->prev_slot * L.slot_bytes;
            k.next_off = (unsigned long long)r->next_slot * L.slot_bytes;
            k.track = r->base + o;
            k.x = r->prev_xy[2 * (size_t)o]; k.y = r->prev_xy[2 * (size_t)o + 1];
            k.pad = 0;
        };
        for (LKReq* r : lk) {
            if (r->rc != PMV_OK) continue;
            if (!E->lk_lpt) for (int o : r->order) put(r, o);
            live.push_back(r);
        }
        if (E->lk_lpt) {
            // longest-predicted tracks first, over the WHOLE round: every request's order is "most expensive first" (engine_lk), the
            // launch takes the k-th track of every request before any (k + 1)-th. Workgroups start in index order, so the tracks that
            // will iterate longest start first and the launch does not end with one of them started last (a launch ends with its slowest
            // track: makespan <= work / slots + longest track for an arbitrary order).
            size_t kmax = 0;
            for (LKReq* r : live) kmax = std::max(kmax, r->order.size());
            for (size_t k = 0; k < kmax; k++)
                for (LKReq* r : live)
                    if (k < r->order.size()) put(r, r->order[k]);
        }
        LKParams P;
        P.max_iter = 30; P.eps2 = 1e-4f; P.eps2d = 0.01 * 0.01; P.min_eig = 1e-4f;
        static const bool lk_stamps = getenv("PMV_LK_STAMPS") != nullptr;   // diagnostic: phase timers of every 64th track (pmv_debug_lk_stamps)
        if (lk_stamps) {
            static std::mutex stamps_mu;   // two LK lanes
            std::lock_guard<std::mutex> lk_(stamps_mu);
            if (!ctx->d_lk_stamps && hipMalloc(&ctx->d_lk_stamps, 16 * 8) == hipSuccess) (void)hipMemset(ctx->d_lk_stamps, 0, 16 * 8);
        }
        P.stamps = lk_stamps ? ctx->d_lk_stamps : nullptr;
        // mapped pinned: every workgroup reads its 32-byte record once, no copy launch
        EK(launch_lk_batch(s, ctx->d_slots, (const LKBlock*)C.h_front.dev, bpos, L, P, C.dm_out_xy, C.dm_status, C.dm_err, C.dm_work));
    }
    SYNC_TIMED(C);
    for (LKReq* r : lk) {
        if (r->rc != PMV_OK) continue;
        memcpy(r->out_xy, C.h_out_xy + (size_t)2 * r->base, (size_t)r->n * 8);
        memcpy(r->status, C.h_status + r->base, (size_t)r->n);
        memcpy(r->err_out, C.h_err + r->base, (size_t)r->n * 4);
        if (r->iters_out) for (int i = 0; i < r->n; i++) r->iters_out[i] = (uint8_t)(C.h_work[(size_t)r->base + i] & 0xffu);
        ctx->add_lk_work(C.h_work + r->base, (size_t)r->n);
    }
}

// ---- detectors --------------------------------------------------------------------------------------------------------------------
void process_det(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    pmv_ctx* ctx = E->ctx;
    hipStream_t s = C.s;
    std::vector<DetReq*> det;
    for (Req* r : batch) det.push_back((DetReq*)r);
    // ---- detectors: requests with the same parameters AND the same frame geometry share a launch (cells of several frames); the
    // geometry is the one actually staged in each request's slot (KITTI 00-02, 03 and 04-10 have three different sizes)
    struct Group { int kind, max_per_cell, unlimited; double quality, min_dist; PyrLayout L; std::vector<DetReq*> reqs; int n_cells = 0; size_t out_off = 0; };
    std::vector<Group> groups;
    for (DetReq* r : det) {
        PyrLayout Lr = ctx->slot_layout[r->slot];
        if (Lr.n_levels < 0 && (E->slot_round.empty() || E->slot_round[(size_t)r->slot] < 0)) {
            r->rc = PMV_ERR_INVALID; snprintf(r->err, sizeof(r->err), "batch detect: slot %d was staged but its pyramid was never built", r->slot); continue;
        }
        if (Lr.n_levels < 0) Lr.n_levels = -Lr.n_levels;   // staged, its build round is awaited below
        Group* g = nullptr;
        for (Group& x : groups)
            if (x.kind == r->kind && x.max_per_cell == r->max_per_cell && x.unlimited == r->unlimited && x.quality == r->quality && x.min_dist == r->min_dist &&
                x.L.w[0] == Lr.w[0] && x.L.h[0] == Lr.h[0] && x.L.n_levels == Lr.n_levels) { g = &x; break; }
        if (!g) { groups.push_back(Group{r->kind, r->max_per_cell, r->unlimited, r->quality, r->min_dist, Lr, {}, 0, 0}); g = &groups.back(); }
        r->cell_base = g->n_cells;
        g->n_cells += r->n_cells;
        g->reqs.push_back(r);
    }
    {
        size_t tot_cells = 0, tot_out = 0;
        for (Group& g : groups) { g.out_off = tot_out; tot_cells += g.n_cells; tot_out += (size_t)g.n_cells * g.max_per_cell; }
        EK(C.h_cells.ensure(tot_cells * CELL_STRIDE * 4));
        EK(C.d_eig.ensure(tot_cells * CELL_PIX * sizeof(double))); EK(C.d_cellmax.ensure(tot_cells * 8)); EK(C.d_spill.ensure(tot_cells * CELL_PIX * 4));
        EK(C.h_det.ensure(tot_out * 16 + tot_cells * 4 + 64));   // [xy | score | count | flags], written by the kernels through the mapped alias
        int* hc = (int*)C.h_cells.p;
        size_t cpos = 0;
        for (Group& g : groups)
            for (DetReq* r : g.reqs)
                for (int i = 0; i < r->n_cells; i++, cpos++) {
                    int* d = hc + cpos * CELL_STRIDE;
                    d[0] = r->cells[4 * i]; d[1] = r->cells[4 * i + 1]; d[2] = r->cells[4 * i + 2]; d[3] = r->cells[4 * i + 3]; d[4] = r->slot; d[5] = d[6] = d[7] = 0;
                }
        EK(hipMemsetAsync(C.d_flags, 0, 16, s));
        {
            int need_round = -1;
            if (!E->slot_round.empty()) for (DetReq* r : det) need_round = std::max(need_round, E->slot_round[(size_t)r->slot]);
            EK(wait_built(E, s, need_round));
        }
        size_t c0 = 0;
        char* hd = (char*)C.h_det.p;
        char* dd = C.h_det.dev;
        for (Group& g : groups) {
            const int* dc = (const int*)C.h_cells.dev + c0 * CELL_STRIDE;
            int* dxy = (int*)dd + g.out_off * 2;
            double* dsc = (double*)(dd + tot_out * 8) + g.out_off;
            int* dcnt = (int*)(dd + tot_out * 16) + c0;
            if (g.kind == 1)
                EK(launch_gftt(s, ctx->d_slots, g.L, dc, g.n_cells, g.max_per_cell, g.quality, g.min_dist, g.unlimited, (float*)C.d_eig.p + c0 * CELL_PIX,
                               (unsigned*)C.d_cellmax.p + 2 * c0, dxy, dcnt, C.d_flags, (unsigned*)C.d_spill.p + c0 * CELL_PIX));
            else
                EK(launch_shitomasi(s, ctx->d_slots, g.L, dc, g.n_cells, g.max_per_cell, g.quality, (double*)C.d_eig.p + c0 * CELL_PIX,
                                    (unsigned long long*)C.d_cellmax.p + c0, dxy, dsc, dcnt, C.d_flags, (unsigned*)C.d_spill.p + c0 * CELL_PIX));
            c0 += g.n_cells;
        }
        EK(hipMemcpyAsync(hd + tot_out * 16 + tot_cells * 4, C.d_flags, 4, hipMemcpyDeviceToHost, s));   // (the kernels set the bits with atomics: device memory)
    }
    SYNC_TIMED(C);
    {
        size_t tot_cells = 0, tot_out = 0;
        for (Group& g : groups) { tot_cells += g.n_cells; tot_out += (size_t)g.n_cells * g.max_per_cell; }
        const char* hd = (const char*)C.h_det.p;
        const int flags = *(const int*)(hd + tot_out * 16 + tot_cells * 4);
        const int* hxy = (const int*)hd;
        const double* hsc = (const double*)(hd + tot_out * 8);
        const int* hcnt = (const int*)(hd + tot_out * 16);
        size_t c0 = 0;
        for (Group& g : groups) {
            for (DetReq* r : g.reqs) {
                if (g.kind == 1 && (flags & 4)) { r->rc = PMV_ERR_OVERFLOW; snprintf(r->err, sizeof(r->err), "more corners than the no-limit capacity in a cell (batched launch)"); continue; }
                const size_t o = g.out_off + (size_t)r->cell_base * g.max_per_cell;
                memcpy(r->out_xy, hxy + o * 2, (size_t)r->n_cells * g.max_per_cell * 8);
                if (r->out_score) memcpy(r->out_score, hsc + o, (size_t)r->n_cells * g.max_per_cell * 8);
                memcpy(r->out_count, hcnt + c0 + r->cell_base, (size_t)r->n_cells * 4);
            }
            c0 += g.n_cells;
        }
    }
}

// ---- back-end classes: the callers prepared their inputs in their slot's pinned block; one gather kernel pulls them into HBM ----
// Descriptor block of a batch in mapped pinned memory: [problem records | stage-in jobs]. No DMA call is made: k_stage_in reads
// its job list through the host alias; job 0 copies the problem records into device memory (the back-end chains read them in every
// launch), jobs 1..n pull the requests' input blocks. (A hipMemcpyAsync is a blit kernel of its own: 83 k of them, 15 us each in
// stream time, in a B = 64 run before this.)
template <class Prob> struct DescBlock { Prob* hprob; StageJob* hjobs; const Prob* dprob; const StageJob* djobs; size_t bytes; };
template <class Prob>
hipError_t desc_block(Combiner& C, size_t n, DescBlock<Prob>& D) {
    const size_t off_jobs = (sizeof(Prob) * n + 255) & ~(size_t)255;
    D.bytes = off_jobs + sizeof(StageJob) * (n + 1);
    hipError_t e = C.h_desc.ensure(D.bytes + 256);
    if (e != hipSuccess) return e;
    e = C.d_desc.ensure(off_jobs + 256);
    if (e != hipSuccess) return e;
    D.hprob = (Prob*)C.h_desc.p; D.hjobs = (StageJob*)((char*)C.h_desc.p + off_jobs);
    D.dprob = (const Prob*)C.d_desc.p; D.djobs = (const StageJob*)(C.h_desc.dev + off_jobs);
    D.hjobs[0] = StageJob{C.h_desc.dev, (char*)C.d_desc.p, (unsigned)(sizeof(Prob) * n), 0};
    D.hjobs += 1;   // the callers fill jobs 1..n
    return hipSuccess;
}
template <class Prob>
hipError_t stage_in(Combiner& C, const DescBlock<Prob>& D, size_t n, int blocks_per_job) {
    hipLaunchKernelGGL(k_stage_in, dim3(blocks_per_job, (unsigned)(n + 1)), dim3(256), 0, C.s, D.djobs);
    return hipGetLastError();
}

void process_pnp(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    DescBlock<PnPProblem> D;
    EK(desc_block(C, batch.size(), D));
    int max_hyp = 0;
    for (size_t i = 0; i < batch.size(); i++) {
        PnPReq* r = (PnPReq*)batch[i];
        D.hprob[i] = r->P;
        D.hjobs[i] = StageJob{(const char*)r->b->d_h_stage, r->b->d_pnp_in, (unsigned)r->in_bytes, 0};
        max_hyp = std::max(max_hyp, r->P.n_hyp);
    }
    EK(stage_in(C, D, batch.size(), 2));
    EK(launch_pnp_batch(C.s, D.dprob, (int)batch.size(), max_hyp));
    SYNC_TIMED(C);   // the refit kernel wrote every result straight into the request's pinned block
}

void process_ba(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    if (E->ctx->ba_mode == 1) {   // one workgroup per problem: ONE launch for the whole round (pmv_set_ba_mode)
        DescBlock<BAArgs> D;
        EK(desc_block(C, batch.size(), D));
        int max_m = 6;
        for (size_t i = 0; i < batch.size(); i++) {
            BAReq* r = (BAReq*)batch[i];
            D.hprob[i] = r->A;
            D.hjobs[i] = StageJob{(const char*)r->b->d_h_stage, r->b->d_ba_io, (unsigned)r->io_bytes, 0};
            max_m = std::max(max_m, 6 * r->A.nc);
        }
        EK(stage_in(C, D, batch.size(), 8));
        EK(launch_ba_lm_batch(C.s, D.dprob, (int)batch.size(), max_m));
        SYNC_TIMED(C);
        return;
    }
    // one launch chain per distinct iteration cap (in practice one)
    std::vector<int> iters;
    for (Req* q : batch) { const int it = ((BAReq*)q)->max_iterations; if (std::find(iters.begin(), iters.end(), it) == iters.end()) iters.push_back(it); }
    std::vector<BAReq*> sorted;
    for (int it : iters) for (Req* q : batch) if (((BAReq*)q)->max_iterations == it) sorted.push_back((BAReq*)q);
    DescBlock<BAProb> D;
    EK(desc_block(C, sorted.size(), D));
    for (size_t i = 0; i < sorted.size(); i++) {
        BAReq* r = sorted[i];
        ba_fill_prob(D.hprob[i], r->A, r->b->d_bastate, r->b->d_bapart);
        D.hjobs[i] = StageJob{(const char*)r->b->d_h_stage, r->b->d_ba_io, (unsigned)r->io_bytes, 0};
    }
    EK(stage_in(C, D, sorted.size(), 8));
    size_t i0 = 0;
    for (int it : iters) {
        BABatchDims dims{0, 0, 0, 0, 0, 0, it};
        size_t i1 = i0;
        while (i1 < sorted.size() && sorted[i1]->max_iterations == it) {
            const BAProb& P = D.hprob[i1];
            dims.max_eval_blocks = std::max(dims.max_eval_blocks, P.nbo + P.clear_blocks);
            dims.max_nc = std::max(dims.max_nc, P.A.nc); dims.max_nbp = std::max(dims.max_nbp, P.nbp); dims.max_tiles = std::max(dims.max_tiles, P.tiles);
            dims.max_m = std::max(dims.max_m, 6 * P.A.nc);
            i1++;
        }
        dims.n_probs = (int)(i1 - i0);
        EK(launch_ba_multi_batch(C.s, D.dprob + i0, dims));
        i0 = i1;
    }
    SYNC_TIMED(C);
}

void process_dlt(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    DescBlock<DltProblem> D;
    EK(desc_block(C, batch.size(), D));
    int max_n = 0;
    for (size_t i = 0; i < batch.size(); i++) {
        DltReq* r = (DltReq*)batch[i];
        D.hprob[i] = r->P;
        D.hjobs[i] = StageJob{(const char*)r->b->d_h_stage, r->b->d_tri_in, (unsigned)r->in_bytes, 0};
        max_n = std::max(max_n, r->P.n);
    }
    EK(stage_in(C, D, batch.size(), 2));
    EK(launch_tri_dlt_batch(C.s, D.dprob, (int)batch.size(), max_n));
    SYNC_TIMED(C);
}

void process_fp(BatchEngine* E, Combiner& C, std::vector<Req*>& batch) {
    DescBlock<FivePointProblem> D;
    EK(desc_block(C, batch.size(), D));
    int max_hyp = 0;
    for (size_t i = 0; i < batch.size(); i++) {
        FPReq* r = (FPReq*)batch[i];
        D.hprob[i] = r->P;
        D.hjobs[i] = StageJob{(const char*)r->b->d_h_stage, r->b->d_tri_in, (unsigned)r->in_bytes, 0};
        max_hyp = std::max(max_hyp, r->P.n_hyp);
    }
    EK(stage_in(C, D, batch.size(), 2));
    EK(launch_fivepoint_batch(C.s, D.dprob, (int)batch.size(), max_hyp));
    SYNC_TIMED(C);
}

static inline void futex_wait_while(std::atomic<int>* w, int v) {
    while (w->load(std::memory_order_acquire) == v) (void)syscall(SYS_futex, (int*)w, FUTEX_WAIT_PRIVATE, v, nullptr, nullptr, 0);
}
static inline void futex_wake_one(std::atomic<int>* w) { (void)syscall(SYS_futex, (int*)w, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }

void combiner_loop(BatchEngine* E, int role, int lane) {
    Combiner* C = &E->comb[role][lane];
    Queue* Q = &E->queue[role];
    (void)hipSetDevice(E->ctx->device);
    tl_prof = &E->ctx->prof;
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   // 1 us instead of the default 50 us: the timed sleeps of wait_stream
    for (;;) {
        std::vector<Req*> batch;
        const auto ti = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(Q->mu);
            Q->cv_new.wait(lk, [&] { return !Q->pending.empty() || Q->stop; });
            if (Q->pending.empty() && Q->stop) return;
            if (E->min_batch[role] > 1 && (int)Q->pending.size() < E->min_batch[role] && !Q->stop) {
                const auto deadline = Q->first_arrival + std::chrono::microseconds(E->max_linger_us);
                Q->cv_new.wait_until(lk, deadline, [&] { return (int)Q->pending.size() >= E->min_batch[role] || Q->stop; });
                if (Q->pending.empty()) { if (Q->stop) return; continue; }   // another lane of the class took them meanwhile
            }
            if (E->linger_us > 0 && (int)Q->pending.size() < E->B) {   // optional: give stragglers a moment to join the batch
                lk.unlock();
                std::this_thread::sleep_for(std::chrono::microseconds(E->linger_us));
                lk.lock();
            }
            batch.swap(Q->pending);
        }
        const auto tw = std::chrono::steady_clock::now();
        C->t_idle += std::chrono::duration<double>(tw - ti).count();
        // diagnostic (PMV_BATCH_EXCLUSIVE=1): one class on the GPU at a time, so a round's duration is that of its kernels alone
        std::unique_lock<std::mutex> excl(E->exclusive_mu, std::defer_lock);
        if (E->exclusive) excl.lock();
        switch (role) {
        case R_LK: process_lk(E, *C, batch); break;
        case R_DET: process_det(E, *C, batch); break;
        case R_PNP: process_pnp(E, *C, batch); break;
        case R_BA: process_ba(E, *C, batch); break;
        case R_DLT: process_dlt(E, *C, batch); break;
        default: process_fp(E, *C, batch); break;
        }
        if (E->exclusive) excl.unlock();
        C->batches++; C->requests += (long)batch.size();
        for (Req* r : batch) {
            std::atomic<int>* w = &r->done;   // (after the store the owner may return and the request, which lives on its stack, is gone)
            w->store(1, std::memory_order_release);
            futex_wake_one(w);
        }
        C->t_work += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); C->t_cpu = ts.tv_sec + 1e-9 * ts.tv_nsec; }
    }
}

int submit(pmv_ctx* ctx, Queue& Q, Req* r) {
    {
        std::lock_guard<std::mutex> lk(Q.mu);
        if (Q.pending.empty()) Q.first_arrival = std::chrono::steady_clock::now();
        Q.pending.push_back(r);
        const int sz = (int)Q.pending.size();
        if (sz == 1) Q.cv_new.notify_one();
        else if (Q.min_batch <= 1 || sz == Q.min_batch) Q.cv_new.notify_all();   // (a lingering combiner is waiting for exactly this)
    }
    futex_wait_while(&r->done, 0);
    if (r->rc != PMV_OK) set_err(ctx, "%s", r->err);
    return r->rc;
}

}  // namespace

#define CKC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(ctx, "%s: %s", #x, hipGetErrorString(e_)); return PMV_ERR_HIP; } } while (0)
#define REQ(cond, code, ...) do { if (!(cond)) { set_err(ctx, __VA_ARGS__); return code; } } while (0)

void batch_engine_destroy(pmv_ctx* ctx) {
    BatchEngine* E = ctx->engine;
    if (!E) return;
    for (Queue& Q : E->queue) {
        { std::lock_guard<std::mutex> lk(Q.mu); Q.stop = true; }
        Q.cv_new.notify_all();
    }
    for (auto& role : E->comb)
        for (Combiner& C : role)
            if (C.th.joinable()) C.th.join();   // all of them first: lanes of a class may share a stream
    for (auto& role : E->comb)
        for (Combiner& C : role) {
            if (getenv("PMV_BATCH_TIMING") && C.batches)
                fprintf(stderr, "[batch-timing] class %d lane %d: %ld rounds, %.1f req/round, per round: work %.0f us, sync %.0f us (first sleep %.0f us, then %.1f polls), cpu %.0f us, ema %.0f us\n",
                        (int)(&role - &E->comb[0]), (int)(&C - &role[0]), C.batches, (double)C.requests / C.batches, C.t_work / C.batches * 1e6, C.t_sync / C.batches * 1e6,
                        C.t_first_sleep / C.batches * 1e6, (double)C.n_polls / C.batches, C.t_cpu / C.batches * 1e6, C.ema_wait_us);
            if (C.s && C.owns_stream) { (void)hipStreamSynchronize(C.s); (void)hipStreamDestroy(C.s); }
            if (C.ev) (void)hipEventDestroy(C.ev);
            if (C.h_done) (void)hipHostFree(C.h_done);
            for (Growable* g : {&C.h_desc, &C.d_desc, &C.h_front, &C.d_front, &C.h_cells, &C.d_cells, &C.d_eig, &C.d_cellmax, &C.d_spill, &C.d_det_xy, &C.d_det_score, &C.d_det_count, &C.h_det}) g->release();
            if (C.h_out_xy) (void)hipHostFree(C.h_out_xy);
            if (C.h_err) (void)hipHostFree(C.h_err);
            if (C.h_status) (void)hipHostFree(C.h_status);
            if (C.h_work) (void)hipHostFree(C.h_work);
            if (C.d_flags) (void)hipFree(C.d_flags);
        }
    if (E->build_thread.joinable()) E->build_thread.join();
    for (hipEvent_t ev : E->build_ev) (void)hipEventDestroy(ev);
    for (BackendBuffers* b : E->slots) backend_free(b);
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
    {
        const double frac = getenv("PMV_BATCH_MIN_FRAC") ? atof(getenv("PMV_BATCH_MIN_FRAC")) : 0.12;   // of the B sequences, per class with long rounds
        if (const char* e = getenv("PMV_BATCH_MAX_LINGER_US")) E->max_linger_us = atoi(e);
        for (int r : {(int)R_LK, (int)R_PNP, (int)R_BA}) { E->min_batch[r] = (int)(frac * B); E->queue[r].min_batch = E->min_batch[r]; }
    }
    for (int i = 0; i < B; i++) {
        BackendBuffers* b = nullptr;
        const int rc = backend_alloc(ctx, &b);
        if (b) E->slots.push_back(b);
        if (rc != PMV_OK) { batch_engine_destroy(ctx); return rc; }
    }
    E->cap_tracks = (size_t)B * ctx->max_tracks;
    if (const char* e = getenv("PMV_BATCH_EXCLUSIVE")) E->exclusive = atoi(e) != 0;
    if (const char* e = getenv("PMV_LK_LPT")) E->lk_lpt = atoi(e) != 0;
    E->lanes[R_LK] = 2;   // (measured, B = 128: see DESIGN.md §5)
    E->lanes[R_BA] = 2;   // (B = 192, profiles/r03_batch_exp_ae.log: 64.8 k -> 67.6 k frames/s; three lanes: 52 k)
    if (const char* e = getenv("PMV_BATCH_LANES")) for (int& l : E->lanes) l = std::max(1, std::min(MAX_LANES, atoi(e)));
    if (const char* e = getenv("PMV_BATCH_LANES_LK")) E->lanes[R_LK] = std::max(1, std::min(MAX_LANES, atoi(e)));
    if (const char* e = getenv("PMV_BATCH_LANES_PNP")) E->lanes[R_PNP] = std::max(1, std::min(MAX_LANES, atoi(e)));
    if (const char* e = getenv("PMV_BATCH_LANES_BA")) E->lanes[R_BA] = std::max(1, std::min(MAX_LANES, atoi(e)));
    if (const char* e = getenv("PMV_BATCH_STREAMS_LK")) E->streams[R_LK] = atoi(e);
    if (const char* e = getenv("PMV_BATCH_STREAMS_BA")) E->streams[R_BA] = atoi(e);
    if (const char* e = getenv("PMV_BATCH_STREAMS_PNP")) E->streams[R_PNP] = atoi(e);
    if (const char* e = getenv("PMV_BATCH_WAIT")) E->wait_mode = !strcmp(e, "spin") ? 0 : !strcmp(e, "yield") ? 1 : !strcmp(e, "block") ? 2 : 3;
    for (int r = 0; r < R_COUNT; r++)
        for (int l = 0; l < E->lanes[r]; l++) {
            Combiner& C = E->comb[r][l];
            // the back-end classes are chains of small launches (a BA solve: 23 of them): they get the higher stream priority so their
            // workgroups are not queued behind the thousands of LK / detector waves of the front-end classes (PMV_BATCH_PRIO=0: all equal)
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // numerically lower = higher priority
            static const bool use_prio = !(getenv("PMV_BATCH_PRIO") && atoi(getenv("PMV_BATCH_PRIO")) == 0);
            const int prio = !use_prio ? prio_lo : (r == R_BA || r == R_PNP || r == R_DLT) ? prio_hi : prio_lo;
            // (measured and dropped: confining the back-end classes to 64 / 96 CUs with hipExtStreamCreateWithCUMask: 25.4 k -> 16.2 k / 21.3 k
            // frames/s at B = 64 - the PnP hypotheses need the whole chip; equal priorities: no difference either)
            const int n_streams = E->streams[r] > 0 ? std::min(E->streams[r], E->lanes[r]) : E->lanes[r];
            if (l < n_streams) CKC(hipStreamCreateWithPriority(&C.s, hipStreamNonBlocking, prio));
            else { C.s = E->comb[r][l % n_streams].s; C.owns_stream = false; }
            CKC(hipEventCreateWithFlags(&C.ev, hipEventBlockingSync | hipEventDisableTiming));
            CKC(hipHostMalloc(&C.h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
            *C.h_done = 0;
            CKC(hipHostGetDevicePointer((void**)&C.dm_done, C.h_done, 0));
            if (r == R_LK) {
                CKC(hipHostMalloc(&C.h_out_xy, E->cap_tracks * 8, hipHostMallocMapped | hipHostMallocCoherent));
                CKC(hipHostMalloc(&C.h_status, E->cap_tracks, hipHostMallocMapped | hipHostMallocCoherent));
                CKC(hipHostMalloc(&C.h_err, E->cap_tracks * 4, hipHostMallocMapped | hipHostMallocCoherent));
                CKC(hipHostMalloc(&C.h_work, E->cap_tracks * 2, hipHostMallocMapped | hipHostMallocCoherent));
                CKC(hipHostGetDevicePointer((void**)&C.dm_work, C.h_work, 0));
                CKC(hipHostGetDevicePointer((void**)&C.dm_out_xy, C.h_out_xy, 0));
                CKC(hipHostGetDevicePointer((void**)&C.dm_status, C.h_status, 0));
                CKC(hipHostGetDevicePointer((void**)&C.dm_err, C.h_err, 0));
            }
            if (r == R_DET) CKC(hipMalloc(&C.d_flags, 16));
        }
    for (int r = 0; r < R_COUNT; r++)
        for (int l = 0; l < E->lanes[r]; l++) E->comb[r][l].th = std::thread(combiner_loop, E, r, l);
    *out = E;
    return PMV_OK;
}

// Start building the pyramids of B sequences (frames first_slot[b] .. + n_frames[b] - 1, only where build[b] != 0) in the background;
// engine_build_end() joins. The slots must have been staged with one frame geometry.
int engine_build_begin(BatchEngine* E, int B, const int* first_slot, const int* n_frames, const int* build) {
    pmv_ctx* ctx = E->ctx;
    CKC(hipSetDevice(ctx->device));
    E->slot_round.assign((size_t)ctx->n_slots, -1);
    int max_n = 0;
    for (int b = 0; b < B; b++) if (build[b]) {
        max_n = std::max(max_n, n_frames[b]);
        for (int f = 0; f < n_frames[b]; f++) E->slot_round[(size_t)(first_slot[b] + f)] = f / BatchEngine::BUILD_CHUNK;
    }
    const int rounds = (max_n + BatchEngine::BUILD_CHUNK - 1) / BatchEngine::BUILD_CHUNK;
    while ((int)E->build_ev.size() < rounds) {
        hipEvent_t ev;
        CKC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        E->build_ev.push_back(ev);
    }
    E->build_enqueued.store(0);
    E->build_error.store(0);
    if (rounds == 0) return PMV_OK;
    std::vector<int> fs(first_slot, first_slot + B), nf(n_frames, n_frames + B), bd(build, build + B);
    E->build_thread = std::thread([E, ctx, B, rounds, fs, nf, bd] {
        (void)hipSetDevice(ctx->device);
        tl_prof = &ctx->prof;
        for (int r = 0; r < rounds; r++) {
            for (int b = 0; b < B; b++) {
                const int f0 = r * BatchEngine::BUILD_CHUNK, n = std::min(BatchEngine::BUILD_CHUNK, nf[(size_t)b] - f0);
                if (!bd[(size_t)b] || n <= 0) continue;
                if (pmv_frames_build_on(ctx, ctx->s_front, fs[(size_t)b] + f0, n) != PMV_OK) { E->build_error.store(1); return; }
            }
            if (hipEventRecord(E->build_ev[(size_t)r], ctx->s_front) != hipSuccess) { E->build_error.store(1); return; }
            E->build_enqueued.store(r + 1, std::memory_order_release);
        }
    });
    return PMV_OK;
}
int engine_build_end(BatchEngine* E) {
    if (E->build_thread.joinable()) E->build_thread.join();
    E->slot_round.clear();
    return E->build_error.load() ? PMV_ERR_HIP : PMV_OK;
}

void batch_engine_stats(BatchEngine* E, long long* counts10, double* times15) {
    for (int r = 0; r < 5; r++) {   // the five classes every run uses; the optional five-point class is reported separately

        counts10[2 * r] = counts10[2 * r + 1] = 0;
        if (times15) times15[3 * r] = times15[3 * r + 1] = times15[3 * r + 2] = 0;
        for (int l = 0; l < E->lanes[r]; l++) {   // summed over the class's combiners
            const Combiner& C = E->comb[r][l];
            counts10[2 * r] += C.batches; counts10[2 * r + 1] += C.requests;
            if (times15) { times15[3 * r] += C.t_cpu; times15[3 * r + 1] += C.t_work; times15[3 * r + 2] += C.t_sync; }
        }
    }
}

// ---- request entry points (called from the sequences' own host threads) ---------------------------------------------------------
int engine_lk(BatchEngine* E, int prev_slot, int next_slot, const float* prev_xy, int n, float* out_xy, uint8_t* status, float* err,
              const uint8_t* predicted_iters, uint8_t* iters_out) {
    pmv_ctx* ctx = E->ctx;
    REQ(n >= 0 && n <= ctx->max_tracks, PMV_ERR_CAPACITY, "pmv_lk_track: n=%d exceeds max_tracks=%d", n, ctx->max_tracks);
    REQ(prev_slot >= 0 && prev_slot < ctx->n_slots && next_slot >= 0 && next_slot < ctx->n_slots, PMV_ERR_CAPACITY, "pmv_lk_track: slot out of range");
    if (n == 0) return PMV_OK;
    LKReq r;
    r.kind = 0; r.prev_slot = prev_slot; r.next_slot = next_slot; r.n = n; r.prev_xy = prev_xy; r.out_xy = out_xy; r.status = status; r.err_out = err;
    r.iters_out = iters_out;
    if (E->lk_lpt) {
        // most expensive first: a counting sort on the predicted iteration count (255 .. 0), ties in track order
        r.order.resize((size_t)n);
        int cnt[257] = {0};
        for (int i = 0; i < n; i++) cnt[256 - (predicted_iters ? predicted_iters[i] : 0)]++;   // bucket b = 255 - pred, offset by one for the prefix
        for (int b = 1; b <= 256; b++) cnt[b] += cnt[b - 1];
        for (int i = 0; i < n; i++) r.order[(size_t)cnt[255 - (predicted_iters ? predicted_iters[i] : 0)]++] = i;
        return submit(ctx, E->queue[R_LK], &r);
    }
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
    return submit(ctx, E->queue[R_LK], &r);
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
    REQ(slot >= 0 && slot < ctx->n_slots && ctx->slot_layout[slot].n_levels != 0, PMV_ERR_INVALID, "detect: slot %d has no frame", slot);
    const PyrLayout& L = ctx->slot_layout[slot];
    for (int i = 0; i < n_cells; i++) {
        const int* c = cells + 4 * i;
        REQ(c[2] >= 3 && c[3] >= 3 && c[2] <= CELL_MAX && c[3] <= CELL_MAX && c[0] >= 0 && c[1] >= 0 && c[0] + c[2] <= L.w[0] && c[1] + c[3] <= L.h[0],
            PMV_ERR_INVALID, "detect: cell %d invalid", i);
    }
    r.quality = quality; r.min_dist = min_dist; r.out_xy = out_xy; r.out_score = out_score; r.out_count = out_count;
    return submit(ctx, E->queue[R_DET], &r);
}

int engine_pnp(BatchEngine* E, int seq, const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec, int iterations,
               float reproj_err, double confidence, int* out_inliers, int* out_n_inliers) {
    pmv_ctx* ctx = E->ctx;
    int rc = pnp_check(ctx, obj_xyz, img_xy, m, K, rvec, tvec, iterations, confidence, out_inliers, out_n_inliers);
    if (rc) return rc;
    PnPReq r;
    r.kind = 10; r.b = E->slots[seq];
    pnp_prepare(r.b, obj_xyz, img_xy, m, K, iterations, reproj_err, confidence, &r.P, &r.in_bytes);
    rc = submit(ctx, E->queue[R_PNP], &r);
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
    rc = ba_prepare(ctx, r.b, cams, nc, pts, np, obs_xy, cam_idx, pt_idx, n_obs, K, huber, max_iterations, ctx->ba_mode != 1, &r.A, &r.io_bytes);
    if (rc) return rc;
    if (ctx->ba_mode == 1) r.A.out = (double*)r.b->d_h_stage;   // k_ba_lm_batch copies [summary | cams | pts] into the request's pinned block
    rc = submit(ctx, E->queue[R_BA], &r);
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
    const int rc = submit(ctx, E->queue[R_DLT], &r);
    if (rc) return rc;
    dlt_finish(ctx, r.b, q1, q2, n, P1x4, mask_in, r.in_bytes, out_Q, out_mask, out_good);
    return PMV_OK;
}

int engine_fivepoint(BatchEngine* E, int seq, const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models,
                     int* n_models, int* counts) {
    pmv_ctx* ctx = E->ctx;
    FPReq r;
    r.kind = 13; r.b = E->slots[seq];
    int rc = fivepoint_prepare(ctx, r.b, q1, q2, n, samples, n_hyp, thr, &r.P, &r.in_bytes);
    if (rc) return rc;
    rc = submit(ctx, E->queue[R_FP], &r);
    if (rc) return rc;
    fivepoint_finish(r.b, n_hyp, r.in_bytes, models, n_models, counts);
    return PMV_OK;
}

}  // namespace pmv
