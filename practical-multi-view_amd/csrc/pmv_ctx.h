// Internal context definition shared by the C-ABI translation units.
#pragma once
#include <atomic>
#include "pmv_device.h"
#include "pmv_prof.h"
#include "../../include/pmv_hip.h"
#include <vector>
#include <mutex>

namespace pmv {
struct Ingest;                      // streamed frame ingest (ingest.hip)
struct BatchEngine;                 // multi-sequence combiners (batch_engine.hip)
constexpr int MAX_CELLS = 64;       // 1920x1080 -> 8x5 = 40 cells of 255x255
constexpr int MAX_PER_CELL = 4096;    // also the capacity of an "unlimited" (max_per_cell <= 0) goodFeaturesToTrack call
struct BackendBuffers;              // PnP / BA device workspaces (backend.hip)
}

// In-memory log of the back-end plugin calls (pmv_record_*): every pmv_pnp_ransac / pmv_ba_solve / pmv_triangulate_candidates
// call made while recording is on leaves one self-describing blob (layout: include/pmv_hip.h) with its inputs AND outputs, so a
// test can replay exactly the calls a pipeline run made through another implementation. Back-end thread only.
struct pmv_call_log {
    bool on = false;
    std::mutex mu;                          // several back-end threads log in batch mode
    std::vector<std::vector<char>> blobs;
    static void put(std::vector<char>& b, const void* p, size_t n) { const char* c = (const char*)p; b.insert(b.end(), c, c + n); }
};

struct pmv_ctx {
    int device = 0;
    int max_w = 0, max_h = 0, n_slots = 0, max_tracks = 0, max_ba_cams = 0, max_ba_points = 0, max_ba_obs = 0;
    hipStream_t s_front = nullptr, s_back = nullptr;
    pmv::PyrLayout cap;                       // geometry of the largest frame; cap.slot_bytes = slot pitch
    std::vector<pmv::PyrLayout> slot_layout;  // per slot: n_levels 0 = empty, <0 = staged only, >0 = pyramid built
    uint8_t* d_slots = nullptr;
    // landing area of host frames on their way into the slots: TIGHT_FRAMES tight gray frames (H2D copies are contiguous; k_pad_level0 takes
    // level 0 from here). A 2-D copy straight into the padded level is a DMA per image row: 128 x 1101 frames did not finish in 200 s.
    static constexpr int TIGHT_FRAMES = 64;
    uint8_t* d_tight = nullptr;
    // LK
    float *d_prev_xy = nullptr, *d_out_xy = nullptr, *d_err = nullptr;
    uint8_t* d_status = nullptr;
    float *h_prev_xy = nullptr, *h_out_xy = nullptr, *h_err = nullptr;
    uint8_t* h_status = nullptr;
    float *dm_out_xy = nullptr, *dm_err = nullptr;   // device aliases of the mapped pinned result buffers
    uint8_t* dm_status = nullptr;
    int *h_knn = nullptr, *d_knn = nullptr;      // kNN matcher coordinates: [src 2n | cmp 2m] ints, 2 * max_tracks pairs
    unsigned long long* d_lk_stamps = nullptr;   // diagnostic (PMV_LK_STAMPS=1)
    uint16_t* h_work = nullptr; uint16_t* dm_work = nullptr;   // per-track LK work of pmv_lk_track (mapped pinned, see launch_lk)
    std::atomic<unsigned long long> lk_work[3];  // host-side sums: LK iterations, level passes, tracks (pmv_lk_counters)
    void add_lk_work(const uint16_t* w, size_t n) { unsigned long long it = 0, lv = 0; for (size_t i = 0; i < n; i++) { it += w[i] & 0xffu; lv += w[i] >> 8; } lk_work[0] += it; lk_work[1] += lv; lk_work[2] += n; }
    // detectors
    int* d_cells = nullptr;
    int* h_cells = nullptr;   // pinned staging of the device cell records
    double* d_eig = nullptr;
    void* d_cellmax = nullptr;
    unsigned* d_spill = nullptr;   // detector candidates beyond the LDS lists: MAX_CELLS * CELL_PIX pixel indices
    int *d_det_xy = nullptr, *d_det_count = nullptr, *d_flags = nullptr;
    double* d_det_score = nullptr;
    int *h_det_xy = nullptr, *h_det_count = nullptr;
    double* h_det_score = nullptr;
    pmv::BackendBuffers* be = nullptr;
    // second back-end lane (own workspace + stream) for work a helper thread runs ahead of the back-end: pmv_triangulate_candidates_ahead
    pmv::BackendBuffers* be_ahead = nullptr;
    hipStream_t s_ahead = nullptr;
    std::mutex ahead_mu;
    // pmv_set_ba_mode: 0 = the multi-kernel LM chain (shortest latency for ONE solve: every phase spread over many CUs),
    // 1 = the whole solve in one workgroup per problem (k_ba_lm / k_ba_lm_batch: ONE launch per solve or per round of B solves).
    // Both are checked against the oracle to the same bars; their floating-point sums are ordered differently, so runs are compared
    // bit for bit only within one mode.
    int ba_mode = 0;
    pmv::BatchEngine* engine = nullptr; // created by the first pmv_pipeline_run_batch
    pmv::Ingest* ingest = nullptr;      // non-null while a pmv_frames_stream_begin .. _end bracket is open
    pmv::Profiler prof;
    pmv_call_log log;
    std::mutex err_mu;                  // set_err from several host threads (batch engine)
    char err[512] = "";
};

namespace pmv {
void set_err(pmv_ctx* c, const char* fmt, ...);
const char* thread_error();         // the last message set_err wrote on the calling thread
PyrLayout make_layout(int w, int h);
int backend_create(pmv_ctx* c);     // allocates PnP/BA workspaces
void backend_destroy(pmv_ctx* c);
// pyramid levels of `n` consecutive slots with identical geometry L, on `stream` (pmv_frames_build and the ingest thread)
int build_levels_on(pmv_ctx* ctx, hipStream_t stream, int first_slot, int n, const PyrLayout& L, const uint8_t* tight = nullptr);
PyrLayout layout_for(pmv_ctx* ctx, int w, int h);
// pmv_frames_build on a given stream (no host synchronisation)
int pmv_frames_build_on(pmv_ctx* ctx, hipStream_t stream, int first_slot, int n);
// streamed ingest: make the front-end stream wait until `slot` has been copied and its pyramid built (no-op without a stream)
int ingest_require(pmv_ctx* ctx, int slot);
void ingest_destroy(pmv_ctx* ctx);
void batch_engine_destroy(pmv_ctx* ctx);
hipError_t frontend_prepare_device();   // per-device kernel attributes (LDS opt-in), called with the context's device current
hipError_t backend_prepare_device();
}
