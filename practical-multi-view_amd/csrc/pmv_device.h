// Shared declarations for the gfx950 kernels of the VO hot path (internal; the public boundary is include/pmv_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pmv {

// ---- frame slot layout in HBM -------------------------------------------------------------------------
// Every pyramid level is stored with a PAD-pixel BORDER_REFLECT_101 frame on all four sides (cv::buildOpticalFlowPyramid
// pads by winSize=32; we pad by 64 so that the 64x64 LDS search tile of the LK kernel never leaves the buffer) and a
// row stride that is a multiple of 64 bytes, so every tile row starts dword-aligned and no kernel needs a border branch.
constexpr int PAD = 64;
constexpr int MAX_LEVELS = 5;   // maxLevel 4 -> levels 0..4
constexpr int LK_WIN = 32;

struct PyrLayout {
    int n_levels;                 // levels actually built = maxLevel + 1
    int w[MAX_LEVELS], h[MAX_LEVELS], stride[MAX_LEVELS];
    uint32_t off[MAX_LEVELS];     // byte offset of the padded level buffer inside the slot
    uint32_t gray_off;            // byte offset of pixel (0,0) of level 0: a frame is staged straight into the interior of its padded level 0 (row pitch stride[0])
    uint32_t slot_bytes;
};

__host__ __device__ inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// pointer to pixel (0,0) of a padded level
__host__ __device__ inline const uint8_t* level_origin(const uint8_t* slot, const PyrLayout& L, int l) {
    return slot + L.off[l] + (size_t)PAD * L.stride[l] + PAD;
}
__host__ __device__ inline uint8_t* level_origin(uint8_t* slot, const PyrLayout& L, int l) {
    return slot + L.off[l] + (size_t)PAD * L.stride[l] + PAD;
}

// ---- wave64 helpers -------------------------------------------------------------------------------------
__device__ inline long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// DPP all-reduce of an int32 over the wavefront (EXEC must be full): quad butterflies, half-row mirror, row mirror leave the
// 16-lane row sum in every lane of the row; the four row sums are combined on the scalar unit. ~10 instructions instead of
// six ds_bpermute round trips.
__device__ inline int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
// exact 64-bit sum of per-lane int32 partials: v = (v >> 16) * 65536 + (v & 0xffff), both halves summed in int32
__device__ inline long long wave_sum_i32_wide(int v) {
    const int hi = wave_sum_i32(v >> 16), lo = wave_sum_i32(v & 0xffff);
    return (long long)hi * 65536 + lo;
}
// Wavefront all-reduce of doubles on DPP row operations (a ds_bpermute shuffle of an f64 costs ~160 clk per step, a DPP move
// ~10): quad butterflies, row_half_mirror, row_mirror give every lane its 16-lane row sum; the four row sums are fetched with
// lane reads and added in a fixed order. Every lane receives the same bits.
__device__ inline double dpp_f64(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {   // the DPP control must be an immediate
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break;
    }
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64_c(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_sum_f64(double v) {
    v += dpp_f64(v, 0);   // quad_perm [1,0,3,2]
    v += dpp_f64(v, 1);   // quad_perm [2,3,0,1]
    v += dpp_f64(v, 2);   // row_half_mirror
    v += dpp_f64(v, 3);   // row_mirror
    return (readlane_f64_c(v, 0) + readlane_f64_c(v, 16)) + (readlane_f64_c(v, 32) + readlane_f64_c(v, 48));
}
// Wavefront all-reduce maximum of 64-bit keys on the same DPP row operations (the __shfl_xor form was six dependent pairs of ds_bpermute,
// ~1.5 k clk per call: most of a selection round of k_gftt_pick). Every lane receives the maximum.
__device__ inline unsigned long long dpp_u64(unsigned long long v, const int ctrl_sel) {
    int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
    switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break;
    }
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
// the same for 32-bit keys: one DPP move + v_max_u32 per step
__device__ inline unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ inline unsigned long long readlane_u64(unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64(v, 0); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = dpp_u64(v, 1); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = dpp_u64(v, 2); v = t > v ? t : v;   // row_half_mirror
    t = dpp_u64(v, 3); v = t > v ? t : v;   // row_mirror: every lane holds its 16-lane row's maximum
    const unsigned long long a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
    const unsigned long long ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// ---- launch entry points (frontend.hip) -------------------------------------------------------------------
struct LKParams {
    int max_iter;     // 30
    float eps2;       // not used in float: see eps2d
    double eps2d;     // epsilon^2 = 1e-4
    float min_eig;    // 1e-4
    unsigned long long* stamps;   // optional (diagnostic): 16 shader-clock phase timers accumulated by the block of track 0
};

// one workgroup of k_lk_batch: everything it needs to start in ONE 32-byte record (the records sit in mapped pinned host memory: a wave's
// first load is a round trip over PCIe, and it used to make three dependent ones - block -> sequence -> coordinates)
struct __attribute__((aligned(32))) LKBlock {
    unsigned long long prev_off, next_off;   // byte offsets of the sequence's prev / next frame slot
    float x, y;                              // the track's position in the prev frame
    int track;                               // index into the concatenated result arrays
    int pad;
};
hipError_t launch_lk_batch(hipStream_t s, const uint8_t* slots, const LKBlock* d_blocks, int n_blocks, const PyrLayout& L, const LKParams& P,
                           float* d_out_xy, uint8_t* d_status, float* d_err, uint16_t* d_work = nullptr);
hipError_t launch_bgr2gray(hipStream_t s, const uint8_t* d_bgr, int w, int h, int stride, uint8_t* d_gray);   // cv::cvtColor(BGR2GRAY), 8-bit
hipError_t launch_pad_level0(hipStream_t s, uint8_t* slots, const PyrLayout& L, int first_slot, int n, const uint8_t* tight = nullptr /* tight gray frames to take level 0 from; null: in place */);
hipError_t launch_pyrdown(hipStream_t s, uint8_t* slots, const PyrLayout& L, int level_dst, int first_slot, int n);
hipError_t launch_lk(hipStream_t s, const uint8_t* prev_slot, const uint8_t* next_slot, const PyrLayout& L,
                     const float* d_prev_xy, const int* d_order, int n_blocks, int n, const LKParams& P, float* d_out_xy,
                     uint8_t* d_status, float* d_err, uint16_t* d_work = nullptr);
// d_work (optional), per track: LK iterations executed over all levels | (level passes that iterated) << 8 - what the track cost. The
// roofline's OPS_lk is summed from these on the host (three atomics per track on shared counters cost the batched launch 14 % of the
// whole run's throughput: every wavefront of the chip ended on the same three L2 lines).

// Detector cells on the device: CELL_STRIDE ints per cell = (x0, y0, w, h, frame slot index, 0, 0, 0) — the slot index lets one
// launch serve cells of different frames (several sequences in one batch); `slots` is the base of the frame-slot array.
constexpr int CELL_STRIDE = 8;
// The back-end chains (PnP hypotheses + refit, the ~23 launches of an LM solve, two-view DLT) and the detector's selection pass are short,
// serially dependent launches whose waves share SIMDs with thousands of LK waves that keep the vector ALU issuing: in the mix every k_bamB_*
// kernel ran 3-4x longer than alone (kernel trace of a B = 128 run: campoint 64 vs 18 us, backsub 69 vs 15 us, no dispatch gap between them).
// s_setprio raises the issue priority of the wave that executes it, so the few waves of a chain get the instruction slots they ask for and the
// LK waves fill the rest.
#define BACKEND_PRIO() __builtin_amdgcn_s_setprio(3)
// GFTT: eig maps (n_cells * 255*255 floats), cell max (n_cells uint32 ordered keys), outputs
hipError_t launch_gftt(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells,
                       int max_per_cell, double quality, double min_dist, int unlimited, float* d_eig, unsigned* d_cellmax,
                       int* d_out_xy, int* d_out_count, int* d_flags, unsigned* d_spill /* n_cells * CELL_PIX candidate overflow area */);
hipError_t launch_shitomasi(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells,
                            int max_per_cell, double quality, double* d_resp, unsigned long long* d_cellmax,
                            int* d_out_xy, double* d_out_score, int* d_out_count, int* d_flags, unsigned* d_spill);

hipError_t launch_gftt_response(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells, float* d_eig, unsigned* d_cellmax);
constexpr int CELL_MAX = 255;                 // OdometryPipeline.h:31 grid_size
constexpr int CELL_PIX = CELL_MAX * CELL_MAX;

}  // namespace pmv
